"""Board power and clocks while the configs/nerf/frequency.json renderer runs back to back (wide_kernels.hip): is the kernel's MFMA rate set by
the chip's power management rather than by its issue stream? Polls the amdgpu hwmon files (or rocm-smi) from a thread while frames render for
SECONDS (default 6), once with two workgroups per CU (the shipped launch) and once with one (NGP_BLOCKS_PER_CU=1 in a child process)."""
import glob
import importlib
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "temp1_input"):
            v = read(os.path.join(hw, name))
            if v is not None:
                out.setdefault(name, []).append(int(v))
    return out


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
        return r.stdout.strip()[:600]
    except Exception as e:  # noqa: BLE001
        return "rocm-smi: %r" % (e,)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        native = importlib.import_module(PKG + ".native")
        S = importlib.import_module(PKG + ".scene")
        syn = importlib.import_module(PKG + ".synthetic")
        sc = syn.make_scene(aabb_scale=1, seed=1234, cfg=S.frequency_network_config())
        ctx = native.Context(0)
        ctx.set_model(sc)
        cam = native.make_camera(S.orbit_camera(45.0), 1920, 1080, S.focal_from_fov_x(1920, 0.6911))
        opts = native.make_opts(to_srgb=True)
        ctx.render(cam, opts)
        samples, stop = [], False

        def poll():
            while not stop:
                samples.append((time.perf_counter(), sensors()))
                time.sleep(0.05)

        th = threading.Thread(target=poll)
        th.start()
        t0 = time.perf_counter()
        n = 0
        secs = float(os.environ.get("SECONDS_", 6))
        while time.perf_counter() - t0 < secs:
            ctx.render(cam, opts)
            n += 1
        ms = ctx.render_stats()["kernel_device_ms"]
        st = ctx.render_stats()
        stop = True
        th.join()
        tail = [s for t, s in samples if t - t0 > 0.5 * secs]  # the second half: clocks have settled
        def stat(key, scale):
            v = [s[key][0] * scale for s in tail if key in s]
            return "n/a" if not v else "%.0f (min %.0f max %.0f)" % (sum(v) / len(v), min(v), max(v))
        print("blocks/CU %s: %d frames, last kernel %.2f ms = %.0f TFLOP/s | power W %s, cap %s | sclk MHz %s | temp C %s" % (
            os.environ.get("NGP_BLOCKS_PER_CU", "2"), n, ms, st["n_samples"] * 434176 * 2 / ms / 1e9, stat("power1_average", 1e-6) if any("power1_average" in s for s in tail) else stat("power1_input", 1e-6),
            stat("power1_cap", 1e-6), stat("freq1_input", 1e-6), stat("temp1_input", 1e-3)))
        if not tail or not any(tail):
            print(smi())
        return
    for env in ({}, {"NGP_BLOCKS_PER_CU": "1"}):
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, check=False)


if __name__ == "__main__":
    main()
