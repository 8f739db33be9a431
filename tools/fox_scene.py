"""One real scene end to end: the reference's own sample dataset (data/nerf/fox, committed as data under tests/golden/fox/, photographs
downscaled by 2) trained with this build's trainer and evaluated the way scripts/run.py:210-268 evaluates (--test_transforms: black
background, pixel centres, 8 spp, min_transmittance 1e-4, PSNR on sRGB-encoded, clipped images), through `pyngp`.

usage: python tools/fox_scene.py [--steps N] [--table 16|19] [--save PATH.ingp] [--load PATH.ingp] [--perf]
  --perf adds BASELINE.json's configurations 4 (3840x2160) and 1 (256x256) on the trained model: Mrays/s, samples per hit ray, hit fraction."""
import argparse, importlib, json, math, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
FOX = os.path.join(ROOT, "tests", "golden", "fox")


def linear_to_srgb(x):  # scripts/common.py:144-146
    return np.where(x < 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 1e-12), 0.41666) - 0.055)


PYNGP = None


def evaluate(testbed, transforms):
    """scripts/run.py:210-268 (SSIM left out: scripts/common.py takes it from a library that is not in the image)"""
    testbed.background_color = [0.0, 0.0, 0.0, 1.0]
    # upstream's default render mode. This fork's default is ShadeGridEnvMap (testbed.h:880), which -- unlike Shade -- does not take the
    # network's sRGB colours back to linear (shade_kernel_nerf, src/testbed_nerf.cu:1393): its own run.py would compare an sRGB-encoded
    # render, encoded once more, with the photograph. The procedure's intent is Shade.
    testbed.render_mode = PYNGP.RenderMode.Shade
    testbed.snap_to_pixel_centers = True
    testbed.nerf.render_min_transmittance = 1e-4
    testbed.shall_train = False
    testbed.load_training_data(transforms)
    psnrs = []
    for i in range(testbed.nerf.training.dataset.n_images):
        res = testbed.nerf.training.dataset.metadata[i].resolution
        testbed.render_ground_truth = True
        testbed.set_camera_to_training_view(i)
        ref = testbed.render(res[0], res[1], 1, True)
        testbed.render_ground_truth = False
        img = testbed.render(res[0], res[1], 8, True)
        a = np.clip(linear_to_srgb(img[..., :3]), 0.0, 1.0)
        r = np.clip(linear_to_srgb(ref[..., :3]), 0.0, 1.0)
        psnrs.append(-10.0 * math.log10(max(float(((a - r) ** 2).mean()), 1e-12)))
    return psnrs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--table", type=int, default=19)
    ap.add_argument("--save", default="")
    ap.add_argument("--load", default="")
    ap.add_argument("--perf", action="store_true")
    args = ap.parse_args()
    import torch

    torch.zeros(1, device="cuda")
    global PYNGP
    pyngp = PYNGP = importlib.import_module(PKG + ".build").import_pyngp()
    out = {"scene": "data/nerf/fox (reference sample dataset; photographs downscaled x2: 540x960), 44 training views / 6 held out"}
    testbed = pyngp.Testbed()
    testbed.root_dir = FOX
    if args.load:
        testbed.load_snapshot(args.load)
        out["snapshot"] = os.path.relpath(args.load, ROOT)
    else:
        testbed.load_training_data(os.path.join(FOX, "transforms_train.json"))
        if args.table != 19:
            testbed.reload_network_from_file(os.path.join(FOX, "base_t16.json") if args.table == 16 else "")
        testbed.shall_train = True
        t0 = time.perf_counter()
        while testbed.frame():  # scripts/run.py:172-203
            if testbed.training_step >= args.steps:
                break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.update({"training_steps": int(testbed.training_step), "training_seconds": round(dt, 2), "steps_per_s": round(testbed.training_step / dt, 1), "loss": float(testbed.loss),
                    "log2_hashmap_size": args.table})
        if args.save:
            testbed.save_snapshot(args.save, False)
            out["snapshot"] = os.path.relpath(args.save, ROOT)
            out["snapshot_bytes"] = os.path.getsize(args.save)
    psnrs = evaluate(testbed, os.path.join(FOX, "transforms_test.json"))
    out["heldout_psnr_db"] = {"mean": round(float(np.mean(psnrs)), 2), "min": round(min(psnrs), 2), "max": round(max(psnrs), 2), "views": len(psnrs),
                              "procedure": "scripts/run.py:210-268 (black background, pixel centres, 8 spp, min_transmittance 1e-4, sRGB, clipped)"}
    psnrs_tr = evaluate(testbed, os.path.join(FOX, "transforms_train.json"))
    out["training_view_psnr_db"] = round(float(np.mean(psnrs_tr)), 2)
    if args.perf:
        # BASELINE.json configurations 4 and 1 on the trained model: the camera of training view 0 at 3840x2160 / 256x256, two frames in flight
        native = importlib.import_module(PKG + ".native")
        snap = args.load or args.save
        if not snap:
            snap = "/tmp/fox_perf_%d.ingp" % os.getpid()
            testbed.save_snapshot(snap, False)
        ctx = native.Context(0)
        ctx.load_snapshot_file(snap)
        ctx.load_training_data(os.path.join(FOX, "transforms.json"))
        perf = {}
        for name, (w, h) in (("config 4 (3840x2160)", (3840, 2160)), ("config 1 (256x256)", (256, 256)), ("1920x1080", (1920, 1080))):
            views = [ctx.training_view(v) for v in (0, 10, 20, 30)]  # the view's pose and vertical field of view, pinhole, at the configuration's resolution
            cams = [native.make_camera(tv["matrix"], w, h, (float(tv["focal_length"][1]) * h / float(tv["resolution"][1]),) * 2) for tv in views]
            streams = [torch.cuda.Stream() for _ in range(2)]
            bufs = [(torch.zeros((h, w, 4), device="cuda"), torch.zeros((h, w), device="cuda")) for _ in streams]
            opts = native.make_opts()
            def go(i):
                ctx.render_device(cams[i % 4], opts, bufs[i % 2][0].data_ptr(), bufs[i % 2][1].data_ptr(), streams[i % 2].cuda_stream)
            for i in range(4):
                go(i)
            torch.cuda.synchronize()
            n = 16
            t0 = time.perf_counter()
            for i in range(n):
                go(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            hist = ctx.render_history(n)
            hit = float(np.mean([s["n_rays_hit"] for s in hist])); smp = float(np.mean([s["n_samples"] for s in hist]))
            perf[name] = {"Mrays_s": round(w * h / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3, 3), "samples_per_hit_ray": round(smp / max(hit, 1), 2), "hit_fraction": round(hit / (w * h), 4),
                          "gather_ceiling_frac": round((smp * 512 + w * h * 80) / dt / 9.437e12, 3), "hbm_roofline_frac": round((smp * 512 + w * h * 80) / dt / 8e12, 3)}
        out["performance_two_frames_in_flight"] = perf
        ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
