"""Diagnostic: timelines of single waves of the fused render kernel (its stamped twin), for a frame size / a rank's share.

usage: python tools/wave_trace.py W H [shard_count] [level]
  level 1: section stamps (refill + spawn, march, network, composite); level 2: also inside the network section
  (gather issue / gather wait / corner sums / MFMA chains / hand-back -- that build serialises what the shipped kernel overlaps).
The environment variables the library reads (NGP_PROFILE_SECTIONS, NGP_PROFILE_TRACE) are set here before it is loaded.
Every stride-th wave that is dealt rays records one 64-byte record per loop round (csrc/ngp_kernels.h FrameParams::trace)."""
import importlib, os, sys

w, h = int(sys.argv[1]), int(sys.argv[2])
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 1
level = int(sys.argv[4]) if len(sys.argv) > 4 else 1
tiles = ((w + 7) // 8) * ((h + 7) // 8) // shards
os.environ["NGP_PROFILE_SECTIONS"] = str(level)
os.environ["NGP_PROFILE_TRACE"] = str(max(1, min(tiles, 3072) // 64))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
import torch

torch.zeros(1, device="cuda")  # (torch bundles its own HIP runtime: it has to initialise first)
aabb = int(os.environ.get("NGP_TRACE_AABB", "1"))  # 1: the bench model; 4 / 16: the fox- / garden-shaped scenes of tools/run_configs.py (the general kernel's stamped twin)
sc = synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19) if aabb == 1 else synthetic.make_scene(aabb_scale=aabb, seed=7 if aabb == 4 else 11, log2_hashmap_size=19, pls_rule="upstream")
ctx = native.Context(0)
ctx.set_model(sc)
if os.environ.get("NGP_SCHEDULE"):
    ctx.set_schedule(*[int(a) for a in os.environ["NGP_SCHEDULE"].split(",")])
cam = native.make_camera(scene.orbit_camera(45.0), w, h, scene.focal_from_fov_x(w, 0.6911))
opts = native.make_opts(shard_index=0, shard_count=shards, packed_output=True) if shards > 1 else native.make_opts()
rgba = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
depth = torch.zeros((h, w), dtype=torch.float32, device="cuda")
for _ in range(4):
    ctx.render_device(cam, opts, rgba.data_ptr(), depth.data_ptr(), 0)
    torch.cuda.synchronize()
st = ctx.render_stats()
n_working, hd, rec = ctx.profile_trace()
print(f"# {w}x{h} shards {shards} level {level}: kernel {st['kernel_device_ms']:.4f} ms (device clock, stamped build), {st['n_rays']} rays, {st['n_rays_hit']} hit, {st['n_samples']} samples; "
      f"{n_working} waves were dealt rays, stride {os.environ['NGP_PROFILE_TRACE']}")
used = [i for i in range(hd.shape[0]) if hd[i, 4] > 0]
if not used:
    sys.exit("no traced wave")
rt0 = min((int(hd[i, 6]) << 32) | int(hd[i, 5]) for i in used)
clk = []
rows = []
print("# wave: xcd cu simd | arrives at us (100 MHz clock, first traced wave = 0) | staging cyc | rounds | samples | busy us | cycles/round")
for i in used:
    hw = int(hd[i, 0])
    rt_in = ((int(hd[i, 6]) << 32) | int(hd[i, 5])) - rt0
    rt_out = ((int(hd[i, 8]) << 32) | int(hd[i, 7])) - rt0
    n_it = min(int(hd[i, 4]), rec.shape[1])
    cyc = (int(hd[i, 12]) - int(hd[i, 9])) & 0xffffffff
    if rt_out > rt_in:
        clk.append(cyc / ((rt_out - rt_in) * 10.0))  # cycles per ns
    rows.append((i, rt_in * 0.01, (int(hd[i, 10]) - int(hd[i, 9])) & 0xffffffff, n_it, int(hd[i, 13]), (rt_out - rt_in) * 0.01, cyc / max(n_it, 1)))
    if len(rows) <= 12:
        print(f"  {int(hd[i, 1])} {(hw >> 8) & 15:2d} {(hw >> 4) & 3} | {rows[-1][1]:8.2f} | {rows[-1][2]:6d} | {n_it:4d} | {rows[-1][4]:5d} | {rows[-1][5]:8.2f} | {rows[-1][6]:8.0f}")
ghz = float(np.median(clk)) if clk else 0.0
print(f"# shader clock during the launch: {ghz:.3f} GHz (median of s_memtime / s_memrealtime over the traced waves)")
a = np.array([r[1:] for r in rows], np.float64)
print(f"# {len(rows)} traced waves: arrival spread {a[:, 0].min():.1f} .. {a[:, 0].max():.1f} us | staging {np.median(a[:, 1]):.0f} cycles | rounds median {np.median(a[:, 2]):.0f} max {a[:, 2].max():.0f} | "
      f"busy median {np.median(a[:, 4]):.1f} us max {a[:, 4].max():.1f} us | cycles/round median {np.median(a[:, 5]):.0f}")

# ---- what a round costs, by kind
kinds = {"network": [], "stall (waits for marching lanes / chain growth)": [], "march only (nothing ready)": []}
per_section = {k: [] for k in kinds}
inner = []
n_run_hist = np.zeros(65, np.int64)
for (i, *_rest) in rows:
    n_it = min(int(hd[i, 4]), rec.shape[1])
    r = rec[i, :n_it].astype(np.int64)
    for j in range(n_it):
        t1, t2, t3, t4 = r[j, 1], r[j, 2], r[j, 3], r[j, 4]
        n_ready, n_run, n_pass = r[j, 5] & 255, (r[j, 5] >> 8) & 255, (r[j, 5] >> 16) & 255
        nxt = (int(r[j + 1, 0]) - int(r[j, 0])) & 0xffffffff if j + 1 < n_it else t4
        if t3 > t2:
            kind = "network"
            n_run_hist[n_run] += 1
            inner.append(r[j, 8:13])
        elif n_ready > 0:
            kind = "stall (waits for marching lanes / chain growth)"
        else:
            kind = "march only (nothing ready)"
        kinds[kind].append(nxt)
        per_section[kind].append((t1, t2 - t1, t3 - t2, t4 - t3, nxt - t4, (r[j, 7] & 255), (r[j, 7] >> 8)))
tot = sum(sum(v) for v in kinds.values())
print("# rounds by kind: count | mean cycles | share of wave time | mean cycles in refill+spawn, march, network, composite+chains, loop edge | march iterations, lane-steps per round")
for k, v in kinds.items():
    if not v:
        continue
    s = np.array(per_section[k], np.float64).mean(axis=0)
    print(f"  {k:48s} {len(v):6d} | {np.mean(v):7.0f} | {100.0 * sum(v) / tot:5.1f} % | {s[0]:6.0f} {s[1]:6.0f} {s[2]:6.0f} {s[3]:6.0f} {s[4]:6.0f} | {s[5]:.2f} {s[6]:.1f}")
nz = np.nonzero(n_run_hist)[0]
print("# samples per network round (n_run: count): " + " ".join(f"{k}:{n_run_hist[k]}" for k in nz))
if level >= 2 and inner:
    m = np.array(inner, np.float64).mean(axis=0)
    print(f"# inside a network round (mean cycles; stamps serialise): address arithmetic + gather issue {m[0]:.0f} | gather wait {m[1]:.0f} | corner sums {m[2]:.0f} | MFMA chains {m[3]:.0f} | hand-back {m[4]:.0f}")
# ---- the longest wave, round by round (first 24 rounds and the last 8)
longest = max(rows, key=lambda r: r[3])[0]
n_it = min(int(hd[longest, 4]), rec.shape[1])
print(f"# longest traced wave ({n_it} rounds): round | cycles since its first round | refill march network composite | ready run passes stall | alive marching continuations | flags (1 queue empty, 4 retired, 8 refilled)")
r = rec[longest, :n_it].astype(np.int64)
for j in list(range(min(24, n_it))) + list(range(max(24, n_it - 8), n_it)):
    print(f"  {j:4d} | {(int(r[j, 0]) - int(r[0, 0])) & 0xffffffff:9d} | {r[j, 1]:6d} {r[j, 2] - r[j, 1]:6d} {r[j, 3] - r[j, 2]:6d} {r[j, 4] - r[j, 3]:6d} | {r[j, 5] & 255:3d} {(r[j, 5] >> 8) & 255:3d} {(r[j, 5] >> 16) & 255:2d} {(r[j, 5] >> 24) & 255:2d} | "
          f"{r[j, 6] & 255:3d} {(r[j, 6] >> 8) & 255:3d} {(r[j, 6] >> 16) & 255:3d} | {(r[j, 6] >> 24) & 255:2d}")
# ---- the first rounds of a few waves (cold start: first touches of code, occupancy words, table lines)
print("# first 5 rounds of 8 traced waves: wave | per round: refill/march/network/composite cycles [slots]")
for (i, *_r) in rows[:64:8]:
    n_it = min(int(hd[i, 4]), rec.shape[1], 5)
    r = rec[i, :n_it].astype(np.int64)
    print(f"  wave {i:2d} (arrives {rows[[q[0] for q in rows].index(i)][1]:.2f} us): " + " | ".join(f"{r[j, 1]}/{r[j, 2] - r[j, 1]}/{r[j, 3] - r[j, 2]}/{r[j, 4] - r[j, 3]} [{r[j, 5] & 255}]" for j in range(n_it)))
ctx.close()
