"""Diagnostic: per-rank render time of an N-way tile-sharded 1080p frame, measured on ONE GPU (rank r of N rendered
alone). Tells how much of the strong-scaling loss is the kernel itself (tail, under-filled grid) before any gather."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
torch.zeros(1, device="cuda")
sc = synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19)
ctx = native.Context(0)
ctx.set_model(sc)
w, h = 1920, 1080
cams = [native.make_camera(scene.orbit_camera(az), w, h, scene.focal_from_fov_x(w, 0.6911)) for az in (0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0)]
INFLIGHT = int(os.environ.get("NGP_BENCH_INFLIGHT", "1"))
streams = [torch.cuda.Stream() for _ in range(INFLIGHT)]
bufs = [(torch.zeros((h, w, 4), dtype=torch.float32, device="cuda"), torch.zeros((h, w), dtype=torch.float32, device="cuda")) for _ in streams]
whole = None  # ms per step of the unsharded frame (N = 1 comes first), the numerator of the compute-side bound
for n in [int(a) for a in os.environ.get("NGP_SHARD_NS", "1,2,4,8").split(",")]:
    worst = 0.0
    for r in range(n if os.environ.get("NGP_SHARD_ALL_RANKS", "1") == "1" else 1):
        opts = native.make_opts(shard_index=r, shard_count=n, packed_output=n > 1)
        def go(i):
            rgba, depth = bufs[i % INFLIGHT]
            ctx.render_device(cams[i % 8], opts, rgba.data_ptr(), depth.data_ptr(), streams[i % INFLIGHT].cuda_stream)
        for i in range(4):
            go(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 32
        for i in range(K):
            go(i)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / K * 1e3
        hist = ctx.render_history(K)
        km = np.mean([s["kernel_ms"] for s in hist]); fm = np.mean([s["frame_ms"] for s in hist])
        worst = max(worst, wall)
        print(f"N={n} rank {r}: kernel {km:.4f} ms  frame(events) {fm:.4f} ms  wall/step {wall:.4f} ms", flush=True)
    if n == 1:
        whole = worst
    print(f"N={n}, {INFLIGHT} frame(s) in flight: slowest rank {worst:.4f} ms/step" + (f" -> compute-side speed-up bound {whole / worst:.2f}x of {n} (no gather)" if whole else ""), flush=True)
