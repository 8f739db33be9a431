"""How fast can one host thread issue frames? A tiny frame (64x64) on 8 streams, no synchronisation inside the loop: wall time per
ngp_render_device call = the host-side cost of a frame (ctypes + the library's bookkeeping + events + the launch)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
torch.zeros(1, device="cuda")
sc = synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19)
ctx = native.Context(0)
ctx.set_model(sc)
for (w, h) in ((64, 64), (1920, 1080)):
    cam = native.make_camera(scene.orbit_camera(45.0), w, h, scene.focal_from_fov_x(w, 0.6911))
    k = 8
    streams = [torch.cuda.Stream() for _ in range(k)]
    bufs = [(torch.zeros((h, w, 4), device="cuda"), torch.zeros((h, w), device="cuda")) for _ in range(k)]
    for shards in (1, 8):
        opts = native.make_opts(shard_index=0, shard_count=shards, packed_output=shards > 1)
        for i in range(16):
            ctx.render_device(cam, opts, bufs[i % k][0].data_ptr(), bufs[i % k][1].data_ptr(), streams[i % k].cuda_stream)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for i in range(n):
            ctx.render_device(cam, opts, bufs[i % k][0].data_ptr(), bufs[i % k][1].data_ptr(), streams[i % k].cuda_stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{w}x{h} shards {shards}: issue {1e6 * (t1 - t0) / n:.1f} us/call, incl. drain {1e6 * (t2 - t0) / n:.1f} us/call")
