"""Diagnostic: the fixed cost of a frame. Solo frame time (one frame at a time, image left in HBM) of a 256x256 frame, of a
1080p frame, and of rank 0's share of a 1080p frame cut 8 ways -- the quantities that cap strong scaling over 8 GPUs.
usage: python tools/latency_probe.py [k_drain ...]   (schedule knob 5: samples a ray may emit per round once the tile queue is empty)"""
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
rgba = torch.zeros((1080, 1920, 4), dtype=torch.float32, device="cuda")
depth = torch.zeros((1080, 1920), dtype=torch.float32, device="cuda")
stream = torch.cuda.Stream()


def solo(w, h, opts, n=24):
    cams = [native.make_camera(scene.orbit_camera(az), w, h, scene.focal_from_fov_x(w, 0.6911)) for az in (0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0)]
    for i in range(3):
        ctx.render_device(cams[i % 8], opts, rgba.data_ptr(), depth.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        ctx.render_device(cams[i % 8], opts, rgba.data_ptr(), depth.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    hist = ctx.render_history(n)
    return wall, float(np.mean([s["kernel_device_ms"] for s in hist])), float(np.mean([s["n_samples"] for s in hist]))


for links in [int(a) for a in sys.argv[1:]] or [8]:
    ctx.set_schedule(64, 4, 32, 1, int(os.environ.get('NGP_KBUSY', '8')), links, 1, int(os.environ.get('NGP_THIN', '1')))
    a = solo(256, 256, native.make_opts())
    b = solo(1920, 1080, native.make_opts())
    c = solo(1920, 1080, native.make_opts(shard_index=0, shard_count=8, packed_output=True))
    print(f"k_drain {links}: 256x256 wall {a[0]:.3f} ms (kernel {a[1]:.3f}) | 1080p wall {b[0]:.3f} ms (kernel {b[1]:.3f}, {1920 * 1080 / b[0] / 1e3:.0f} Mrays/s) | "
          f"1/8 share wall {c[0]:.3f} ms (kernel {c[1]:.3f}) -> 8-way bound {b[0] / c[0]:.2f}x", flush=True)
ctx.close()
