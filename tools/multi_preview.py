"""A live preview over two devices while the primary trains (the reference's multi-device render_frame beside Testbed::train): a few
training steps, then a frame through the multi-device context, repeated. Under `rocprofv3 --hip-trace --stats` the HIP API table
shows which host synchronisations the loop performs (tools/multi_preview_trace.sh): none of them device-wide inside the loop.
On a one-GPU box the same device is listed twice (separate contexts, streams and buffers; the peer copies are local)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
torch.zeros(1, device="cuda")
n_dev = min(torch.cuda.device_count(), 2)
devices = [0, 1] if n_dev > 1 else [0, 0]
gt = native.Context(0)
gt.set_model(synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=15))
res, views = 128, 8
focal = scene.focal_from_fov_x(res, 0.6911)
mats = [scene.orbit_camera(360.0 * k / views, 20.0 + 5.0 * k) for k in range(views)]
opts = native.make_opts(background=(0.0, 0.0, 0.0, 0.0))
imgs = [gt.render(native.make_camera(m, res, res, focal), opts) for m in mats]
path = scene.write_transforms("/tmp/multi_preview_%d.json" % os.getpid(), mats, res, res, 0.6911)
ctx = native.Context(devices=devices)
ctx.load_training_data(path)
os.remove(path)
for i, im in enumerate(imgs):
    ctx.set_training_image(i, im)
ctx.reset_network(log2_hashmap_size=15, seed=1337)
cam = native.make_camera(scene.orbit_camera(33.0, 25.0), 256, 144, scene.focal_from_fov_x(256, 0.6911))
ctx.train(16, 1 << 16)
ctx.render(cam)
print("LOOP BEGIN", flush=True)
t0 = time.perf_counter()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    ctx.train(8, 1 << 16)
    img = ctx.render(cam)
dt = time.perf_counter() - t0
print(f"LOOP END: {dt * 1e3 / 30:.2f} ms per (8 training steps + one 256x144 frame over {len(devices)} devices), frame mean {img[..., :3].mean():.4f}", flush=True)
ctx.close(); gt.close()
