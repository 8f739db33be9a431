"""Frame time of the bench scene through the kernels a non-pinhole camera selects (depth of field, OpenCV lens, environment map) next to the plain one:
how much do the out-of-line camera paths and the spills of render_nerf_fused_unit cost?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
torch.zeros(1, device="cuda")
ctx = native.Context(0)
ctx.set_model(synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19))
w, h = 1920, 1080
focal = scene.focal_from_fov_x(w, 0.6911)
rgba = torch.zeros((h, w, 4), device="cuda")
depth = torch.zeros((h, w), device="cuda")
cases = {
    "plain pinhole": dict(),
    "depth of field (aperture 0.01)": dict(aperture_size=0.01, focus_z=3.0),
    "depth of field (aperture 1e-6: the code path without the divergence)": dict(aperture_size=1e-6, focus_z=3.0),
    "OpenCV lens (k1 = 0.05)": dict(lens_mode=1, lens_params=(0.05, 0.0, 0.0, 0.0)),
}
for name, kw in cases.items():
    cams = [native.make_camera(scene.orbit_camera(az), w, h, focal, **kw) for az in (0, 45, 90, 135)]
    opts = native.make_opts()
    for i in range(4):
        ctx.render_device(cams[i % 4], opts, rgba.data_ptr(), depth.data_ptr(), None)
    torch.cuda.synchronize()
    n = 16
    t0 = time.perf_counter()
    for i in range(n):
        ctx.render_device(cams[i % 4], opts, rgba.data_ptr(), depth.data_ptr(), None)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = ctx.render_stats()
    print(f"{name}: {dt * 1e3:.3f} ms/frame one at a time, {st['n_samples']} samples, kernel {st['kernel_device_ms']:.3f} ms")
