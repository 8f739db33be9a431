"""The PCIe-inclusive rate of the host-buffer entry point (ngp_render: render + D2H of the 33 MB frame), for DESIGN.md section 4."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
ctx = native.Context(0)
ctx.set_model(synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19))
w, h = 1920, 1080
cams = [native.make_camera(scene.orbit_camera(az), w, h, scene.focal_from_fov_x(w, 0.6911)) for az in (0.0, 45.0, 90.0, 135.0)]
for c in cams:
    ctx.render(c)
t0 = time.perf_counter()
n = 12
for i in range(n):
    ctx.render(cams[i % 4])
dt = (time.perf_counter() - t0) / n
print(f"ngp_render (host image, pageable numpy buffer): {dt * 1e3:.2f} ms/frame = {w * h / dt / 1e6:.1f} Mrays/s")
