"""Diagnostic: section shares / wave timeline of the fused kernel (NGP_PROFILE_SECTIONS=1 selects the stamped twin) for a frame size.
usage: NGP_PROFILE_SECTIONS=1 python tools/section_probe.py W H [shard_count]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
w, h = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sc = synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19)
ctx = native.Context(0)
ctx.set_model(sc)
cam = native.make_camera(scene.orbit_camera(45.0), w, h, scene.focal_from_fov_x(w, 0.6911))
opts = native.make_opts(shard_index=0, shard_count=n) if n > 1 else native.make_opts()
for _ in range(3):
    ctx.render(cam, opts)
    st = ctx.render_stats()
print(st)
