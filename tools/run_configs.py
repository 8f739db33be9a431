"""the five configurations of BASELINE.json on one MI355X (synthetic stand-ins of the same
shape: the reference ships no snapshots, and /root/reference does not exist on the GPU box).
Prints one JSON object per configuration; PSNR / irradiance are against the CPU oracle at a reduced resolution."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene, meshio = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene", "meshio"))
import oracle as O

torch.zeros(1, device="cuda")
orc = O.Oracle()
CORES = min(len(os.sched_getaffinity(0)), 16)
FOV = 0.6911


def psnr(a, b):
    a = np.clip(a, 0, 1); b = np.clip(b, 0, 1)
    return float(-10 * np.log10(max(np.mean((a - b) ** 2), 1e-12)))


def srgb(x):
    return np.where(x < 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 1e-12), 0.41666) - 0.055)


def with_bitfield(sc):
    sc = dict(sc)
    grid = np.asarray(sc["density_grid"], np.float16).astype(np.float32)
    sc["density_grid_bitfield"], _ = orc.density_grid_to_bitfield(grid, sc["max_cascade"])
    return sc


def timed(ctx, cams, w, h, opts, steps=24, inflight=2):
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    outs = [(torch.zeros((h, w, 4), device="cuda"), torch.zeros((h, w), device="cuda")) for _ in streams]
    def go(i):
        b = i % inflight
        ctx.render_device(cams[i % len(cams)], opts, outs[b][0].data_ptr(), outs[b][1].data_ptr(), streams[b].cuda_stream)
    for i in range(4):
        go(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        go(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    hist = ctx.render_history(steps)
    return dt, {k: float(np.mean([s[k] for s in hist])) for k in ("n_rays_hit", "n_samples", "kernel_ms")}


def nerf_config(name, sc, w, h, azimuths, radius=4.03, check_wh=(240, 135)):
    sc = with_bitfield(sc)
    ctx = native.Context(0)
    ctx.set_model(sc)
    focal = scene.focal_from_fov_x(w, FOV)
    cams = [native.make_camera(scene.orbit_camera(az, 30.0, radius), w, h, focal) for az in azimuths]
    dt, st = timed(ctx, cams, w, h, native.make_opts())
    # parity at reduced resolution (the oracle is a scalar CPU program)
    cw, ch = check_wh
    mat = scene.orbit_camera(azimuths[0], 30.0, radius)
    img = ctx.render(native.make_camera(mat, cw, ch, scene.focal_from_fov_x(cw, FOV)))
    m = orc.make_model(sc)
    t0 = time.perf_counter()
    fb, _, ost = orc.render_nerf(m, orc.make_camera(mat, cw, ch, scene.focal_from_fov_x(cw, FOV)), orc.make_opts(n_threads=CORES))
    cpu_s = time.perf_counter() - t0
    ref = orc.tonemap(orc.accumulate(fb.reshape(-1, 4), np.zeros((cw * ch, 4), np.float32), 0)).reshape(ch, cw, 4)
    orc.release(m)
    rays = w * h
    S = st["n_samples"] / max(st["n_rays_hit"], 1)
    bytes_alg = st["n_samples"] * 512 + rays * 80
    out = {"config": name, "resolution": [w, h], "Mrays_s": round(rays / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3, 3), "kernel_ms": round(st["kernel_ms"], 3),
           "samples_per_hit_ray": round(S, 2), "hit_fraction": round(st["n_rays_hit"] / rays, 4),
           # over the frame time with two frames in flight (the HIP-event duration of a launch that overlaps its neighbour is not a rate)
           "gather_ceiling_frac": round(bytes_alg / dt / 9.437e12, 3), "hbm_roofline_frac": round(bytes_alg / dt / 8e12, 3), "mfma_roofline_frac": round(st["n_samples"] * 20480 / dt / 2.5e15, 4),
           "psnr_vs_oracle_db": round(psnr(srgb(img[..., :3]), srgb(ref[..., :3])), 1), "max_abs_diff": float(np.abs(img - ref).max()),
           "cpu_oracle_Mrays_s": round(cw * ch / cpu_s / 1e6, 4), "cpu_cores": CORES, "levels_hashed": int(sum(1 for r in scene.grid_layout(sc["encoding"])[1] if r ** 3 > 2 ** sc["encoding"]["log2_hashmap_size"]))}
    print(json.dumps(out), flush=True)
    return ctx, sc


def mesh_config(ctx, sc):
    # config 3: the Lego-shaped model + two inserted meshes of the bunny's / armadillo's triangle counts, sun shadow ray per hit,
    # 256x128-texel irradiance probe lighting the meshes (ShadeEnvMap)
    meshes = [(meshio.icosphere(4), (0.62, -0.05, 0.05)), (meshio.torus(224, 224), (-0.45, 0.3, 0.5))]
    for tris, c in meshes:
        ctx.add_mesh(tris, c)
    t0 = time.perf_counter()
    ctx.compute_envmap(0, 256, 128)
    probe_ms = (time.perf_counter() - t0) * 1e3
    w, h = 1920, 1080
    focal = scene.focal_from_fov_x(w, 0.8)
    cams = [native.make_camera(scene.orbit_camera(az, 25.0, 5.5), w, h, focal) for az in (0.0, 60.0, 120.0, 180.0, 240.0, 300.0)]
    opts = native.make_opts(testbed_mode=native.MODE_GEOMETRY, render_mode=native.RENDER_SHADE_ENVMAP)
    dt, st = timed(ctx, cams, w, h, opts, inflight=1)
    # parity at low resolution
    cw, ch = 192, 108
    mat = scene.orbit_camera(60.0, 25.0, 5.5)
    img = ctx.render(native.make_camera(mat, cw, ch, scene.focal_from_fov_x(cw, 0.8)), opts)
    _, irr = ctx.get_envmap()
    hnd = orc.mesh_scene(meshes)
    ocam = orc.make_camera(mat, cw, ch, scene.focal_from_fov_x(cw, 0.8))
    fb, db = orc.render_mesh(hnd, ocam, orc.make_mesh_opts(irradiance=irr))
    s2 = dict(sc)
    lo, hi = orc.mesh_scene_aabb(hnd)
    s2["render_aabb"] = (tuple(lo.tolist()), tuple(hi.tolist()))
    m = orc.make_model(s2)
    # (render_mode 1 = ShadeEnvMap: only ERenderMode::Shade linearises the NeRF's sRGB output in shade_kernel_nerf, :1393)
    fb2, _, _ = orc.render_nerf(m, ocam, orc.make_opts(depth_test=True, n_threads=CORES, render_mode=1), frame_buffer=fb, depth_buffer=db)
    ref = orc.tonemap(orc.accumulate(fb2.reshape(-1, 4), np.zeros((cw * ch, 4), np.float32), 0)).reshape(ch, cw, 4)
    # irradiance parity: E(n) on 512 random normals, HIP probe vs oracle probe
    m0 = orc.make_model(sc)
    env_o, _ = orc.compute_envmap(m0, orc.make_probe(0, 256, 128), orc.make_opts(n_threads=CORES))
    orc.release(m0)
    rng = np.random.default_rng(0)
    n = rng.normal(size=(512, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    irr_linf = float(np.abs(ctx.irradiance(n) - orc.irradiance(env_o, n)).max())
    orc.release(m)
    out = {"config": "3: Lego-shaped + 2 meshes (5120 + 100352 triangles), sun shadow ray, irradiance probe 256x128", "resolution": [w, h],
           "Mrays_s": round(w * h / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3, 3), "nerf_kernel_ms": round(st["kernel_ms"], 3), "probe_ms_incl_alloc": round(probe_ms, 2),
           "psnr_vs_oracle_db": round(psnr(srgb(img[..., :3]), srgb(ref[..., :3])), 1), "irradiance_linf": irr_linf}
    print(json.dumps(out), flush=True)
    ctx.clear_meshes()


if __name__ == "__main__":
    only = set(os.environ.get("RUN_ONLY", "1,2,3,4,5").split(","))
    az8 = (0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0)
    if "2" in only or "3" in only:
        ctx, sc = nerf_config("2: Lego-shaped (aabb 1, T19, b=2.0), 1080p", synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19), 1920, 1080, az8)
        if "3" in only:
            mesh_config(ctx, sc)
        ctx.close()
    if "4" in only:
        ctx, _ = nerf_config("4: fox-shaped (aabb 4, cone 1/256, 3 cascades, upstream b=2.44), 4K", synthetic.make_scene(aabb_scale=4, seed=7, log2_hashmap_size=19, pls_rule="upstream"), 3840, 2160, az8)
        ctx.close()
    if "5" in only:
        ctx, _ = nerf_config("5: garden-shaped (aabb 16, 5 cascades, upstream b=2.97), 1080p", synthetic.make_scene(aabb_scale=16, seed=11, log2_hashmap_size=19, pls_rule="upstream"), 1920, 1080, az8)
        ctx.close()
    if "1" in only:
        ctx, _ = nerf_config("1: fox-shaped plumbing case, 256x256", synthetic.make_scene(aabb_scale=4, seed=7, log2_hashmap_size=19, pls_rule="fork"), 256, 256, (45.0,), check_wh=(256, 256))
        ctx.close()
