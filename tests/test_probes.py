"""Irradiance probes: ray fans, probe texture and irradiance E(n).
north_star bar: per-pixel irradiance L-inf < 1e-3 against the reference renderer (here: the oracle)."""
import numpy as np
import pytest

from conftest import pkg, psnr


def test_probe_ray_fans_oracle(oracle, scene_unit):
    m = oracle.make_model(scene_unit)
    d = oracle.make_probe(0, 16, 8)
    pl = oracle.probe_payloads(m, d)
    assert pl.shape[0] == 128 and pl["alive"].all() and (pl["t"] == 0).all()
    assert np.allclose(pl["origin"], 0.5)  # render_aabb.center()
    assert np.allclose(np.linalg.norm(pl["dir"], axis=1), 1.0, atol=1e-6)
    assert np.array_equal(pl["idx"], np.arange(128))
    # cos(theta)-uniform rows: z = 1 - 2 i / n_theta
    z = pl["dir"][:, 2].reshape(8, 16)
    assert np.allclose(z, (1 - 2 * np.arange(16) / 16)[None, :], atol=1e-6)
    # equal-area parameterisation: the mean direction of the full fan vanishes except for the polar offset of the grid
    assert np.abs(pl["dir"][:, :2].mean(0)).max() < 1e-6
    # multi-centre: n_origin^2 rays per texel share idx, origins are Halton-jittered around the centre
    d2 = oracle.make_probe(2, 8, 4, n_origin=3)
    pl2 = oracle.probe_payloads(m, d2)
    assert pl2.shape[0] == 8 * 4 * 9
    counts = np.bincount(pl2["idx"], minlength=32)
    assert (counts == 9).all()
    off = pl2["origin"] - 0.5
    assert np.abs(off).max() <= 0.5 and len(np.unique(off.round(6), axis=0)) == 9
    # outward: rays start at the shell position and point back towards the scene (negated local frame)
    d1 = oracle.make_probe(1, 8, 4, origin=(1.4, 1.0, 0.3))
    pl1 = oracle.probe_payloads(m, d1)
    assert np.allclose(pl1["origin"], [1.4, 1.0, 0.3])
    n = np.float32([1.4, 1.0, 0.3]) / np.linalg.norm([1.4, 1.0, 0.3])
    assert np.allclose((pl1["dir"] @ n).reshape(4, 8), -(1 - 2 * np.arange(8) / 8)[None, :], atol=1e-5)
    oracle.release(m)


def test_irradiance_of_constant_environment(oracle):
    # E(n) of a constant radiance L is pi L for every normal: checks dOmega and the equal-area parameterisation
    env = np.zeros((64, 128, 4), np.float32)
    env[..., :3] = [0.5, 1.0, 2.0]
    rng = np.random.default_rng(1)
    n = rng.normal(size=(16, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    E = oracle.irradiance(env, n)
    # (texel directions sit at i/n_theta, not at cell centres -- as the reference's generators do -- so the
    # quadrature is first order: ~1/n_theta)
    assert np.allclose(E, np.pi * np.float32([0.5, 1.0, 2.0]), rtol=2e-2)
    # a single bright texel: E = L max(0, n.w) dOmega
    env2 = np.zeros((8, 16, 4), np.float32)
    env2[3, 5, :3] = 7.0
    w = oracle.texel_directions(16, 8)[3, 5]
    E2 = oracle.irradiance(env2, np.stack([w, -w]))
    assert np.allclose(E2[0], 7.0 * 4 * np.pi / 128, rtol=1e-5) and np.all(E2[1] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,kw", [(0, {}), (2, {"n_origin": 2}), (1, {"origin": (0.9, 0.85, 0.8)})])
def test_envmap_and_irradiance_parity(mode, kw, gpu_ctx, oracle, scene_unit):
    gpu_ctx.set_model(scene_unit)
    m = oracle.make_model(scene_unit)
    nt, nph = 64, 32
    env = gpu_ctx.compute_envmap(mode, nt, nph, **kw)
    st = gpu_ctx.render_stats()
    ref, ost = oracle.compute_envmap(m, oracle.make_probe(mode, nt, nph, **kw))
    assert st["n_rays"] == ost["n_rays"] and st["n_samples"] > 0
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 2e-3 * ost["n_samples"] + 2
    assert np.array_equal(env[..., 3] > 0, ref[..., 3] > 0) or (np.not_equal(env[..., 3] > 0, ref[..., 3] > 0).mean() < 0.002)
    assert np.abs(env - ref).max() < 1e-2 and np.abs(env - ref).mean() < 2e-4
    rng = np.random.default_rng(8)
    n = rng.normal(size=(64, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    E = gpu_ctx.irradiance(n)
    org = kw.get("origin") if mode == 1 else None  # an outward probe's texels hold radiance along -frame(normalize(origin)) * texel direction
    Eref = oracle.irradiance(ref, n, origin=org)
    assert np.abs(E - Eref).max() < 1e-3          # north_star: irradiance L-inf < 1e-3
    # the irradiance operator itself (same texture on both sides) agrees to rounding
    assert np.abs(E - oracle.irradiance(env, n, origin=org)).max() < 2e-6
    env2, irr = gpu_ctx.get_envmap()
    assert np.array_equal(env2, env)
    dirs = oracle.texel_directions(nt, nph).reshape(-1, 3)
    assert np.abs(irr.reshape(-1, 4)[:, :3] - oracle.irradiance(env, dirs, origin=org)).max() < 2e-6
    # the lookup mesh shading does: bilinear in the tabulated map (read_envmap's scheme)
    got = gpu_ctx.irradiance_at(np.zeros_like(n), n)
    want = np.stack([oracle.irradiance_read(irr, v) for v in n])
    assert np.abs(got - want).max() < 2e-6
    oracle.release(m)


def test_irradiance_read_is_bilinear(oracle):
    """orc_irradiance_read: at a texel direction it returns the texel; between two texels of a row the mean; phi wraps, theta clamps"""
    nt, nph = 8, 6
    rng = np.random.default_rng(2)
    tab = rng.uniform(0, 1, (nph, nt, 4)).astype(np.float32)
    dirs = oracle.texel_directions(nt, nph)
    for (a, b) in ((1, 0), (3, 2), (7, 5), (4, 3)):  # (texel a = 0 is the pole: every b is the same direction)
        assert np.allclose(oracle.irradiance_read(tab, dirs[b, a]), tab[b, a, :3], atol=2e-5)
    # halfway in phi between texels (3, 5) and (3, 0): periodic
    ang = 2 * np.pi * (5.5 / nph - 0.5)  # texel b sits at phi = 2 pi (b / n_phi - 0.5)
    z = dirs[5, 3][2]
    v = np.float32([np.sqrt(1 - z * z) * np.cos(ang), np.sqrt(1 - z * z) * np.sin(ang), z])
    assert np.allclose(oracle.irradiance_read(tab, v), 0.5 * (tab[5, 3, :3] + tab[0, 3, :3]), atol=1e-4)
    # beyond the last theta row the read clamps to it
    south = np.float32([0.0, 1e-4, -1.0])
    south /= np.linalg.norm(south)
    got = oracle.irradiance_read(tab, south)
    assert np.all(got >= tab[:, nt - 1, :3].min(0) - 1e-6) and np.all(got <= tab[:, nt - 1, :3].max(0) + 1e-6)


BIG_BOX = ((-1.5, -1.5, -1.5), (2.5, 2.5, 2.5))  # a render box that holds the shell positions (in Geometry mode: the inflated scene box)


def test_probe_grid_oracle_properties(oracle, scene_unit):
    """The grid of probes on the CPU: shell positions, one K11 fan per position, constant-radiance sanity of the blended lookup."""
    m = oracle.make_model(dict(scene_unit, render_aabb=BIG_BOX))
    d = oracle.make_probe_grid(4, 6, 16, 8, shell_radius=0.9)
    org = oracle.probe_grid_origins(d)
    assert org.shape == (24, 3) and np.allclose(np.linalg.norm(org - 0.5, axis=1), 0.9, atol=1e-5)
    # rows of constant cos(theta): z = 0.5 + R (1 - 2 (i + 0.5) / grid_x)
    assert np.allclose(org[:, 2].reshape(6, 4), 0.5 + 0.9 * (1 - 2 * (np.arange(4) + 0.5) / 4)[None, :], atol=1e-5)
    env, st = oracle.compute_envmap_grid(m, d)
    assert env.shape == (24, 8, 16, 4) and st["n_rays"] == 24 * 128
    one, st1 = oracle.compute_envmap(m, oracle.make_probe(1, 16, 8, origin=tuple(org[7])))
    assert np.array_equal(env[7], one)
    # every probe looks at the scene from outside: probes on different sides of the object see different texels lit
    lit = (env[..., 3] > 0).reshape(24, -1)
    assert lit.any(1).all() and len({tuple(r) for r in lit}) > 12
    const = np.zeros_like(env)
    const[..., :3] = [0.2, 0.4, 0.8]
    tab = oracle.irradiance_grid_tabulate(d, const)
    rng = np.random.default_rng(0)
    p = rng.uniform(-0.5, 1.5, (64, 3)).astype(np.float32)
    n = rng.normal(size=(64, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    E = oracle.irradiance_grid_lookup(d, tab, p, n)
    assert np.allclose(E, np.pi * np.float32([0.2, 0.4, 0.8]), rtol=0.12)  # first-order quadrature at 16 x 8 texels
    oracle.release(m)


@pytest.mark.gpu
def test_probe_grid_parity(gpu_ctx, oracle, scene_unit):
    """Testbed::computeEnvmapGrid: all probes in one launch vs the oracle's probe-by-probe trace; E tables; blended lookup
    (north_star: per-pixel irradiance L-inf < 1e-3)."""
    sc = dict(scene_unit, render_aabb=BIG_BOX)
    gpu_ctx.set_model(sc)
    gpu_ctx.clear_meshes()
    m = oracle.make_model(sc)
    gx, gy, nt, nph, R = 3, 4, 32, 16, 0.95
    env = gpu_ctx.compute_envmap_grid(gx, gy, nt, nph, shell_radius=R)
    st = gpu_ctx.render_stats()
    d = oracle.make_probe_grid(gx, gy, nt, nph, shell_radius=R)
    ref, ost = oracle.compute_envmap_grid(m, d)
    assert st["n_rays"] == ost["n_rays"] == gx * gy * nt * nph
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 2e-3 * ost["n_samples"] + 2
    assert (np.not_equal(env[..., 3] > 0, ref[..., 3] > 0)).mean() < 0.002
    assert np.abs(env - ref).max() < 1e-2 and np.abs(env - ref).mean() < 2e-4
    desc, org, env2, irr = gpu_ctx.get_envmap_grid()
    assert (desc.grid_x, desc.grid_y, desc.n_theta, desc.n_phi) == (gx, gy, nt, nph) and np.array_equal(env2, env)
    assert np.abs(org - oracle.probe_grid_origins(d)).max() < 1e-6
    tab_ref = oracle.irradiance_grid_tabulate(d, ref)
    assert np.abs(irr[..., :3] - tab_ref[..., :3]).max() < 1e-3
    assert np.abs(irr[..., :3] - oracle.irradiance_grid_tabulate(d, env)[..., :3]).max() < 2e-6  # the operator alone
    rng = np.random.default_rng(5)
    p = rng.uniform(-0.6, 1.6, (2048, 3)).astype(np.float32)
    p[:4] = [[0.5, 0.5, 0.5], [0.5, 0.5, 2.0], [0.5, 0.5, -1.0], [1.5, 0.5, 0.5]]  # centre (no direction), poles, seam
    n = rng.normal(size=(2048, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[:3] = [[0, 0, 1], [0, 0, -1], [-1, 0, 0]]
    E = gpu_ctx.irradiance_at(p, n)
    Eref = oracle.irradiance_grid_lookup(d, tab_ref, p, n)
    assert np.abs(E - Eref).max() < 1e-3
    assert np.abs(E - oracle.irradiance_grid_lookup(d, irr, p, n)).max() < 5e-6
    # position dependence is real: the same normal receives different light on opposite sides of the object
    a = oracle.irradiance_grid_lookup(d, tab_ref, np.float32([[1.4, 0.5, 0.5], [-0.4, 0.5, 0.5]]), np.float32([[0, 1, 0], [0, 1, 0]]))
    assert np.abs(a[0] - a[1]).max() > 1e-3
    with pytest.raises(RuntimeError, match="ngp_irradiance_at"):
        gpu_ctx.irradiance(n[:4])
    oracle.release(m)


@pytest.mark.gpu
def test_mesh_lit_by_probe_grid(gpu_ctx, oracle, native, scene_mod, scene_unit):
    """ShadeGridEnvMap (the fork's default render mode): a mesh spanning several probes' sectors, ambient light = blended E(N)/pi."""
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    mi = pkg("meshio")
    meshes = [(mi.torus(48, 24, R=1.0, r=0.12), (0.0, 0.0, 0.0)), (mi.icosphere(3), (0.9, 0.2, 0.1))]  # a ring around the object + a ball beside it
    for tris, c in meshes:
        gpu_ctx.add_mesh(tris, c)
    gx, gy, nt, nph, R = 4, 6, 32, 16, 0.95
    gpu_ctx.compute_envmap_grid(gx, gy, nt, nph, shell_radius=R)
    desc, org, env, irr = gpu_ctx.get_envmap_grid()
    w, hgt = 160, 90
    mat = scene_mod.orbit_camera(25.0, 35.0, 6.5)
    focal = scene_mod.focal_from_fov_x(w, 0.7)
    cam = native.make_camera(mat, w, hgt, focal)
    img = gpu_ctx.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, render_mode=native.RENDER_SHADE_GRID_ENVMAP, background=(0, 0, 0, 0)))
    img_sky = gpu_ctx.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, background=(0, 0, 0, 0)))
    h = oracle.mesh_scene(meshes)
    ocam = oracle.make_camera(mat, w, hgt, focal)
    fb, db = oracle.render_mesh(h, ocam, oracle.make_mesh_opts(irradiance=irr, grid=(gx, gy), probe_center=(0.5, 0.5, 0.5)))
    fb_sky, _ = oracle.render_mesh(h, ocam)
    sc = dict(scene_unit)
    lo, hi = oracle.mesh_scene_aabb(h)
    sc["render_aabb"] = (tuple(lo.tolist()), tuple(hi.tolist()))
    m = oracle.make_model(sc)
    ref, _, _ = oracle.render_nerf(m, ocam, oracle.make_opts(depth_test=True, render_mode=1), frame_buffer=fb, depth_buffer=db)  # (1: the NeRF pass of these modes does not linearise, src/testbed_geometry_training.cu:1862)
    on_mesh = fb_sky[..., :3].sum(-1) > 0
    assert on_mesh.mean() > 0.05 and (np.abs(fb - fb_sky)[on_mesh].max(-1) > 2e-5).mean() > 0.9  # (the shell probes see a mostly empty sky: little light)
    # the mesh pass alone (north_star: per-pixel irradiance L-inf < 1e-3): pixels the NeRF does not cover
    mesh_only = on_mesh & (np.abs(ref - fb).max(-1) == 0)
    assert mesh_only.sum() > 500 and np.abs(img - ref)[mesh_only].max() < 1e-3
    assert psnr(img[..., :3], ref[..., :3]) > 48.0 and (np.abs(img - ref).max(-1) < 1e-2).mean() > 0.995
    assert not np.array_equal(img, img_sky)
    # one probe for the whole scene (ShadeEnvMap) lights the ring differently from the grid
    gpu_ctx.compute_envmap(0, nt, nph)
    img_one = gpu_ctx.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, render_mode=native.RENDER_SHADE_ENVMAP, background=(0, 0, 0, 0)))
    assert (np.abs(img_one - img)[on_mesh].max(-1) > 2e-5).mean() > 0.5
    with pytest.raises(RuntimeError, match="ngp_compute_envmap_grid first"):
        gpu_ctx.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, render_mode=native.RENDER_SHADE_GRID_ENVMAP))
    oracle.release(m)
    oracle.mesh_scene_destroy(h)
    gpu_ctx.clear_meshes()


@pytest.mark.gpu
def test_mesh_lit_by_nerf_irradiance(gpu_ctx, oracle, native, scene_mod, scene_unit):
    """ShadeEnvMap: inserted meshes receive ambient light E(N)/pi from the NeRF-derived probe."""
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    mi = pkg("meshio")
    meshes = [(mi.icosphere(3), (0.6, 0.0, 0.1))]
    for tris, c in meshes:
        gpu_ctx.add_mesh(tris, c)
    env = gpu_ctx.compute_envmap(0, 64, 32)
    _, irr = gpu_ctx.get_envmap()
    w, hgt = 128, 72
    mat = scene_mod.orbit_camera(20.0, 15.0, 6.0)
    focal = scene_mod.focal_from_fov_x(w, 0.7)
    cam = native.make_camera(mat, w, hgt, focal)
    # mesh only (the NeRF pass is exercised elsewhere): unload nothing, but compare the mesh pass through a zero-NeRF view
    img_env = gpu_ctx.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, render_mode=native.RENDER_SHADE_ENVMAP, background=(0, 0, 0, 0)))
    img_sky = gpu_ctx.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, background=(0, 0, 0, 0)))
    h = oracle.mesh_scene(meshes)
    ocam = oracle.make_camera(mat, w, hgt, focal)
    fb_env, db = oracle.render_mesh(h, ocam, oracle.make_mesh_opts(irradiance=irr))
    fb_sky, _ = oracle.render_mesh(h, ocam)
    sc = dict(scene_unit)
    lo, hi = oracle.mesh_scene_aabb(h)
    sc["render_aabb"] = (tuple(lo.tolist()), tuple(hi.tolist()))
    m = oracle.make_model(sc)
    ref_env, _, _ = oracle.render_nerf(m, ocam, oracle.make_opts(depth_test=True, render_mode=1), frame_buffer=fb_env, depth_buffer=db)
    on_mesh = (fb_sky[..., :3].sum(-1) > 0)
    assert on_mesh.mean() > 0.03
    assert (np.abs(fb_env - fb_sky)[on_mesh].max(-1) > 1e-3).mean() > 0.9  # the probe light changes the shading
    assert psnr(img_env[..., :3], ref_env[..., :3]) > 48.0
    assert (np.abs(img_env - ref_env).max(-1) < 1e-2).mean() > 0.995
    assert not np.array_equal(img_env, img_sky)
    with pytest.raises(RuntimeError, match="ngp_compute_envmap first"):
        ctx2 = native.Context(0)
        ctx2.add_mesh(meshes[0][0])
        ctx2.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, render_mode=native.RENDER_SHADE_ENVMAP))
    oracle.release(m)
    oracle.mesh_scene_destroy(h)
    gpu_ctx.clear_meshes()
