"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for integer/bit work (occupancy bitfield, ray setup; fp16 hash-grid
features of the hash grid); floating point within the stated tolerances: PSNR vs oracle >= 50 dB (the 0.1 dB budget against ground
truth allows ~36 dB), per-pixel |d| < 1e-2 on radiance, fp16 network outputs within 4 ulp.
"""
import os

import numpy as np
import pytest

from conftest import pkg, psnr

pytestmark = pytest.mark.gpu


def assert_image_close(img, ref, min_psnr, tol=1e-2, frac=0.9995, hard=0.3):
    """PSNR bar plus a per-pixel bound. The fused kernel leaves aligned empty 4^3/16^3 occupancy blocks in one jump
    where the reference takes voxel steps; both land on the same lattice point except when a lattice point coincides
    with a block face to fp32 rounding (measured: ~6e-6 of the samples). Such a ray gains or loses one boundary sample,
    so a handful of pixels may differ by one sample's weight; everything else must agree to `tol`."""
    assert psnr(img[..., :3], ref[..., :3]) >= min_psnr
    d = np.abs(img - ref).max(-1)
    assert (d < tol).mean() >= frac, f"only {(d < tol).mean():.5f} of the pixels within {tol}"
    assert d.max() < hard


def _cam_pair(native, oracle, scene_mod, w, h, az=45.0, el=30.0, radius=4.03, spp=0, snap=True):
    mat = scene_mod.orbit_camera(az, el, radius)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    return native.make_camera(mat, w, h, focal, spp_index=spp, snap=snap), oracle.make_camera(mat, w, h, focal, spp_index=spp, snap=snap)


@pytest.mark.parametrize("which", ["unit", "big"])
def test_bitfield_bit_exact(which, gpu_ctx, oracle, scene_unit, scene_big):
    sc = scene_unit if which == "unit" else scene_big
    gpu_ctx.set_model(sc)
    bf, mean = gpu_ctx.density_bitfield()
    assert mean == sc["density_grid_mean"]
    assert np.array_equal(bf, sc["density_grid_bitfield"])
    # mip pyramid is non-trivial
    n = 128 ** 3 // 8
    assert bf[:n].any() and bf[n:2 * n].any() and bf[7 * n:].any()


def assert_encode_close(got, ref):
    """The shipped build sums a level's corners as tvec-era tiny-cuda-nn does -- result = fma((T)weight, val, result), one
    v_pk_fma_f16 per feature pair -- and the oracle's default ("fma") restates exactly that: compared as values (so that
    -0 == +0) and required to be EQUAL. (The older published sum is the legacy build: test_legacy_encode_variant.)"""
    got, ref = np.asarray(got, np.float32), np.asarray(ref, np.float32)
    assert got.shape == ref.shape
    bad = got != ref
    assert not bad.any(), f"{int(bad.sum())} of {bad.size} features differ, max |d| {np.abs(got - ref).max()}"


LEGACY_SCRIPT = """
import importlib, os, sys
import os

import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native = importlib.import_module(PKG + ".native"); syn = importlib.import_module(PKG + ".synthetic")
import oracle as O
orc = O.Oracle()
for kw in (dict(aabb_scale=1, seed=1234, log2_hashmap_size=15), dict(aabb_scale=4, seed=99, log2_hashmap_size=16, pls_rule="upstream")):
    sc = syn.make_scene(**kw)
    sc["grid_accumulate"] = "legacy"
    g = np.asarray(sc["density_grid"], np.float16).astype(np.float32)
    sc["density_grid_bitfield"], sc["density_grid_mean"] = orc.density_grid_to_bitfield(g, sc["max_cascade"])
    ctx = native.Context(0); ctx.set_model(sc); m = orc.make_model(sc)
    rng = np.random.default_rng(5)
    pos = rng.uniform(0, 1, (20000, 3)).astype(np.float32)
    pos[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.999999, 0.999999, 0.999999], [1e-7, 1e-7, 1e-7]]
    pos[8:16] = [[1.5, 0.2, 0.3], [-0.25, 0.5, 0.5], [0.3, 2.75, 0.1], [0.9, 0.9, -1.5], [3.0, 3.0, 3.0], [-0.01, -0.01, -0.01], [1.0001, 0.5, 0.5], [0.5, 0.5, 1.2]]
    pos[4000:4064] = rng.uniform(-2, 3, (64, 3)).astype(np.float32)
    assert np.array_equal(ctx.grid_encode(pos).astype(np.float32), orc.grid_encode(m, pos).astype(np.float32))
    for n in (1, 15, 17, 63, 65, 257):
        p = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        assert np.array_equal(ctx.grid_encode(p).astype(np.float32), orc.grid_encode(m, p).astype(np.float32))
    ctx.close()
print("LEGACY-OK")
"""


def test_legacy_encode_variant(native):
    """libngp_hip_legacy.so (-DNGP_TCNN_LEGACY_ENCODE): the corner sum tiny-cuda-nn published before its tvec refactor,
    result[f] += (T)(weight * (float)val[f]) -- bit for bit the oracle's "legacy" mode, in a process of its own. The
    reference pins no tiny-cuda-nn version, so both published sequences stay buildable and tested."""
    import subprocess
    import sys

    lib = pkg("build").build(legacy=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", LEGACY_SCRIPT.format(root=root)], env=dict(os.environ, NGP_HIP_LIBRARY=lib), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LEGACY-OK" in r.stdout, r.stderr[-1500:]


@pytest.mark.parametrize("which", ["unit", "big"])
def test_grid_encode(which, gpu_ctx, oracle, scene_unit, scene_big):
    sc = scene_unit if which == "unit" else scene_big
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    rng = np.random.default_rng(5)
    pos = rng.uniform(0, 1, (20000, 3)).astype(np.float32)
    pos[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.999999, 0.999999, 0.999999], [1e-7, 1e-7, 1e-7]]
    # positions outside [0, 1] leave the xor layout's range: the wave takes the tcnn-order table (wrap / hash of any coordinate)
    pos[8:16] = [[1.5, 0.2, 0.3], [-0.25, 0.5, 0.5], [0.3, 2.75, 0.1], [0.9, 0.9, -1.5], [3.0, 3.0, 3.0], [-0.01, -0.01, -0.01], [1.0001, 0.5, 0.5], [0.5, 0.5, 1.2]]
    pos[4000:4064] = rng.uniform(-2, 3, (64, 3)).astype(np.float32)  # one whole wave out of range, the rest mixed in
    got = gpu_ctx.grid_encode(pos)
    ref = oracle.grid_encode(m, pos)
    assert_encode_close(got, ref)
    oracle.release(m)


def test_dense_grid_encoding(gpu_ctx, oracle, native, scene_mod):
    """tcnn DenseGrid (log2_hashmap_size 31 on the ABI): no level is hashed, every level is a padded lattice."""
    from conftest import _with_bitfield, pkg

    cfg = scene_mod.base_network_config()
    cfg["encoding"] = dict(cfg["encoding"], otype="DenseGrid", base_resolution=4, per_level_scale=1.5)
    sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=77, log2_hashmap_size=31, cfg=cfg))
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    rng = np.random.default_rng(8)
    pos = rng.uniform(0, 1, (8192, 3)).astype(np.float32)
    pos[:4] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0.999999, 1e-7, 1.0]]
    pos[64:128] = rng.uniform(-1, 2, (64, 3)).astype(np.float32)
    assert_encode_close(gpu_ctx.grid_encode(pos), oracle.grid_encode(m, pos))
    oracle.release(m)
    img, depth, st, ref, db, ost = _render_both(gpu_ctx, oracle, native, scene_mod, sc, 128, 72, 35.0)
    assert st["n_rays_hit"] > 500 and abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
    assert_image_close(img, ref, 50.0)


@pytest.mark.parametrize("n_hidden", [1, 3])
def test_rgb_heads_with_one_and_three_hidden_layers(n_hidden, gpu_ctx, oracle, native, scene_mod):
    """configs/nerf/base_1layer.json / base_3layer.json: rgb_network.n_hidden_layers 1 and 3 (2 in base.json)."""
    from conftest import _with_bitfield

    cfg = scene_mod.base_network_config()
    cfg["rgb_network"] = dict(cfg["rgb_network"], n_hidden_layers=n_hidden)
    sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=31 + n_hidden, log2_hashmap_size=15, cfg=cfg))
    gpu_ctx.set_model(sc)
    assert gpu_ctx.get_model().n_hidden_rgb == n_hidden
    m = oracle.make_model(sc)
    rng = np.random.default_rng(3)
    pos = rng.uniform(0, 1, (4096 + 5, 3)).astype(np.float32)
    d = rng.normal(size=(pos.shape[0], 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dir01 = ((d + 1) * 0.5).astype(np.float32)
    got = gpu_ctx.network(pos, dir01).astype(np.float32)
    ref = oracle.network(m, pos, dir01).astype(np.float32)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)
    err = np.abs(got - ref)
    assert err.max() <= 8e-2 and (err <= 4 * ulp).mean() > 0.8 and np.median(err / ulp) <= 2.0  # one more layer of fp16 roundings than base.json at 3
    oracle.release(m)
    img, depth, st, refimg, db, ost = _render_both(gpu_ctx, oracle, native, scene_mod, sc, 160, 90, 70.0)
    assert st["n_rays_hit"] > 1000 and abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
    assert_image_close(img, refimg, 50.0)
    with pytest.raises(RuntimeError, match="irradiance probes are built for"):
        gpu_ctx.compute_envmap(n_theta=8, n_phi=4)
    with pytest.raises(RuntimeError, match="training is built for"):
        gpu_ctx.train(1, 1 << 14)


@pytest.mark.parametrize("hidden_density", [0, 1])
def test_heads_without_a_hidden_layer(hidden_density, gpu_ctx, oracle, native, scene_mod, tmp_path):
    """configs/nerf/linear.json (both heads a single matrix) and base_0layer.json (the rgb head is; a CutlassMLP, so its output is
    padded to 8 rows rather than 16): network outputs, the rendered image, the Normals gradient, the snapshot round trip."""
    from conftest import _with_bitfield

    cfg = scene_mod.linear_network_config(hidden_density)
    sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=51 + hidden_density, log2_hashmap_size=15, cfg=cfg))
    assert scene_mod.n_params(sc)[:2] == ((512, 256) if hidden_density == 0 else (3072, 256))
    gpu_ctx.set_model(sc)
    d = gpu_ctx.get_model()
    assert (d.n_hidden_density, d.n_hidden_rgb, d.mlp_alignment) == (hidden_density, 0, 8)
    m = oracle.make_model(sc)
    rng = np.random.default_rng(13)
    pos = rng.uniform(0, 1, (4096 + 21, 3)).astype(np.float32)
    dr = rng.normal(size=(pos.shape[0], 3)).astype(np.float32)
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    dir01 = ((dr + 1) * 0.5).astype(np.float32)
    got = gpu_ctx.network(pos, dir01).astype(np.float32)
    ref = oracle.network(m, pos, dir01).astype(np.float32)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)
    err = np.abs(got - ref)
    # one or two dot products of 32 terms: the MFMA's fp32 sum against the oracle's exact one, a last-place flip at most
    assert err.max() <= 3e-2 and (err <= ulp).mean() > 0.97 and (err == 0).mean() > 0.7
    g = gpu_ctx.density_gradient(pos[:2048])
    gr = oracle.density_gradient(m, pos[:2048])
    scale = np.abs(gr).max()
    assert np.abs(g - gr).max() <= 2e-3 * scale
    oracle.release(m)
    img, depth, st, refimg, db, ost = _render_both(gpu_ctx, oracle, native, scene_mod, sc, 160, 90, 70.0)
    assert st["n_rays_hit"] > 1000 and abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
    assert_image_close(img, refimg, 50.0)
    # ERenderMode::Normals through the linear density head (its input gradient is 128 W[0][:], unmasked)
    m = oracle.make_model(sc)
    cam, ocam = _cam_pair(native, oracle, scene_mod, 96, 54, az=70.0)
    nimg = gpu_ctx.render(cam, native.make_opts(render_mode=native.RENDER_NORMALS))
    nfb, _, _ = oracle.render_nerf(m, ocam, oracle.make_opts(render_mode=7))
    nref = oracle.tonemap(oracle.accumulate(nfb.reshape(-1, 4), np.zeros((96 * 54, 4), np.float32), 0)).reshape(54, 96, 4)
    oracle.release(m)
    assert_image_close(nimg, nref, 38.0, tol=5e-2, frac=0.99, hard=1.01)
    # density grid refresh through the linear density head (update_density_grid_nerf, src/testbed_nerf.cu:2772-2861)
    if hidden_density == 0:
        c = native.Context(0)  # (a context of its own: the refresh advances the context's grid rng)
        try:
            c.set_model(sc)
            mc = sc["max_cascade"]
            grid0 = c.density_grid(mc)
            c.update_density_grid(0.95, 40000, 20000, 1)
            gotg = c.density_grid(mc)
        finally:
            c.close()
        m = oracle.make_model(sc)
        refg, _ = oracle.update_density_grid(m, grid0, mc, oracle.grid_rng(), 0, 0.95, 40000, 20000)
        oracle.release(m)
        decayed = np.float32(0.95) * grid0
        touched = refg != decayed
        assert np.array_equal(gotg != decayed, touched) and touched.sum() > 20000
        rel = np.abs(gotg[touched] - refg[touched]) / np.maximum(refg[touched], 1e-12)
        assert np.median(rel) < 5e-3 and (rel < 0.1).mean() > 0.995
    path = str(tmp_path / "linear.ingp")
    gpu_ctx.save_snapshot_file(path)
    gpu_ctx.load_snapshot_file(path)
    d2 = gpu_ctx.get_model()
    assert (d2.n_hidden_density, d2.n_hidden_rgb, d2.mlp_alignment, d2.n_params) == (hidden_density, 0, 8, d.n_params)
    if hidden_density == 1:
        assert np.array_equal(gpu_ctx.network(pos, dir01).astype(np.float32), got)
    with pytest.raises(RuntimeError, match="irradiance probes are built for"):
        gpu_ctx.compute_envmap(n_theta=8, n_phi=4)
    with pytest.raises(RuntimeError, match="training is built for"):
        gpu_ctx.train(1, 1 << 14)


def test_grid_encode_ragged_sizes(gpu_ctx, oracle, scene_unit):
    gpu_ctx.set_model(scene_unit)
    m = oracle.make_model(scene_unit)
    rng = np.random.default_rng(6)
    for n in (1, 15, 16, 17, 63, 64, 65, 255, 257):
        pos = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        assert_encode_close(gpu_ctx.grid_encode(pos), oracle.grid_encode(m, pos))
    assert gpu_ctx.grid_encode(np.zeros((0, 3), np.float32)).shape == (0, 32)
    oracle.release(m)


@pytest.mark.parametrize("which", ["unit", "big"])
def test_network_outputs(which, gpu_ctx, oracle, scene_unit, scene_big):
    sc = scene_unit if which == "unit" else scene_big
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    rng = np.random.default_rng(11)
    n = 8192 + 37
    pos = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dir01 = ((d + 1) * 0.5).astype(np.float32)
    got = gpu_ctx.network(pos, dir01).astype(np.float32)
    ref = oracle.network(m, pos, dir01).astype(np.float32)
    assert np.isfinite(got).all()
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)  # fp16 spacing at |ref|
    err = np.abs(got - ref)
    # the encoded features are bit-identical (test_grid_encode); the MFMA's fp32 accumulation order differs from the oracle's
    # exact sum, so a hidden activation can flip by one fp16 ulp and move an output logit (|logit| ~ 5, fp16 spacing 2^-8
    # there) by a few spacings
    assert err.max() <= 3e-2, f"max err {err.max()} at ref {ref.ravel()[err.argmax()]}"
    assert (err == 0).mean() > 0.5 and (err <= ulp).mean() > 0.90
    oracle.release(m)


@pytest.mark.parametrize("which", ["unit", "big"])
def test_init_rays(which, gpu_ctx, oracle, native, scene_mod, scene_unit, scene_big):
    sc = scene_unit if which == "unit" else scene_big
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    for spp, snap in ((0, True), (3, False)):
        cam, ocam = _cam_pair(native, oracle, scene_mod, 80, 45, spp=spp, snap=snap)
        got = gpu_ctx.init_rays(cam)
        ref = oracle.init_rays(m, ocam)
        assert np.array_equal(got["alive"], ref["alive"])
        alive = ref["alive"] == 1
        assert alive.any() and (which == "big" or (~alive).any())  # the camera sits inside the aabb_scale-4 box
        assert np.array_equal(got["idx"][alive], ref["idx"][alive])
        assert np.array_equal(got["origin"], ref["origin"])
        assert np.array_equal(got["dir"][alive], ref["dir"][alive])
        if sc["cone_angle_constant"] == 0.0:
            assert np.array_equal(got["t"][alive], ref["t"][alive])  # no transcendental on the path: bit-exact
        else:
            # logf/expf of the exponential stepping differ by ulps between glibc and the device library; a ceilf in the
            # voxel skip can then land one step (1/256 of t) further for a few rays
            rel = np.abs(got["t"][alive] - ref["t"][alive]) / ref["t"][alive]
            assert (rel < 2e-5).mean() > 0.99 and rel.max() < 1e-2
    oracle.release(m)


def _render_both(gpu_ctx, oracle, native, scene_mod, sc, w, h, az, **cam_kw):
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    cam, ocam = _cam_pair(native, oracle, scene_mod, w, h, az=az, **cam_kw)
    img, depth = gpu_ctx.render(cam, native.make_opts(), want_depth=True)
    st = gpu_ctx.render_stats()
    fb, db, ost = oracle.render_nerf(m, ocam)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    oracle.release(m)
    return img, depth, st, ref, db, ost


@pytest.mark.parametrize("az", [45.0, 200.0])
def test_render_unit_scene(az, gpu_ctx, oracle, native, scene_mod, scene_unit):
    w, h = 256, 144
    img, depth, st, ref, db, ost = _render_both(gpu_ctx, oracle, native, scene_mod, scene_unit, w, h, az)
    assert st["n_rays"] == ((w + 7) // 8) * ((h + 7) // 8) * 64
    assert abs(int(st["n_rays_alive_after_init"]) - int(ost["n_rays_alive_after_init"])) <= 2
    assert abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 1e-4 * ost["n_samples"]
    assert st["n_samples"] / max(st["n_rays_hit"], 1) > 10
    assert_image_close(img, ref, 50.0)               # radiance: fp16 network + early-termination flips
    assert (np.abs(img[..., 3] - ref[..., 3]) < 5e-3).mean() > 0.9995
    both = (depth < 16000) & (db < 16000)
    assert both.sum() > 1000 and (np.not_equal(depth >= 16000, db >= 16000)).sum() <= 3
    assert np.median(np.abs(depth[both] - db[both])) < 1e-4


def test_render_big_scene_exponential_stepping(gpu_ctx, oracle, native, scene_mod, scene_big):
    w, h = 192, 108
    img, depth, st, ref, db, ost = _render_both(gpu_ctx, oracle, native, scene_mod, scene_big, w, h, 120.0)
    assert abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 0.002 * ost["n_rays_hit"] + 2
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 5e-3 * ost["n_samples"]
    assert psnr(img[..., :3], ref[..., :3]) >= 45.0  # logf/expf in the stepping differ by ulps between libm and the device
    assert np.abs(img - ref).mean() < 1e-3


def test_render_odd_resolution_and_subpixel_jitter(gpu_ctx, oracle, native, scene_mod, scene_unit):
    w, h = 101, 67  # not multiples of the 8x8 tile
    img, depth, st, ref, db, ost = _render_both(gpu_ctx, oracle, native, scene_mod, scene_unit, w, h, 300.0, spp=5, snap=False)
    assert abs(int(st["n_rays_alive_after_init"]) - int(ost["n_rays_alive_after_init"])) <= 2
    assert_image_close(img, ref, 50.0)


def test_render_camera_inside_and_empty_view(gpu_ctx, oracle, native, scene_mod, scene_unit):
    # camera inside the unit cube, and a camera looking away from it (no ray enters the AABB)
    w, h = 64, 36
    img, _, st, ref, _, ost = _render_both(gpu_ctx, oracle, native, scene_mod, scene_unit, w, h, 10.0, radius=0.9)
    assert abs(int(st["n_rays_alive_after_init"]) - int(ost["n_rays_alive_after_init"])) <= 2
    assert_image_close(img, ref, 48.0)
    gpu_ctx.set_model(scene_unit)
    mat = scene_mod.orbit_camera(45.0)
    mat[:, 2] *= -1.0  # flip the view direction
    mat[:, 0] *= -1.0
    cam = native.make_camera(mat, w, h, scene_mod.focal_from_fov_x(w, 0.6911))
    img = gpu_ctx.render(cam)
    st = gpu_ctx.render_stats()
    assert st["n_rays_hit"] == 0 and st["n_samples"] == 0
    assert np.array_equal(img[..., :3], np.zeros((h, w, 3), np.float32)) and np.all(img[..., 3] == 1.0)  # black, opaque background


def test_multi_spp_accumulation_and_srgb(gpu_ctx, oracle, native, scene_mod, scene_unit):
    w, h, spp = 96, 54, 4
    gpu_ctx.set_model(scene_unit)
    m = oracle.make_model(scene_unit)
    mat = scene_mod.orbit_camera(60.0)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    img = gpu_ctx.render(native.make_camera(mat, w, h, focal, snap=False), native.make_opts(spp=spp, to_srgb=True, background=(0.2, 0.4, 0.6, 1.0), exposure=0.5))
    acc = np.zeros((w * h, 4), np.float32)
    for s in range(spp):
        fb, _, _ = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal, spp_index=s, snap=False))
        acc = oracle.accumulate(fb.reshape(-1, 4), acc, s)
    ref = oracle.tonemap(acc, (0.2, 0.4, 0.6, 1.0), 0.5, True).reshape(h, w, 4)
    oracle.release(m)
    assert_image_close(img, ref, 48.0, tol=2e-2)


def test_tile_sharding_covers_frame(gpu_ctx, native, scene_mod, scene_unit):
    # rendering with 3 shards and summing equals the unsharded frame bit for bit (rays are independent)
    w, h = 120, 72
    gpu_ctx.set_model(scene_unit)
    cam = native.make_camera(scene_mod.orbit_camera(45.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
    bg0 = (0.0, 0.0, 0.0, 0.0)
    full = gpu_ctx.render(cam, native.make_opts(background=bg0))
    total = np.zeros_like(full)
    hits = 0
    for r in range(3):
        part = gpu_ctx.render(cam, native.make_opts(background=bg0, shard_index=r, shard_count=3))
        hits += gpu_ctx.render_stats()["n_rays_hit"]
        assert not (np.abs(total).sum(-1) > 0)[np.abs(part).sum(-1) > 0].any()  # shards are disjoint
        total += part
    assert np.array_equal(total, full)
    gpu_ctx.render(cam, native.make_opts(background=bg0))
    assert hits == gpu_ctx.render_stats()["n_rays_hit"]


def test_tile_packed_output_matches_image_layout(gpu_ctx, native, scene_mod, scene_unit):
    """packed_output (the layout the RCCL all_gather moves) holds exactly the pixels of the plain image."""
    import ctypes as C

    import torch
    from conftest import pkg

    par = pkg("parallel")
    w, h, world = 120, 67, 3  # not a multiple of the tile in y
    gpu_ctx.set_model(scene_unit)
    cam = native.make_camera(scene_mod.orbit_camera(45.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
    full, full_depth = gpu_ctx.render(cam, native.make_opts(), want_depth=True)
    dev = torch.device("cuda", 0)
    g = par.PackedFrameGather(w, h, world, dev)
    parts_rgba, parts_depth = [], []
    for r in range(world):
        rgba, depth = g.buffers()
        assert native.load_library().ngp_packed_tiles(w, h, r, world) <= g.n_slots
        gpu_ctx.render_device(cam, native.make_opts(shard_index=r, shard_count=world, packed_output=True), rgba.data_ptr(), depth.data_ptr(), None)
        gpu_ctx.render_stats()  # synchronises the context's stream
        parts_rgba.append(rgba.clone())
        parts_depth.append(depth.clone())
    img, dep = g.unpack(torch.cat(parts_rgba), torch.cat(parts_depth))
    assert np.array_equal(img.cpu().numpy(), full)
    assert np.array_equal(dep.cpu().numpy(), full_depth)


def test_render_is_deterministic(gpu_ctx, native, scene_mod, scene_unit):
    gpu_ctx.set_model(scene_unit)
    cam = native.make_camera(scene_mod.orbit_camera(45.0), 128, 72, scene_mod.focal_from_fov_x(128, 0.6911))
    a = gpu_ctx.render(cam)
    b = gpu_ctx.render(cam)
    assert np.array_equal(a, b)


def test_snapshot_roundtrip(tmp_path, gpu_ctx, native, scene_mod, scene_big):
    gpu_ctx.set_model(scene_big)
    cam = native.make_camera(scene_mod.orbit_camera(80.0), 96, 54, scene_mod.focal_from_fov_x(96, 0.6911))
    a = gpu_ctx.render(cam)
    for name in ("snap.msgpack", "snap.ingp"):
        p = str(tmp_path / name)
        gpu_ctx.save_snapshot_file(p)
        ctx2 = native.Context(0)
        ctx2.load_snapshot_file(p)
        d = ctx2.get_model()
        assert d.aabb_scale == 4 and abs(d.per_level_scale - scene_big["encoding"]["per_level_scale"]) < 1e-7
        assert np.array_equal(ctx2.render(cam), a)
        ctx2.close()


def test_errors_are_reported(gpu_ctx, native, scene_unit):
    bad = dict(scene_unit)
    bad["network"] = dict(scene_unit["network"], n_neurons=128)
    with pytest.raises(RuntimeError, match="unsupported network architecture"):
        gpu_ctx.set_model(bad)
    bad = dict(scene_unit)
    bad["params"] = scene_unit["params"][:-8]
    with pytest.raises(RuntimeError, match="parameter count mismatch"):
        gpu_ctx.set_model(bad)
    ctx2 = native.Context(0)
    with pytest.raises(RuntimeError, match="No network available"):
        ctx2.render(native.make_camera(np.eye(3, 4, dtype=np.float32), 8, 8, (8.0, 8.0)))
    with pytest.raises(RuntimeError, match="msgpack|snapshot"):
        ctx2.load_snapshot_bytes(b"\x81\xa1a\x01")
    ctx2.close()


def test_direct_output_matches_general_path_and_oracle(gpu_ctx, oracle, native, scene_mod, scene_unit):
    """1 spp without meshes takes the direct-output path (clear + accumulate + tonemap folded into the fused kernel);
    geometry mode without meshes renders the same frame through the general path (separate clear / tonemap passes)."""
    w, h = 136, 77
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    mat = scene_mod.orbit_camera(75.0)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    cam = native.make_camera(mat, w, h, focal)
    kw = dict(to_srgb=True, background=(0.3, 0.5, 0.1, 0.8), exposure=-0.25)
    direct, d_depth = gpu_ctx.render(cam, native.make_opts(**kw), want_depth=True)
    m = oracle.make_model(scene_unit)
    fb, db, _ = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal))
    oracle.release(m)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0), kw["background"], kw["exposure"], True).reshape(h, w, 4)
    assert_image_close(direct, ref, 48.0, tol=2e-2)
    assert np.median(np.abs(d_depth - db.reshape(h, w))) < 1e-4
    general, g_depth = gpu_ctx.render(cam, native.make_opts(testbed_mode=native.MODE_GEOMETRY, **kw), want_depth=True)  # no meshes: same frame, general path
    assert np.abs(general - direct).max() < 1e-6 and np.array_equal(g_depth, d_depth)


def test_frames_on_two_streams_overlap_safely(gpu_ctx, native, scene_mod, scene_unit):
    """bench.py keeps two frames in flight (streams i % 2, own output buffers): every frame must equal its serial render."""
    import torch

    w, h, n = 256, 144, 6
    gpu_ctx.set_model(scene_unit)
    cams = [native.make_camera(scene_mod.orbit_camera(30.0 + 50.0 * i), w, h, scene_mod.focal_from_fov_x(w, 0.6911)) for i in range(n)]
    serial = [gpu_ctx.render(c, native.make_opts(), want_depth=True) for c in cams]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [(torch.zeros((h, w, 4), device="cuda"), torch.zeros((h, w), device="cuda")) for _ in range(n)]
    for i, c in enumerate(cams):
        gpu_ctx.render_device(c, native.make_opts(), outs[i][0].data_ptr(), outs[i][1].data_ptr(), streams[i % 2].cuda_stream)
    torch.cuda.synchronize()
    hist = gpu_ctx.render_history(n)
    for i in range(n):
        assert np.array_equal(outs[i][0].cpu().numpy(), serial[i][0]) and np.array_equal(outs[i][1].cpu().numpy(), serial[i][1])
        assert hist[i]["n_samples"] > 0 and hist[i]["n_rays"] == ((w + 7) // 8) * ((h + 7) // 8) * 64


def test_slot_ring_survives_an_unsynchronised_loop(native, scene_mod, scene_unit):
    """Call k binds queue word / exit counter / accumulators k % 256. 600 small frames issued without a wait over 6 streams (what
    bench.py does beyond two GPUs, and what a video loop does) put calls k and k + 256 on different streams: the library orders the
    later one behind the earlier one's end, so every frame still equals its serial render and its own counters."""
    import torch

    torch.zeros(1, device="cuda")
    ctx = native.Context(0)
    ctx.set_model(scene_unit)
    w, h, n, n_cams = 64, 36, 600, 8
    cams = [native.make_camera(scene_mod.orbit_camera(45.0 * i), w, h, scene_mod.focal_from_fov_x(w, 0.6911)) for i in range(n_cams)]
    serial = []
    for c in cams:
        img, dep = ctx.render(c, native.make_opts(), want_depth=True)
        serial.append((img, dep, ctx.render_stats()))
    streams = [torch.cuda.Stream() for _ in range(6)]
    rgba = torch.zeros((n, h, w, 4), device="cuda")
    depth = torch.zeros((n, h, w), device="cuda")
    torch.cuda.synchronize()
    for i in range(n):
        ctx.render_device(cams[i % n_cams], native.make_opts(), rgba[i].data_ptr(), depth[i].data_ptr(), streams[i % 6].cuda_stream)
    torch.cuda.synchronize()
    hist = ctx.render_history(256)
    got, got_d = rgba.cpu().numpy(), depth.cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], serial[i % n_cams][0]) and np.array_equal(got_d[i], serial[i % n_cams][1]), f"frame {i}"
    for j, st in enumerate(hist):  # the last 256 calls: i = n - 256 + j
        ref = serial[(n - 256 + j) % n_cams][2]
        assert (st["n_samples"], st["n_rays_hit"], st["n_rays"]) == (ref["n_samples"], ref["n_rays_hit"], ref["n_rays"]), f"call {n - 256 + j}"
    ctx.close()


def test_density_grid_refresh_parity(native, oracle, scene_mod, scene_unit):
    """ngp_update_density_grid vs the oracle's update_density_grid_nerf: the same cells are sampled (pcg32 stream, cell
    hash and positions are bit-exact), their optical thickness agrees to the fp16 network tolerance, and the rebuilt
    bitfield differs only where a cell sits on the threshold. Rendering afterwards still matches the oracle."""
    ctx = native.Context(0)
    ctx.set_model(scene_unit)
    mc = scene_unit["max_cascade"]
    grid0 = ctx.density_grid(mc)
    assert np.array_equal(grid0, np.asarray(scene_unit["density_grid"], np.float16).astype(np.float32))
    ctx.update_density_grid(0.95, 60000, 30000, 2)
    got = ctx.density_grid(mc)
    m = oracle.make_model(scene_unit)
    rng = oracle.grid_rng()
    ref, step = oracle.update_density_grid(m, grid0, mc, rng, 0, 0.95, 60000, 30000)
    ref, step = oracle.update_density_grid(m, ref, mc, rng, step, 0.95, 60000, 30000)
    decayed = np.float32(0.95) * (np.float32(0.95) * grid0)
    assert np.array_equal(got != decayed, ref != decayed)  # exactly the same cells were touched
    touched = ref != decayed
    assert touched.sum() > 50000
    rel = np.abs(got[touched] - ref[touched]) / np.maximum(ref[touched], 1e-12)
    assert np.median(rel) < 2e-3 and (rel < 5e-2).mean() > 0.999  # exp(logit): one fp16 ulp of the logit is 0.1 %..1.6 %
    bf, mean = ctx.density_bitfield()
    obf, omean = oracle.density_grid_to_bitfield(ref, mc)
    assert abs(mean - omean) <= 2e-3 * omean
    assert np.unpackbits(bf ^ obf).sum() <= 1e-4 * 128 ** 3
    # the refreshed grid renders like the oracle with the oracle's refreshed grid
    sc = dict(scene_unit)
    sc["density_grid_bitfield"] = obf
    m2 = oracle.make_model(sc)
    w, h = 128, 72
    cam, ocam = _cam_pair(native, oracle, scene_mod, w, h, az=100.0)
    img = ctx.render(cam)
    fb, _, _ = oracle.render_nerf(m2, ocam)
    refimg = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    oracle.release(m)
    oracle.release(m2)
    assert_image_close(img, refimg, 45.0, tol=2e-2)
    # schedule form (0, 0): the first 256 updates sample 128^3 cells per cascade uniformly
    ctx.update_density_grid(0.95, 0, 0, 1)
    assert np.isfinite(ctx.density_grid(mc)).all()
    ctx.close()


@pytest.mark.parametrize("mode", ["AO", "POSITIONS", "DEPTH", "COST"])
def test_gbuffer_render_modes(mode, gpu_ctx, oracle, native, scene_mod, scene_unit):
    """ERenderMode::{AO, Positions, Depth}: composite_kernel_nerf replaces the per-sample colour (src/testbed_nerf.cu:689-702),
    shade_kernel_nerf leaves it un-linearised (:1393)."""
    w, h = 120, 68
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    cam, ocam = _cam_pair(native, oracle, scene_mod, w, h, az=140.0)
    rm = {"AO": native.RENDER_AO, "POSITIONS": native.RENDER_POSITIONS, "DEPTH": native.RENDER_DEPTH, "COST": native.RENDER_COST}[mode]
    img = gpu_ctx.render(cam, native.make_opts(render_mode=rm))
    shade = gpu_ctx.render(cam, native.make_opts())
    m = oracle.make_model(scene_unit)
    fb, _, _ = oracle.render_nerf(m, ocam, oracle.make_opts(render_mode=rm))
    oracle.release(m)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    assert_image_close(img, ref, 48.0 if mode != "COST" else 40.0, tol=2e-2)  # Cost: a ray's sample count may differ by one (1 / 128)
    if mode == "COST":
        assert set(np.unique(img[..., 3])) <= {0.0, 1.0} and (np.abs(img[..., 0] * 128 - np.round(img[..., 0] * 128)) < 1e-4).all()
    hit = shade[..., :3].sum(-1) > 0
    assert hit.mean() > 0.2 and np.abs(img[hit][:, :3] - shade[hit][:, :3]).max() > 0.05  # it is a different image from Shade
    if mode != "POSITIONS":
        assert np.abs(img[..., 0] - img[..., 1]).max() < 1e-6 and np.abs(img[..., 0] - img[..., 2]).max() < 1e-6  # grey
    with pytest.raises(RuntimeError, match="NeRF mode"):
        gpu_ctx.render(cam, native.make_opts(render_mode=rm, testbed_mode=native.MODE_GEOMETRY))


@pytest.mark.parametrize("spp", [1, 3])
def test_srgb_color_space_accumulation(spp, gpu_ctx, oracle, native, scene_mod, scene_unit):
    """EColorSpace::SRGB (run.py --nerf_compatibility): samples are averaged as sRGB values and blended with the sRGB
    background, then linearised (src/render_buffer.cu:241-248, 324-340, 537-541). spp 1 takes the direct-output path."""
    w, h = 96, 54
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    m = oracle.make_model(scene_unit)
    mat = scene_mod.orbit_camera(250.0)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    bg = (0.25, 0.5, 0.75, 1.0)
    img = gpu_ctx.render(native.make_camera(mat, w, h, focal, snap=False), native.make_opts(spp=spp, color_space=1, background=bg, exposure=0.3))
    lin = gpu_ctx.render(native.make_camera(mat, w, h, focal, snap=False), native.make_opts(spp=spp, color_space=0, background=bg, exposure=0.3))
    acc = np.zeros((w * h, 4), np.float32)
    for s in range(spp):
        fb, _, _ = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal, spp_index=s, snap=False))
        acc = oracle.accumulate(fb.reshape(-1, 4), acc, s, color_space=1)
    ref = oracle.tonemap(acc, bg, 0.3, False, color_space=1).reshape(h, w, 4)
    oracle.release(m)
    assert_image_close(img, ref, 48.0, tol=2e-2)
    assert np.abs(img - lin).max() > 1e-2  # blending with the background in sRGB is visibly different from linear


def test_cone_angle_override(native, oracle, scene_mod, scene_big):
    """testbed.nerf.cone_angle_constant = 0 (run.py:167): the fox-shaped model marched with the fixed step."""
    ctx = native.Context(0)
    ctx.set_model(scene_big)
    w, h = 128, 72
    cam, ocam = _cam_pair(native, oracle, scene_mod, w, h, az=120.0)
    before = ctx.render(cam)
    ctx.set_cone_angle_constant(0.0)
    img = ctx.render(cam)
    st = ctx.render_stats()
    sc = dict(scene_big)
    sc["cone_angle_constant"] = 0.0
    m = oracle.make_model(sc)
    fb, _, ost = oracle.render_nerf(m, ocam)
    oracle.release(m)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 1e-3 * ost["n_samples"]
    assert_image_close(img, ref, 48.0, tol=2e-2)
    assert st["n_samples"] > 0 and np.abs(img - before).max() > 1e-3
    ctx.close()


@pytest.mark.parametrize("lens", ["opencv", "fisheye", "latlong", "equirectangular", "ftheta"])
def test_lens_models(lens, gpu_ctx, oracle, native, scene_mod, scene_unit):
    """uv_to_ray's lenses (common_device.cuh:441-462): OpenCV / OpenCV-fisheye undistortion (Newton), lat-long, equirectangular,
    F-theta (angle = polynomial of the pixel radius; pixels beyond 90 degrees have no ray)."""
    w, h = 112, 64
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    kw = {"opencv": dict(lens_mode=native.LENS_OPENCV, lens_params=(0.0578421, -0.0805099, -0.000980296, 0.00015575)),
          "fisheye": dict(lens_mode=native.LENS_OPENCV_FISHEYE, lens_params=(0.05, -0.01, 0.003, -0.0005)),
          "latlong": dict(lens_mode=native.LENS_LATLONG), "equirectangular": dict(lens_mode=native.LENS_EQUIRECTANGULAR),
          # r(pixels) -> angle: 0.8 rad at the intrinsics' half width (res 1000 x 600), slightly non-linear
          "ftheta": dict(lens_mode=native.LENS_FTHETA, lens_params=(0.0, 1.5e-3, 2.0e-7, -1.0e-10, 0.0, 1000.0, 600.0))}[lens]
    mat = scene_mod.orbit_camera(20.0, 20.0, 2.2 if lens in ("opencv", "fisheye") else 0.9)
    focal = scene_mod.focal_from_fov_x(w, 1.2)
    img = gpu_ctx.render(native.make_camera(mat, w, h, focal, **kw))
    plain = gpu_ctx.render(native.make_camera(mat, w, h, focal))
    m = oracle.make_model(scene_unit)
    fb, _, ost = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal, **kw))
    oracle.release(m)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    assert ost["n_rays_hit"] > 500
    assert_image_close(img, ref, 45.0, tol=2e-2)
    assert np.abs(img - plain).max() > 0.05  # the lens changes the picture
    with pytest.raises(RuntimeError, match="unknown lens mode"):
        gpu_ctx.render(native.make_camera(mat, w, h, focal, lens_mode=7))
    if lens == "ftheta":  # a polynomial that passes 90 degrees inside the image: those pixels have no ray and stay background
        wide = dict(lens_mode=native.LENS_FTHETA, lens_params=(0.0, 4.0e-3, 0.0, 0.0, 0.0, 1000.0, 600.0))
        img_w = gpu_ctx.render(native.make_camera(mat, w, h, focal, **wide), native.make_opts(background=(0, 0, 0, 0)))
        m = oracle.make_model(scene_unit)
        fb, _, _ = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal, **wide))
        oracle.release(m)
        assert_image_close(img_w, fb.reshape(h, w, 4), 45.0, tol=2e-2)
        assert (img_w[:, :8, 3] == 0).all() and (img_w[h // 2, w // 2 - 4:w // 2 + 4, 3] > 0).any()


@pytest.mark.parametrize("rotated", [False, True])
def test_render_aabb_crop(rotated, native, oracle, scene_mod, scene_unit):
    """testbed.render_aabb / render_aabb_to_local: the crop box of the render (rays start at its entry, stop at its exit)."""
    ctx = native.Context(0)
    ctx.set_model(scene_unit)
    w, h = 128, 72
    cam, ocam = _cam_pair(native, oracle, scene_mod, w, h, az=60.0)
    full = ctx.render(cam)
    lo, hi = (0.2, 0.05, 0.3), (0.62, 0.8, 0.95)
    a = 0.5
    rot = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], np.float32) if rotated else None
    ctx.set_render_aabb(lo, hi, rot)
    img = ctx.render(cam)
    st = ctx.render_stats()
    sc = dict(scene_unit)
    sc["render_aabb"] = (lo, hi)
    if rotated:
        sc["render_aabb_to_local"] = rot
    m = oracle.make_model(sc)
    fb, _, ost = oracle.render_nerf(m, ocam)
    oracle.release(m)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    assert ost["n_rays_hit"] > 300 and abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
    assert_image_close(img, ref, 48.0, tol=2e-2)
    assert np.abs(img - full).max() > 0.05 and st["n_samples"] > 0
    with pytest.raises(RuntimeError, match="min must not exceed max"):
        ctx.set_render_aabb((0.5, 0.5, 0.5), (0.4, 0.6, 0.6))
    ctx.close()


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (8, 8), (9, 17), (1031, 2)])
def test_tiny_and_thin_resolutions(w, h, gpu_ctx, oracle, native, scene_mod, scene_unit):
    """Frames smaller than a tile, single rows, widths that leave ragged tiles: every pixel is written exactly like the oracle's."""
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    mat = scene_mod.orbit_camera(45.0)
    focal = scene_mod.focal_from_fov_x(max(w, 16), 0.6911)
    img, depth = gpu_ctx.render(native.make_camera(mat, w, h, focal), native.make_opts(background=(0.1, 0.2, 0.3, 1.0)), want_depth=True)
    st = gpu_ctx.render_stats()
    m = oracle.make_model(scene_unit)
    fb, db, ost = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal))
    oracle.release(m)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0), (0.1, 0.2, 0.3, 1.0)).reshape(h, w, 4)
    assert img.shape == (h, w, 4) and np.isfinite(img).all()
    assert abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 1
    assert np.abs(img - ref).max() < 2e-2
    assert np.array_equal(depth >= 16000, db.reshape(h, w) >= 16000)


def test_depth_of_field(gpu_ctx, oracle, native, scene_mod, scene_unit):
    """uv_to_ray's thin-lens step (common_device.cuh:471-477): the ray origin moves on the aperture disk (2-d Sobol point per
    pixel and sample), the direction keeps the focus point; several samples accumulate to the blur."""
    w, h, spp = 96, 54, 4
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    mat = scene_mod.orbit_camera(310.0)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    kw = dict(aperture_size=0.03, focus_z=1.1)
    img = gpu_ctx.render(native.make_camera(mat, w, h, focal, snap=False, **kw), native.make_opts(spp=spp))
    sharp = gpu_ctx.render(native.make_camera(mat, w, h, focal, snap=False), native.make_opts(spp=spp))
    m = oracle.make_model(scene_unit)
    acc = np.zeros((w * h, 4), np.float32)
    for s in range(spp):
        fb, _, _ = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal, spp_index=s, snap=False, **kw))
        acc = oracle.accumulate(fb.reshape(-1, 4), acc, s)
    oracle.release(m)
    ref = oracle.tonemap(acc).reshape(h, w, 4)
    assert_image_close(img, ref, 45.0, tol=2e-2)
    assert np.abs(img - sharp).max() > 0.05
    one = gpu_ctx.render(native.make_camera(mat, w, h, focal, aperture_size=0.03, focus_z=-1.0))  # plane_z < 0 switches the aperture off
    assert np.array_equal(one, gpu_ctx.render(native.make_camera(mat, w, h, focal)))


@pytest.mark.parametrize("spp", [1, 3])
def test_environment_map_background(spp, native, oracle, scene_mod, scene_unit):
    """m_envmap behind the NeRF: every valid ray starts from read_envmap(dir) (envmap.cuh:24-50, src/testbed_nerf.cu:1526-1528) --
    through the direct-output path (1 spp) and the general one (several samples per pixel)."""
    ctx = native.Context(0)
    ctx.set_model(scene_unit)
    rng = np.random.default_rng(12)
    eh, ew = 16, 32
    env = np.zeros((eh, ew, 4), np.float32)
    env[..., :3] = rng.uniform(0, 1, (eh, ew, 3))
    env[..., 3] = rng.uniform(0.5, 1.0, (eh, ew))
    env[..., :3] *= env[..., 3:4]  # premultiplied, like every radiance buffer of the renderer
    w, h = 144, 80
    mat = scene_mod.orbit_camera(250.0, 10.0, 2.6)  # close: the picture has object, grazing rays and plain background
    focal = scene_mod.focal_from_fov_x(w, 1.1)
    kw = dict(background=(0.1, 0.2, 0.3, 1.0), exposure=0.25, to_srgb=(spp == 1))
    plain = ctx.render(native.make_camera(mat, w, h, focal, snap=(spp == 1)), native.make_opts(spp=spp, **kw))
    ctx.set_envmap(env)
    img = ctx.render(native.make_camera(mat, w, h, focal, snap=(spp == 1)), native.make_opts(spp=spp, **kw))
    m = oracle.make_model(scene_unit)
    acc = np.zeros((w * h, 4), np.float32)
    for s in range(spp):
        fb, _, ost = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal, spp_index=s, snap=(spp == 1)), oracle.make_opts(envmap=env))
        acc = oracle.accumulate(fb.reshape(-1, 4), acc, s)
    oracle.release(m)
    ref = oracle.tonemap(acc, kw["background"], kw["exposure"], kw["to_srgb"]).reshape(h, w, 4)
    assert ost["n_rays_hit"] > 1000 and ost["n_rays_hit"] < 0.9 * w * h
    assert_image_close(img, ref, 48.0, tol=2e-2)
    assert np.abs(img - plain).max() > 0.1  # the map shows
    # where no ray is composited the pixel is the map alone: agree to rounding (atan2 / acos of the device vs libm)
    miss = np.abs(fb.reshape(h, w, 4) - 0).sum(-1) > 0
    assert np.abs(img - ref)[miss].max() < 0.3
    ctx.set_envmap(None)
    assert np.array_equal(ctx.render(native.make_camera(mat, w, h, focal, snap=(spp == 1)), native.make_opts(spp=spp, **kw)), plain)
    ctx.set_envmap(env)
    ctx.add_mesh(pkg("meshio").icosphere(1))
    with pytest.raises(RuntimeError, match="environment map applies to NeRF mode"):
        ctx.render(native.make_camera(mat, w, h, focal), native.make_opts(testbed_mode=native.MODE_GEOMETRY))
    ctx.close()


@pytest.mark.parametrize("shutter", [(0.0, 0.0, 0.0, 1.0), (0.1, 0.3, 0.5, 0.2)])
def test_moving_camera_and_rolling_shutter(shutter, native, oracle, scene_mod, scene_unit):
    """camera_matrix0 -> camera_matrix1 with a rolling shutter (get_xform_given_rolling_shutter, common_device.cuh:651-659;
    src/testbed_nerf.cu:1468): every pixel is rendered by the camera of its own time; depth is measured along camera_matrix1."""
    ctx = native.Context(0)
    ctx.set_model(scene_unit)
    w, h = 128, 72
    m0, m1 = scene_mod.orbit_camera(40.0, 25.0, 3.6), scene_mod.orbit_camera(58.0, 32.0, 3.9)  # turns and moves within the frame
    focal = scene_mod.focal_from_fov_x(w, 0.8)
    kw = dict(rolling_shutter=shutter)
    m = oracle.make_model(scene_unit)
    for spp in (0, 5):
        cam = native.make_camera(m0, w, h, focal, spp_index=spp, snap=False, matrix1_3x4=m1, **kw)
        ocam = oracle.make_camera(m0, w, h, focal, spp_index=spp, snap=False, matrix1_4x3=m1, **kw)
        got = ctx.init_rays(cam)
        ref = oracle.init_rays(m, ocam)
        assert np.array_equal(got["alive"], ref["alive"])
        alive = ref["alive"] == 1
        # quaternion slerp: sinf / acosf of the device against libm differ by an ulp
        assert np.abs(got["origin"] - ref["origin"]).max() < 2e-6 and np.abs(got["dir"][alive] - ref["dir"][alive]).max() < 2e-6
        img, depth = ctx.render(cam, native.make_opts(), want_depth=True)
        fb, db, ost = oracle.render_nerf(m, ocam)
        ref_img = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
        assert ost["n_rays_hit"] > 1500
        assert_image_close(img, ref_img, 45.0, tol=2e-2)
        both = (depth < 16000) & (db < 16000)
        assert np.median(np.abs(depth[both] - db[both])) < 1e-4
    still = ctx.render(native.make_camera(m0, w, h, focal, spp_index=5, snap=False))
    assert np.abs(img - still).max() > 0.05
    # camera1 == camera0 is the static frame, bit for bit (no quaternion round trip)
    same = ctx.render(native.make_camera(m0, w, h, focal, spp_index=5, snap=False, matrix1_3x4=m0, **kw))
    assert np.array_equal(same, still)
    oracle.release(m)
    ctx.add_mesh(pkg("meshio").icosphere(1))
    with pytest.raises(RuntimeError, match="moving camera"):
        ctx.render(native.make_camera(m0, w, h, focal, matrix1_3x4=m1), native.make_opts(testbed_mode=native.MODE_GEOMETRY))
    ctx.close()


@pytest.mark.parametrize("which", ["unit", "big"])
def test_density_gradient(which, gpu_ctx, oracle, scene_unit, scene_big):
    """the stage behind ERenderMode::Normals (tcnn input_gradient of the density logit): same fp16 backward pass and dy_dx as the
    oracle; the MFMA's fp32 accumulation order can move one fp16 dL/dy by an ulp and the four lanes of a sample add their level
    pairs in a different order"""
    sc = scene_unit if which == "unit" else scene_big
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    rng = np.random.default_rng(21)
    pos = rng.uniform(0, 1, (4096 + 7, 3)).astype(np.float32)
    got = gpu_ctx.density_gradient(pos)
    ref = oracle.density_gradient(m, pos)
    oracle.release(m)
    assert np.isfinite(got).all()
    nr = np.linalg.norm(ref, axis=1)
    ok = nr > 1e-3 * np.median(nr)
    cos = (got[ok] * ref[ok]).sum(1) / (np.linalg.norm(got[ok], axis=1) * nr[ok])
    assert ok.mean() > 0.95 and np.median(cos) > 0.99999 and cos.min() > 0.995, (np.median(cos), cos.min())
    rel = np.linalg.norm(got[ok] - ref[ok], axis=1) / nr[ok]
    assert np.median(rel) < 2e-3 and rel.max() < 0.1, (np.median(rel), rel.max())
    assert gpu_ctx.density_gradient(np.zeros((0, 3), np.float32)).shape == (0, 3)


def test_render_mode_normals(gpu_ctx, oracle, native, scene_mod, scene_unit):
    """ERenderMode::Normals (composite_kernel_nerf :688-693, shade_kernel_nerf :1379-1381) against the oracle"""
    w, h = 96, 54
    gpu_ctx.set_model(scene_unit)
    m = oracle.make_model(scene_unit)
    cam, ocam = _cam_pair(native, oracle, scene_mod, w, h, az=70.0)
    img = gpu_ctx.render(cam, native.make_opts(render_mode=native.RENDER_NORMALS))
    st = gpu_ctx.render_stats()
    fb, db, ost = oracle.render_nerf(m, ocam, oracle.make_opts(render_mode=7))
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    oracle.release(m)
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 1e-4 * ost["n_samples"] + 2
    assert np.isfinite(img).all()
    hit = ref[..., 3] > 0.5
    assert hit.sum() > 500
    # the pixel is (0.5 n + 0.5) alpha: undo it and compare directions
    n_got = img[..., :3][hit] / img[..., 3:][hit] * 2 - 1
    n_ref = ref[..., :3][hit] / ref[..., 3:][hit] * 2 - 1
    cos = (n_got * n_ref).sum(1) / (np.linalg.norm(n_got, axis=1) * np.linalg.norm(n_ref, axis=1))
    assert np.median(cos) > 0.9999 and (cos > 0.99).mean() > 0.995, (np.median(cos), (cos > 0.99).mean())
    # (a sample whose gradient nearly vanishes has a direction that one fp16 ulp of dL/dy can turn: such samples move a pixel visibly)
    # and where a ray's normals nearly cancel, the renormalisation in the shade step amplifies it: measured 41.6 dB, 99.4 % within 0.05)
    assert_image_close(img, ref, 38.0, tol=5e-2, frac=0.99, hard=1.01)
