"""GPU parity of the second architecture: configs/nerf/frequency.json -- Frequency encodings of position and
direction, MLPs 256 (or 128) wide -- through the C ABI against the CPU oracle on the same seeded inputs.

Bars: the encoding within one fp16 spacing (two correctly rounded sines of the same fp32 argument can differ in the last fp32
bit, which flips an fp16 rounding in ~1e-4 of the features); network outputs within a few fp16 spacings of the logits (eight
layers of fp16 activations, fp32 MFMA accumulation against the oracle's exact sums); images: identical ray statistics -- the
march takes the reference's one-voxel steps, so the sample SETS are the oracle's -- and per-pixel |d| < 1e-2, PSNR >= 50 dB.
"""
import numpy as np
import pytest

from conftest import _with_bitfield, pkg, psnr

pytestmark = pytest.mark.gpu


def _scene(oracle, scene_mod, seed=3, aabb_scale=1, **kw):
    cfg = scene_mod.frequency_network_config(**kw)
    return _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=aabb_scale, seed=seed, cfg=cfg))


@pytest.fixture(scope="module")
def scene_freq(oracle, scene_mod):
    return _scene(oracle, scene_mod)


@pytest.fixture(scope="module")
def ctx(native, gpu_ctx):  # (gpu_ctx first: torch has to initialise its HIP runtime before the library does, conftest.py)
    c = native.Context(0)
    yield c
    c.close()


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return pos, ((d + 1) * 0.5).astype(np.float32)


def test_frequency_encoding(ctx, oracle, scene_freq):
    ctx.set_model(scene_freq)
    pos, _ = _rays(8192 + 5, 1)
    pos[:4] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0.999999, 1e-7, 0.25]]
    got = ctx.grid_encode(pos).astype(np.float32)
    ref = oracle.frequency_encode(pos, 16).astype(np.float32)
    assert got.shape == (pos.shape[0], 96) and ref.shape == got.shape
    err = np.abs(got - ref)
    assert err.max() <= 2.0 ** -10, err.max()  # one fp16 spacing below 1
    assert (err == 0).mean() > 0.999
    assert ctx.grid_encode(np.zeros((0, 3), np.float32)).shape == (0, 96)


@pytest.mark.parametrize("n", [1, 255, 256, 257, 20000])
def test_network_outputs(n, ctx, oracle, scene_freq):
    ctx.set_model(scene_freq)
    m = oracle.make_model(scene_freq)
    pos, dir01 = _rays(n, 11 + n)
    got = ctx.network(pos, dir01).astype(np.float32)
    ref = oracle.network(m, pos, dir01).astype(np.float32)
    oracle.release(m)
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)
    assert err.max() <= 6e-2, f"max err {err.max()} at ref {ref.ravel()[err.argmax()]}"
    if n >= 256:
        assert (err <= 2 * ulp).mean() > 0.9


@pytest.mark.parametrize("variant", ["w128_sh4_fullyfused", "w128_freq_cutlass_3_2", "w256_1_1"])
def test_network_variants(variant, ctx, oracle, scene_mod):
    """other shapes of the same family: 128 neurons, SphericalHarmonics directions with FullyFusedMLP alignment (16), other depths"""
    if variant == "w128_sh4_fullyfused":
        cfg = scene_mod.frequency_network_config(n_neurons=128, n_hidden_density=2, n_hidden_rgb=2)
        cfg["dir_encoding"] = {"otype": "SphericalHarmonics", "degree": 4}
        cfg["network"]["otype"] = cfg["rgb_network"]["otype"] = "FullyFusedMLP"
        cfg["encoding"]["n_frequencies"] = 10  # 60 inputs -> padded to 64 with ones
    elif variant == "w128_freq_cutlass_3_2":
        cfg = scene_mod.frequency_network_config(n_neurons=128, n_hidden_density=3, n_hidden_rgb=2)
        cfg["encoding"]["n_frequencies"] = 9   # 54 -> 56: the padding ones meet zero-padded weight columns up to the K block (64)
        cfg["dir_encoding"]["n_frequencies"] = 3  # 18 -> 24
    else:
        cfg = scene_mod.frequency_network_config(n_neurons=256, n_hidden_density=1, n_hidden_rgb=1)
    sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=21, cfg=cfg))
    ctx.set_model(sc)
    m = oracle.make_model(sc)
    pos, dir01 = _rays(3000, 5)
    got = ctx.network(pos, dir01).astype(np.float32)
    ref = oracle.network(m, pos, dir01).astype(np.float32)
    oracle.release(m)
    err = np.abs(got - ref)
    assert np.isfinite(got).all() and err.max() <= 6e-2, err.max()


@pytest.mark.parametrize("aabb_scale", [1, 4])
def test_render_matches_oracle(aabb_scale, ctx, oracle, native, scene_mod, scene_freq):
    sc = scene_freq if aabb_scale == 1 else _scene(oracle, scene_mod, seed=8, aabb_scale=4)
    ctx.set_model(sc)
    m = oracle.make_model(sc)
    w, h = (96, 54) if aabb_scale == 1 else (64, 36)
    mat = scene_mod.orbit_camera(40.0, 25.0, 3.2 if aabb_scale == 1 else 5.0)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    img, depth = ctx.render(native.make_camera(mat, w, h, focal), native.make_opts(), want_depth=True)
    st = ctx.render_stats()
    fb, db, ost = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal))
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    oracle.release(m)
    # one-voxel marching: the samples are the oracle's unless a ray's termination flips on a network rounding
    assert st["n_rays_alive_after_init"] == ost["n_rays_alive_after_init"]
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= max(4, ost["n_samples"] // 2000)
    assert abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 2
    assert ost["n_samples"] > 10000 and img[..., 3].max() > 0.9
    assert psnr(img[..., :3], ref[..., :3]) >= 50.0
    d = np.abs(img - ref).max(-1)
    tol = 1e-2 if aabb_scale == 1 else 3e-2  # (exponential stepping: logf / expf differ by ulps between libm and the device, as for base.json's big scenes)
    assert (d < tol).mean() >= 0.999 and d.max() < 0.1, (float((d < tol).mean()), float(d.max()))
    db = db.reshape(h, w)
    both = (depth < 16000) & (db < 16000)
    assert both.sum() > 100 and np.median(np.abs(depth[both] - db[both])) < 1e-3


def test_snapshot_round_trip_and_modes(ctx, native, scene_mod, scene_freq, tmp_path):
    ctx.set_model(scene_freq)
    w, h = 64, 36
    cam = native.make_camera(scene_mod.orbit_camera(10.0, 20.0, 3.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
    img0 = ctx.render(cam, native.make_opts(to_srgb=True))
    path = str(tmp_path / "freq.ingp")
    ctx.save_snapshot_file(path)
    other = native.Context(0)
    try:
        other.load_snapshot_file(path)
        d = other.get_model()
        assert (d.pos_encoding, d.pos_n_frequencies, d.dir_encoding, d.dir_n_frequencies, d.mlp_alignment) == (1, 16, 1, 4, 8)
        assert (d.n_neurons, d.n_hidden_density, d.n_hidden_rgb) == (256, 7, 1) and d.n_params == 421888 + 12288
        img1 = other.render(cam, native.make_opts(to_srgb=True))
        assert np.array_equal(img0, img1)
        # the G-buffer modes and several samples per pixel go through the same kernel
        for mode in (native.RENDER_AO, native.RENDER_POSITIONS, native.RENDER_DEPTH, native.RENDER_COST):
            g = other.render(cam, native.make_opts(render_mode=mode))
            assert np.isfinite(g).all() and g[..., 3].max() > 0.9
        img4 = other.render(cam, native.make_opts(to_srgb=True, spp=4))
        assert psnr(img4[..., :3], img0[..., :3]) > 25.0
        with pytest.raises(RuntimeError, match="inference only"):
            other.train(1)
    finally:
        other.close()


def test_identity_encodings(ctx, oracle, native, scene_mod, tmp_path):
    """configs/nerf/none.json: tcnn Identity encodings -- the warped position and the direction themselves, padded with ones to the
    CutlassMLPs' alignment (3 -> 8) -- in front of frequency.json's MLPs: encoding bit for bit, network, a frame, the snapshot round trip."""
    sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=17, cfg=scene_mod.identity_network_config()))
    assert scene_mod.network_shapes(sc) == (8, 8, 24, 8) and scene_mod.n_params(sc) == (8 * 256 + 6 * 256 * 256 + 16 * 256, 24 * 256 + 8 * 256, 0)
    ctx.set_model(sc)
    d = ctx.get_model()
    assert (d.pos_encoding, d.dir_encoding, d.mlp_alignment) == (2, 2, 8)
    pos, dir01 = _rays(5000, 3)
    enc = ctx.grid_encode(pos)
    assert enc.shape == (5000, 8) and np.array_equal(enc[:, :3], pos.astype(np.float16)) and (enc[:, 3:] == 1).all()
    m = oracle.make_model(sc)
    got = ctx.network(pos, dir01).astype(np.float32)
    ref = oracle.network(m, pos, dir01).astype(np.float32)
    err = np.abs(got - ref)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)
    assert np.isfinite(got).all() and err.max() <= 6e-2 and (err <= 2 * ulp).mean() > 0.9, (err.max(), (err <= 2 * ulp).mean())
    w, h = 96, 54
    mat = scene_mod.orbit_camera(40.0, 25.0, 3.2)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    img = ctx.render(native.make_camera(mat, w, h, focal), native.make_opts())
    st = ctx.render_stats()
    fb, db, ost = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal))
    ref_img = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    oracle.release(m)
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= max(4, ost["n_samples"] // 2000) and ost["n_samples"] > 10000
    assert psnr(img[..., :3], ref_img[..., :3]) >= 50.0
    path = str(tmp_path / "none.ingp")
    ctx.save_snapshot_file(path)
    other = native.Context(0)
    try:
        other.load_snapshot_file(path)
        d2 = other.get_model()
        assert (d2.pos_encoding, d2.dir_encoding, d2.mlp_alignment, d2.n_params) == (2, 2, 8, d.n_params)
        assert np.array_equal(other.network(pos, dir01).astype(np.float32), got)
    finally:
        other.close()


@pytest.mark.parametrize("which", ["frequency", "identity", "w128_3_sh"])
def test_density_gradient_and_normals(which, ctx, oracle, native, scene_mod, scene_freq):
    """ERenderMode::Normals for this architecture (tcnn input_gradient(stream, 3, ...), src/testbed_nerf.cu:2106-2107, works for any network): the
    density network's backward pass on the transposed layers + the encoding's derivative, against the oracle -- the gradient at explicit positions
    (ngp_density_gradient) and a Normals frame."""
    if which == "frequency":
        sc = scene_freq
    elif which == "identity":
        sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=17, cfg=scene_mod.identity_network_config()))
    else:
        cfg = scene_mod.frequency_network_config(n_neurons=128, n_hidden_density=3, n_hidden_rgb=1)
        cfg["dir_encoding"] = {"otype": "SphericalHarmonics", "degree": 4}
        cfg["encoding"]["n_frequencies"] = 6
        sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=23, cfg=cfg))
    ctx.set_model(sc)
    m = oracle.make_model(sc)
    pos, _ = _rays(3000 + 7, 9)
    g = ctx.density_gradient(pos)
    gr = oracle.density_gradient(m, pos)
    assert np.isfinite(g).all() and g.shape == gr.shape
    nr = np.linalg.norm(gr, axis=1)
    err = np.linalg.norm(g - gr, axis=1) / np.maximum(nr, 1e-3 * np.median(nr))
    # fp16 gradients through up to eight layers, fp32 MFMA sums against exact ones: a last-place flip of one activation's gradient moves the result by ~1e-3
    assert np.median(err) < 1e-4 and (err < 0.05).mean() > 0.99, (float(np.median(err)), float((err < 0.05).mean()))  # (measured: median 9e-8, 99.8 %)
    cos = (g * gr).sum(1) / np.maximum(np.linalg.norm(g, axis=1) * nr, 1e-30)
    assert np.median(cos) > 0.9999
    w, h = 80, 45
    mat = scene_mod.orbit_camera(40.0, 25.0, 3.2)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    img = ctx.render(native.make_camera(mat, w, h, focal), native.make_opts(render_mode=native.RENDER_NORMALS))
    st = ctx.render_stats()
    fb, db, ost = oracle.render_nerf(m, oracle.make_camera(mat, w, h, focal), oracle.make_opts(render_mode=7))
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    oracle.release(m)
    assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= max(4, ost["n_samples"] // 2000)
    hit = ref[..., 3] > 0.5
    assert hit.sum() > 300 and np.isfinite(img).all()
    n_got = img[..., :3][hit] / img[..., 3:][hit] * 2 - 1
    n_ref = ref[..., :3][hit] / ref[..., 3:][hit] * 2 - 1
    c = (n_got * n_ref).sum(1) / (np.linalg.norm(n_got, axis=1) * np.linalg.norm(n_ref, axis=1))
    print(which, "normals frame: median cos %.7f, cos > 0.98: %.4f, cos > 0.9: %.4f, psnr %.1f dB; gradient: median rel err %.2e, < 5 %%: %.4f" % (
        float(np.median(c)), float((c > 0.98).mean()), float((c > 0.9).mean()), psnr(img[..., :3], ref[..., :3]), float(np.median(err)), float((err < 0.05).mean())))
    # a pixel's colour is the renormalised SUM of its samples' unit normals, and this network's gradient (frequencies up to 2^15 pi) turns by a large
    # angle from one sample to the next: where the sum nearly cancels, one sample whose gradient moved in the last fp16 place turns the pixel
    # (measured: 16 frequencies 98.8 % of the pixels within cos 0.9 and 29 dB; 6 frequencies / Identity: every pixel within 0.98, 60 / 81 dB)
    assert np.median(c) > 0.999 and (c > 0.9).mean() > 0.97, (float(np.median(c)), float((c > 0.9).mean()))
    if which != "frequency":
        assert (c > 0.98).mean() > 0.995 and psnr(img[..., :3], ref[..., :3]) > 50.0


def test_density_grid_refresh(native, oracle, scene_freq):
    """update_density_grid_nerf works for any NerfNetwork (src/testbed_nerf.cu:2772-2861): for the Frequency architecture the sampled
    cells and positions are the grid model's (same pcg32 stream), the density comes from the wide-MLP kernel; against the oracle."""
    c = native.Context(0)
    try:
        c.set_model(scene_freq)
        mc = scene_freq["max_cascade"]
        grid0 = c.density_grid(mc)
        c.update_density_grid(0.95, 40000, 20000, 2)
        got = c.density_grid(mc)
        m = oracle.make_model(scene_freq)
        rng = oracle.grid_rng()
        ref, step = oracle.update_density_grid(m, grid0, mc, rng, 0, 0.95, 40000, 20000)
        ref, step = oracle.update_density_grid(m, ref, mc, rng, step, 0.95, 40000, 20000)
        oracle.release(m)
        decayed = np.float32(0.95) * (np.float32(0.95) * grid0)
        assert np.array_equal(got != decayed, ref != decayed)  # exactly the same cells were touched
        touched = ref != decayed
        assert touched.sum() > 30000
        rel = np.abs(got[touched] - ref[touched]) / np.maximum(ref[touched], 1e-12)
        assert np.median(rel) < 5e-3 and (rel < 0.1).mean() > 0.995  # exp(logit) of a 9-layer fp16 network: a few fp16 ulps of the logit
        bf, mean = c.density_bitfield()
        obf, omean = oracle.density_grid_to_bitfield(ref, mc)
        assert abs(mean - omean) <= 5e-3 * omean and np.unpackbits(bf ^ obf).sum() <= 2e-4 * 128 ** 3
    finally:
        c.close()


def test_probe_envmap(ctx, native, scene_freq):
    """the irradiance probes (K10: rays from the centre) run on this architecture too"""
    ctx.set_model(scene_freq)
    env = ctx.compute_envmap(n_theta=32, n_phi=16)
    assert env.shape == (16, 32, 4)
    assert np.isfinite(env).all() and env[..., 3].max() > 0.5


def test_tile_sharding_and_block_jump_knob(ctx, native, scene_mod, scene_freq):
    """camera-tile shards (what bench.py --gpus N and ngp_create_multi deal out) are disjoint and sum to the unsharded frame bit for
    bit; with block_jumps off the march takes the reference's one-voxel steps and the image moves by rounding noise only"""
    w, h = 120, 72
    ctx.set_model(scene_freq)
    cam = native.make_camera(scene_mod.orbit_camera(45.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
    bg0 = (0.0, 0.0, 0.0, 0.0)
    full = ctx.render(cam, native.make_opts(background=bg0))
    total = np.zeros_like(full)
    for r in range(3):
        part = ctx.render(cam, native.make_opts(background=bg0, shard_index=r, shard_count=3))
        assert not (np.abs(total).sum(-1) > 0)[np.abs(part).sum(-1) > 0].any()
        total += part
    assert np.array_equal(total, full)
    st_on = ctx.render_stats()
    ctx.set_schedule(64, 4, 32, 1, 1, 4, 0)  # block_jumps = 0 (the other knobs, at their defaults, belong to the base.json kernel)
    try:
        exact = ctx.render(cam, native.make_opts(background=bg0))
    finally:
        ctx.set_schedule(64, 4, 32, 1, 1, 4, 1)
    d = np.abs(exact - full).max(-1)
    assert np.median(d) < 1e-3 and (d > 1e-2).mean() < 2e-3


def test_hybrid_frame_with_meshes(ctx, oracle, native, scene_mod, scene_freq):
    """Geometry mode with this architecture: mesh pass, then the NeRF pass inside the inflated scene box, depth-tested against the
    meshes (shade_kernel_nerf_geometry); the march starts outside the occupancy grid's cube"""
    ctx.set_model(scene_freq)
    ctx.clear_meshes()
    mi = pkg("meshio")
    meshes = [(mi.icosphere(3), (0.55, -0.1, 0.0)), (mi.torus(32, 16), (-0.3, 0.25, 0.4))]
    for tris, c in meshes:
        ctx.add_mesh(tris, c)
    hnd = oracle.mesh_scene(meshes)
    w, h = 128, 72
    mat = scene_mod.orbit_camera(60.0, 25.0, 5.5)
    focal = scene_mod.focal_from_fov_x(w, 0.8)
    try:
        img = ctx.render(native.make_camera(mat, w, h, focal), native.make_opts(testbed_mode=native.MODE_GEOMETRY))
        st = ctx.render_stats()
    finally:
        ctx.clear_meshes()
    ocam = oracle.make_camera(mat, w, h, focal)
    fb, db = oracle.render_mesh(hnd, ocam)
    sc = dict(scene_freq)
    lo, hi = oracle.mesh_scene_aabb(hnd)
    sc["render_aabb"] = (tuple(lo.tolist()), tuple(hi.tolist()))
    m = oracle.make_model(sc)
    fb2, db2, ost = oracle.render_nerf(m, ocam, oracle.make_opts(depth_test=True), frame_buffer=fb, depth_buffer=db)
    ref = oracle.tonemap(oracle.accumulate(fb2.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    oracle.release(m)
    oracle.mesh_scene_destroy(hnd)
    assert ost["n_rays_hit"] > 300 and abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
    assert (np.abs(fb2 - fb).sum(-1) > 0).mean() > 0.05  # the NeRF shows in front of / beside the meshes
    assert psnr(img[..., :3], ref[..., :3]) > 45.0
    assert (np.abs(img - ref).max(-1) < 1e-2).mean() > 0.995


def test_multi_device_context(native, scene_mod, scene_freq, gpu_ctx):
    """ngp_create_multi with this architecture (the same device listed twice on a one-GPU box): the assembled frame is the
    single-device frame bit for bit"""
    single = native.Context(0)
    multi = native.Context(devices=[0, 0])
    try:
        single.set_model(scene_freq)
        multi.set_model(scene_freq)
        w, h = 101, 67
        cam = native.make_camera(scene_mod.orbit_camera(60.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
        assert np.array_equal(multi.render(cam), single.render(cam))
        assert multi.render_stats()["n_samples"] == single.render_stats()["n_samples"]
    finally:
        multi.close()
        single.close()


def test_envmap_moving_camera_and_lens(ctx, oracle, native, scene_mod, scene_freq):
    """what sits around the network in the base kernel sits around this one too: an environment map behind the NeRF (1 spp direct
    output and 2 spp through accumulate + tonemap), a camera that moves within the frame with a rolling shutter, and a lens
    (OpenCV distortion) -- against the oracle"""
    ctx.set_model(scene_freq)
    m = oracle.make_model(scene_freq)
    rng = np.random.default_rng(12)
    env = np.zeros((16, 32, 4), np.float32)
    env[..., :3] = rng.uniform(0, 1, (16, 32, 3))
    env[..., 3] = rng.uniform(0.5, 1.0, (16, 32))
    env[..., :3] *= env[..., 3:4]
    w, h = 96, 54
    focal = scene_mod.focal_from_fov_x(w, 1.0)
    m0, m1 = scene_mod.orbit_camera(250.0, 10.0, 2.6), scene_mod.orbit_camera(262.0, 14.0, 2.8)
    lens = dict(lens_mode=native.LENS_OPENCV, lens_params=(0.1, -0.05, 0.01, -0.02)) if hasattr(native, "LENS_OPENCV") else {}
    try:
        ctx.set_envmap(env)
        for spp in (1, 2):
            kw = dict(background=(0.1, 0.2, 0.3, 1.0), exposure=0.25, to_srgb=True)
            cam_kw = dict(snap=(spp == 1), matrix1_3x4=m1, rolling_shutter=(0.0, 0.0, 0.3, 0.7))
            img = ctx.render(native.make_camera(m0, w, h, focal, **cam_kw, **lens), native.make_opts(spp=spp, **kw))
            acc = np.zeros((w * h, 4), np.float32)
            for s in range(spp):
                okw = dict(spp_index=s, snap=(spp == 1), matrix1_4x3=m1, rolling_shutter=(0.0, 0.0, 0.3, 0.7))
                if lens:
                    okw.update(lens_mode=1, lens_params=lens["lens_params"])
                fb, _, ost = oracle.render_nerf(m, oracle.make_camera(m0, w, h, focal, **okw), oracle.make_opts(envmap=env))
                acc = oracle.accumulate(fb.reshape(-1, 4), acc, s)
            ref = oracle.tonemap(acc, kw["background"], kw["exposure"], kw["to_srgb"]).reshape(h, w, 4)
            assert ost["n_rays_hit"] > 300
            assert psnr(img[..., :3], ref[..., :3]) > 45.0
            assert (np.abs(img - ref).max(-1) < 2e-2).mean() > 0.995
    finally:
        ctx.set_envmap(None)
        oracle.release(m)
