"""GPU parity on the configurations BASELINE.json names and on the committed fixtures (round-2 additions):

* the HIP path against the COMMITTED golden vectors (tests/golden/nerf_unit_v2.npz) -- no oracle is built or run;
* the benchmark model itself (Lego-shaped, T = 2^19: levels 0-2 dense on the padded lattice, 3-7 hashed) under pytest;
* the garden-shaped scene (aabb_scale 16, five cascades, upstream per_level_scale: render_nerf_fused_c5) and an
  aabb_scale-128 scene (the general eight-cascade kernel in NeRF mode) against the oracle;
* the reference's own meshes (bunny.obj, armadillo.obj as committed data) traced and shaded on the GPU against the
  committed hit records;
* empty-block jumps on against off: what the jumps change, isolated from everything else.
"""
import os

import numpy as np
import pytest

from conftest import _with_bitfield, pkg, psnr

HERE = os.path.dirname(os.path.abspath(__file__))


def _cams(native, oracle, scene_mod, w, h, az, el=30.0, radius=4.03):
    mat = scene_mod.orbit_camera(az, el, radius)
    focal = scene_mod.focal_from_fov_x(w, 0.6911)
    return native.make_camera(mat, w, h, focal), oracle.make_camera(mat, w, h, focal)


def _oracle_frame(oracle, m, ocam, w, h):
    fb, db, ost = oracle.render_nerf(m, ocam)
    ref = oracle.tonemap(oracle.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
    return ref, db.reshape(h, w), ost


# ---------------------------------------------------------------------------------------- committed golden vectors
@pytest.mark.gpu
def test_hip_path_matches_committed_golden_vectors(gpu_ctx, native):
    """Everything the fixture holds, HIP against the committed bytes: the GPU box needs no oracle for this test."""
    import importlib.util

    g = np.load(os.path.join(HERE, "golden", "nerf_unit_v2.npz"))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    sc, pos, dir01, cam_matrix, focal = mg.build_inputs()  # seeded inputs; the fixture pins them by hash and by value
    assert np.array_equal(pos, g["pos"]) and np.array_equal(dir01, g["dir01"]) and np.array_equal(cam_matrix.astype(np.float32), g["cam_matrix"])
    gpu_ctx.set_model(sc)
    # occupancy bits: bit-exact (hash of the 2 MB bitfield)
    import hashlib

    bf, mean = gpu_ctx.density_bitfield()
    assert np.array_equal(np.frombuffer(hashlib.sha256(bf.tobytes()).digest(), np.uint8), g["bitfield_sha256"]) and np.float32(mean) == g["bitfield_mean"]
    # hash-grid features: bit-exact as values
    enc = gpu_ctx.grid_encode(pos).view(np.uint16)
    assert np.array_equal(enc.view(np.float16).astype(np.float32), g["enc"].view(np.float16).astype(np.float32))
    # network outputs: fp16 logits, MFMA fp32 accumulation order vs the oracle's exact sums
    net = gpu_ctx.network(pos, dir01).astype(np.float32)
    ref = g["net"].view(np.float16).astype(np.float32).reshape(net.shape)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -14))) - 10)
    err = np.abs(net - ref)
    assert err.max() <= 3e-2 and (err == 0).mean() > 0.5 and (err <= ulp).mean() > 0.9
    # ray setup: the 40-byte NerfPayload records, bit for bit (unit scene: no transcendental on the path)
    w, h = mg.W, mg.H
    cam = native.make_camera(cam_matrix, w, h, tuple(focal))
    got = gpu_ctx.init_rays(cam)
    want = g["payloads"].view(got.dtype).reshape(got.shape)
    assert np.array_equal(got["alive"], want["alive"])
    alive = want["alive"] == 1
    for f in ("origin", "dir", "t", "idx"):
        assert np.array_equal(got[f][alive], want[f][alive]), f
    # frame (pre-tonemap radiance, premultiplied) + depth + counters
    img, depth = gpu_ctx.render(cam, native.make_opts(background=(0, 0, 0, 0)), want_depth=True)
    st = gpu_ctx.render_stats()
    n_rays, n_alive, n_hit, n_samples = g["stats"].tolist()
    assert abs(int(st["n_rays_hit"]) - n_hit) <= 1 and abs(int(st["n_samples"]) - n_samples) <= 2e-4 * n_samples + 2
    frame = g["frame"].reshape(h, w, 4)
    assert psnr(img[..., :3], frame[..., :3]) >= 50.0
    d = np.abs(img - frame).max(-1)
    assert (d < 1e-2).mean() >= 0.999 and d.max() < 0.3
    gd = g["depth"].reshape(h, w)
    both = (depth < 16000) & (gd < 16000)
    assert both.sum() > 300 and np.median(np.abs(depth[both] - gd[both])) < 1e-4


# ---------------------------------------------------------------------------------------- the benchmark model
@pytest.fixture(scope="module")
def scene_bench(oracle):
    """bench.py's model: make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19)"""
    return _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19))


@pytest.mark.gpu
def test_bench_model_encode_and_render(gpu_ctx, oracle, native, scene_mod, scene_bench):
    gpu_ctx.set_model(scene_bench)
    m = oracle.make_model(scene_bench)
    off, res, scl = oracle.grid_layout(m)
    size = np.diff(off)
    assert (res[:3].astype(np.uint64) ** 3 <= size[:3]).all() and (res[3:].astype(np.uint64) ** 3 > size[3:]).all() and (size[3:] == 1 << 19).all()
    rng = np.random.default_rng(21)
    pos = rng.uniform(0, 1, (20000, 3)).astype(np.float32)
    pos[:6] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0.5], [0.999999, 1e-7, 1.0], [0.25, 0.75, 0.125]]
    pos[6000:6064] = rng.uniform(-1, 2, (64, 3)).astype(np.float32)  # a wave outside the xor layout's range
    got = gpu_ctx.grid_encode(pos).astype(np.float32)
    ref = oracle.grid_encode(m, pos).astype(np.float32)
    assert np.array_equal(got, ref)
    w, h = 256, 144
    for az in (45.0, 225.0):
        cam, ocam = _cams(native, oracle, scene_mod, w, h, az)
        img, depth = gpu_ctx.render(cam, native.make_opts(), want_depth=True)
        st = gpu_ctx.render_stats()
        ref_img, db, ost = _oracle_frame(oracle, m, ocam, w, h)
        assert abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
        assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 1e-4 * ost["n_samples"] + 2
        assert st["n_samples"] / max(st["n_rays_hit"], 1) > 15
        assert psnr(img[..., :3], ref_img[..., :3]) >= 50.0
        d = np.abs(img - ref_img).max(-1)
        assert (d < 1e-2).mean() >= 0.9995 and d.max() < 0.3
    oracle.release(m)


# ---------------------------------------------------------------------------------------- fp16 vs fp32 accumulation in the MLPs
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["bench", "garden"])
def test_mlp_accumulation_bracket(which, gpu_ctx, oracle, scene_mod, scene_bench, scene_garden):
    """The reference's FullyFusedMLP accumulates in fp16 WMMA fragments (nerf_network.h:120,130), this build in fp32 MFMA
    accumulators; the reference cannot run here. Both are measured against the float64 network (oracle modes, oracle.h): the HIP
    frame must be at least as close to it as the fp16-accumulating model of the reference is, and its distance from that model --
    the bound on 'PSNR within 0.1 dB of the CUDA reference' that BASELINE.md states -- must stay above 65 dB."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("mlp_bracket", os.path.join(os.path.dirname(HERE), "tools", "mlp_bracket.py"))
    mb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mb)
    sc, view = (scene_bench, (45.0, 30.0, 4.03)) if which == "bench" else (scene_garden, (40.0, 25.0, 4.03))
    r = mb.bracket(oracle, gpu_ctx, sc, scene_mod.orbit_camera(*view), 256, 144)
    print(which, r)
    assert r["hit"] > 0.2 * 256 * 144
    assert r["psnr_exact_vs_ideal"] >= r["psnr_fp16_k16_vs_ideal"]
    if which == "bench":  # cone angle 0: with block_jumps = 0 the HIP frame has the oracle's sample sets, the comparison is arithmetic only
        assert r["psnr_hip_vs_ideal"] >= r["psnr_fp16_k16_vs_ideal"] - 0.5
        assert r["psnr_hip_vs_ideal"] >= 70.0 and r["psnr_hip_vs_fp16_k16"] >= 70.0
    else:
        # exponential stepping: device logf / expf differ from glibc by an ulp, some rays gain or lose a sample (max |d| ~ 0.09 on
        # a handful of pixels, which is all a PSNR at this level sees): compare the typical covered pixel instead
        assert r["q50_hip_vs_ideal"] <= 1.25 * r["q50_fp16_k16_vs_ideal"]
        assert r["psnr_hip_vs_ideal"] >= 60.0 and r["psnr_hip_vs_fp16_k16"] >= 60.0


# ---------------------------------------------------------------------------------------- garden-shaped / 8 cascades
@pytest.fixture(scope="module")
def scene_garden(oracle):
    """BASELINE.json config 5: aabb_scale 16 (five cascades, cone angle 1/256), upstream per_level_scale 2.97 => six of the
    eight levels hashed at T = 2^19"""
    return _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=16, seed=5, log2_hashmap_size=19, pls_rule="upstream"))


@pytest.fixture(scope="module")
def scene_128(oracle):
    """aabb_scale 128: all eight cascades, the general kernel (render_nerf_fused) in NeRF mode"""
    return _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=128, seed=6, log2_hashmap_size=17, pls_rule="upstream"))


def _check_large_scene(gpu_ctx, oracle, native, scene_mod, sc, w, h, views):
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    for az, el, radius in views:
        cam, ocam = _cams(native, oracle, scene_mod, w, h, az, el, radius)
        img = gpu_ctx.render(cam, native.make_opts())
        st = gpu_ctx.render_stats()
        ref, _, ost = _oracle_frame(oracle, m, ocam, w, h)
        assert ost["n_rays_hit"] > 0.3 * w * h
        # logf / expf of the exponential stepping differ by ulps between glibc and the device library: a ceilf in the voxel
        # skip can land a step further, so counts agree to a fraction of a percent rather than to the ray
        assert abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 0.002 * ost["n_rays_hit"] + 2
        assert abs(int(st["n_samples"]) - int(ost["n_samples"])) <= 5e-3 * ost["n_samples"]
        assert psnr(img[..., :3], ref[..., :3]) >= 45.0
        assert np.abs(img - ref).mean() < 1e-3
        assert (np.abs(img - ref).max(-1) < 2e-2).mean() > 0.995
    oracle.release(m)


@pytest.mark.gpu
def test_garden_shaped_scene_five_cascades(gpu_ctx, oracle, native, scene_mod, scene_garden):
    assert scene_garden["max_cascade"] == 4 and abs(scene_garden["encoding"]["per_level_scale"] - 2.9719) < 1e-3
    # outside looking in, and from inside the scene box (camera at radius 4 sits inside the 16-unit box)
    _check_large_scene(gpu_ctx, oracle, native, scene_mod, scene_garden, 160, 90, [(40.0, 25.0, 4.03), (200.0, 12.0, 9.0)])


@pytest.mark.gpu
def test_aabb_scale_128_scene_eight_cascades(gpu_ctx, oracle, native, scene_mod, scene_128):
    assert scene_128["max_cascade"] == 7
    # per_level_scale is exactly 4 here (the upstream rule at aabb_scale 128), so level 6 has resolution 65536: tcnn's
    # grid_index forms res^2 in uint32, gets 0, and indexes the level as (x + 65536 y) % T -- z drops out. The oracle runs
    # the same uint32 loop; the HIP table layout serves it through the hashed form (ngp_api.cpp build_xor_layout).
    gpu_ctx.set_model(scene_128)
    m = oracle.make_model(scene_128)
    off, res, scl = oracle.grid_layout(m)
    assert res[6] == 65536
    rng = np.random.default_rng(33)
    pos = rng.uniform(0, 1, (8192, 3)).astype(np.float32)
    pos[:5] = [[0, 0, 0], [1, 1, 1], [0.9999999, 0.5, 0.25], [0.99999, 0.99999, 0.99999], [0.5, 0.9999999, 0.1]]  # the far-edge corner x + 1 == res
    pos[64:128, 0] = np.float32(1.0) - rng.uniform(0, 2e-5, 64).astype(np.float32)
    assert np.array_equal(gpu_ctx.grid_encode(pos).astype(np.float32), oracle.grid_encode(m, pos).astype(np.float32))
    same_z = pos.copy()
    same_z[:, 2] = rng.uniform(0, 1, pos.shape[0]).astype(np.float32)
    a, b = oracle.grid_encode(m, pos).astype(np.float32), oracle.grid_encode(m, same_z).astype(np.float32)
    frac6 = (a[:, 24:28] == b[:, 24:28]).all(1).mean()  # level 6 ignores z except through the trilinear weights: rarely equal, never garbage
    assert np.isfinite(a).all() and frac6 < 0.5
    oracle.release(m)
    _check_large_scene(gpu_ctx, oracle, native, scene_mod, scene_128, 128, 72, [(70.0, 20.0, 4.03), (310.0, 35.0, 30.0)])


# ---------------------------------------------------------------------------------------- the reference's meshes
def _mesh_fixture(name):
    g = np.load(os.path.join(HERE, "golden", "mesh_%s_v1.npz" % name))
    return g, g["verts"][g["faces"]].astype(np.float32)


@pytest.mark.parametrize("name,n_tris", [("bunny", 4968), ("armadillo", 99976)])
def test_oracle_reproduces_mesh_fixture(name, n_tris, oracle):
    """CPU: the committed hit records are what the oracle answers today (pins the oracle's BVH path across platforms)."""
    g, tris = _mesh_fixture(name)
    assert tris.shape == (n_tris, 3, 3)  # bunny 4968 triangles, armadillo 99 976
    h = oracle.mesh_scene([(tris, (0.0, 0.0, 0.0))])
    hp, hn = oracle.trace_mesh(h, g["ray_o"], g["ray_d"])
    lo, hi = oracle.mesh_scene_aabb(h)
    oracle.mesh_scene_destroy(h)
    assert np.array_equal(hp, g["hit_pos"]) and np.array_equal(hn, g["hit_normal"]) and np.array_equal(np.concatenate([lo, hi]), g["scene_aabb"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bunny", "armadillo"])
def test_reference_mesh_traced_on_gpu(name, gpu_ctx, native, scene_mod):
    """bunny.obj / armadillo.obj (committed as data) through the HIP BVH4: hit position + face normal records and the
    shaded, shadowed frame against the committed oracle outputs."""
    g, tris = _mesh_fixture(name)
    shared_ctx = gpu_ctx  # (keeps the session's device initialisation order; this test wants a context WITHOUT a NeRF model)
    gpu_ctx = native.Context(0)
    gpu_ctx.add_mesh(tris, (0.0, 0.0, 0.0))
    lo, hi = gpu_ctx.mesh_info(-1)["aabb"]
    assert np.array_equal(np.concatenate([lo, hi]), g["scene_aabb"])
    gp, gd = gpu_ctx.trace_mesh_rays(g["ray_o"], g["ray_d"])
    rp, rd = g["hit_pos"], g["hit_normal"]
    hit = ~np.all(rd == g["ray_d"], axis=1)
    assert hit.sum() > 1000
    same = np.all(gp == rp, axis=1) & np.all(gd == rd, axis=1)
    assert same.mean() > 0.999  # two valid BVH partitions only differ on exact ties between triangles
    fin = np.isfinite(rp).all(1)
    assert np.abs(gp - rp)[fin].max() < 1e-4
    mg_shade = dict(sun_dir=(0.3, 0.8, 0.5), roughness=0.4, metallic=0.1, sheen=0.2, clearcoat=0.3, clearcoat_gloss=0.6, subsurface=0.1,
                    basecolor=(0.8, 0.6, 0.4), ambientcolor=(0.1, 0.1, 0.15))  # tests/golden/make_golden_mesh.py SHADE
    gpu_ctx.set_geometry_opts(**mg_shade)
    frame = g["frame"]
    h, w = frame.shape[:2]
    mat = g["cam_matrix"]
    img, depth = gpu_ctx.render(native.make_camera(mat, w, h, tuple(g["focal"])), native.make_opts(testbed_mode=native.MODE_GEOMETRY, background=(0, 0, 0, 0)), want_depth=True)
    assert (frame[..., 3] == 1).sum() > 1000 and np.array_equal(img[..., 3], frame[..., 3])
    assert (np.abs(img - frame).max(-1) < 1e-4).mean() > 0.998  # silhouette / edge ties only
    assert psnr(img[..., :3], frame[..., :3]) > 50.0
    assert (np.abs(depth - g["depth"]) < 1e-4).mean() > 0.998
    gpu_ctx.close()
    assert shared_ctx is not None


# ---------------------------------------------------------------------------------------- block jumps on / off
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["unit", "big"])
def test_block_jumps_are_the_only_source_of_sample_set_changes(which, gpu_ctx, oracle, native, scene_mod, scene_unit, scene_big):
    """With block_jumps off the march walks empty space voxel by voxel exactly like if_unoccupied_advance_to_next_occupied_voxel:
    on the unit scene (no transcendental in the stepping) every ray then takes exactly the oracle's samples at exactly the
    oracle's positions (measured: same sample count, max |d| 1e-4). Switching the jumps on perturbs t in its last bits and
    changes the sample set of a few rays in a million (a lattice point that coincides with a block face to fp32 rounding)."""
    sc = scene_unit if which == "unit" else scene_big
    w, h = 320, 180
    gpu_ctx.set_model(sc)
    m = oracle.make_model(sc)
    cam, ocam = _cams(native, oracle, scene_mod, w, h, 135.0)
    ref, _, ost = _oracle_frame(oracle, m, ocam, w, h)
    oracle.release(m)
    try:
        gpu_ctx.set_schedule(64, 4, 32, 1, 1, 4, 0)
        off = gpu_ctx.render(cam, native.make_opts())
        st_off = gpu_ctx.render_stats()
        gpu_ctx.set_schedule(64, 4, 32, 1, 1, 4, 1)
        on = gpu_ctx.render(cam, native.make_opts())
        st_on = gpu_ctx.render_stats()
    finally:
        gpu_ctx.set_schedule(64, 4, 32, 1, 1, 4, 1)
    d_off = np.abs(off - ref).max(-1)
    d_on = np.abs(on - ref).max(-1)
    changed = np.abs(on - off).max(-1) > 0
    print(f"block jumps [{which}]: off vs oracle max {d_off.max():.4f}, on vs oracle max {d_on.max():.4f}, {int(changed.sum())} of {changed.size} pixels change, "
          f"samples off {st_off['n_samples']} on {st_on['n_samples']} oracle {ost['n_samples']}")
    # jumps off: the march is the oracle's, so what remains is fp16 network noise (MFMA accumulation order) and a ray now
    # and then whose early termination flips on it -- no pixel is off by a sample's worth of radiance
    # (big: the device's logf / expf differ from glibc's by an ulp in the exponential stepping, which moves a sample by a step
    # on a few rays whatever the jumps do -- measured 0.25 on one pixel with jumps off and on alike)
    tol = 5e-3 if which == "unit" else 0.3
    assert d_off.max() < tol and (d_off > 2e-2).mean() < 5e-3 and psnr(off[..., :3], ref[..., :3]) >= (55.0 if which == "unit" else 45.0)
    assert abs(int(st_off["n_samples"]) - int(ost["n_samples"])) <= (1e-4 if which == "unit" else 5e-3) * ost["n_samples"] + 2
    # on vs off (same network arithmetic on both sides). A jump lands on the same lattice point of the ray's step grid as
    # the chain of voxel steps it replaces, but reaches it through fewer fp32 roundings, so t -- and with it the sample
    # position -- can differ in the last bits (the reference's own --use_fast_math division moves t by as much): through the
    # fp16 features that is colour noise of a few 1e-3 on many pixels. The sample SET changes for a few rays in a million
    # (a lattice point within rounding of a block face); those rays are the outliers assert_image_close's `hard` bound
    # allows for, and there are no others.
    delta = np.abs(on - off).max(-1)
    # (cascaded scene: since the climb keeps the farthest-reaching empty block the jumps are longer -- 16^3 blocks of outer cascades -- and a
    # few more lattice points coincide with a block face: measured 15 of 57600 pixels, 31 of 586512 samples)
    assert np.median(delta) < 1e-3 and (delta > 1e-2).mean() < (2e-4 if which == "unit" else 5e-4), f"{(delta > 1e-2).sum()} of {delta.size} pixels move by more than 1e-2"
    assert abs(int(st_on["n_samples"]) - int(st_off["n_samples"])) <= 1e-4 * st_off["n_samples"] + 2
    assert d_on[delta <= 1e-2].max() < max(tol, 2e-2) and d_on.max() < 0.3


def test_schedule_knobs_are_validated(native):
    """out-of-range knobs would spin the persistent kernel; they are refused before they reach it."""
    ctx = native.Context(-1)
    for bad in ((65,), (15,), (64, 0), (64, 4, 0), (64, 4, 32, -1), (64, 4, 32, 1, 0), (64, 4, 32, 1, 9), (64, 4, 32, 1, 8, 8, 2), (64, 4, 32, 1, 8, 8, 1, 2)):
        with pytest.raises(RuntimeError, match="schedule knob"):
            ctx.set_schedule(*bad)
    ctx.set_schedule(32, 8, 16, 2, 1, 2, 0, 0)
    ctx.set_schedule(64, 4, 32, 1, 1, 4, 1)
    ctx.close()


@pytest.mark.gpu
def test_render_arguments_are_validated_before_any_launch(native, scene_mod, scene_unit, gpu_ctx):
    """A frame's arguments come from a caller, not from the library: resolutions, shard coordinates and modes that the kernels' indexing does
    not cover are refused with a message; a camera of NaNs renders an empty frame (every ray invalid) and leaves the context usable."""
    gpu_ctx.set_model(scene_unit)
    focal = scene_mod.focal_from_fov_x(64, 0.6911)
    good = native.make_camera(scene_mod.orbit_camera(45.0), 64, 36, focal)
    ref = gpu_ctx.render(good, native.make_opts())
    for w, h in ((0, 36), (64, 0), (-8, 36), (70000, 36), (64, 70000)):
        with pytest.raises(RuntimeError, match="invalid render resolution"):
            gpu_ctx.render(native.make_camera(scene_mod.orbit_camera(45.0), w, h, focal), native.make_opts())
    with pytest.raises(RuntimeError, match="shard_index out of range"):
        gpu_ctx.render(good, native.make_opts(shard_index=3, shard_count=2))
    with pytest.raises(RuntimeError, match="render modes implemented"):
        gpu_ctx.render(good, native.make_opts(render_mode=99))
    nan_cam = native.make_camera(np.full((3, 4), np.nan, np.float32), 64, 36, focal)
    img = gpu_ctx.render(nan_cam, native.make_opts())
    assert img.shape == (36, 64, 4) and np.isfinite(img).all() and (img == img[0, 0]).all()  # (the background, everywhere)
    nan_focal = native.make_camera(scene_mod.orbit_camera(45.0), 64, 36, (float("nan"), float("nan")))
    gpu_ctx.render(nan_focal, native.make_opts())
    assert np.array_equal(gpu_ctx.render(good, native.make_opts()), ref)


# ---------------------------------------------------------------------------------------- several devices behind one context
@pytest.mark.gpu
@pytest.mark.parametrize("n_dev", [2, 3])
def test_multi_device_context_assembles_the_single_device_frame(n_dev, native, scene_mod, scene_unit):
    """ngp_create_multi (Testbed's device list, src/testbed.cu:5490-5616): replicas kept in step by generation, every device
    renders its tiles tile-packed, peer copies bring them to the primary, which scatters them into the image. Rehearsed on
    ONE GPU by listing its ordinal several times (separate contexts, streams and buffers; the peer copies are local)."""
    single = native.Context(0)
    single.set_model(scene_unit)
    multi = native.Context(devices=[0] * n_dev)
    assert multi.n_devices() == n_dev
    multi.set_model(scene_unit)
    for (w, h) in ((200, 112), (101, 67)):  # the second: neither a multiple of the tile nor of the device count
        cam = native.make_camera(scene_mod.orbit_camera(60.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
        ref, ref_depth = single.render(cam, native.make_opts(), want_depth=True)
        st1 = single.render_stats()
        img, depth = multi.render(cam, native.make_opts(), want_depth=True)
        st = multi.render_stats()
        assert np.array_equal(img, ref) and np.array_equal(depth, ref_depth)
        assert st["n_rays_hit"] == st1["n_rays_hit"] and st["n_samples"] == st1["n_samples"]
        shares = [multi.device_render_stats(i)["n_rays"] for i in range(n_dev)]
        assert sum(shares) == st["n_rays"] and max(shares) - min(shares) <= 64  # tiles dealt round-robin
    # the replicas follow the primary: a new model, a render box, a refreshed occupancy grid
    other = dict(scene_unit, render_aabb=((0.1, 0.0, 0.0), (0.9, 1.0, 1.0)))
    single.set_model(other)
    multi.set_model(other)
    cam = native.make_camera(scene_mod.orbit_camera(200.0), 160, 90, scene_mod.focal_from_fov_x(160, 0.6911))
    assert np.array_equal(multi.render(cam), single.render(cam))
    single.set_render_aabb((0.0, 0.0, 0.2), (1.0, 1.0, 0.8))
    multi.set_render_aabb((0.0, 0.0, 0.2), (1.0, 1.0, 0.8))
    assert np.array_equal(multi.render(cam), single.render(cam))
    single.update_density_grid(0.95, 1 << 18, 1 << 16, 2)
    multi.update_density_grid(0.95, 1 << 18, 1 << 16, 2)
    assert np.array_equal(multi.render(cam), single.render(cam))
    # several samples per pixel and sRGB output go through the same assembly
    o = native.make_opts(spp=3, to_srgb=True, background=(0.2, 0.3, 0.4, 1.0))
    a, b = multi.render(native.make_camera(scene_mod.orbit_camera(200.0), 96, 54, scene_mod.focal_from_fov_x(96, 0.6911), snap=False), o), None
    b = single.render(native.make_camera(scene_mod.orbit_camera(200.0), 96, 54, scene_mod.focal_from_fov_x(96, 0.6911), snap=False), o)
    assert np.abs(a - b).max() < 1e-6
    # Geometry mode (the reference's per-device render_frame serves every mode, src/testbed.cu:4833-4889, 5575-5616): meshes and their
    # BVHs, the BRDF / sun parameters and the irradiance tables follow to every device; mesh pass, shadow rays and the NeRF pass run per share
    meshio = pkg("meshio")
    for c in (single, multi):
        c.add_mesh(meshio.icosphere(3), (0.62, -0.05, 0.05))
        c.add_mesh(meshio.torus(24, 16), (-0.45, 0.3, 0.5))
        c.compute_envmap(0, 32, 16)
    for mode in (native.RENDER_SHADE, native.RENDER_SHADE_ENVMAP):
        o = native.make_opts(testbed_mode=native.MODE_GEOMETRY, render_mode=mode)
        gcam = native.make_camera(scene_mod.orbit_camera(60.0, 25.0, 5.5), 144, 81, scene_mod.focal_from_fov_x(144, 0.8))
        a, da = multi.render(gcam, o, want_depth=True)
        b, db = single.render(gcam, o, want_depth=True)
        assert np.abs(a - b).max() < 1e-6 and np.array_equal(da, db) and a[..., 3].max() > 0.9
    single.clear_meshes()
    multi.clear_meshes()
    assert np.array_equal(multi.render(cam), single.render(cam))
    multi.close()
    single.close()


@pytest.mark.gpu
def test_render_into_page_locked_images(native, scene_mod, scene_unit, gpu_ctx):
    """ngp_render with a destination from ngp_host_alloc: the kernels write the device-mapped image themselves (direct output at
    1 spp, accumulate + tonemap at several, the tile scatter of a multi-device context) and no copy follows; the pixels are those of
    the render-then-copy path into ordinary memory, bit for bit"""
    import ctypes as C

    single = native.Context(0)
    multi = native.Context(devices=[0, 0])
    try:
        single.set_model(scene_unit)
        multi.set_model(scene_unit)
        w, h = 200, 112
        cam = native.make_camera(scene_mod.orbit_camera(30.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911), snap=False)
        for opts in (native.make_opts(to_srgb=True), native.make_opts(spp=3, background=(0.2, 0.3, 0.4, 1.0)), native.make_opts(render_mode=native.RENDER_DEPTH)):
            ref = single.render(cam, opts)
            assert np.array_equal(single.render_pinned(cam, opts), ref)
            assert np.array_equal(multi.render_pinned(cam, opts), ref)
        # a destination INSIDE a page-locked buffer (an image of a batch): the alias carries the offset
        batch = native.host_image((3, h, w, 4))
        batch[:] = -1.0
        opts = native.make_opts()
        ref = single.render(cam, opts)
        single._check(single.L.ngp_render(single.h, C.byref(cam), C.byref(opts), batch[1].ctypes.data_as(C.c_void_p), None))
        assert np.array_equal(batch[1], ref) and (batch[0] == -1.0).all() and (batch[2] == -1.0).all()
    finally:
        multi.close()
        single.close()


@pytest.mark.gpu
def test_fox_shaped_4k_across_eight_shards(native, scene_mod, scene_big, gpu_ctx):
    """BASELINE config 4's tiling (fox-shaped model: aabb_scale 4, exponential stepping, 3 cascades; 3840 x 2160 over 8 ranks),
    rehearsed on one GPU: every rank's tile-packed share rendered in turn, assembled the way the gather does, equals the frame one
    device renders -- the c5 kernel, packed output and the unpack indices at the size the scaling run uses."""
    import torch
    from conftest import pkg

    par = pkg("parallel")
    w, h, world = 3840, 2160, 8
    ctx = native.Context(0)
    try:
        ctx.set_model(scene_big)
        cam = native.make_camera(scene_mod.orbit_camera(120.0, 25.0, 5.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
        full, full_depth = ctx.render(cam, native.make_opts(), want_depth=True)
        n_full = ctx.render_stats()["n_samples"]
        g = par.PackedFrameGather(w, h, world, torch.device("cuda", 0))
        parts_rgba, parts_depth, n_parts = [], [], 0
        for r in range(world):
            rgba, depth = g.buffers()
            ctx.render_device(cam, native.make_opts(shard_index=r, shard_count=world, packed_output=True), rgba.data_ptr(), depth.data_ptr(), None)
            n_parts += ctx.render_stats()["n_samples"]  # (synchronises the context's stream)
            parts_rgba.append(rgba.clone())
            parts_depth.append(depth.clone())
        img, dep = g.unpack(torch.cat(parts_rgba), torch.cat(parts_depth))
        assert n_parts == n_full and full[..., 3].min() >= 0.0 and (full[..., :3] > 0).mean() > 0.3
        assert np.array_equal(img.cpu().numpy(), full)
        assert np.array_equal(dep.cpu().numpy(), full_depth)
    finally:
        ctx.close()
