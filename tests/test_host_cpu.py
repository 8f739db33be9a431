"""CPU tests of the host side: the C ABI exports everything include/ngp_hip.h declares, file formats
(transforms.json, msgpack / .ingp snapshots) parse and round-trip without a GPU, errors are reported, and the
N > 1 tile-sharding path works over gloo with world_size 2."""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest

from conftest import ROOT, pkg

REF = "/root/reference"


def test_library_exports_every_declared_symbol(native):
    L = native.load_library()
    names = native.header_exports()
    assert len(names) >= 20 and "ngp_render" in names and "ngp_load_snapshot" in names and "ngp_load_training_data" in names
    for n in names:
        assert hasattr(L, n), f"libngp_hip.so does not export {n}"
        # every entry point the binding calls must declare its argument types: ctypes would otherwise pass the 64-bit
        # context handle as a C int
        assert n == "ngp_version" or getattr(L, n).argtypes is not None, f"native.py declares no argtypes for {n}"
    assert b"gfx950" in L.ngp_version()


def test_no_cpu_fallback(native):
    ctx = native.Context(-1)  # host-only: formats yes, rendering no
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        ctx.render(native.make_camera(np.eye(3, 4, dtype=np.float32), 8, 8, (8.0, 8.0)))
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        ctx.grid_encode(np.zeros((4, 3), np.float32))
    ctx.close()
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            native.Context(0)
    # the product never touches the oracle
    for root, _, files in os.walk(os.path.join(ROOT, pkg("native").__name__.split(".")[0])):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "orc_" not in txt, f


def _write_transforms(tmp_path, **extra):
    frames = []
    for i, name in enumerate(["r_10", "r_2", "r_1"]):  # deliberately not in natural order
        m = np.eye(4)
        m[:3, 3] = [i + 1.0, 2.0 * i, -1.0]
        frames.append({"file_path": f"./train/{name}", "transform_matrix": m.tolist()})
    d = {"camera_angle_x": 0.6911, "w": 800, "h": 600, "aabb_scale": 2, "frames": frames}
    d.update(extra)
    p = tmp_path / "transforms.json"
    p.write_text("// a comment, accepted like nlohmann's ignore_comments\n" + json.dumps(d))
    return str(p)


def test_transforms_json(tmp_path, native, scene_mod):
    ctx = native.Context(-1)
    ctx.load_training_data(_write_transforms(tmp_path))
    assert ctx.n_training_views() == 3
    info = ctx.dataset_info()
    assert info["aabb_scale"] == 2 and abs(info["scale"] - 0.33) < 1e-7 and np.allclose(info["offset"], 0.5)
    # natural sort: r_1, r_2, r_10 -> translations x = 3, 2, 1
    xs = []
    for i in range(3):
        v = ctx.training_view(i)
        assert v["resolution"].tolist() == [800, 600]
        f = 0.5 * 800 / np.tan(0.5 * 0.6911)
        assert np.allclose(v["focal_length"], f, rtol=1e-5)
        xs.append(v["matrix"])
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = [3.0, 4.0, -1.0]
    assert np.allclose(xs[0], scene_mod.nerf_matrix_to_ngp(m), atol=1e-6)
    # per-json overrides: scale/offset/fl_x/cx
    ctx.load_training_data(_write_transforms(tmp_path, scale=0.5, offset=[0.1, 0.2, 0.3], fl_x=1000.0, cx=200.0, cy=150.0))
    info = ctx.dataset_info()
    assert abs(info["scale"] - 0.5) < 1e-7 and np.allclose(info["offset"], [0.1, 0.2, 0.3])
    v = ctx.training_view(2)
    assert np.allclose(v["focal_length"], 1000.0) and np.allclose(v["principal_point"], [0.25, 0.25])
    with pytest.raises(RuntimeError, match="json file or a directory"):
        ctx.load_training_data(str(tmp_path / "nope.txt"))
    with pytest.raises(RuntimeError, match="cannot open"):
        ctx.load_training_data(str(tmp_path / "missing.json"))
    (tmp_path / "bad.json").write_text("{\"frames\": [")
    with pytest.raises(RuntimeError, match="json parse error"):
        ctx.load_training_data(str(tmp_path / "bad.json"))
    ctx.close()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference mount not present (GPU box)")
def test_reference_datasets_load(native):
    """The reference's own data files (SURVEY 8c 'usable inputs'): fox keeps the 50 frames whose images exist."""
    ctx = native.Context(-1)
    ctx.load_training_data(os.path.join(REF, "data/nerf/fox"))
    assert ctx.n_training_views() == 50
    assert ctx.dataset_info()["aabb_scale"] == 4
    v = ctx.training_view(0)
    assert v["resolution"].tolist() == [1080, 1920] and abs(v["focal_length"][0] - 1375.52) < 1e-2
    assert np.allclose(v["principal_point"], [554.558 / 1080, 965.268 / 1920], atol=1e-6)
    # read_lens (src/nerf_loader.cu:175-240): fox carries OpenCV distortion coefficients
    assert v["lens_mode"] == native.LENS_OPENCV and np.allclose(v["lens_params"][:4], [0.0578421, -0.0805099, -0.000980296, 0.00015575], rtol=1e-6)
    ctx.load_training_data(os.path.join(REF, "data/test3/images/transforms_train.json"))
    assert ctx.n_training_views() == 150
    assert ctx.training_view(3)["lens_mode"] == native.LENS_PERSPECTIVE
    info = ctx.dataset_info()
    assert info["aabb_scale"] == 1 and np.allclose(info["offset"], [0.5, 0.5, 0.0])
    ctx.close()


def test_lens_undistortion_inverts_the_distortion(oracle):
    """iterative_lens_undistortion (common_device.cuh:300-329): x + delta(x) == x0 after the Newton iteration, for the fox
    dataset's OpenCV coefficients and for a fisheye lens."""
    import ctypes as C

    for mode, q in ((1, (0.0578421, -0.0805099, -0.000980296, 0.00015575)), (4, (0.05, -0.01, 0.003, -0.0005))):
        cam = oracle.make_camera(np.eye(3, 4, dtype=np.float32), 1000, 1000, (1000.0, 1000.0), lens_mode=mode, lens_params=q)
        persp = oracle.make_camera(np.eye(3, 4, dtype=np.float32), 1000, 1000, (1000.0, 1000.0))
        d = (C.c_float * 3)()
        p = (C.c_float * 3)()
        for u, v in ((0.5, 0.5), (0.1, 0.9), (0.95, 0.3), (0.0, 0.0)):
            oracle.lib.orc_lens_direction(C.byref(cam), C.c_float(u), C.c_float(v), d)
            oracle.lib.orc_lens_direction(C.byref(persp), C.c_float(u), C.c_float(v), p)
            x, y = d[0], d[1]
            if mode == 1:
                k1, k2, p1, p2 = q
                r2 = x * x + y * y
                rad = k1 * r2 + k2 * r2 * r2
                dx = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
                dy = y * rad + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
            else:
                r = float(np.hypot(x, y))
                if r > 1e-12:
                    th = np.arctan(r)
                    thd = th * (1 + q[0] * th ** 2 + q[1] * th ** 4 + q[2] * th ** 6 + q[3] * th ** 8)
                    dx, dy = x * thd / r - x, y * thd / r - y
                else:
                    dx = dy = 0.0
            assert abs(x + dx - p[0]) < 2e-5 and abs(y + dy - p[1]) < 2e-5 and d[2] == 1.0


def test_snapshot_formats_roundtrip(tmp_path, native):
    import msgpack

    sc = pkg("synthetic").make_scene(aabb_scale=2, seed=5, log2_hashmap_size=12, pls_rule="upstream")
    ctx = native.Context(-1)
    ctx.set_model(sc)
    for name in ("s.msgpack", "s.ingp"):
        p = str(tmp_path / name)
        ctx.save_snapshot_file(p)
        raw = open(p, "rb").read()
        if name.endswith(".ingp"):
            raw = gzip.decompress(raw)  # zstr writes a gzip container
        root = msgpack.unpackb(raw, raw=False)
        snap = root["snapshot"]
        assert snap["version"] == 1 and snap["mode"] == "nerf" and snap["density_grid_size"] == 128
        assert snap["n_params"] == sc["params"].size and snap["params_type"] == "__half"
        assert np.array_equal(np.frombuffer(snap["params_binary"], np.uint16), sc["params"])
        assert np.array_equal(np.frombuffer(snap["density_grid_binary"], np.float16), np.asarray(sc["density_grid"], np.float16))
        assert snap["nerf"]["aabb_scale"] == 2 and root["encoding"]["n_levels"] == 8
        assert abs(root["encoding"]["per_level_scale"] - sc["encoding"]["per_level_scale"]) < 1e-7
        ctx2 = native.Context(-1)
        ctx2.load_snapshot_file(p)
        d = ctx2.get_model()
        assert d.n_params == sc["params"].size and d.aabb_scale == 2 and d.n_density_grid == 2 * 128 ** 3
        assert abs(d.cone_angle_constant - 1 / 256) < 1e-9 and list(d.aabb_min) == [-0.5] * 3 and list(d.aabb_max) == [1.5] * 3
        ctx2.close()
    # a snapshot written by another msgpack encoder (python) with float32 params loads too
    root["snapshot"]["params_type"] = "float"
    root["snapshot"]["params_binary"] = sc["params"].view(np.float16).astype(np.float32).tobytes()
    blob = msgpack.packb(root, use_bin_type=True)
    ctx3 = native.Context(-1)
    ctx3.load_snapshot_bytes(blob)
    assert ctx3.get_model().n_params == sc["params"].size
    # error paths (Testbed::load_snapshot throws, src/testbed.cu:5285-5352)
    bad = dict(root)
    bad["snapshot"] = dict(root["snapshot"], version=0)
    with pytest.raises(RuntimeError, match="old format"):
        ctx3.load_snapshot_bytes(msgpack.packb(bad, use_bin_type=True))
    bad["snapshot"] = dict(root["snapshot"], density_grid_size=64)
    with pytest.raises(RuntimeError, match="Incompatible grid size"):
        ctx3.load_snapshot_bytes(msgpack.packb(bad, use_bin_type=True))
    bad["snapshot"] = dict(root["snapshot"], density_grid_binary=b"\0" * 100)
    with pytest.raises(RuntimeError, match="grid cascades"):
        ctx3.load_snapshot_bytes(msgpack.packb(bad, use_bin_type=True))
    bad["snapshot"] = dict(root["snapshot"], params_binary=b"\0" * 64)
    with pytest.raises(RuntimeError, match="parameter count mismatch|n_params"):
        ctx3.load_snapshot_bytes(msgpack.packb(bad, use_bin_type=True))
    with pytest.raises(RuntimeError, match="does not contain a snapshot"):
        ctx3.load_snapshot_bytes(msgpack.packb({"encoding": {}}, use_bin_type=True))
    with pytest.raises(RuntimeError, match="truncated|msgpack"):
        ctx3.load_snapshot_bytes(blob[: len(blob) // 2])
    with pytest.raises(RuntimeError, match="inflate|corrupt"):
        ctx3.load_snapshot_bytes(b"not a gzip stream", compressed=True)
    ctx3.close()
    ctx.close()


def _dense_scene():
    S = pkg("scene")
    cfg = S.base_network_config()
    cfg["encoding"] = dict(cfg["encoding"], otype="DenseGrid", base_resolution=4, per_level_scale=1.5)
    return pkg("synthetic").make_scene(aabb_scale=1, seed=77, log2_hashmap_size=31, cfg=cfg)


def test_snapshot_dense_and_unsupported_encodings(tmp_path, native):
    """tcnn's GridEncoding family (create_grid_encoding, dependencies/tiny-cuda-nn encodings/grid.h): HashGrid and DenseGrid
    (no level is ever hashed) are implemented; TiledGrid and non-grid position encodings are refused at load."""
    import msgpack

    sc = _dense_scene()
    offs, res, _ = pkg("scene").grid_layout(sc["encoding"])
    assert all(o2 - o1 >= r ** 3 for o1, o2, r in zip(offs, offs[1:], res))  # every level holds its full lattice
    ctx = native.Context(-1)
    ctx.set_model(sc)
    p = str(tmp_path / "dense.msgpack")
    ctx.save_snapshot_file(p)
    root = msgpack.unpackb(open(p, "rb").read(), raw=False)
    assert root["encoding"]["otype"] == "DenseGrid" and "log2_hashmap_size" not in root["encoding"]
    ctx2 = native.Context(-1)
    ctx2.load_snapshot_file(p)
    d = ctx2.get_model()
    assert d.log2_hashmap_size == 31 and d.n_params == sc["params"].size and d.base_resolution == 4
    # the generic spelling tcnn also accepts: otype Grid + type Dense
    root["encoding"]["otype"], root["encoding"]["type"] = "Grid", "Dense"
    ctx2.load_snapshot_bytes(msgpack.packb(root, use_bin_type=True))
    assert ctx2.get_model().log2_hashmap_size == 31
    for enc in ({"otype": "TiledGrid"}, {"otype": "Grid", "type": "Tiled"}, {"otype": "OneBlob"}):
        bad = dict(root, encoding=dict(root["encoding"], **enc))
        with pytest.raises(RuntimeError, match="unsupported (encoding|grid type)"):
            ctx2.load_snapshot_bytes(msgpack.packb(bad, use_bin_type=True))
    # choices that leave the parameter count unchanged must not load silently: direction encoding, activations
    assert root["dir_encoding"]["nested"][0]["degree"] == 4
    for key, val, msg in (("dir_encoding", {"otype": "SphericalHarmonics", "degree": 3}, "unsupported dir_encoding"),
                          ("dir_encoding", {"otype": "Frequency", "n_frequencies": 4}, "unsupported dir_encoding"),
                          ("dir_encoding", {"otype": "Composite", "nested": [{"otype": "SphericalHarmonics", "degree": 4, "n_dims_to_encode": 3}, {"otype": "OneBlob"}]}, "unsupported dir_encoding"),
                          ("network", dict(root["network"], activation="Sine"), "unsupported network activation"),
                          ("rgb_network", dict(root["rgb_network"], output_activation="Sigmoid"), "unsupported network output_activation")):
        with pytest.raises(RuntimeError, match=msg):
            ctx2.load_snapshot_bytes(msgpack.packb(dict(root, **{key: val}), use_bin_type=True))
    for ok in ({"otype": "SphericalHarmonics", "degree": 4}, {"otype": "Composite", "nested": [{"otype": "SphericalHarmonics", "degree": 4}]}):
        ctx2.load_snapshot_bytes(msgpack.packb(dict(root, dir_encoding=ok), use_bin_type=True))
    ctx2.close()
    ctx.close()


def test_frequency_architecture_snapshot(tmp_path, native, scene_mod):
    """configs/nerf/frequency.json (SURVEY a-19): Frequency encodings + CutlassMLP 256 x 7 / 256 x 1. Shapes follow NerfNetwork's
    padding to the MLP alignment (8 for CutlassMLP): 96 + 24 inputs, rgb input 40, rgb output 8."""
    import msgpack

    cfg = scene_mod.frequency_network_config()
    assert scene_mod.network_shapes(cfg) == (96, 24, 40, 8)
    assert scene_mod.n_params(cfg) == (96 * 256 + 6 * 256 * 256 + 16 * 256, 40 * 256 + 8 * 256, 0)
    ff = scene_mod.frequency_network_config(n_neurons=128, n_hidden_density=2, n_hidden_rgb=2)
    ff["network"]["otype"] = ff["rgb_network"]["otype"] = "FullyFusedMLP"
    ff["encoding"]["n_frequencies"] = 10
    assert scene_mod.network_shapes(ff) == (64, 32, 48, 16)  # alignment 16
    sc = pkg("synthetic").make_scene(aabb_scale=2, seed=4, cfg=cfg)
    ctx = native.Context(-1)
    ctx.set_model(sc)
    p = str(tmp_path / "freq.msgpack")
    ctx.save_snapshot_file(p, compress=False)
    root = msgpack.unpackb(open(p, "rb").read(), raw=False)
    assert root["encoding"] == {"otype": "Frequency", "n_frequencies": 16} and root["dir_encoding"] == {"otype": "Frequency", "n_frequencies": 4}
    assert root["network"]["otype"] == "CutlassMLP" and root["network"]["n_hidden_layers"] == 7 and root["rgb_network"]["n_neurons"] == 256
    assert root["snapshot"]["n_params"] == 421888 + 12288
    ctx2 = native.Context(-1)
    ctx2.load_snapshot_file(p)
    d = ctx2.get_model()
    assert (d.pos_encoding, d.pos_n_frequencies, d.dir_encoding, d.dir_n_frequencies, d.mlp_alignment, d.aabb_scale) == (1, 16, 1, 4, 8, 2)
    # the direction encoding inside a Composite (how base.json spells its own), and a parameter count that does not follow the shapes
    comp = {"otype": "Composite", "nested": [{"otype": "Frequency", "n_frequencies": 4, "n_dims_to_encode": 3}, {"otype": "Identity"}]}
    ctx2.load_snapshot_bytes(msgpack.packb(dict(root, dir_encoding=comp), use_bin_type=True))
    assert ctx2.get_model().dir_n_frequencies == 4
    with pytest.raises(RuntimeError, match="parameter count mismatch"):
        ctx2.load_snapshot_bytes(msgpack.packb(dict(root, encoding={"otype": "Frequency", "n_frequencies": 12}), use_bin_type=True))
    with pytest.raises(RuntimeError, match="unsupported network architecture"):
        ctx2.load_snapshot_bytes(msgpack.packb(dict(root, network=dict(root["network"], n_neurons=64), rgb_network=dict(root["rgb_network"], n_neurons=64)), use_bin_type=True))
    # configs/nerf/none.json: Identity encodings in front of the same MLPs (3 -> 8 inputs each, rgb input 16 + 8)
    ident = scene_mod.identity_network_config()
    assert scene_mod.network_shapes(ident) == (8, 8, 24, 8)
    ctx.set_model(pkg("synthetic").make_scene(aabb_scale=1, seed=4, cfg=ident))
    ctx.save_snapshot_file(p, compress=False)
    root = msgpack.unpackb(open(p, "rb").read(), raw=False)
    assert root["encoding"] == {"otype": "Identity"} and root["dir_encoding"] == {"otype": "Identity"} and root["snapshot"]["n_params"] == 399360 + 8192
    ctx2.load_snapshot_file(p)
    d = ctx2.get_model()
    assert (d.pos_encoding, d.dir_encoding, d.mlp_alignment, d.n_params) == (2, 2, 8, 399360 + 8192)
    with pytest.raises(RuntimeError, match="unsupported Identity encoding"):
        ctx2.load_snapshot_bytes(msgpack.packb(dict(root, encoding={"otype": "Identity", "scale": 2.0}), use_bin_type=True))
    ctx2.close()
    ctx.close()


def test_snapshot_session_state_roundtrip(tmp_path, native):
    """save_snapshot / load_snapshot carry the session (src/testbed.cu:5245-5263, 5395-5418): background, exposure, sun / up
    direction, camera with scale / aperture / autofocus depth -- under the reference's key names."""
    import msgpack

    sc = pkg("synthetic").make_scene(aabb_scale=1, seed=3, log2_hashmap_size=12)
    ctx = native.Context(-1)
    ctx.set_model(sc)
    assert ctx.session_state().valid == 0
    st = native.SessionState()
    st.background_color[:] = [0.1, 0.2, 0.3, 0.5]
    st.exposure = 1.25
    st.sun_dir[:] = [0.0, 0.6, 0.8]
    st.up_dir[:] = [0.0, 0.0, 1.0]
    st.camera_scale, st.aperture_size, st.autofocus_depth = 2.5, 0.04, 0.7
    cam = pkg("scene").orbit_camera(33.0)
    ctx.set_session_state(st, cam, relative_focal_length=(1.3, 1.3), fov_axis=0, screen_center=(0.45, 0.55), zoom=1.5)
    p = str(tmp_path / "session.msgpack")
    ctx.save_snapshot_file(p, compress=False)
    snap = msgpack.unpackb(open(p, "rb").read(), raw=False)["snapshot"]
    assert np.allclose(snap["background_color"], [0.1, 0.2, 0.3, 0.5]) and abs(snap["exposure"] - 1.25) < 1e-7
    assert np.allclose(snap["sun_dir"], [0.0, 0.6, 0.8]) and np.allclose(snap["up_dir"], [0.0, 0.0, 1.0])
    c = snap["camera"]
    assert abs(c["scale"] - 2.5) < 1e-7 and abs(c["aperture_size"] - 0.04) < 1e-7 and abs(c["autofocus_depth"] - 0.7) < 1e-7
    assert c["fov_axis"] == 0 and abs(c["zoom"] - 1.5) < 1e-7 and np.allclose(c["screen_center"], [0.45, 0.55])
    ctx2 = native.Context(-1)
    ctx2.load_snapshot_file(p)
    s2 = ctx2.session_state()
    assert s2.valid == 1 and np.allclose(list(s2.background_color), [0.1, 0.2, 0.3, 0.5]) and abs(s2.exposure - 1.25) < 1e-7
    assert np.allclose(list(s2.sun_dir), [0.0, 0.6, 0.8]) and np.allclose(list(s2.up_dir), [0.0, 0.0, 1.0])
    assert abs(s2.camera_scale - 2.5) < 1e-7 and abs(s2.aperture_size - 0.04) < 1e-7 and abs(s2.autofocus_depth - 0.7) < 1e-7
    c2 = ctx2.snapshot_camera()
    assert np.allclose(c2["matrix"], cam, atol=1e-7) and c2["fov_axis"] == 0 and np.allclose(c2["relative_focal_length"], [1.3, 1.3]) and abs(c2["zoom"] - 1.5) < 1e-7


def test_model_validation(native, scene_mod, tmp_path):
    sc = pkg("synthetic").make_scene(aabb_scale=1, seed=5, log2_hashmap_size=12)
    ctx = native.Context(-1)
    for key, val, msg in (("aabb_scale", 3, "power of two"), ("aabb_scale", 256, "aabb_scale <= 128")):
        bad = dict(sc)
        bad[key] = val
        with pytest.raises(RuntimeError, match=msg):
            ctx.set_model(bad)
    bad = dict(sc)
    bad["rgb_network"] = dict(sc["rgb_network"], n_hidden_layers=4)
    with pytest.raises(RuntimeError, match="unsupported network architecture"):
        ctx.set_model(bad)
    bad["rgb_network"] = dict(sc["rgb_network"], n_hidden_layers=3)  # base_3layer.json: accepted, but the parameter count must follow
    with pytest.raises(RuntimeError, match="parameter count mismatch"):
        ctx.set_model(bad)
    bad["rgb_network"] = dict(sc["rgb_network"], n_hidden_layers=2)
    bad["network"] = dict(sc["network"], n_hidden_layers=0)  # a linear density head comes with a linear rgb head (configs/nerf/linear.json)
    with pytest.raises(RuntimeError, match="unsupported network architecture"):
        ctx.set_model(bad)
    # configs/nerf/linear.json and base_0layer.json: heads without a hidden layer; CutlassMLP pads the rgb output to 8 rows
    for hidden_density, n_mlp in ((0, 16 * 32 + 8 * 32), (1, 64 * 32 + 16 * 64 + 8 * 32)):
        lin = pkg("synthetic").make_scene(aabb_scale=1, seed=5, log2_hashmap_size=14, cfg=scene_mod.linear_network_config(hidden_density))
        assert sum(scene_mod.n_params(lin)[:2]) == n_mlp
        ctx.set_model(lin)
        d = ctx.get_model()
        assert (d.n_hidden_density, d.n_hidden_rgb, d.mlp_alignment, d.n_params) == (hidden_density, 0, 8, lin["params"].size)
        path = str(tmp_path / f"linear{hidden_density}.ingp")
        ctx.save_snapshot_file(path)
        ctx.load_snapshot_file(path)
        d = ctx.get_model()
        assert (d.n_hidden_density, d.n_hidden_rgb, d.mlp_alignment, d.n_params) == (hidden_density, 0, 8, lin["params"].size)
        assert np.array_equal(ctx.get_scene()["params"], lin["params"])
    ctx.close()


def test_scene_conventions(scene_mod):
    # nerf_matrix_to_ngp (nerf_loader.h:101-120) on the identity pose
    m = scene_mod.nerf_matrix_to_ngp(np.eye(4, dtype=np.float32))
    assert np.allclose(m, [[0, -1, 0, 0.5], [0, 0, -1, 0.5], [1, 0, 0, 0.5]])
    cam = scene_mod.orbit_camera(45.0)
    R = cam[:, :3]
    assert np.allclose(R.T @ R, np.eye(3), atol=1e-5)
    to_centre = np.array([0.5, 0.5, 0.5]) - cam[:, 3]
    assert np.allclose(to_centre / np.linalg.norm(to_centre), R[:, 2], atol=1e-5)  # +z looks at the scene centre
    assert abs(np.linalg.norm(to_centre) - 4.03 * 0.33) < 1e-4
    assert scene_mod.per_level_scale(16, 8, 16, "fork") == 2.0
    assert abs(scene_mod.per_level_scale(16, 8, 16, "upstream") - 2.9720) < 1e-3
    assert abs(scene_mod.per_level_scale(4, 8, 16, "upstream") - 2.4380) < 1e-3
    off, res, _ = scene_mod.grid_layout(dict(scene_mod.base_network_config()["encoding"], per_level_scale=2.0))
    assert res == [16, 32, 64, 128, 256, 512, 1024, 2048] and off[-1] == 2920448  # 23.36 MB of fp16x4


def _gloo_worker(rank, world, port, w, h, q):
    import torch
    import torch.distributed as dist

    par = pkg("parallel")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    full = torch.rand((h, w, 5), generator=g)
    # a rank "renders" only its own tiles (what ngp_render_device does with shard_index/shard_count)
    mine = torch.zeros_like(full)
    tx, ty = par.tile_grid(w, h)
    for t in par.local_tiles(w, h, rank, world).tolist():
        y0, x0 = (t // tx) * 8, (t % tx) * 8
        mine[y0:y0 + 8, x0:x0 + 8] = full[y0:y0 + 8, x0:x0 + 8]
    out = par.gather_frame(mine, w, h, rank, world)
    ok = bool(torch.equal(out, full))
    # the tile-packed path of bench.py: two gatherers used alternately (two frames in flight), one collective per frame
    gs = [par.PackedFrameGather(w, h, world, torch.device("cpu")) for _ in range(2)]
    for frame in range(3):
        g_ = gs[frame % 2]
        img_full = full[..., :4] + float(frame)
        dep_full = full[..., 4] + float(frame)
        rgba, depth = g_.buffers()
        for slot, t in enumerate(par.local_tiles(w, h, rank, world).tolist()):  # what the fused kernel writes (packed_output)
            y0, x0 = (t // tx) * 8, (t % tx) * 8
            th, tw = min(8, h - y0), min(8, w - x0)
            rgba.view(-1, 8, 8, 4)[slot, :th, :tw] = img_full[y0:y0 + th, x0:x0 + tw]
            depth.view(-1, 8, 8)[slot, :th, :tw] = dep_full[y0:y0 + th, x0:x0 + tw]
        res = g_.gather(dst=0, rank=rank)  # what bench.py does: the frame is assembled at rank 0 only
        if rank == 0:
            img, dep = res
            ok = ok and bool(torch.equal(img, img_full)) and bool(torch.equal(dep, dep_full))
        else:
            ok = ok and res is None
        img, dep = g_.gather(dst=None)  # all_gather: every rank gets the frame
        ok = ok and bool(torch.equal(img, img_full)) and bool(torch.equal(dep, dep_full))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _bcast_worker(rank, world, port, snap_path, q):
    import torch.distributed as dist

    par, native = pkg("parallel"), pkg("native")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = native.Context(-1)  # host-only: parses and validates what arrives
    nbytes = par.broadcast_snapshot(ctx, snap_path if rank == 0 else "/nonexistent/only-rank-0-has-the-file", rank)
    d = ctx.get_model()
    q.put((rank, nbytes, int(d.n_params), int(d.aabb_scale), int(d.log2_hashmap_size)))
    dist.barrier()
    dist.destroy_process_group()


def test_snapshot_broadcast_gloo_world2(tmp_path, native):
    """one-off broadcast of the model after load -- only rank 0 can read the file."""
    import torch.multiprocessing as mp

    sc = pkg("synthetic").make_scene(aabb_scale=2, seed=9, log2_hashmap_size=12)
    ctx0 = native.Context(-1)
    ctx0.set_model(sc)
    path = str(tmp_path / "model.ingp")
    ctx0.save_snapshot_file(path)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 29500 + (os.getpid() + 77) % 2000
    procs = [mpctx.Process(target=_bcast_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    size = os.path.getsize(path)
    assert res == [(0, size, sc["params"].size, 2, 12), (1, size, sc["params"].size, 2, 12)]


@pytest.mark.parametrize("w,h", [(64, 48), (101, 67)])
def test_tile_sharding_gather_gloo_world2(w, h):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + w) % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, w, h, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


def test_tile_pack_unpack_roundtrip():
    import torch

    par = pkg("parallel")
    for (w, h, world) in ((1920, 1080, 8), (37, 21, 3), (8, 8, 2)):
        img = torch.rand((h, w, 5))
        n = par.slots_per_rank(w, h, world)
        packed = torch.stack([par.pack_tiles(img, w, h, r, world, n) for r in range(world)])
        assert torch.equal(par.unpack_tiles(packed, w, h, world), img)
        assert sum(par.local_tiles(w, h, r, world).numel() for r in range(world)) == par.tile_grid(w, h)[0] * par.tile_grid(w, h)[1]


def test_constant_division_is_exact(tmp_path):
    """csrc/nerf_device.h:div_const replaces `/` by a reciprocal multiply and one FMA correction for the three
    constant divisors of the march (STEPSIZE, MAX_CONE_STEPSIZE, the dt warp): it must be the IEEE quotient for
    every float the march can produce, checked exhaustively on [1e-9, 65536) and its negatives."""
    import subprocess

    with open("/proc/cpuinfo") as f:
        has_fma = " fma " in f.read()
    exe = tmp_path / "div_const_check"
    flags = ["-O2", "-fopenmp", "-ffp-contract=off"] + (["-mfma"] if has_fma else [])
    subprocess.check_call(["gcc", *flags, "-o", str(exe), os.path.join(ROOT, "tests", "aux", "div_const_check.c"), "-lm"])
    out = subprocess.run([str(exe), "1" if has_fma else "251"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr


def test_image_decoders_against_pillow(native):
    """ngp_decode_image (what ngp_load_training_images feeds on): PNG exactly, baseline JPEG within the +-3 levels that
    separate two conforming IDCT / upsampling implementations; unsupported files are refused with a reason."""
    import io

    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    h, w = 67, 101
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 9.0), 128 + 100 * np.cos(yy / 7.0), (xx * 2 + yy * 3) % 256, 255 * (xx > yy)], -1)
    img = (img + rng.normal(0, 6, img.shape)).clip(0, 255).astype(np.uint8)

    def png(im, **kw):
        b = io.BytesIO()
        im.save(b, format="PNG", **kw)
        return b.getvalue()

    assert np.array_equal(native.decode_image(png(Image.fromarray(img))), img)  # RGBA
    rgb = native.decode_image(png(Image.fromarray(img[..., :3]), compress_level=1))
    assert np.array_equal(rgb[..., :3], img[..., :3]) and (rgb[..., 3] == 255).all()
    grey = native.decode_image(png(Image.fromarray(img[..., 0])))
    assert np.array_equal(grey[..., 0], img[..., 0]) and np.array_equal(grey[..., 1], grey[..., 2])
    la = native.decode_image(png(Image.fromarray(img[..., [0, 3]], "LA")))
    assert np.array_equal(la[..., 0], img[..., 0]) and np.array_equal(la[..., 3], img[..., 3])
    pal_im = Image.fromarray(img[..., :3]).quantize(32)
    assert np.array_equal(native.decode_image(png(pal_im))[..., :3], np.asarray(pal_im.convert("RGB")))
    g16 = (img[..., 0].astype(np.uint16) << 8) | 0x5A
    assert np.array_equal(native.decode_image(png(Image.fromarray(g16)))[..., 0], img[..., 0])  # 16-bit: the high byte
    for sub in (0, 1, 2):  # 4:4:4, 4:2:2, 4:2:0
        for extra in ({}, {"restart_marker_blocks": 5}):
            b = io.BytesIO()
            try:
                Image.fromarray(img[..., :3]).save(b, format="JPEG", quality=90, subsampling=sub, **extra)
            except TypeError:
                continue
            ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB")).astype(int)
            got = native.decode_image(b.getvalue())
            d = np.abs(got[..., :3].astype(int) - ref)
            assert got.shape == (h, w, 4) and d.max() <= 3 and d.mean() < 0.1 and (got[..., 3] == 255).all()
    b = io.BytesIO()
    Image.fromarray(img[..., 1]).save(b, format="JPEG", quality=85)
    assert np.abs(native.decode_image(b.getvalue())[..., 0].astype(int) - np.asarray(Image.open(io.BytesIO(b.getvalue()))).astype(int)).max() <= 2
    b = io.BytesIO()
    Image.fromarray(img[..., :3]).save(b, format="JPEG", progressive=True)
    with pytest.raises(RuntimeError, match="progressive"):
        native.decode_image(b.getvalue())
    with pytest.raises(RuntimeError, match="interlaced"):
        native.decode_image(_interlaced_png())
    with pytest.raises(RuntimeError, match="only PNG and JPEG"):
        native.decode_image(b"GIF89a" + b"\0" * 64)
    base = io.BytesIO()
    Image.fromarray(img[..., :3]).save(base, format="JPEG", quality=90)
    with pytest.raises(RuntimeError, match="truncated|corrupt|without image data"):
        native.decode_image(base.getvalue()[:200])


def _interlaced_png():
    import struct
    import zlib

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")


def test_resolution_comes_from_the_image_when_the_json_has_none(tmp_path, native):
    """NeRF-synthetic style transforms (camera_angle_x only): the loader reads w / h from the PNG / JPEG header."""
    Image = pytest.importorskip("PIL.Image")
    os.makedirs(tmp_path / "train")
    Image.fromarray(np.zeros((30, 40, 4), np.uint8)).save(tmp_path / "train" / "r_0.png")
    Image.fromarray(np.zeros((30, 40, 3), np.uint8)).save(tmp_path / "train" / "r_1.jpg", quality=80)
    frames = [{"file_path": "./train/r_0", "transform_matrix": np.eye(4).tolist()}, {"file_path": "./train/r_1.jpg", "transform_matrix": np.eye(4).tolist()}]
    (tmp_path / "transforms_train.json").write_text(json.dumps({"camera_angle_x": 0.6911, "frames": frames}))
    ctx = native.Context(-1)
    ctx.load_training_data(str(tmp_path / "transforms_train.json"))
    for i in range(2):
        v = ctx.training_view(i)
        assert v["resolution"].tolist() == [40, 30] and np.allclose(v["focal_length"], 0.5 * 40 / np.tan(0.5 * 0.6911), rtol=1e-5)
    frames.append({"file_path": "./train/missing", "transform_matrix": np.eye(4).tolist()})
    (tmp_path / "transforms_train.json").write_text(json.dumps({"camera_angle_x": 0.6911, "frames": frames}))
    with pytest.raises(RuntimeError, match="cannot be read"):
        ctx.load_training_data(str(tmp_path / "transforms_train.json"))
    ctx.close()


def test_parsers_survive_mutated_files(tmp_path, native):
    """tests/aux/parse_fuzz.cpp under AddressSanitizer + UBSan: the JSON and msgpack parsers (csrc/minijson.h) on truncations, byte flips,
    insertions and deep nesting of a transforms.json and of a snapshot -- parsed or refused, no crash (2 M nested brackets used to overflow the stack)."""
    import subprocess

    exe = str(tmp_path / "pfuzz")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux", "parse_fuzz.cpp")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all", src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    frames = [{"file_path": f"./images/{i:04d}.jpg", "sharpness": 30.5 + i, "transform_matrix": np.eye(4).tolist()} for i in range(6)]
    (tmp_path / "seed.json").write_text(json.dumps({"camera_angle_x": 0.6911, "fl_x": 1375.5, "k1": 0.03, "aabb_scale": 4, "w": 1080.0, "h": 1920.0, "frames": frames}, indent=2))
    ctx = native.Context(-1)
    ctx.set_model(pkg("synthetic").make_scene(aabb_scale=1, seed=5, log2_hashmap_size=8))
    ctx.save_snapshot_file(str(tmp_path / "seed.msgpack"), compress=False)
    ctx.close()
    (tmp_path / "deep.json").write_text("[" * 2000000)
    (tmp_path / "deep.msgpack").write_bytes(b"\x91" * 2000000)
    for seed, n in (("seed.json", 2000), ("seed.msgpack", 120), ("deep.json", 1), ("deep.msgpack", 1)):
        r = subprocess.run([exe, str(tmp_path / seed), str(n)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "parsed" in r.stdout, (seed, r.stdout[-300:], r.stderr[-800:])


def test_snapshot_loader_survives_mutated_snapshots(tmp_path, native):
    """Testbed::load_snapshot's counterpart on hostile files: every field of a valid snapshot replaced by values of other types and sizes, keys
    removed, binary blobs shortened, the byte stream truncated and flipped -- the loader either loads the file or throws; the context stays usable."""
    import msgpack

    ctx = native.Context(-1)
    sc = pkg("synthetic").make_scene(aabb_scale=2, seed=5, log2_hashmap_size=8)
    ctx.set_model(sc)
    p = str(tmp_path / "seed.msgpack")
    ctx.save_snapshot_file(p, compress=False)
    blob = open(p, "rb").read()
    root = msgpack.unpackb(blob, raw=False)
    weird = [None, True, -1, 0, 1, 3, 2 ** 31, 2 ** 40, -2 ** 40, 1e30, float("nan"), float("inf"), "", "x" * 70000, b"", b"\x00" * 7, [], [1, 2, 3], {}, {"a": {"b": [None]}}]

    def paths(node, prefix=()):
        if isinstance(node, dict):
            for k, v in node.items():
                yield prefix + (k,)
                yield from paths(v, prefix + (k,))
        elif isinstance(node, list) and len(node) <= 16:
            for i, v in enumerate(node):
                yield prefix + (i,)
                yield from paths(v, prefix + (i,))

    def mutated(path, value, delete=False):
        import copy

        r = copy.copy(root)
        node = r
        for k in path[:-1]:
            child = copy.copy(node[k])
            node[k] = child
            node = child
        if delete:
            if isinstance(node, dict):
                del node[path[-1]]
            else:
                node.pop(path[-1])
        else:
            node[path[-1]] = value
        return msgpack.packb(r, use_bin_type=True)

    rng = np.random.default_rng(3)
    all_paths = list(paths(root))
    assert len(all_paths) > 60
    n_ok = n_refused = 0
    for path in all_paths:
        variants = [mutated(path, None, delete=True)] + [mutated(path, weird[int(i)]) for i in rng.choice(len(weird), 5, replace=False)]
        leaf = root
        for k in path:
            leaf = leaf[k]
        if isinstance(leaf, (bytes, str)) and len(leaf) > 4:  # a blob one element short, and one element long
            variants += [mutated(path, leaf[:-2]), mutated(path, leaf + leaf[:2])]
        for data in variants:
            try:
                ctx.load_snapshot_bytes(data)
                n_ok += 1
            except RuntimeError:
                n_refused += 1
    for _ in range(150):  # the raw stream: truncations and flips in the header region
        b = bytearray(blob)
        if rng.random() < 0.4:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            for _k in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, min(len(b), 3000)))] = int(rng.integers(0, 256))
        try:
            ctx.load_snapshot_bytes(bytes(b))
            n_ok += 1
        except RuntimeError:
            n_refused += 1
    assert n_refused > 100 and n_ok > 20, (n_ok, n_refused)
    ctx.load_snapshot_bytes(blob)  # and the context still takes the intact file
    assert ctx.get_model().n_params == sc["params"].size
    ctx.close()


def test_dataset_loader_survives_mutated_transforms(tmp_path, native):
    """Testbed::load_training_data's counterpart on hostile transforms.json files: every field replaced by values of other types and ranges
    (negative / huge / NaN resolutions and intrinsics, matrices of the wrong shape, missing keys): loaded or refused, the context stays usable."""
    Image = pytest.importorskip("PIL.Image")
    os.makedirs(tmp_path / "images")
    for i in range(2):
        Image.fromarray(np.full((12, 16, 3), 40 * i + 20, np.uint8)).save(tmp_path / "images" / f"{i:04d}.png")
    good = {"camera_angle_x": 0.69, "fl_x": 20.0, "fl_y": 20.0, "cx": 8.0, "cy": 6.0, "w": 16, "h": 12, "k1": 0.01, "k2": 0.0, "p1": 0.0, "p2": 0.0, "aabb_scale": 2, "scale": 0.33,
            "offset": [0.5, 0.5, 0.5], "enable_depth_loading": False, "white_transparent": False, "rolling_shutter": [0, 0, 0, 0],
            "frames": [{"file_path": f"./images/{i:04d}.png", "sharpness": 10.0, "transform_matrix": np.eye(4).tolist()} for i in range(2)]}
    weird = [None, True, -1, 0, 3, 2 ** 31, 2 ** 40, -2 ** 40, 1e30, -1e30, 1e-30, "", "x", [], [1, 2, 3], [[1, 2], [3]], {}, {"a": 1}]
    ctx = native.Context(-1)
    path = str(tmp_path / "transforms.json")
    n_ok = n_refused = 0

    def attempt(obj, raw=None):
        nonlocal n_ok, n_refused
        open(path, "w").write(raw if raw is not None else json.dumps(obj))
        try:
            ctx.load_training_data(path)
            n_ok += 1
        except RuntimeError:
            n_refused += 1

    for key in list(good):
        for v in weird:
            attempt(dict(good, **{key: v}))
        attempt({k: val for k, val in good.items() if k != key})
    for key in list(good["frames"][0]):
        for v in weird:
            fr = [dict(good["frames"][0], **{key: v}), good["frames"][1]]
            attempt(dict(good, frames=fr))
    for raw in ("", "{", "[]", "null", json.dumps(good)[:-20], json.dumps(good).replace("0.69", "NaN"), json.dumps(good).replace("16", "1e400"), "{\"frames\": " + "[" * 100000):
        attempt(None, raw)
    assert n_refused > 40 and n_ok > 40, (n_ok, n_refused)
    attempt(good)
    assert ctx.training_view(1)["resolution"].tolist() == [16, 12]
    ctx.close()


def test_image_decoders_survive_mutated_files(tmp_path):
    """tests/aux/decode_fuzz.cpp under AddressSanitizer + UBSan: truncations, byte flips, 0xFF runs and insertions of
    PNG / JPEG seeds either decode or are refused -- no crash, no hang (dataset images are untrusted input)."""
    import subprocess

    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 255, (40, 56, 4)).astype(np.uint8)
    Image.fromarray(img).save(tmp_path / "seed.png")
    Image.fromarray(img[..., :3]).save(tmp_path / "seed420.jpg", quality=85, subsampling=2)
    try:
        Image.fromarray(img[..., :3]).save(tmp_path / "seed444.jpg", quality=95, subsampling=0, restart_marker_blocks=3)
    except TypeError:
        Image.fromarray(img[..., :3]).save(tmp_path / "seed444.jpg", quality=95, subsampling=0)
    exe = str(tmp_path / "fuzz")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux", "decode_fuzz.cpp")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", src, "-o", exe, "-lz"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for seed in ("seed.png", "seed420.jpg", "seed444.jpg"):
        r = subprocess.run([exe, str(tmp_path / seed), "1500"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "decoded" in r.stdout, (seed, r.stdout[-300:], r.stderr[-800:])
