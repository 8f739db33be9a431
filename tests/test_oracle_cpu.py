"""CPU tests of the oracle: committed golden vectors, fp16 emulation, and an independent numpy restatement of the
hash-grid / SH / MLP arithmetic (the oracle is PARITY UNPINNED by the reference, so it is pinned here against a
second, differently written implementation and against size-independent properties)."""
import hashlib
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=[1, 2])
def golden(request):
    """v1: the pre-tvec corner sum of the grid encoding ("legacy"), v2: the tvec-era fma sum (tests/golden/make_golden.py)"""
    g = dict(np.load(os.path.join(HERE, "golden", "nerf_unit_v%d.npz" % request.param)))
    g["grid_accumulate"] = {1: "legacy", 2: "fma"}[request.param]
    return g


@pytest.fixture(scope="module")
def golden_inputs(oracle):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    sc, pos, dir01, cam_matrix, focal = mg.build_inputs()
    grid = np.asarray(sc["density_grid"], np.float16).astype(np.float32)
    sc["density_grid_bitfield"], sc["density_grid_mean"] = oracle.density_grid_to_bitfield(grid, sc["max_cascade"])
    return sc, pos, dir01, cam_matrix, focal, mg


def test_fixture_inputs_are_reproducible(golden, golden_inputs):
    sc = golden_inputs[0]
    sha = hashlib.sha256(np.ascontiguousarray(sc["params"]).tobytes()).digest()
    assert np.array_equal(np.frombuffer(sha, np.uint8), golden["params_sha256"]), "seeded parameter stream changed (numpy version?)"
    sha = hashlib.sha256(np.asarray(sc["density_grid"], np.float16).tobytes()).digest()
    assert np.array_equal(np.frombuffer(sha, np.uint8), golden["density_grid_sha256"])


def test_oracle_matches_golden_vectors(oracle, golden, golden_inputs):
    sc, pos, dir01, cam_matrix, focal, mg = golden_inputs
    sc = dict(sc, grid_accumulate=golden["grid_accumulate"])
    assert np.array_equal(pos, golden["pos"]) and np.array_equal(dir01, golden["dir01"])
    sha = hashlib.sha256(sc["density_grid_bitfield"].tobytes()).digest()
    assert np.array_equal(np.frombuffer(sha, np.uint8), golden["bitfield_sha256"])
    assert np.float32(sc["density_grid_mean"]) == golden["bitfield_mean"]
    m = oracle.make_model(sc)
    assert np.array_equal(oracle.grid_encode(m, pos).view(np.uint16), golden["enc"])
    assert np.array_equal(oracle.sh4(dir01).view(np.uint16), golden["sh"])
    assert np.array_equal(oracle.network(m, pos, dir01).view(np.uint16), golden["net"])
    cam = oracle.make_camera(cam_matrix, mg.W, mg.H, focal)
    assert np.array_equal(oracle.init_rays(m, cam).view(np.uint8), golden["payloads"])
    fb, db, st = oracle.render_nerf(m, cam)
    assert np.array_equal(fb, golden["frame"]) and np.array_equal(db, golden["depth"])
    assert [st["n_rays"], st["n_rays_alive_after_init"], st["n_rays_hit"], st["n_samples"]] == golden["stats"].tolist()
    vals = np.array([oracle.ld_random_val(i, s) for i in range(8) for s in (0, 786433, 0xdeadbeef)], np.float32)
    assert np.array_equal(vals, golden["ld_vals"])
    assert np.array_equal(np.stack([oracle.pixel_offset(s) for s in range(6)]), golden["pixel_offsets"])
    oracle.release(m)


def test_fp16_emulation_matches_ieee(oracle):
    import ctypes as C

    # the emulation lives in a header; exercise it through the SH encoder's float->half and the grid's half->float
    lib = oracle.lib
    # 1) every binary16 value survives half->float->half: build a 1-level dense grid holding all 65536 patterns
    # (done implicitly by test_grid_encode_against_numpy); here: float->half rounding on random floats vs numpy
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.normal(scale=s, size=4000) for s in (1e-7, 1e-4, 1.0, 100.0, 3e4)]).astype(np.float32)
    x = np.concatenate([x, np.float32([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, 5.96e-8, 2.98e-8, 2.9802325e-8, 6.1e-5, np.inf, -np.inf])])
    # orc_sh4: out[3] = -0.4886..*x with dir01 -> use the dedicated rounding path instead: SH coefficient 2 is 0.4886*z
    # simpler and exact: coefficient 0 is a constant, so test conversion through the pixel path is not possible;
    # use the exported tonemap? No -- call the static inline via a tiny shim compiled on the fly.
    import subprocess, tempfile, textwrap
    src = textwrap.dedent('''
        #include "orc_common.h"
        void conv(int n, const float* in, unsigned short* out) { for (int i = 0; i < n; ++i) out[i] = orc_float_to_half(in[i]); }
        void back(int n, const unsigned short* in, float* out) { for (int i = 0; i < n; ++i) out[i] = orc_half_to_float(in[i]); }
        void add(int n, const unsigned short* a, const unsigned short* b, unsigned short* out) { for (int i = 0; i < n; ++i) out[i] = orc_half_add(a[i], b[i]); }
        void fma3(int n, const unsigned short* a, const unsigned short* b, const unsigned short* c, unsigned short* out) { for (int i = 0; i < n; ++i) out[i] = orc_half_fma(a[i], b[i], c[i]); }
    ''')
    with tempfile.TemporaryDirectory() as td:
        cpath = os.path.join(td, "shim.c")
        open(cpath, "w").write(src)
        so = os.path.join(td, "shim.so")
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(os.path.dirname(HERE), "oracle"), cpath, "-o", so, "-lm"], check=True)
        shim = C.CDLL(so)
        out = np.zeros(x.size, np.uint16)
        shim.conv(x.size, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        with np.errstate(over="ignore"):
            ref = x.astype(np.float16).view(np.uint16)
        assert np.array_equal(out, ref)
        allh = np.arange(65536, dtype=np.uint16)
        f = np.zeros(65536, np.float32)
        shim.back(65536, allh.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        reff = allh.view(np.float16).astype(np.float32)
        assert np.array_equal(f.view(np.uint32)[~np.isnan(reff)], reff.view(np.uint32)[~np.isnan(reff)])
        a = rng.integers(0, 0x7c00, 200000).astype(np.uint16) | (rng.integers(0, 2, 200000).astype(np.uint16) << 15)
        b = rng.integers(0, 0x7c00, 200000).astype(np.uint16) | (rng.integers(0, 2, 200000).astype(np.uint16) << 15)
        s = np.zeros(200000, np.uint16)
        shim.add(200000, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
        with np.errstate(over="ignore"):
            refs = (a.view(np.float16).astype(np.float64) + b.view(np.float16).astype(np.float64)).astype(np.float16)
        assert np.array_equal(s.view(np.float16).astype(np.float32), refs.astype(np.float32))
        # single-rounding fma (the tvec-era corner sum) against exact integer arithmetic: random operands below 64, plus ties
        # -- products that land exactly on an fp16 rounding boundary and an addend far too small to show in a double sum
        n = 300000
        def rnd(n):  # magnitudes in [2^-24, 64): exponent field 0..20
            return (rng.integers(0, 21 << 10, n).astype(np.uint16) | (rng.integers(0, 2, n).astype(np.uint16) << 15))
        fa, fb, fc = rnd(n), rnd(n), rnd(n)
        # ties: a = 1 + 2^-10 k (11 bits), b = 1 + 2^-1 -> a*b has 12+ bits; c = +-2^-24 nudges the tie either way
        k = rng.integers(0, 1024, 4096)
        ta = np.float16(1.0) + (k * 2.0 ** -10).astype(np.float16)
        tb = np.full(4096, 1.5, np.float16)
        tc = np.where(rng.integers(0, 2, 4096) == 0, np.float16(2.0 ** -24), np.float16(-(2.0 ** -24))).astype(np.float16)
        tc[::3] = 0
        fa = np.concatenate([fa, ta.view(np.uint16), (ta * np.float16(32)).view(np.uint16)])
        fb = np.concatenate([fb, tb.view(np.uint16), tb.view(np.uint16)])
        fc = np.concatenate([fc, tc.view(np.uint16), tc.view(np.uint16)])
        out = np.zeros(fa.size, np.uint16)
        shim.fma3(fa.size, fa.ctypes.data_as(C.c_void_p), fb.ctypes.data_as(C.c_void_p), fc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        ref = _half_fma_exact(fa.view(np.float16), fb.view(np.float16), fc.view(np.float16))
        assert np.array_equal(out.view(np.float16).astype(np.float32), ref.astype(np.float32))
        # larger magnitudes (product ~2^12 on or next to a tie, addends from 2^-24 to 1): checked with Fractions
        from fractions import Fraction
        import math

        def round_half(x):
            if x == 0:
                return 0.0
            sgn, x = (-1.0, -x) if x < 0 else (1.0, x)
            e = math.floor(math.log2(float(x)))
            while Fraction(2) ** e > x:
                e -= 1
            while Fraction(2) ** (e + 1) <= x:
                e += 1
            quantum = Fraction(2) ** max(e - 10, -24)
            q = x / quantum
            fl = q.numerator // q.denominator
            rem = q - fl
            if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1):
                fl += 1
            v = float(fl * quantum)
            return sgn * (math.inf if v >= 65520.0 else v)

        ks = rng.integers(0, 1024, 600)
        wa = (np.float16(2.0 ** 6) * (np.float16(1.0) + (ks * 2.0 ** -10).astype(np.float16))).astype(np.float16)
        wb = np.full(600, 48.0, np.float16)  # 1.5 * 2^5: the product sits on or next to a tie of the 2^11..2^13 binades
        wc = np.tile(np.array([2.0 ** -24, -(2.0 ** -24), 0.0, 3.0 * 2.0 ** -24, -(2.0 ** -14), 1.0], np.float16), 100)
        wa[1::2] = -wa[1::2]
        out = np.zeros(600, np.uint16)
        shim.fma3(600, wa.ctypes.data_as(C.c_void_p), wb.ctypes.data_as(C.c_void_p), wc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        want = np.array([round_half(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))) for a, b, c in zip(wa, wb, wc)], np.float32)
        assert np.array_equal(out.view(np.float16).astype(np.float32), want)


def _half_fma_exact(a16, b16, c16):
    """fma(a, b, c) on binary16 with ONE rounding, in exact integer arithmetic (a second implementation next to the
    oracle's TwoSum one): every finite half is a multiple of 2^-24, so a*b + c is N * 2^-48 with N an int64 as long as
    the magnitudes stay below 64 (asserted). Round-half-even of N to the fp16 grid of its binade."""
    def to_int(h):  # h * 2^24 as int64 (exact)
        v = h.astype(np.float64) * 2.0 ** 24
        assert np.all(v == np.rint(v))
        return v.astype(np.int64)
    a, b, c = to_int(a16), to_int(b16), to_int(c16)
    assert np.abs(a).max(initial=0) < 2 ** 30 and np.abs(b).max(initial=0) < 2 ** 30 and np.abs(c).max(initial=0) < 2 ** 30
    n = a * b + (c << 24)
    sign = np.where(n < 0, -1, 1).astype(np.int64)
    mag = np.abs(n)
    # floor(log2 mag): bit length - 1 (0 for mag 0)
    bl = np.zeros(mag.shape, np.int64)
    t = mag.copy()
    for sh in (32, 16, 8, 4, 2, 1):
        m = t >= (np.int64(1) << sh)
        bl = np.where(m, bl + sh, bl)
        t = np.where(m, t >> sh, t)
    e = bl - 48                                   # value in [2^e, 2^(e+1))
    q_exp = np.maximum(e - 10, -24)               # fp16 quantum of that binade (subnormals: 2^-24)
    shift = q_exp + 48                            # >= 24
    half = np.int64(1) << (shift - 1)
    q = mag >> shift
    rem = mag & ((np.int64(1) << shift) - 1)
    q = q + ((rem > half) | ((rem == half) & ((q & 1) == 1))).astype(np.int64)
    val = sign.astype(np.float64) * q.astype(np.float64) * np.exp2(q_exp.astype(np.float64))
    with np.errstate(over="ignore"):
        return val.astype(np.float16)  # exact: val is already on the fp16 grid (or beyond the range -> inf)


def _grid_encode_numpy(sc, scene_mod, pos, scales=None, mode="fma"):
    """Independent restatement of tcnn's kernel_grid with numpy float16 arithmetic, in both published corner sums
    (oracle.h orc_nerf_model::grid_accumulate): "fma" = result = fma((T)weight, val, result), "legacy" =
    result[f] += (T)(weight * (float)val[f])."""
    enc = sc["encoding"]
    F = enc["n_features_per_level"]
    offsets, resolutions, py_scales = scene_mod.grid_layout(enc)
    scales = py_scales if scales is None else scales
    nd, nr, ng = scene_mod.n_params(sc)
    table = sc["params"][nd + nr:].view(np.float16)
    out = np.zeros((pos.shape[0], enc["n_levels"] * F), np.float16)
    for l in range(enc["n_levels"]):
        size = offsets[l + 1] - offsets[l]
        lvl = table[offsets[l] * F:offsets[l + 1] * F].reshape(size, F)
        p = (np.float64(np.float32(scales[l])) * pos.astype(np.float64) + 0.5).astype(np.float32)  # fmaf: single rounding
        fl = np.floor(p)
        w = (p - fl).astype(np.float32)
        g = fl.astype(np.int64).astype(np.uint32)
        res = np.uint32(resolutions[l])
        acc = np.zeros((pos.shape[0], F), np.float16)
        for c in range(8):
            wt = np.ones(pos.shape[0], np.float32)
            gl = []
            for d in range(3):
                if c & (1 << d):
                    wt = (wt * w[:, d]).astype(np.float32)
                    gl.append(g[:, d] + np.uint32(1))
                else:
                    wt = (wt * (np.float32(1) - w[:, d])).astype(np.float32)
                    gl.append(g[:, d])
            if res.astype(np.uint64) ** 3 <= size:
                idx = gl[0] + gl[1] * res + gl[2] * res * res
            else:
                idx = gl[0] ^ (gl[1] * np.uint32(2654435761)) ^ (gl[2] * np.uint32(805459861))
            idx = idx % np.uint32(size)
            if mode == "fma":
                wh = np.broadcast_to(wt.astype(np.float16)[:, None], acc.shape)
                acc = _half_fma_exact(wh, lvl[idx], acc)
            else:
                prod = (wt[:, None] * lvl[idx].astype(np.float32)).astype(np.float32).astype(np.float16)
                acc = (acc.astype(np.float64) + prod.astype(np.float64)).astype(np.float16)
        out[:, l * F:(l + 1) * F] = acc
    return out


@pytest.mark.parametrize("mode", ["fma", "legacy"])
@pytest.mark.parametrize("which", ["unit", "big"])
def test_grid_encode_against_numpy(which, mode, oracle, scene_mod, scene_unit, scene_big):
    sc = dict(scene_unit if which == "unit" else scene_big, grid_accumulate=mode)
    m = oracle.make_model(sc)
    rng = np.random.default_rng(9)
    pos = rng.uniform(0, 1, (3000, 3)).astype(np.float32)
    pos[:3] = [[0, 0, 0], [1, 1, 1], [1, 0, 0.5]]
    # the per-level scale exp2f(l * log2f(b)) * N_min - 1 is evaluated by the C library (glibc here, as in the HIP
    # build's host code); numpy's float32 exp2/log2 can differ by an ulp, so the restatement takes the scales as data
    off, res, scl = oracle.grid_layout(m)
    with np.errstate(over="ignore"):
        ref = _grid_encode_numpy(sc, scene_mod, pos, scl, mode)
    got = oracle.grid_encode(m, pos)
    assert np.array_equal(got.astype(np.float32), ref.astype(np.float32))
    # layout agrees with the host-side table used for sizing
    o2, r2, s2 = scene_mod.grid_layout(sc["encoding"])
    assert off.tolist() == o2 and res.tolist() == r2 and np.allclose(scl, s2, rtol=1e-6)
    oracle.release(m)


def test_network_against_numpy_float64(oracle, scene_mod, scene_unit):
    sc = scene_unit
    m = oracle.make_model(sc)
    rng = np.random.default_rng(10)
    pos = rng.uniform(0.2, 0.8, (512, 3)).astype(np.float32)
    d = rng.normal(size=(512, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dir01 = ((d + 1) * 0.5).astype(np.float32)
    enc = oracle.grid_encode(m, pos).astype(np.float64)
    sh = oracle.sh4(dir01).astype(np.float64)
    p = sc["params"].view(np.float16).astype(np.float64)
    W0, W1 = p[:2048].reshape(64, 32), p[2048:3072].reshape(16, 64)
    R0, R1, R2 = p[3072:5120].reshape(64, 32), p[5120:9216].reshape(64, 64), p[9216:10240].reshape(16, 64)

    def h(x):
        return x.astype(np.float32).astype(np.float16).astype(np.float64)

    dens = h(h(np.maximum(enc @ W0.T, 0)) @ W1.T)
    rin = np.concatenate([dens, sh], axis=1)
    rgb = h(h(np.maximum(h(np.maximum(rin @ R0.T, 0)) @ R1.T, 0)) @ R2.T)
    ref = np.concatenate([rgb[:, :3], dens[:, :1]], axis=1)
    got = oracle.network(m, pos, dir01).astype(np.float64)
    assert np.array_equal(got, ref)
    # real spherical harmonics: band energies of a unit vector are rotation invariant, sum_m Y_lm^2 = (2l+1)/(4 pi)
    shf = oracle.sh4(dir01).astype(np.float64)
    for l, (a, b) in enumerate(((0, 1), (1, 4), (4, 9), (9, 16))):
        assert np.allclose((shf[:, a:b] ** 2).sum(1), (2 * l + 1) / (4 * np.pi), rtol=4e-3)
    oracle.release(m)


@pytest.mark.parametrize("hidden_density", [0, 1])
def test_heads_without_a_hidden_layer_against_numpy(hidden_density, oracle, scene_mod):
    """configs/nerf/linear.json / base_0layer.json: tcnn's CutlassMLP with n_hidden_layers 0 is one (padded output) x (input) matrix
    (rgb: 8 x 32), output_activation None. Both accumulation modes against numpy, bit for bit."""
    from conftest import _with_bitfield, pkg

    sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=9, log2_hashmap_size=12, cfg=scene_mod.linear_network_config(hidden_density)))
    rng = np.random.default_rng(4)
    pos = rng.uniform(0.1, 0.9, (512, 3)).astype(np.float32)
    d = rng.normal(size=(512, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dir01 = ((d + 1) * 0.5).astype(np.float32)
    p = sc["params"].view(np.float16).astype(np.float64)
    nd, nr, _ = scene_mod.n_params(sc)
    assert nr == 8 * 32
    R = p[nd:nd + nr].reshape(8, 32)

    def layer(x, W, relu, k_block):
        acc = np.zeros((x.shape[0], W.shape[0]), np.float16)
        for k0 in range(0, W.shape[1], k_block):  # k_block = the whole row: one exact sum, rounded to fp32 and then to fp16
            acc = (acc.astype(np.float64) + x[:, k0:k0 + k_block] @ W[:, k0:k0 + k_block].T).astype(np.float32).astype(np.float16) if k_block == W.shape[1] else \
                  (acc.astype(np.float64) + x[:, k0:k0 + k_block] @ W[:, k0:k0 + k_block].T).astype(np.float16)
        a = acc.astype(np.float64)
        return np.maximum(a, 0) if relu else a

    for mode, kb in (("exact", None), ("fp16_k16", 16)):
        sc["mlp_accumulate"] = mode
        m = oracle.make_model(sc)
        enc = oracle.grid_encode(m, pos).astype(np.float64)
        sh = oracle.sh4(dir01).astype(np.float64)
        if hidden_density == 0:
            dens = layer(enc, p[:512].reshape(16, 32), False, kb or 32)
        else:
            dens = layer(layer(enc, p[:2048].reshape(64, 32), True, kb or 32), p[2048:3072].reshape(16, 64), False, kb or 64)
        rgb = layer(np.concatenate([dens, sh], axis=1), R, False, kb or 32)
        ref = np.concatenate([rgb[:, :3], dens[:, :1]], axis=1)
        got = oracle.network(m, pos, dir01).astype(np.float64)
        oracle.release(m)
        assert np.array_equal(got, ref), (mode, np.abs(got - ref).max())


def test_mlp_accumulation_modes_against_numpy(oracle, scene_mod, scene_unit):
    """The two modes that bracket / measure the reference's fp16-accumulating FullyFusedMLP (oracle.h, orc_nerf_model::mlp_accumulate):
    "fp16_k16" (the running sum rounded to fp16 after every 16-wide block of the inner dimension) against an independent numpy
    restatement, bit for bit; "ideal" (float64, no intermediate rounding) against float64 numpy on the oracle's own encodings being
    replaced by float64 interpolation -- checked through its distance from the default mode, which is bounded by fp16 rounding."""
    rng = np.random.default_rng(11)
    pos = rng.uniform(0.2, 0.8, (384, 3)).astype(np.float32)
    d = rng.normal(size=(384, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dir01 = ((d + 1) * 0.5).astype(np.float32)
    sc = dict(scene_unit)
    m_exact = oracle.make_model(sc)
    enc = oracle.grid_encode(m_exact, pos).astype(np.float64)
    sh = oracle.sh4(dir01).astype(np.float64)
    p = sc["params"].view(np.float16).astype(np.float64)
    W0, W1 = p[:2048].reshape(64, 32), p[2048:3072].reshape(16, 64)
    R0, R1, R2 = p[3072:5120].reshape(64, 32), p[5120:9216].reshape(64, 64), p[9216:10240].reshape(16, 64)

    def layer_k16(x, W, relu):  # x: [n, k] float64 holding fp16 values
        acc = np.zeros((x.shape[0], W.shape[0]), np.float16)
        for k0 in range(0, W.shape[1], 16):
            acc = (acc.astype(np.float64) + x[:, k0:k0 + 16] @ W[:, k0:k0 + 16].T).astype(np.float16)  # one rounding per block (float64 -> fp16 is a single RN)
        a = acc.astype(np.float64)
        return np.maximum(a, 0) if relu else a

    dens = layer_k16(layer_k16(enc, W0, True), W1, False)
    rin = np.concatenate([dens, sh], axis=1)
    rgb = layer_k16(layer_k16(layer_k16(rin, R0, True), R1, True), R2, False)
    ref = np.concatenate([rgb[:, :3], dens[:, :1]], axis=1)
    sc["mlp_accumulate"] = "fp16_k16"
    m_k16 = oracle.make_model(sc)
    got_k16 = oracle.network(m_k16, pos, dir01).astype(np.float64)
    assert np.array_equal(got_k16, ref)
    got_exact = oracle.network(m_exact, pos, dir01).astype(np.float64)
    sc["mlp_accumulate"] = "ideal"
    m_ideal = oracle.make_model(sc)
    got_ideal = oracle.network(m_ideal, pos, dir01).astype(np.float64)  # (this interface rounds the float64 logits to fp16 once)
    # float64 everything: a numpy restatement of the MLPs on float64 interpolation is what "ideal" is; here the cheap property --
    # both fp16 pipelines scatter around it, fp16 accumulation further than exact sums
    e_exact, e_k16 = np.abs(got_exact - got_ideal), np.abs(got_k16 - got_ideal)
    assert e_exact.max() < 0.05 and e_k16.max() < 0.1, (e_exact.max(), e_k16.max())
    assert np.sqrt((e_exact ** 2).mean()) <= np.sqrt((e_k16 ** 2).mean())
    assert not np.array_equal(got_k16, got_exact)
    for m in (m_exact, m_k16, m_ideal):
        oracle.release(m)


def test_sampling_sequences(oracle):
    # Owen-scrambled Sobol: values in [0,1], deterministic, and (0,2)-stratified: any 2^k consecutive indices hit
    # every dyadic interval of length 2^-k once
    for seed in (0, 786433 * 5, 0xdeadbeef):
        v = np.array([oracle.ld_random_val(i, seed) for i in range(64)])
        assert (v >= 0).all() and (v <= 1).all()
        for k in (3, 4, 5, 6):
            assert sorted(np.minimum((v[: 1 << k] * (1 << k)).astype(int), (1 << k) - 1).tolist()) == list(range(1 << k))
    off = oracle.pixel_offset(0)
    assert np.allclose(off, 0.5, atol=1e-6)  # sample 0 / snap_to_pixel_centers is the pixel centre
    assert not np.allclose(oracle.pixel_offset(1), 0.5)


def test_bitfield_properties(oracle, scene_mod, scene_big):
    sc = scene_big
    grid = np.asarray(sc["density_grid"], np.float16).astype(np.float32)
    bf = sc["density_grid_bitfield"]
    n = 128 ** 3
    assert bf.size == n // 8 * 8
    bits = np.unpackbits(bf, bitorder="little").reshape(8, n)
    thresh = min(0.01, sc["density_grid_mean"])
    # level 0 is the thresholded grid
    assert np.array_equal(bits[0], (grid[:n] > thresh).astype(np.uint8))
    # max-pool: a set cell at level k implies its parent cell at level k+1 is set (parent = centred half)
    idx = np.arange(128, dtype=np.uint32)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    mort = scene_mod.morton3d(X.ravel(), Y.ravel(), Z.ravel())
    parent = scene_mod.morton3d(X.ravel() // 2 + 32, Y.ravel() // 2 + 32, Z.ravel() // 2 + 32)
    for k in range(7):
        child_set = bits[k][mort] == 1
        assert bits[k + 1][parent][child_set].all()
    assert bits[7].any()
    # an empty grid gives an empty bitfield
    bf0, mean0 = oracle.density_grid_to_bitfield(np.zeros(n, np.float32), 0)
    assert mean0 == 0.0 and not bf0.any()


def test_render_properties(oracle, scene_mod, scene_unit):
    sc = dict(scene_unit)
    m = oracle.make_model(sc)
    w, h = 40, 24
    cam = oracle.make_camera(scene_mod.orbit_camera(45.0), w, h, scene_mod.focal_from_fov_x(w, 0.6911))
    fb, db, st = oracle.render_nerf(m, cam)
    assert st["n_rays_hit"] > 0
    a = fb[..., 3]
    assert (a >= 0).all() and (a <= 1.0 + 1e-6).all()
    assert ((a > 0.989) | (a < 0.99)).all()
    # alpha == 1 exactly where the ray terminated early (rgba /= a); premultiplied colours never exceed alpha
    assert (fb[..., :3].max(-1) <= a + 1e-5).all()
    assert np.array_equal(db[a == 0], np.full((a == 0).sum(), 16384.0, np.float32))
    # stricter transmittance threshold -> at least as many samples
    fb2, _, st2 = oracle.render_nerf(m, cam, oracle.make_opts(min_transmittance=1e-4))
    assert st2["n_samples"] > st["n_samples"] and st2["n_rays_hit"] == st["n_rays_hit"]
    # threads do not change the result
    fb3, db3, _ = oracle.render_nerf(m, cam, oracle.make_opts(n_threads=1))
    assert np.array_equal(fb, fb3) and np.array_equal(db, db3)
    # empty occupancy -> nothing is rendered
    sc["density_grid_bitfield"] = np.zeros_like(sc["density_grid_bitfield"])
    m2 = oracle.make_model(sc)
    fb4, _, st4 = oracle.render_nerf(m2, cam)
    assert st4["n_samples"] == 0 and not fb4.any()
    oracle.release(m)
    oracle.release(m2)


def test_post_pass(oracle):
    rng = np.random.default_rng(2)
    fb = rng.uniform(0, 1, (50, 4)).astype(np.float32)
    acc = oracle.accumulate(fb, np.zeros_like(fb), 0)
    assert np.array_equal(acc, fb)
    acc2 = oracle.accumulate(np.zeros_like(fb), acc, 1)
    assert np.allclose(acc2, fb / 2)
    out = oracle.tonemap(acc, (1.0, 1.0, 1.0, 1.0), 0.0, False)
    assert np.allclose(out[:, 3], 1.0) and np.allclose(out[:, :3], fb[:, :3] + (1 - fb[:, 3:4]), atol=1e-6)
    srgb = oracle.tonemap(acc, (0, 0, 0, 0), 1.0, True)
    lin = oracle.tonemap(acc, (0, 0, 0, 0), 1.0, False)
    back = np.array([[oracle.lib.orc_srgb_to_linear(float(v)) for v in row[:3]] for row in srgb])
    assert np.allclose(back, lin[:, :3], rtol=2e-3, atol=1e-4)  # the reference's 0.41666 exponent is not the exact inverse
    assert np.allclose(lin[:, :3], acc[:, :3] * 2.0)


def test_pcg32_known_answers(oracle):
    """pcg32 (the reference's default_rng_t) restated in the oracle: the published generator's first outputs for the demo
    seeding pcg32(42, 54) -- the known-answer vector of the PCG reference implementation -- and skip-ahead == stepping."""
    import ctypes as C

    L = oracle.lib
    L.orc_pcg32_next_uint.restype = C.c_uint32
    r = (C.c_uint64 * 2)()
    L.orc_pcg32_seed(r, C.c_uint64(42), C.c_uint64(54))
    got = [L.orc_pcg32_next_uint(r) for _ in range(6)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    a = (C.c_uint64 * 2)()
    b = (C.c_uint64 * 2)()
    L.orc_pcg32_seed(a, C.c_uint64(1337), C.c_uint64(1))
    L.orc_pcg32_seed(b, C.c_uint64(1337), C.c_uint64(1))
    for _ in range(1000):
        L.orc_pcg32_next_uint(a)
    L.orc_pcg32_advance(b, C.c_uint64(1000))
    assert (a[0], a[1]) == (b[0], b[1])


def test_density_grid_refresh_oracle(oracle, scene_unit):
    """update_density_grid_nerf on the oracle: cells the synthetic occupancy marks empty but where the network is dense
    get switched on, touched cells hold MIN_CONE_STEPSIZE * exp(logit) > 0, untouched cells only decay."""
    m = oracle.make_model(scene_unit)
    grid0 = np.asarray(scene_unit["density_grid"], np.float16).astype(np.float32)
    rng = oracle.grid_rng()
    g1, step = oracle.update_density_grid(m, grid0, scene_unit["max_cascade"], rng, 0, 0.95, 50000, 20000)
    oracle.release(m)
    assert step == 1
    touched = g1 != np.float32(0.95) * grid0
    assert 30000 < touched.sum() <= 70000
    assert np.all(g1 >= np.float32(0.95) * grid0 - 1e-6) and np.all(g1[touched] > 0)
    # the non-uniform half only lands in cells that were above the optical-thickness threshold
    assert np.isfinite(g1).all()


def test_density_gradient_against_central_differences(oracle, scene_mod):
    """ERenderMode::Normals' input gradient (orc_density_gradient: fp16 backward pass + tcnn's dy_dx) against central differences of a
    float64 restatement of the same density head (no fp16 rounding anywhere: the two can differ by the fp16 noise only)."""
    from conftest import pkg

    from conftest import _with_bitfield

    sc = _with_bitfield(oracle, pkg("synthetic").make_scene(aabb_scale=1, seed=17, log2_hashmap_size=14))
    m = oracle.make_model(sc)
    enc = sc["encoding"]
    F = enc["n_features_per_level"]
    offsets, resolutions, scales = scene_mod.grid_layout(enc)
    params = np.asarray(sc["params"], np.uint16).view(np.float16).astype(np.float64)
    nd, nr, ng = scene_mod.n_params(sc)
    W1 = params[:64 * 32].reshape(64, 32)
    W2 = params[64 * 32:64 * 32 + 16 * 64].reshape(16, 64)
    grid = params[nd + nr:]

    def logit(x):  # x: float64[3]
        feats = []
        for l in range(enc["n_levels"]):
            size = offsets[l + 1] - offsets[l]
            table = grid[offsets[l] * F:offsets[l + 1] * F].reshape(size, F)
            p = x * float(np.float32(scales[l])) + 0.5
            pg = np.floor(p)
            w = p - pg
            pg = pg.astype(np.int64)
            acc = np.zeros(F)
            res = resolutions[l]
            for c in range(8):
                wt, g = 1.0, []
                for d in range(3):
                    bit = (c >> d) & 1
                    wt *= w[d] if bit else 1 - w[d]
                    g.append(int(pg[d]) + bit)
                stride, index = 1, 0
                for d in range(3):
                    if stride > size:
                        break
                    index = (index + g[d] * stride) & 0xFFFFFFFF
                    stride *= res
                if size < stride:
                    index = (g[0] ^ ((g[1] * 2654435761) & 0xFFFFFFFF) ^ ((g[2] * 805459861) & 0xFFFFFFFF))
                acc += wt * table[index % size]
            feats.append(acc)
        h = np.maximum(W1 @ np.concatenate(feats), 0)
        return (W2 @ h)[0]

    rng = np.random.default_rng(2)
    pos = rng.uniform(0.2, 0.8, (48, 3)).astype(np.float32)
    got = oracle.density_gradient(m, pos)
    oracle.release(m)
    eps = 1e-6
    cos = []
    for i in range(pos.shape[0]):
        x = pos[i].astype(np.float64)
        fd = np.array([(logit(x + eps * e) - logit(x - eps * e)) / (2 * eps) for e in np.eye(3)])
        if np.linalg.norm(fd) < 1e-3:
            continue
        cos.append(float(fd @ got[i] / (np.linalg.norm(fd) * np.linalg.norm(got[i]))))
        assert abs(np.linalg.norm(got[i]) / np.linalg.norm(fd) - 1) < 0.1
    assert len(cos) > 30 and np.median(cos) > 0.999 and min(cos) > 0.97, (np.median(cos), min(cos))
