"""The training oracle checked against itself on the CPU: the float64 backward against central differences, the C loss
kernel's dL/d(network output) against differences of its own composited loss, Adam / Ema against their closed forms."""
import os
import sys

import numpy as np
import pytest

from conftest import pkg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))


def test_backward_oracle_matches_central_differences():
    import train_oracle as T

    enc = {"n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 12, "base_resolution": 16, "per_level_scale": 1.6}
    offsets, resolutions, _ = T.layout(enc)
    assert any(r ** 3 > (b - a) for a, b, r in zip(offsets, offsets[1:], resolutions)) and resolutions[0] ** 3 <= offsets[1]  # hashed and dense levels
    rng = np.random.default_rng(3)
    params = rng.normal(size=10240 + offsets[-1] * 4) * 0.3
    coords = rng.uniform(0, 1, (40, 7))
    dloss = rng.normal(size=(40, 4))
    g = T.backward(params, enc, coords, dloss)
    touched = 10240 + np.flatnonzero(g[10240:])
    idx = np.concatenate([rng.integers(0, 10240, 30), rng.choice(touched, 30, replace=False)])
    fd = T.check_gradients(params, enc, coords, dloss, idx, eps=1e-5)
    assert np.allclose(fd, g[idx], rtol=1e-5, atol=1e-7)
    # rows 3..15 of the rgb output layer receive no gradient (extract_rgb, nerf_network.h:206)
    assert not g[9216 + 3 * 64:10240].any() and g[9216:9216 + 3 * 64].any()
    # the density logit's gradient enters through row 0 of the density network's output (add_density_gradient, :235)
    only_sigma = np.zeros_like(dloss)
    only_sigma[:, 3] = dloss[:, 3]
    gs = T.backward(params, enc, coords, only_sigma)
    assert not gs[3072:10240].any() and gs[2048:2048 + 64].any() and not gs[2048 + 64:3072].any()


def test_adam_and_ema_closed_forms():
    import train_oracle as T

    rng = np.random.default_rng(4)
    n, n_matrix = 1000, 100
    w = rng.normal(size=n)
    w0 = w.copy()
    g = rng.normal(size=n) * 128.0
    g[n_matrix::3] = 0.0  # untouched grid entries
    m1, m2, steps = np.zeros(n), np.zeros(n), np.zeros(n, np.int64)
    upd = T.adam_step(w, g, m1, m2, steps, n_matrix)
    assert upd[:n_matrix].all() and not upd[n_matrix::3].any() and np.array_equal(w[~upd], w0[~upd])
    # first step: |delta| = lr for every updated weight, sign opposite to the (regularised) gradient
    eff = np.where(np.arange(n) < n_matrix, g / 128.0 + 1e-6 * w0, g / 128.0)
    assert np.allclose(w[upd] - w0[upd], -0.01 * np.sign(eff[upd]), rtol=1e-9, atol=1e-12)
    assert np.array_equal(steps, upd.astype(np.int64))
    # second step with the same gradient keeps the direction; per-parameter step counts differ
    g2 = g.copy()
    g2[n_matrix + 1::3] = 0.0
    T.adam_step(w, g2, m1, m2, steps, n_matrix)
    assert steps.max() == 2 and set(np.unique(steps)) == {0, 1, 2}
    # Ema: debiased running mean; a constant sequence stays put, step 1 returns the weights themselves
    e = np.zeros(5)
    for s in range(1, 6):
        T.ema_step(e, np.full(5, 2.5), s)
        assert np.allclose(e, 2.5)


@pytest.fixture(scope="module")
def loss_setup(oracle):
    """A small model with real marching: a synthetic scene, two float images, 256 rays."""
    import oracle as O

    S = pkg("scene")
    sc = pkg("synthetic").make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=12)
    grid = np.asarray(sc["density_grid"], np.float16).astype(np.float32)
    sc["density_grid_bitfield"], sc["density_grid_mean"] = oracle.density_grid_to_bitfield(grid, sc["max_cascade"])
    m = oracle.make_model(sc)
    rng = np.random.default_rng(9)
    views = []
    for az in (20.0, 200.0):
        px = rng.uniform(0, 1, (24, 32, 4)).astype(np.float32)
        px[..., :3] *= px[..., 3:4]  # premultiplied
        views.append({"pixels": px, "xform": S.orbit_camera(az), "focal": tuple(S.focal_from_fov_x(32, 0.6911))})
    images = oracle.make_train_images(views)
    o = O.TrainOpts()
    o.n_rays, o.n_images, o.rng = 256, 2, oracle.train_rng(1337, 3)
    o.snap_to_pixel_centers, o.random_bg_color, o.linear_colors, o.color_space = 1, 1, 0, 1
    o.near_distance, o.loss_scale, o.density_grid_mean = 0.1, 128.0, float(sc["density_grid_mean"])
    yield oracle, m, images, o
    oracle.release(m)


# L2, L1, Huber, LogL1. The relative losses (Mape, Smape, RelativeL2) treat their normaliser as a constant
# (nerf_device.cuh:82-142), so their "gradient" is deliberately not the derivative of the reported loss.
@pytest.mark.parametrize("loss_type", [0, 1, 4, 5])
def test_loss_kernel_gradient_is_the_derivative_of_its_loss(loss_setup, loss_type):
    oracle, m, images, o = loss_setup
    o.loss_type = loss_type
    gen = oracle.train_generate_samples(m, images, o, 1 << 16)
    total = int(gen["total"])
    assert total > 2000 and (gen["numsteps"] > 0).sum() > 40
    # bases are the running sum of the kept rays' step counts (ray order), coordinates stay in [0, 1]
    kept = np.flatnonzero(gen["numsteps"])
    assert np.array_equal(gen["base"][kept], np.concatenate([[0], np.cumsum(gen["numsteps"][kept])[:-1]]))
    c = gen["coords"][:total]
    assert c[:, :3].min() >= 0 and c[:, :3].max() <= 1 and c[:, 4:].min() >= 0 and c[:, 4:].max() <= 1 and (c[:, 3] >= -1e-6).all()
    rng = np.random.default_rng(11)
    # moderate outputs (no early termination, far from the Huber kink is not required: the kink has measure zero)
    net = np.zeros((1 << 16, 4), np.float16)
    net[:total, :3] = rng.normal(size=(total, 3)).astype(np.float16)
    net[:total, 3] = rng.uniform(-1.0, 3.0, total).astype(np.float16)
    ref = oracle.train_loss(m, images, o, gen, net)
    assert np.array_equal(ref["compacted_numsteps"] > 0, gen["numsteps"] > 0) and (ref["loss"] >= 0).all()
    scale = 128.0 / o.n_rays
    checked = 0
    for ray in kept[:12]:
        b, cn = int(gen["base"][ray]), int(ref["compacted_numsteps"][ray])
        for j in (0, cn // 2, cn - 1):
            for ch in range(4):
                v = np.float32(net[b + j, ch])
                hi, lo = np.float16(v + 0.0625), np.float16(v - 0.0625)
                p = net.copy(); p[b + j, ch] = hi
                q = net.copy(); q[b + j, ch] = lo
                lp, lq = oracle.train_loss(m, images, o, gen, p), oracle.train_loss(m, images, o, gen, q)
                if lp["compacted_numsteps"][ray] != cn or lq["compacted_numsteps"][ray] != cn:
                    continue
                # loss_out is the channel mean / n_rays; the gradient is of the channel SUM, times loss_scale / n_rays
                fd = 3.0 * o.n_rays * (float(lp["loss"][ray]) - float(lq["loss"][ray])) / (float(hi) - float(lo)) * scale
                got = float(ref["dloss"][b + j, ch])
                reg = 1e-4 if (ch == 3 and (np.float32(net[b + j, 3]) < 0 and o.density_grid_mean < 0.01)) else 0.0
                assert abs(got - fd) <= 0.06 * abs(fd) + 2e-3 * scale + reg + 2e-4, (loss_type, ray, j, ch, got, fd)
                checked += 1
    assert checked > 100


def test_training_oracle_matches_golden_vectors():
    """tests/golden/train_unit_v2.npz (made by make_golden_train.py): the training oracle recomputed from the seeded
    inputs -- integers and sample positions exactly, floating point to the last few ulps across compilers / numpy."""
    import importlib.util

    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden_train", os.path.join(here, "golden", "make_golden_train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.compute()
    ref = np.load(os.path.join(here, "golden", "train_unit_v2.npz"))
    for k in ("numsteps", "base", "total", "compacted_numsteps", "grad_index", "n_touched"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.array_equal(got["coords"], ref["coords"]) and np.array_equal(got["net"], ref["net"])
    assert np.allclose(got["loss"], ref["loss"], rtol=1e-5, atol=1e-12)
    d_got, d_ref = got["dloss"].view(np.float16).astype(np.float32), ref["dloss"].view(np.float16).astype(np.float32)
    assert np.mean(d_got != d_ref) < 1e-3 and np.allclose(d_got, d_ref, rtol=2e-3, atol=1e-7)  # an fp16 rounding may flip with libm's last bit
    for k in ("grad_value", "grad_matrix", "adam_value", "ema_value"):
        assert np.allclose(got[k], ref[k], rtol=1e-6, atol=1e-10), k
