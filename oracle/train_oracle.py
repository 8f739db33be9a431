"""ORACLE -- TEST INFRASTRUCTURE ONLY. PARITY UNPINNED.

numpy restatement (float64) of what one training step does to the parameters, for the configs/nerf/base.json network:
NerfNetwork::forward / backward_impl (include/neural-graphics-primitives/nerf_network.h:140-268) over tiny-cuda-nn's
FullyFusedMLP and GridEncoding (un-vendored submodule: forward y = W x with ReLU hidden layers and no bias; trilinear
hash-grid interpolation), and its Adam / Ema optimizers (optimizers/adam.h, ema.h). The reference computes in fp16 with
fp16 gradient accumulation; this oracle keeps float64 throughout, so comparisons carry the fp16 tolerance written in the
tests. `check_gradients` pins the analytic backward against central differences of the oracle's own forward.
"""
import numpy as np

PRIMES = (1, 2654435761, 805459861)


def layout(enc):
    """tcnn GridEncoding level table: offsets (entries), resolutions, scales."""
    log2_pls = np.log2(np.float32(enc["per_level_scale"]))
    offsets, resolutions, scales = [0], [], []
    for l in range(enc["n_levels"]):
        scale = np.float32(np.exp2(np.float32(l) * log2_pls) * np.float32(enc["base_resolution"]) - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        n = min(res ** 3, 0xFFFFFFFF // 2)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << enc["log2_hashmap_size"])
        offsets.append(offsets[-1] + n)
        resolutions.append(res)
        scales.append(float(scale))
    return offsets, resolutions, scales


def split_params(params, enc):
    """params: float array in snapshot order -> (W_D0 64x32, W_D1 16x64, W_R0 64x32, W_R1 64x64, W_R2 16x64, grid (entries, F))."""
    p = np.asarray(params, np.float64)
    shapes = [(64, 32), (16, 64), (64, 32), (64, 64), (16, 64)]
    out, k = [], 0
    for s in shapes:
        out.append(p[k:k + s[0] * s[1]].reshape(s))
        k += s[0] * s[1]
    out.append(p[k:].reshape(-1, enc["n_features_per_level"]))
    return out


def corners(enc, pos01):
    """Per level: (indices (n, 8) into the grid table, weights (n, 8)) -- tcnn grid_index / trilinear weights."""
    offsets, resolutions, scales = layout(enc)
    pos = np.asarray(pos01, np.float32)
    res_out = []
    for l in range(enc["n_levels"]):
        size = offsets[l + 1] - offsets[l]
        res = resolutions[l]
        p = pos * np.float32(scales[l]) + np.float32(0.5)
        fl = np.floor(p)
        w = (p - fl).astype(np.float64)
        g = fl.astype(np.int64).astype(np.uint64) & 0xFFFFFFFF
        idx = np.zeros((pos.shape[0], 8), np.int64)
        wt = np.zeros((pos.shape[0], 8), np.float64)
        hashed = res ** 3 > size
        for c in range(8):
            b = [(c >> d) & 1 for d in range(3)]
            cc = [(g[:, d] + b[d]) & 0xFFFFFFFF for d in range(3)]
            if hashed:
                i = (cc[0] * PRIMES[0]) ^ ((cc[1] * PRIMES[1]) & 0xFFFFFFFF) ^ ((cc[2] * PRIMES[2]) & 0xFFFFFFFF)
            else:
                i = (cc[0] + cc[1] * res + cc[2] * res * res) & 0xFFFFFFFF
            idx[:, c] = (i % np.uint64(size)).astype(np.int64) + offsets[l]
            wt[:, c] = np.prod([w[:, d] if b[d] else 1.0 - w[:, d] for d in range(3)], axis=0)
        res_out.append((idx, wt))
    return res_out


def sh4(dir01):
    d = np.asarray(dir01, np.float64) * 2.0 - 1.0
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = np.zeros((d.shape[0], 16))
    o[:, 0] = 0.28209479177387814
    o[:, 1] = -0.48860251190291987 * y
    o[:, 2] = 0.48860251190291987 * z
    o[:, 3] = -0.48860251190291987 * x
    o[:, 4] = 1.0925484305920792 * xy
    o[:, 5] = -1.0925484305920792 * yz
    o[:, 6] = 0.94617469575755997 * z2 - 0.31539156525251999
    o[:, 7] = -1.0925484305920792 * xz
    o[:, 8] = 0.54627421529603959 * x2 - 0.54627421529603959 * y2
    o[:, 9] = 0.59004358992664352 * y * (-3.0 * x2 + y2)
    o[:, 10] = 2.8906114426405538 * xy * z
    o[:, 11] = 0.45704579946446572 * y * (1.0 - 5.0 * z2)
    o[:, 12] = 0.3731763325901154 * z * (5.0 * z2 - 3.0)
    o[:, 13] = 0.45704579946446572 * x * (1.0 - 5.0 * z2)
    o[:, 14] = 1.4453057213202769 * z * (x2 - y2)
    o[:, 15] = 0.59004358992664352 * x * (-x2 + 3.0 * y2)
    return o


def forward(params, enc, coords, keep=False):
    """coords (n, 7): pos01, dt, dir01 -> network output (n, 4): rgb logits, density logit."""
    WD0, WD1, WR0, WR1, WR2, grid = split_params(params, enc)
    cs = corners(enc, coords[:, :3])
    F = enc["n_features_per_level"]
    x = np.zeros((coords.shape[0], enc["n_levels"] * F))
    for l, (idx, wt) in enumerate(cs):
        x[:, l * F:(l + 1) * F] = np.einsum("nc,ncf->nf", wt, grid[idx])
    a_d = x @ WD0.T
    h_d = np.maximum(a_d, 0)
    dens = h_d @ WD1.T
    rin = np.concatenate([dens, sh4(coords[:, 4:7])], axis=1)
    h1 = np.maximum(rin @ WR0.T, 0)
    h2 = np.maximum(h1 @ WR1.T, 0)
    out = h2 @ WR2.T
    res = np.concatenate([out[:, :3], dens[:, :1]], axis=1)
    if keep:
        return res, dict(cs=cs, x=x, h_d=h_d, rin=rin, h1=h1, h2=h2, W=(WD0, WD1, WR0, WR1, WR2), n_grid=grid.shape[0])
    return res


def backward(params, enc, coords, dloss):
    """dL/d(every parameter), snapshot order, for dL/d(output) = dloss (n, 4) [rgb logits, density logit]."""
    _, k = forward(params, enc, coords, keep=True)
    WD0, WD1, WR0, WR1, WR2 = k["W"]
    n = coords.shape[0]
    dl = np.asarray(dloss, np.float64)
    dout = np.zeros((n, 16))
    dout[:, :3] = dl[:, :3]
    gWR2 = dout.T @ k["h2"]
    dh2 = (dout @ WR2) * (k["h2"] > 0)
    gWR1 = dh2.T @ k["h1"]
    dh1 = (dh2 @ WR1) * (k["h1"] > 0)
    gWR0 = dh1.T @ k["rin"]
    drin = dh1 @ WR0
    ddens = drin[:, :16].copy()
    ddens[:, 0] += dl[:, 3]
    gWD1 = ddens.T @ k["h_d"]
    dhd = (ddens @ WD1) * (k["h_d"] > 0)
    gWD0 = dhd.T @ k["x"]
    dx = dhd @ WD0
    F = enc["n_features_per_level"]
    ggrid = np.zeros((k["n_grid"], F))
    for l, (idx, wt) in enumerate(k["cs"]):
        contrib = wt[:, :, None] * dx[:, None, l * F:(l + 1) * F]
        np.add.at(ggrid, idx.reshape(-1), contrib.reshape(-1, F))
    return np.concatenate([g.reshape(-1) for g in (gWD0, gWD1, gWR0, gWR1, gWR2, ggrid)])


def check_gradients(params, enc, coords, dloss, indices, eps=1e-4):
    """Central differences of sum(forward * dloss) w.r.t. the parameters at `indices`."""
    p = np.asarray(params, np.float64).copy()
    out = []
    for i in indices:
        old = p[i]
        p[i] = old + eps
        a = np.sum(forward(p, enc, coords) * dloss)
        p[i] = old - eps
        b = np.sum(forward(p, enc, coords) * dloss)
        p[i] = old
        out.append((a - b) / (2 * eps))
    return np.array(out)


def adam_step(w, grad, m1, m2, steps, n_matrix, lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-15, l2_reg=1e-6, loss_scale=128.0):
    """tcnn adam_step (optimizers/adam.h): in place on float64 arrays; grid entries with a zero gradient are skipped."""
    g = grad / loss_scale
    idx = np.arange(w.size)
    upd = (idx < n_matrix) | (g != 0)
    g = np.where(idx < n_matrix, g + l2_reg * w, g)
    m1[upd] = beta1 * m1[upd] + (1 - beta1) * g[upd]
    m2[upd] = beta2 * m2[upd] + (1 - beta2) * g[upd] ** 2
    steps[upd] += 1
    t = steps[upd].astype(np.float64)
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    w[upd] = w[upd] - lr_t / (np.sqrt(m2[upd]) + eps) * m1[upd]
    return upd


def ema_step(ema_tmp, w_half, step, decay=0.95):
    """tcnn ema_step (optimizers/ema.h): step = the optimizer's step count after this update (1-based)."""
    old = 1 - decay ** (step - 1)
    new = 1 / (1 - decay ** step)
    ema_tmp[:] = (ema_tmp * decay * old + w_half * (1 - decay)) * new
    return ema_tmp
