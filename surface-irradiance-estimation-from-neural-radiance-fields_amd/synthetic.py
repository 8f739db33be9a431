"""Deterministic synthetic NeRF scenes in the reference's model layout.

The reference mount ships no .msgpack/.ingp snapshot and the container has no network, so the
benchmark and parity scenes are generated: seeded random-init weights of the configs/nerf/base.json
architecture (density head calibrated so that rays terminate after a few tens of samples, like a trained
Lego model) and an analytic occupancy grid ("lego-like": plates, blocks, a sphere shell and a torus;
larger aabb_scale scenes add a ground disc and a far shell so that the outer cascades are exercised).
"""
import math

import numpy as np

from . import scene as S


def _shapes_occupancy(p, aabb_scale):
    """p: (...,3) positions in NGP space (unit cube = [0,1]^3 centred at 0.5). Returns bool mask."""
    q = p - 0.5
    x, y, z = q[..., 0], q[..., 1], q[..., 2]
    occ = np.zeros(p.shape[:-1], bool)
    # base plate
    occ |= (np.abs(x) < 0.34) & (np.abs(z) < 0.30) & (y > -0.30) & (y < -0.24)
    # blocks
    occ |= (np.abs(x + 0.12) < 0.10) & (np.abs(z + 0.05) < 0.12) & (y > -0.24) & (y < -0.02)
    occ |= (np.abs(x - 0.16) < 0.07) & (np.abs(z - 0.10) < 0.07) & (y > -0.24) & (y < 0.12)
    # sphere shell
    r = np.sqrt((x + 0.02) ** 2 + (y - 0.10) ** 2 + (z + 0.02) ** 2)
    occ |= (r > 0.13) & (r < 0.17)
    # torus around y axis
    rt = np.sqrt(x ** 2 + z ** 2) - 0.27
    occ |= (rt ** 2 + (y + 0.08) ** 2) < 0.035 ** 2
    if aabb_scale > 1:
        half = 0.5 * aabb_scale
        rr = np.sqrt(x ** 2 + z ** 2)
        occ |= (rr < 0.8 * half) & (y > -0.36) & (y < -0.30)              # ground disc
        rs = np.sqrt(x ** 2 + y ** 2 + z ** 2)
        occ |= (rs > 0.82 * half) & (rs < 0.90 * half) & (y > -0.30)       # far dome
    return occ


def make_density_grid(aabb_scale, inside_value=50.0):
    """float32 density grid, (max_cascade+1) x 128^3, Morton order per cascade
    (layout read by grid_to_bitfield, reference src/testbed_nerf.cu:284-308)."""
    mc = S.max_cascade_for(aabb_scale)
    n = S.NERF_GRIDSIZE
    idx = np.arange(n, dtype=np.uint32)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    morton = S.morton3d(X.ravel(), Y.ravel(), Z.ravel())
    grid = np.zeros((mc + 1, n ** 3), np.float32)
    centers = (np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32) + 0.5) / n - 0.5
    for k in range(mc + 1):
        p = centers * float(1 << k) + 0.5
        occ = _shapes_occupancy(p, aabb_scale)
        # dilate by sampling the 8 cell corners too (conservative voxelisation)
        h = 0.5 / n * float(1 << k)
        for dx in (-h, h):
            for dy in (-h, h):
                for dz in (-h, h):
                    occ |= _shapes_occupancy(p + np.array([dx, dy, dz], np.float32), aabb_scale)
        grid[k, morton] = np.where(occ, np.float32(inside_value), np.float32(0.0))
    return grid.reshape(-1)


def _grid_encode_np(params_f32, cfg, pos01):
    """Approximate (fp32) hash-grid encode, used only to calibrate the density head."""
    enc = cfg["encoding"]
    F = enc["n_features_per_level"]
    offsets, resolutions, scales = S.grid_layout(enc)
    out = np.zeros((pos01.shape[0], enc["n_levels"] * F), np.float32)
    for l in range(enc["n_levels"]):
        size = offsets[l + 1] - offsets[l]
        table = params_f32[offsets[l] * F:offsets[l + 1] * F].reshape(size, F)
        p = pos01 * np.float32(scales[l]) + np.float32(0.5)
        pg = np.floor(p)
        w = p - pg
        pg = pg.astype(np.int64).astype(np.uint32)
        res = resolutions[l]
        acc = np.zeros((pos01.shape[0], F), np.float32)
        for c in range(8):
            wt = np.ones(pos01.shape[0], np.float32)
            g = []
            for d in range(3):
                if c & (1 << d):
                    wt = wt * w[:, d]
                    g.append(pg[:, d] + np.uint32(1))
                else:
                    wt = wt * (1 - w[:, d])
                    g.append(pg[:, d])
            stride, index, hashed = 1, np.zeros(pos01.shape[0], np.uint32), False
            for d in range(3):
                if stride > size:
                    break
                index = index + g[d] * np.uint32(stride)
                stride *= res
            if size < stride:
                index = g[0] ^ (g[1] * np.uint32(2654435761)) ^ (g[2] * np.uint32(805459861))
            index = index % np.uint32(size)
            acc += wt[:, None] * table[index]
        out[:, l * F:(l + 1) * F] = acc
    return out


def make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19, pls_rule="fork", target_logit=4.2, cfg=None):
    """Returns the scene dict consumed by the ctypes binding (and by oracle.make_model in tests)."""
    cfg = cfg or S.base_network_config()
    cfg = {k: dict(v) if isinstance(v, dict) else v for k, v in cfg.items()}
    enc = cfg["encoding"]
    frequency = enc.get("otype") == "Frequency"
    identity = enc.get("otype") == "Identity"
    if not frequency and not identity:
        enc["log2_hashmap_size"] = log2_hashmap_size
        if "per_level_scale" not in enc:
            enc["per_level_scale"] = S.per_level_scale(aabb_scale, enc["n_levels"], enc["base_resolution"], pls_rule)
    nd, nr, ng = S.n_params(cfg)
    rng = np.random.default_rng(seed)
    width = cfg["network"]["n_neurons"]
    enc_dims, dir_dims, rgb_in_dims, rgb_out_dims = S.network_shapes(cfg)
    dens_out = cfg["network"].get("n_output_dims", 16)

    def xavier(n_out, n_in):
        s = math.sqrt(6.0 / (n_in + n_out))
        return rng.uniform(-s, s, size=(n_out, n_in)).astype(np.float32)

    def mlp(n_in, n_hidden, n_out):
        if n_hidden == 0:  # configs/nerf/linear.json: the output layer alone
            return [xavier(n_out, n_in)]
        layers = [xavier(width, n_in)]
        for _ in range(n_hidden - 1):
            layers.append(xavier(width, width))
        layers.append(xavier(n_out, width))
        return layers

    dens_layers = mlp(enc_dims, cfg["network"]["n_hidden_layers"], dens_out)
    rgb_layers = mlp(rgb_in_dims, cfg["rgb_network"]["n_hidden_layers"], rgb_out_dims)
    # zero-sum colour rows (ReLU activations have a positive mean) and a wider spread, so that the image
    # shows spatial and directional colour variation instead of one flat tint
    rgb_layers[-1][:3] -= rgb_layers[-1][:3].mean(axis=1, keepdims=True)
    rgb_layers[-1] *= 12.0
    grid = rng.uniform(-0.5, 0.5, size=ng).astype(np.float32)

    # Calibrate the density logit (output 0 of the density head) on points inside the occupied shapes.
    dens_layers[-1][0] = np.abs(dens_layers[-1][0])
    pts = rng.uniform(0.15, 0.85, size=(20000, 3)).astype(np.float32)
    pts = pts[_shapes_occupancy(pts, 1)][:4096]
    if frequency:  # tcnn FrequencyEncoding: sin(2^k pi x + (j % 2) pi / 2), input-major, then frequency, then sin / cos
        nf = enc["n_frequencies"]
        k = np.arange(nf, dtype=np.float32)
        arg = pts[:, :, None] * np.exp2(k)[None, None, :] * np.float32(np.pi)
        h = np.stack([np.sin(arg), np.sin(arg + np.float32(np.pi / 2))], -1).reshape(pts.shape[0], -1).astype(np.float32)
        if h.shape[1] < enc_dims:
            h = np.concatenate([h, np.ones((pts.shape[0], enc_dims - h.shape[1]), np.float32)], 1)
    elif identity:  # tcnn IdentityEncoding: the inputs, padded with ones
        h = np.concatenate([pts, np.ones((pts.shape[0], enc_dims - 3), np.float32)], 1)
    else:
        h = _grid_encode_np(grid.astype(np.float16).astype(np.float32), cfg, pts)
    for W in dens_layers[:-1]:
        h = np.maximum(h @ W.T, 0)
    logit = h @ dens_layers[-1][0]
    if len(dens_layers) == 1:  # a linear head over zero-mean features has a zero-mean logit: calibrate its spread instead
        dens_layers[-1][0] *= np.float32(target_logit / max(float(logit.std()), 1e-6))
    else:
        dens_layers[-1][0] *= np.float32(target_logit / max(float(logit.mean()), 1e-6))

    params = np.concatenate([W.reshape(-1) for W in dens_layers] + [W.reshape(-1) for W in rgb_layers] + [grid]).astype(np.float16)
    assert params.size == nd + nr + ng
    mc = S.max_cascade_for(aabb_scale)
    half = 0.5 * min(1 << (S.NERF_CASCADES - 1), aabb_scale)
    aabb = ((0.5 - half,) * 3, (0.5 + half,) * 3)
    sc = dict(cfg)
    sc.update({
        "params": params.view(np.uint16),
        "density_grid": make_density_grid(aabb_scale),
        "aabb": aabb,
        "render_aabb": aabb,
        "render_aabb_to_local": np.eye(3, dtype=np.float32),
        "aabb_scale": aabb_scale,
        "max_cascade": mc,
        "cone_angle_constant": 0.0 if aabb_scale <= 1 else 1.0 / 256.0,
        "rgb_activation": S.ACT_LOGISTIC,
        "density_activation": S.ACT_EXPONENTIAL,
        "linear_colors": False,
        "seed": seed,
    })
    return sc
