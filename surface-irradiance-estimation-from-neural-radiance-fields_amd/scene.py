"""Host-side scene description shared by the synthetic generator, the snapshot reader/writer and the
ctypes binding of libngp_hip: grid layout, parameter counts, Morton helpers, camera conventions.

Mirrors what Testbed::reset_network / load_nerf_post derive from a config + dataset
(reference src/testbed.cu:3928-3977, src/testbed_nerf.cu:2720-2738, nerf_loader.h:101-120).
"""
import math

import numpy as np

NERF_GRIDSIZE = 128
NERF_CASCADES = 8
ACT_NONE, ACT_RELU, ACT_LOGISTIC, ACT_EXPONENTIAL = 0, 1, 2, 3


def base_network_config():
    """configs/nerf/base.json (reference configs/nerf/base.json:23-58), render-relevant parts."""
    return {
        "encoding": {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 19, "base_resolution": 16},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1},
        "dir_encoding": {"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4}, {"otype": "Identity"}]},
        "rgb_network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
    }


def linear_network_config(hidden_density=0):
    """configs/nerf/linear.json (hidden_density 0: both heads a single matrix, CutlassMLP) and configs/nerf/base_0layer.json
    (hidden_density 1: the base.json density head, the rgb head a single matrix), merged over base.json."""
    cfg = base_network_config()
    if hidden_density == 0:
        cfg["network"].update({"otype": "CutlassMLP", "n_hidden_layers": 0})
    cfg["rgb_network"].update({"otype": "CutlassMLP", "n_hidden_layers": 0})
    return cfg


def frequency_network_config(n_neurons=256, n_hidden_density=7, n_hidden_rgb=1):
    """configs/nerf/frequency.json merged over base.json: the original NeRF's architecture -- Frequency encodings (16 / 4
    frequencies), CutlassMLPs 256 wide with 7 + 1 hidden layers."""
    return {
        "encoding": {"otype": "Frequency", "n_frequencies": 16},
        "network": {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": n_neurons, "n_hidden_layers": n_hidden_density},
        "dir_encoding": {"otype": "Frequency", "n_frequencies": 4},
        "rgb_network": {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": n_neurons, "n_hidden_layers": n_hidden_rgb},
    }


def identity_network_config(n_neurons=256, n_hidden_density=7, n_hidden_rgb=1):
    """configs/nerf/none.json merged over frequency.json: Identity encodings (the position and the direction themselves, padded with ones to the
    CutlassMLPs' alignment of 8) in front of the same MLPs."""
    cfg = frequency_network_config(n_neurons, n_hidden_density, n_hidden_rgb)
    cfg["encoding"] = {"otype": "Identity"}
    cfg["dir_encoding"] = {"otype": "Identity"}
    return cfg


def network_shapes(cfg):
    """(position encoding width, direction encoding width, rgb network input width, rgb network output width) as NerfNetwork derives
    them (nerf_network.h:81-100): the position encoding is padded to the density network's alignment, the direction encoding, the
    rgb input and the rgb output to the rgb network's -- 16 for FullyFusedMLP, 8 for CutlassMLP."""
    al_pos = 8 if cfg["network"].get("otype", "FullyFusedMLP") == "CutlassMLP" else 16
    al = 8 if cfg["rgb_network"].get("otype", "FullyFusedMLP") == "CutlassMLP" else 16
    up = lambda v: (v + al - 1) // al * al
    enc = cfg["encoding"]
    up_pos = lambda v: (v + al_pos - 1) // al_pos * al_pos
    enc_dims = up_pos(3 * 2 * enc["n_frequencies"]) if enc.get("otype") == "Frequency" else up_pos(3) if enc.get("otype") == "Identity" else enc["n_levels"] * enc["n_features_per_level"]
    de = cfg.get("dir_encoding", {})
    dir_dims = up(3 * 2 * de["n_frequencies"]) if de.get("otype") == "Frequency" else up(3) if de.get("otype") == "Identity" else 16
    dens_out = cfg["network"].get("n_output_dims", 16)
    return enc_dims, dir_dims, up(dens_out + dir_dims), up(3)


def per_level_scale(aabb_scale, n_levels, base_resolution, rule="fork", desired_resolution=2048.0):
    """src/testbed.cu:3951-3966. The fork overwrites the value with one derived from
    m_geometry.nerf...aabb_scale, which is 1 in Nerf mode ('fork' rule); upstream uses the dataset's."""
    s = 1 if rule == "fork" else aabb_scale
    if n_levels <= 1:
        return 1.0
    v = np.exp(np.log(np.float32(desired_resolution) * np.float32(s) / np.float32(base_resolution)) / np.float32(n_levels - 1))
    return float(np.float32(v))


def grid_layout(enc):
    """tcnn GridEncoding level table: offsets (entries), resolutions, scales."""
    n_levels = enc["n_levels"]
    log2_pls = np.log2(np.float32(enc["per_level_scale"]))
    offsets, resolutions, scales = [0], [], []
    for l in range(n_levels):
        scale = np.float32(np.exp2(np.float32(l) * log2_pls) * np.float32(enc["base_resolution"]) - np.float32(1.0))
        res = int(math.ceil(float(scale))) + 1
        n = min(res ** 3, 0xFFFFFFFF // 2)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << enc["log2_hashmap_size"])
        offsets.append(offsets[-1] + n)
        resolutions.append(res)
        scales.append(float(scale))
    return offsets, resolutions, scales


def mlp_n_params(n_in, width, n_hidden, n_out_padded):
    if n_hidden == 0:  # tcnn CutlassMLP without a hidden layer: one (padded output) x (input) matrix
        return n_out_padded * n_in
    return width * n_in + (n_hidden - 1) * width * width + n_out_padded * width


def n_params(cfg):
    enc = cfg["encoding"]
    enc_dims, dir_dims, rgb_in, rgb_out = network_shapes(cfg)
    dens_out = cfg["network"].get("n_output_dims", 16)
    nd = mlp_n_params(enc_dims, cfg["network"]["n_neurons"], cfg["network"]["n_hidden_layers"], dens_out)
    nr = mlp_n_params(rgb_in, cfg["rgb_network"]["n_neurons"], cfg["rgb_network"]["n_hidden_layers"], rgb_out)
    if enc.get("otype") in ("Frequency", "Identity"):
        return nd, nr, 0
    offsets, _, _ = grid_layout(enc)
    return nd, nr, offsets[-1] * enc["n_features_per_level"]


def _expand_bits(v):
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(x, y, z):
    return _expand_bits(x) | (_expand_bits(y) << np.uint32(1)) | (_expand_bits(z) << np.uint32(2))


def max_cascade_for(aabb_scale):
    mc = 0
    while (1 << mc) < aabb_scale:
        mc += 1
    return mc


def nerf_matrix_to_ngp(m, scale=0.33, offset=(0.5, 0.5, 0.5)):
    """nerf_loader.h:101-120: negate columns 1,2; t*scale+offset; cycle rows (x,y,z) <- (y,z,x).
    m: (3,4) or (4,4) camera-to-world in the NeRF/Blender convention. Returns (3,4) float32."""
    m = np.asarray(m, np.float32)[:3, :4].copy()
    m[:, 1] *= -1.0
    m[:, 2] *= -1.0
    m[:, 3] = m[:, 3] * np.float32(scale) + np.asarray(offset, np.float32)
    return m[[1, 2, 0], :].copy()


def orbit_camera(azimuth_deg, elevation_deg=30.0, radius=4.03, scale=0.33, offset=(0.5, 0.5, 0.5)):
    """SURVEY 8(d) synthetic camera: orbit pose looking at the origin (NeRF convention: camera looks down -z,
    y up), converted with nerf_matrix_to_ngp."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    pos = np.array([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)], np.float64)
    fwd = -pos / np.linalg.norm(pos)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    c2w = np.stack([right, true_up, -fwd, pos], axis=1)  # columns: x, y, z(back), t
    return nerf_matrix_to_ngp(c2w.astype(np.float32), scale, offset)


def focal_from_fov_x(width, fov_x_rad):
    """fov_axis = 0: relative_focal_length = 0.5 / tan(fov/2); focal = rel * res[0] (src/testbed.cu:4474-4476)."""
    rel = np.float32(0.5) / np.float32(math.tan(0.5 * fov_x_rad))
    f = float(rel * np.float32(width))
    return (f, f)


def ngp_matrix_to_nerf(m, scale=0.33, offset=(0.5, 0.5, 0.5)):
    """Inverse of nerf_matrix_to_ngp: (3,4) ngp-space camera -> (4,4) NeRF/Blender transform_matrix."""
    m = np.asarray(m, np.float64)[[2, 0, 1], :].copy()
    m[:, 3] = (m[:, 3] - np.asarray(offset, np.float64)) / scale
    m[:, 1] *= -1.0
    m[:, 2] *= -1.0
    out = np.eye(4)
    out[:3, :4] = m
    return out


def write_transforms(path, ngp_matrices, width, height, fov_x, aabb_scale=1):
    """A transforms.json (the reference's dataset format, src/nerf_loader.cu) for the given ngp-space cameras; frame
    names sort in the given order."""
    import json

    frames = [{"file_path": f"./train/r_{i:04d}", "transform_matrix": ngp_matrix_to_nerf(m).tolist()} for i, m in enumerate(ngp_matrices)]
    with open(path, "w") as f:
        json.dump({"camera_angle_x": fov_x, "w": width, "h": height, "aabb_scale": aabb_scale, "frames": frames}, f)
    return path
