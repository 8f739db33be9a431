"""ctypes binding of libngp_hip.so -- the C ABI declared in include/ngp_hip.h.

This is the only way Python reaches the renderer: there is no CPU fallback. Loading fails loudly when the
library has not been built (run the package's build.py / __graft_entry__.build()), and every compute call
fails loudly when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libngp_hip.so")

HEADER_PATH = os.path.join(HERE, "..", "include", "ngp_hip.h")


def header_exports():
    """Every entry point include/ngp_hip.h declares (NGP_API ... name(...))."""
    import re

    with open(HEADER_PATH) as f:
        return re.findall(r"NGP_API\s+[\w\s\*]+?\b(ngp_\w+)\s*\(", f.read())


EXPORTS = header_exports()


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_levels", C.c_uint32), ("n_features_per_level", C.c_uint32), ("log2_hashmap_size", C.c_uint32), ("base_resolution", C.c_uint32),
        ("per_level_scale", C.c_float),
        ("n_neurons", C.c_uint32), ("n_hidden_density", C.c_uint32), ("n_hidden_rgb", C.c_uint32), ("density_out_dims", C.c_uint32),
        ("rgb_activation", C.c_uint32), ("density_activation", C.c_uint32),
        ("params_fp16", C.c_void_p), ("n_params", C.c_uint64),
        ("density_grid_fp16", C.c_void_p), ("n_density_grid", C.c_uint64),
        ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
        ("render_aabb_min", C.c_float * 3), ("render_aabb_max", C.c_float * 3),
        ("render_aabb_to_local", C.c_float * 9),
        ("aabb_scale", C.c_uint32), ("cone_angle_constant", C.c_float), ("linear_colors", C.c_int32),
        ("pos_encoding", C.c_uint32), ("pos_n_frequencies", C.c_uint32), ("dir_encoding", C.c_uint32), ("dir_n_frequencies", C.c_uint32),
        ("mlp_alignment", C.c_uint32),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("matrix", C.c_float * 12), ("width", C.c_int32), ("height", C.c_int32), ("focal_length", C.c_float * 2),
        ("screen_center", C.c_float * 2), ("spp_index", C.c_uint32), ("snap_to_pixel_centers", C.c_int32), ("near_distance", C.c_float),
        ("lens_mode", C.c_int32), ("lens_params", C.c_float * 7), ("aperture_size", C.c_float), ("focus_z", C.c_float),
        ("has_matrix1", C.c_int32), ("matrix1", C.c_float * 12), ("rolling_shutter", C.c_float * 4),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("render_mode", C.c_int32), ("min_transmittance", C.c_float), ("background", C.c_float * 4), ("exposure", C.c_float),
        ("to_srgb", C.c_int32), ("spp", C.c_int32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("testbed_mode", C.c_int32), ("packed_output", C.c_int32), ("depth_scale", C.c_float), ("color_space", C.c_int32),
    ]


class SessionState(C.Structure):
    _fields_ = [
        ("valid", C.c_int32), ("background_color", C.c_float * 4), ("exposure", C.c_float), ("sun_dir", C.c_float * 3), ("up_dir", C.c_float * 3),
        ("camera_scale", C.c_float), ("aperture_size", C.c_float), ("autofocus_depth", C.c_float),
    ]


class GeometryOpts(C.Structure):
    _fields_ = [
        ("sun_dir", C.c_float * 3), ("up_dir", C.c_float * 3),
        ("metallic", C.c_float), ("subsurface", C.c_float), ("specular", C.c_float), ("roughness", C.c_float),
        ("sheen", C.c_float), ("clearcoat", C.c_float), ("clearcoat_gloss", C.c_float),
        ("basecolor", C.c_float * 3), ("ambientcolor", C.c_float * 3),
    ]


class ProbeDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_theta", C.c_uint32), ("n_phi", C.c_uint32), ("n_origin", C.c_uint32), ("origin", C.c_float * 3),
                ("min_transmittance", C.c_float)]


class ProbeGridDesc(C.Structure):
    _fields_ = [("grid_x", C.c_uint32), ("grid_y", C.c_uint32), ("n_theta", C.c_uint32), ("n_phi", C.c_uint32), ("shell_radius", C.c_float),
                ("min_transmittance", C.c_float)]


MODE_NERF, MODE_GEOMETRY = 0, 1
RENDER_SHADE, RENDER_SHADE_ENVMAP, RENDER_AO, RENDER_POSITIONS, RENDER_DEPTH, RENDER_COST, RENDER_SHADE_GRID_ENVMAP, RENDER_NORMALS = 0, 1, 2, 3, 4, 5, 6, 7
PROBE_CENTER, PROBE_CENTER_OUTWARD, PROBE_MULTI_CENTER = 0, 1, 2
BVH_NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("left_idx", "<i4"), ("right_idx", "<i4")])
TRIANGLE_DTYPE = np.dtype([("a", "<f4", 3), ("b", "<f4", 3), ("c", "<f4", 3)])


class RenderStats(C.Structure):
    _fields_ = [
        ("n_rays", C.c_uint64), ("n_rays_alive_after_init", C.c_uint64), ("n_rays_hit", C.c_uint64), ("n_samples", C.c_uint64),
        ("kernel_ms", C.c_float), ("frame_ms", C.c_float), ("kernel_device_ms", C.c_float),
    ]


PAYLOAD_DTYPE = np.dtype([("origin", "<f4", 3), ("dir", "<f4", 3), ("t", "<f4"), ("max_weight", "<f4"), ("idx", "<u4"),
                          ("n_steps", "<u2"), ("alive", "u1"), ("pad", "u1")])

_lib = None


class TrainingOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("loss_type", C.c_int32), ("random_bg_color", C.c_int32), ("linear_colors", C.c_int32), ("snap_to_pixel_centers", C.c_int32),
        ("near_distance", C.c_float), ("density_grid_decay", C.c_float), ("train_network", C.c_int32), ("train_encoding", C.c_int32),
        ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float), ("l2_reg", C.c_float),
        ("ema_decay", C.c_float), ("decay_start", C.c_uint32), ("decay_interval", C.c_uint32), ("decay_base", C.c_float),
        ("background_color", C.c_float * 3), ("color_space", C.c_int32),
    ]


class TrainingState(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("training_step", C.c_uint32), ("rays_per_batch", C.c_uint32), ("measured_batch_size", C.c_uint32),
        ("measured_batch_size_before_compaction", C.c_uint32), ("n_rays_total", C.c_uint32), ("loss", C.c_float), ("learning_rate", C.c_float),
        ("n_params", C.c_uint64), ("n_matrix_params", C.c_uint64),
    ]


LOSS_TYPES = {"L2": 0, "L1": 1, "MAPE": 2, "SMAPE": 3, "Huber": 4, "LogL1": 5, "RelativeL2": 6}
IMAGE_BYTE, IMAGE_FLOAT = 1, 3


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    LIB_PATH = os.environ.get("NGP_HIP_LIBRARY") or globals()["LIB_PATH"]  # e.g. the legacy-encode variant, libngp_hip_legacy.so
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build the HIP extension first (python __graft_entry__.py or the package's build.py). "
                           "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.c_int
    L.ngp_create.argtypes = [ip]; L.ngp_create.restype = vp
    L.ngp_create_multi.argtypes = [vp, ip]; L.ngp_create_multi.restype = vp
    L.ngp_n_devices.argtypes = [vp]
    L.ngp_get_device_render_stats.argtypes = [vp, ip, C.POINTER(RenderStats)]
    L.ngp_destroy.argtypes = [vp]; L.ngp_destroy.restype = None
    L.ngp_last_error.argtypes = [vp]; L.ngp_last_error.restype = C.c_char_p
    L.ngp_version.restype = C.c_char_p
    L.ngp_set_model.argtypes = [vp, C.POINTER(ModelDesc)]
    L.ngp_load_snapshot.argtypes = [vp, vp, C.c_size_t, ip]
    L.ngp_load_snapshot_file.argtypes = [vp, C.c_char_p]
    L.ngp_save_snapshot_file.argtypes = [vp, C.c_char_p, ip]
    L.ngp_get_model.argtypes = [vp, C.POINTER(ModelDesc)]
    L.ngp_update_density_grid.argtypes = [vp, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
    L.ngp_get_density_grid.argtypes = [vp, vp, C.c_uint64]
    L.ngp_set_cone_angle_constant.argtypes = [vp, C.c_float]
    L.ngp_set_envmap.argtypes = [vp, C.c_int32, C.c_int32, vp]
    L.ngp_set_render_aabb.argtypes = [vp, vp, vp, vp]
    L.ngp_get_snapshot_camera.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ngp_get_session_state.argtypes = [vp, C.POINTER(SessionState)]
    L.ngp_set_session_state.argtypes = [vp, C.POINTER(SessionState), vp, vp, C.c_int32, vp, C.c_float]
    L.ngp_load_training_data.argtypes = [vp, C.c_char_p]
    L.ngp_n_training_views.argtypes = [vp]
    L.ngp_get_training_view.argtypes = [vp, ip, vp, vp, vp, vp]
    L.ngp_get_training_view_lens.argtypes = [vp, ip, vp, vp]
    L.ngp_get_dataset_info.argtypes = [vp, vp, vp, vp, vp]
    L.ngp_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderOpts), vp, vp]
    L.ngp_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderOpts), vp, vp, vp]
    L.ngp_host_alloc.argtypes = [C.c_size_t]; L.ngp_host_alloc.restype = vp
    L.ngp_host_free.argtypes = [vp]; L.ngp_host_free.restype = None
    L.ngp_packed_tiles.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.c_uint32]
    L.ngp_packed_tiles.restype = C.c_uint32
    L.ngp_get_render_stats.argtypes = [vp, C.POINTER(RenderStats)]
    L.ngp_get_render_history.argtypes = [vp, ip, C.POINTER(RenderStats)]
    L.ngp_set_schedule.argtypes = [vp, vp, ip]
    if hasattr(L, "ngp_get_profile_trace"):  # (diagnostic entry; an older build loaded through NGP_HIP_LIBRARY for an A/B run lacks it)
        L.ngp_get_profile_trace.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.ngp_grid_encode.argtypes = [vp, C.c_uint32, vp, vp]
    L.ngp_network_inference.argtypes = [vp, C.c_uint32, vp, vp, vp]
    L.ngp_density_gradient.argtypes = [vp, C.c_uint32, vp, vp]
    L.ngp_get_density_bitfield.argtypes = [vp, vp, vp]
    L.ngp_init_rays.argtypes = [vp, C.POINTER(Camera), vp]
    L.ngp_load_scene.argtypes = [vp, C.c_char_p]
    L.ngp_add_mesh.argtypes = [vp, vp, C.c_uint32, vp]
    L.ngp_load_mesh_file.argtypes = [vp, C.c_char_p, vp]
    L.ngp_clear_meshes.argtypes = [vp]
    L.ngp_n_meshes.argtypes = [vp]
    L.ngp_get_mesh_info.argtypes = [vp, ip, vp, vp, vp]
    L.ngp_get_mesh_bvh.argtypes = [vp, ip, vp, vp]
    L.ngp_set_geometry_opts.argtypes = [vp, C.POINTER(GeometryOpts)]
    L.ngp_trace_mesh_rays.argtypes = [vp, C.c_uint32, vp, vp]
    L.ngp_compute_envmap.argtypes = [vp, C.POINTER(ProbeDesc), vp]
    L.ngp_get_envmap.argtypes = [vp, vp, vp, vp, vp]
    L.ngp_irradiance.argtypes = [vp, C.c_uint32, vp, vp]
    L.ngp_compute_envmap_grid.argtypes = [vp, C.POINTER(ProbeGridDesc), vp]
    L.ngp_get_envmap_grid.argtypes = [vp, C.POINTER(ProbeGridDesc), vp]
    L.ngp_irradiance_at.argtypes = [vp, C.c_uint32, vp, vp, vp]
    L.ngp_reset_network.argtypes = [vp, C.c_uint32, C.c_uint64]
    L.ngp_default_training_opts.argtypes = [C.POINTER(TrainingOpts)]; L.ngp_default_training_opts.restype = None
    L.ngp_set_training_opts.argtypes = [vp, C.POINTER(TrainingOpts)]
    L.ngp_get_training_opts.argtypes = [vp, C.POINTER(TrainingOpts)]
    L.ngp_set_training_image.argtypes = [vp, ip, C.c_int32, C.c_int32, vp, C.c_int32]
    L.ngp_train.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.ngp_load_training_images.argtypes = [vp, vp]
    L.ngp_render_ground_truth.argtypes = [vp, ip, C.c_int32, C.c_int32, vp, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp]
    L.ngp_decode_image.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_size_t, vp, C.c_size_t]
    L.ngp_get_training_state.argtypes = [vp, C.POINTER(TrainingState)]
    L.ngp_train_prepare_batch.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.ngp_train_gradients.argtypes = [vp, C.c_uint32, vp]
    L.ngp_train_apply.argtypes = [vp]
    L.ngp_get_training_params.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


LENS_PERSPECTIVE, LENS_OPENCV, LENS_FTHETA, LENS_LATLONG, LENS_OPENCV_FISHEYE, LENS_EQUIRECTANGULAR = range(6)


def host_image(shape):
    """float32 array in page-locked memory from ngp_host_alloc; the buffer returns to the pool when the array is collected"""
    import weakref

    L = load_library()
    count = int(np.prod(shape))
    p = L.ngp_host_alloc(count * 4)
    if not p:
        raise MemoryError("ngp_host_alloc failed")
    buf = (C.c_float * count).from_address(p)
    weakref.finalize(buf, L.ngp_host_free, p)
    return np.ctypeslib.as_array(buf).reshape(shape)


def decode_image(data):
    """PNG / baseline JPEG bytes -> (H, W, 4) uint8 through the library's decoder."""
    L = load_library()
    w, h = C.c_int32(0), C.c_int32(0)
    err = C.create_string_buffer(256)
    if L.ngp_decode_image(data, len(data), C.byref(w), C.byref(h), None, 0, err, 256) != 0:
        raise RuntimeError(err.value.decode())
    out = np.zeros((h.value, w.value, 4), np.uint8)
    if L.ngp_decode_image(data, len(data), C.byref(w), C.byref(h), _p(out), out.size, err, 256) != 0:
        raise RuntimeError(err.value.decode())
    return out


def make_camera(matrix_3x4, width, height, focal_length, screen_center=(0.5, 0.5), spp_index=0, snap=True, near=0.0, lens_mode=0, lens_params=(), aperture_size=0.0, focus_z=1.0,
                matrix1_3x4=None, rolling_shutter=(0.0, 0.0, 0.0, 1.0)):
    cam = Camera()
    mat = np.asarray(matrix_3x4, np.float32)
    assert mat.shape == (3, 4)
    for c in range(4):
        for r in range(3):
            cam.matrix[c * 3 + r] = mat[r, c]
    cam.width, cam.height = int(width), int(height)
    cam.focal_length[0], cam.focal_length[1] = focal_length
    cam.screen_center[0], cam.screen_center[1] = screen_center
    cam.spp_index = spp_index
    cam.snap_to_pixel_centers = 1 if snap else 0
    cam.near_distance = near
    cam.lens_mode = lens_mode
    for i, q in enumerate(lens_params):
        cam.lens_params[i] = q
    cam.aperture_size, cam.focus_z = aperture_size, focus_z
    if matrix1_3x4 is not None:  # the camera at the end of the frame's interval (camera_matrix1) + rolling shutter
        m1 = np.asarray(matrix1_3x4, np.float32)
        cam.has_matrix1 = 1
        for c in range(4):
            for r in range(3):
                cam.matrix1[c * 3 + r] = m1[r, c]
        for i in range(4):
            cam.rolling_shutter[i] = rolling_shutter[i]
    return cam


def make_opts(min_transmittance=0.01, background=(0.0, 0.0, 0.0, 1.0), exposure=0.0, to_srgb=False, spp=1, shard_index=0, shard_count=1,
              testbed_mode=MODE_NERF, render_mode=RENDER_SHADE, packed_output=False, depth_scale=0.0, color_space=0):
    o = RenderOpts()
    o.render_mode = render_mode
    o.min_transmittance = min_transmittance
    for i in range(4):
        o.background[i] = background[i]
    o.exposure = exposure
    o.to_srgb = int(to_srgb)
    o.spp = spp
    o.shard_index, o.shard_count = shard_index, shard_count
    o.testbed_mode = testbed_mode
    o.packed_output = int(packed_output)
    o.depth_scale = depth_scale
    o.color_space = color_space
    return o


class Context:
    """One ngp_ctx. Errors from the C ABI become RuntimeError (like the reference's exceptions through pybind11)."""

    def __init__(self, device=0, devices=None):
        """device: one HIP ordinal (-1: host-only). devices: a list of ordinals -> ngp_create_multi (the first is the primary)."""
        self.L = load_library()
        if devices is not None:
            arr = np.asarray(devices, np.int32)
            self.h = self.L.ngp_create_multi(_p(arr), arr.size)
        else:
            self.h = self.L.ngp_create(device)
        if not self.h:
            raise RuntimeError("ngp_create failed: no HIP device available (libngp_hip has no CPU fallback)")

    def n_devices(self):
        return self.L.ngp_n_devices(self.h)

    def device_render_stats(self, index):
        st = RenderStats()
        self._check(self.L.ngp_get_device_render_stats(self.h, index, C.byref(st)))
        return {k: getattr(st, k) for k, _ in RenderStats._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.L.ngp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.ngp_last_error(self.h).decode(errors="replace"))

    def set_schedule(self, *knobs):
        """refill_min, skip_steps, go_min, max_stall, k_busy, k_drain, block_jumps, share (ngp_set_schedule: validated)"""
        a = np.asarray(knobs, np.int32)
        self._check(self.L.ngp_set_schedule(self.h, _p(a), a.size))

    def profile_trace(self):
        """Wave timelines of the last frame (diagnostic build: NGP_PROFILE_SECTIONS + NGP_PROFILE_TRACE): (n_working_waves, headers [W, 16], records [W, I, 16])"""
        cw, ci = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        self._check(self.L.ngp_get_profile_trace(self.h, None, 0, _p(cw), _p(ci)))
        w, it = int(cw[0]), int(ci[0])
        buf = np.zeros(16 + w * 16 + w * it * 16, np.uint32)
        self._check(self.L.ngp_get_profile_trace(self.h, _p(buf), buf.size, None, None))
        return int(buf[0]), buf[16:16 + w * 16].reshape(w, 16), buf[16 + w * 16:].reshape(w, it, 16)

    # ---------------------------------------------------------------- model
    def set_model(self, scene):
        d = ModelDesc()
        enc = scene["encoding"]
        if enc.get("otype") in ("Frequency", "Identity"):  # configs/nerf/frequency.json, none.json
            if enc["otype"] == "Identity":
                d.pos_encoding = 2
            else:
                d.pos_encoding, d.pos_n_frequencies = 1, enc["n_frequencies"]
            de = scene.get("dir_encoding", {})
            if de.get("otype") == "Frequency":
                d.dir_encoding, d.dir_n_frequencies = 1, de["n_frequencies"]
            elif de.get("otype") == "Identity":
                d.dir_encoding = 2
            if scene["network"].get("otype", "FullyFusedMLP") != scene["rgb_network"].get("otype", "FullyFusedMLP"):
                raise ValueError("Frequency encodings: the density and the rgb network must be of the same otype")
        else:
            d.n_levels, d.n_features_per_level = enc["n_levels"], enc["n_features_per_level"]
            d.log2_hashmap_size, d.base_resolution = enc["log2_hashmap_size"], enc["base_resolution"]
            d.per_level_scale = enc["per_level_scale"]
        # the rgb network's alignment (nerf_network.h:83): 8 for CutlassMLP (linear.json, base_0layer.json, frequency.json)
        d.mlp_alignment = 8 if scene["rgb_network"].get("otype", "FullyFusedMLP") == "CutlassMLP" else 16
        d.n_neurons = scene["network"]["n_neurons"]
        d.n_hidden_density = scene["network"]["n_hidden_layers"]
        d.n_hidden_rgb = scene["rgb_network"]["n_hidden_layers"]
        d.density_out_dims = scene["network"].get("n_output_dims", 16)
        d.rgb_activation = scene.get("rgb_activation", 2)
        d.density_activation = scene.get("density_activation", 3)
        params = np.ascontiguousarray(scene["params"], np.uint16)
        grid = np.ascontiguousarray(np.asarray(scene["density_grid"]).astype(np.float16)).view(np.uint16)
        d.params_fp16, d.n_params = params.ctypes.data, params.size
        d.density_grid_fp16, d.n_density_grid = grid.ctypes.data, grid.size
        for i in range(3):
            d.aabb_min[i], d.aabb_max[i] = scene["aabb"][0][i], scene["aabb"][1][i]
            d.render_aabb_min[i], d.render_aabb_max[i] = scene["render_aabb"][0][i], scene["render_aabb"][1][i]
        r2l = np.asarray(scene.get("render_aabb_to_local", np.eye(3)), np.float32)
        for c in range(3):
            for r in range(3):
                d.render_aabb_to_local[c * 3 + r] = r2l[r, c]
        d.aabb_scale = scene["aabb_scale"]
        d.cone_angle_constant = scene["cone_angle_constant"]
        d.linear_colors = int(scene.get("linear_colors", False))
        self._check(self.L.ngp_set_model(self.h, C.byref(d)))

    def load_snapshot_bytes(self, data, compressed=False):
        buf = np.frombuffer(data, np.uint8)
        self._check(self.L.ngp_load_snapshot(self.h, _p(buf), buf.size, int(compressed)))

    def load_snapshot_file(self, path):
        self._check(self.L.ngp_load_snapshot_file(self.h, os.fsencode(path)))

    def save_snapshot_file(self, path, compress=True):
        self._check(self.L.ngp_save_snapshot_file(self.h, os.fsencode(path), int(compress)))

    def get_model(self):
        d = ModelDesc()
        self._check(self.L.ngp_get_model(self.h, C.byref(d)))
        return d

    def get_scene(self):
        """The loaded model as the dict set_model takes (and the oracle's make_model): what a snapshot left behind, copied out of the context"""
        from . import scene as S

        d = self.get_model()
        if d.pos_encoding != 0:
            raise RuntimeError("get_scene: grid models only")
        params = np.ctypeslib.as_array(C.cast(d.params_fp16, C.POINTER(C.c_uint16)), shape=(int(d.n_params),)).copy()
        grid = np.ctypeslib.as_array(C.cast(d.density_grid_fp16, C.POINTER(C.c_uint16)), shape=(int(d.n_density_grid),)).copy().view(np.float16)
        r2l = np.array([[d.render_aabb_to_local[c * 3 + r] for c in range(3)] for r in range(3)], np.float32)
        return {
            "encoding": {"otype": "HashGrid", "n_levels": d.n_levels, "n_features_per_level": d.n_features_per_level, "log2_hashmap_size": d.log2_hashmap_size,
                         "base_resolution": d.base_resolution, "per_level_scale": float(d.per_level_scale)},
            "network": {"otype": "CutlassMLP" if d.n_hidden_density == 0 else "FullyFusedMLP", "n_neurons": d.n_neurons, "n_hidden_layers": d.n_hidden_density, "n_output_dims": d.density_out_dims},
            "rgb_network": {"otype": "CutlassMLP" if d.mlp_alignment == 8 else "FullyFusedMLP", "n_neurons": d.n_neurons, "n_hidden_layers": d.n_hidden_rgb},
            "rgb_activation": d.rgb_activation, "density_activation": d.density_activation,
            "params": params, "density_grid": grid,
            "aabb": (tuple(d.aabb_min), tuple(d.aabb_max)), "render_aabb": (tuple(d.render_aabb_min), tuple(d.render_aabb_max)), "render_aabb_to_local": r2l,
            "aabb_scale": d.aabb_scale, "max_cascade": S.max_cascade_for(d.aabb_scale), "cone_angle_constant": float(d.cone_angle_constant), "linear_colors": bool(d.linear_colors),
        }

    def session_state(self):
        st = SessionState()
        self._check(self.L.ngp_get_session_state(self.h, C.byref(st)))
        return st

    def set_session_state(self, st, matrix_3x4=None, relative_focal_length=(1.0, 1.0), fov_axis=1, screen_center=(0.5, 0.5), zoom=1.0):
        m = None if matrix_3x4 is None else np.ascontiguousarray(np.asarray(matrix_3x4, np.float32).T.reshape(-1))
        rfl = np.asarray(relative_focal_length, np.float32); sc = np.asarray(screen_center, np.float32)
        self._check(self.L.ngp_set_session_state(self.h, C.byref(st), _p(m) if m is not None else None, _p(rfl), fov_axis, _p(sc), zoom))

    def snapshot_camera(self):
        m = np.zeros(12, np.float32); rfl = np.zeros(2, np.float32); sc = np.zeros(2, np.float32)
        ax = C.c_int32(0); zoom = C.c_float(0)
        if self.L.ngp_get_snapshot_camera(self.h, _p(m), _p(rfl), C.addressof(ax), _p(sc), C.addressof(zoom)) != 0:
            return None
        return {"matrix": m.reshape(4, 3).T.copy(), "relative_focal_length": rfl, "fov_axis": ax.value, "screen_center": sc, "zoom": zoom.value}

    # ---------------------------------------------------------------- data
    def load_training_data(self, path):
        self._check(self.L.ngp_load_training_data(self.h, os.fsencode(path)))

    def n_training_views(self):
        return self.L.ngp_n_training_views(self.h)

    def training_view(self, i):
        m = np.zeros(12, np.float32); res = np.zeros(2, np.int32); fl = np.zeros(2, np.float32); pp = np.zeros(2, np.float32)
        if self.L.ngp_get_training_view(self.h, i, _p(m), _p(res), _p(fl), _p(pp)) != 0:
            raise IndexError(i)
        mode = C.c_int32(0); lp = np.zeros(7, np.float32)
        self.L.ngp_get_training_view_lens(self.h, i, C.addressof(mode), _p(lp))
        return {"matrix": m.reshape(4, 3).T.copy(), "resolution": res, "focal_length": fl, "principal_point": pp, "lens_mode": mode.value, "lens_params": lp}

    def dataset_info(self):
        a = C.c_int32(0); s = C.c_float(0); off = np.zeros(3, np.float32); hdr = C.c_int32(0)
        self._check(self.L.ngp_get_dataset_info(self.h, C.addressof(a), C.addressof(s), _p(off), C.addressof(hdr)))
        return {"aabb_scale": a.value, "scale": s.value, "offset": off, "is_hdr": bool(hdr.value)}

    # ---------------------------------------------------------------- render
    def render(self, cam, opts=None, want_depth=False):
        opts = opts or make_opts()
        if not (0 < cam.width <= 65536 and 0 < cam.height <= 65536):  # (the library refuses it: no image to allocate for the call)
            self._check(self.L.ngp_render(self.h, C.byref(cam), C.byref(opts), None, None))
        rgba = np.zeros((cam.height, cam.width, 4), np.float32)
        depth = np.zeros((cam.height, cam.width), np.float32) if want_depth else None
        self._check(self.L.ngp_render(self.h, C.byref(cam), C.byref(opts), _p(rgba), _p(depth) if want_depth else None))
        return (rgba, depth) if want_depth else rgba

    def render_pinned(self, cam, opts=None):
        """ngp_render into a page-locked buffer from the library's pool (what pyngp's Testbed.render does): one DMA, no staging"""
        opts = opts or make_opts()
        rgba = host_image((cam.height, cam.width, 4))
        self._check(self.L.ngp_render(self.h, C.byref(cam), C.byref(opts), _p(rgba), None))
        return rgba

    def render_device(self, cam, opts, d_rgba_ptr, d_depth_ptr=None, stream=None):
        self._check(self.L.ngp_render_device(self.h, C.byref(cam), C.byref(opts), d_rgba_ptr, d_depth_ptr, stream))

    def render_stats(self):
        st = RenderStats()
        self._check(self.L.ngp_get_render_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in RenderStats._fields_}

    def render_history(self, n):
        arr = (RenderStats * n)()
        self._check(self.L.ngp_get_render_history(self.h, n, arr))
        return [{k: getattr(st, k) for k, _ in RenderStats._fields_} for st in arr]

    # ---------------------------------------------------------------- geometry mode
    def add_mesh(self, triangles, center=(0.0, 0.0, 0.0)):
        v = np.ascontiguousarray(triangles, np.float32).reshape(-1)
        c = np.asarray(center, np.float32)
        self._check(self.L.ngp_add_mesh(self.h, _p(v), v.size // 9, _p(c)))

    def load_mesh_file(self, path, center=(0.0, 0.0, 0.0)):
        c = np.asarray(center, np.float32)
        self._check(self.L.ngp_load_mesh_file(self.h, os.fsencode(path), _p(c)))

    def load_scene(self, path):
        self._check(self.L.ngp_load_scene(self.h, os.fsencode(path)))

    def clear_meshes(self):
        self._check(self.L.ngp_clear_meshes(self.h))

    def n_meshes(self):
        return self.L.ngp_n_meshes(self.h)

    def mesh_info(self, mesh=-1):
        nt, nn = C.c_uint32(0), C.c_uint32(0)
        bb = np.zeros(6, np.float32)
        if self.L.ngp_get_mesh_info(self.h, mesh, C.addressof(nt), C.addressof(nn), _p(bb)) != 0:
            raise IndexError(mesh)
        return {"n_tris": nt.value, "n_nodes": nn.value, "aabb": (bb[:3].copy(), bb[3:].copy())}

    def mesh_bvh(self, mesh):
        info = self.mesh_info(mesh)
        nodes = np.zeros(info["n_nodes"], BVH_NODE_DTYPE)
        tris = np.zeros(info["n_tris"], TRIANGLE_DTYPE)
        if self.L.ngp_get_mesh_bvh(self.h, mesh, _p(nodes), _p(tris)) != 0:
            raise IndexError(mesh)
        return nodes, tris

    def set_geometry_opts(self, sun_dir=(1.0, 1.0, 1.0), up_dir=(0.0, 1.0, 0.0), metallic=0.0, subsurface=0.0, specular=1.0, roughness=0.5,
                          sheen=0.0, clearcoat=0.0, clearcoat_gloss=0.0, basecolor=(0.8, 0.8, 0.8), ambientcolor=(0.0, 0.0, 0.0)):
        o = GeometryOpts()
        for i in range(3):
            o.sun_dir[i], o.up_dir[i], o.basecolor[i], o.ambientcolor[i] = sun_dir[i], up_dir[i], basecolor[i], ambientcolor[i]
        o.metallic, o.subsurface, o.specular, o.roughness = metallic, subsurface, specular, roughness
        o.sheen, o.clearcoat, o.clearcoat_gloss = sheen, clearcoat, clearcoat_gloss
        if self.L.ngp_set_geometry_opts(self.h, C.byref(o)) != 0:
            raise RuntimeError("ngp_set_geometry_opts failed")

    def trace_mesh_rays(self, positions, directions):
        p = np.ascontiguousarray(positions, np.float32).copy()
        d = np.ascontiguousarray(directions, np.float32).copy()
        self._check(self.L.ngp_trace_mesh_rays(self.h, p.shape[0], _p(p), _p(d)))
        return p, d

    # ---------------------------------------------------------------- irradiance probes
    def compute_envmap(self, mode=PROBE_CENTER, n_theta=32, n_phi=16, n_origin=1, origin=(0.0, 0.0, 0.0), min_transmittance=0.01):
        d = ProbeDesc()
        d.mode, d.n_theta, d.n_phi, d.n_origin, d.min_transmittance = mode, n_theta, n_phi, n_origin, min_transmittance
        for i in range(3):
            d.origin[i] = origin[i]
        env = np.zeros((n_phi, n_theta, 4), np.float32)
        self._check(self.L.ngp_compute_envmap(self.h, C.byref(d), _p(env)))
        return env

    def get_envmap(self):
        nt, nph = C.c_uint32(0), C.c_uint32(0)
        self._check(self.L.ngp_get_envmap(self.h, C.addressof(nt), C.addressof(nph), None, None))
        env = np.zeros((nph.value, nt.value, 4), np.float32)
        irr = np.zeros((nph.value, nt.value, 4), np.float32)
        self._check(self.L.ngp_get_envmap(self.h, None, None, _p(env), _p(irr)))
        return env, irr

    def irradiance(self, normals):
        nrm = np.ascontiguousarray(normals, np.float32)
        out = np.zeros((nrm.shape[0], 3), np.float32)
        self._check(self.L.ngp_irradiance(self.h, nrm.shape[0], _p(nrm), _p(out)))
        return out

    def compute_envmap_grid(self, grid_x=4, grid_y=4, n_theta=32, n_phi=16, shell_radius=1.0, min_transmittance=0.01):
        """Testbed::computeEnvmapGrid: (grid_x * grid_y, n_phi, n_theta, 4) probe textures, all traced in one launch"""
        d = ProbeGridDesc()
        d.grid_x, d.grid_y, d.n_theta, d.n_phi, d.shell_radius, d.min_transmittance = grid_x, grid_y, n_theta, n_phi, shell_radius, min_transmittance
        env = np.zeros((grid_x * grid_y, n_phi, n_theta, 4), np.float32)
        self._check(self.L.ngp_compute_envmap_grid(self.h, C.byref(d), _p(env)))
        return env

    def get_envmap_grid(self):
        """(desc, shell positions (G, 3), textures (G, n_phi, n_theta, 4), irradiance tables (G, n_phi, n_theta, 4))"""
        d = ProbeGridDesc()
        self._check(self.L.ngp_get_envmap_grid(self.h, C.byref(d), None))
        g = d.grid_x * d.grid_y
        org = np.zeros((g, 3), np.float32)
        self._check(self.L.ngp_get_envmap_grid(self.h, C.byref(d), _p(org)))
        env = np.zeros((g, d.n_phi, d.n_theta, 4), np.float32)
        irr = np.zeros((g, d.n_phi, d.n_theta, 4), np.float32)
        self._check(self.L.ngp_get_envmap(self.h, None, None, _p(env), _p(irr)))
        return d, org, env, irr

    def irradiance_at(self, positions, normals):
        pos = np.ascontiguousarray(positions, np.float32)
        nrm = np.ascontiguousarray(normals, np.float32)
        out = np.zeros((nrm.shape[0], 3), np.float32)
        self._check(self.L.ngp_irradiance_at(self.h, nrm.shape[0], _p(pos), _p(nrm), _p(out)))
        return out

    # ---------------------------------------------------------------- stages
    def grid_encode(self, pos01):
        pos01 = np.ascontiguousarray(pos01, np.float32)
        d = self.get_model()  # a Frequency position encoding (pos_encoding 1) is as wide as its padded 3 * 2 * n_frequencies
        al = d.mlp_alignment or 16
        width = (6 * d.pos_n_frequencies + al - 1) // al * al if d.pos_encoding == 1 else (3 + al - 1) // al * al if d.pos_encoding == 2 else 32
        out = np.zeros((pos01.shape[0], width), np.uint16)
        self._check(self.L.ngp_grid_encode(self.h, pos01.shape[0], _p(pos01), _p(out)))
        return out.view(np.float16)

    def density_gradient(self, pos01):
        """d density logit / d position (what ERenderMode::Normals composites): n x 3 floats"""
        pos01 = np.ascontiguousarray(pos01, np.float32)
        out = np.zeros((pos01.shape[0], 3), np.float32)
        self._check(self.L.ngp_density_gradient(self.h, pos01.shape[0], _p(pos01), _p(out)))
        return out

    def network(self, pos01, dir01):
        pos01 = np.ascontiguousarray(pos01, np.float32)
        dir01 = np.ascontiguousarray(dir01, np.float32)
        out = np.zeros((pos01.shape[0], 4), np.uint16)
        self._check(self.L.ngp_network_inference(self.h, pos01.shape[0], _p(pos01), _p(dir01), _p(out)))
        return out.view(np.float16)

    def density_bitfield(self):
        bf = np.zeros(128 ** 3 // 8 * 8, np.uint8)
        mean = C.c_float(0)
        self._check(self.L.ngp_get_density_bitfield(self.h, _p(bf), C.addressof(mean)))
        return bf, mean.value

    def set_render_aabb(self, lo, hi, to_local=None):
        lo = np.asarray(lo, np.float32); hi = np.asarray(hi, np.float32)
        r = None if to_local is None else np.ascontiguousarray(np.asarray(to_local, np.float32).T.reshape(-1))  # column-major
        self._check(self.L.ngp_set_render_aabb(self.h, _p(lo), _p(hi), _p(r) if r is not None else None))

    def set_envmap(self, rgba=None):
        """(H, W, 4) lat-long radiance behind the NeRF (m_envmap); None removes it"""
        if rgba is None:
            self._check(self.L.ngp_set_envmap(self.h, 0, 0, None))
            return
        env = np.ascontiguousarray(rgba, np.float32)
        self._check(self.L.ngp_set_envmap(self.h, env.shape[1], env.shape[0], _p(env)))

    def set_cone_angle_constant(self, value):
        self._check(self.L.ngp_set_cone_angle_constant(self.h, value))

    def update_density_grid(self, decay=0.95, n_uniform=0, n_nonuniform=0, n_iterations=1):
        """Testbed::update_density_grid_nerf: refresh the occupancy grid from the density network (0/0 = training_prep_nerf's schedule)."""
        self._check(self.L.ngp_update_density_grid(self.h, decay, n_uniform, n_nonuniform, n_iterations))

    def density_grid(self, max_cascade):
        out = np.zeros(128 ** 3 * (max_cascade + 1), np.float32)
        self._check(self.L.ngp_get_density_grid(self.h, _p(out), out.size))
        return out

    def init_rays(self, cam):
        pl = np.zeros(cam.width * cam.height, PAYLOAD_DTYPE)
        self._check(self.L.ngp_init_rays(self.h, C.byref(cam), _p(pl)))
        return pl

    # ------------------------------------------------------------- training
    def reset_network(self, log2_hashmap_size=19, seed=1337):
        self._check(self.L.ngp_reset_network(self.h, log2_hashmap_size, seed))

    def training_opts(self):
        o = TrainingOpts()
        self._check(self.L.ngp_get_training_opts(self.h, C.byref(o)))
        return o

    def set_training_opts(self, **kw):
        o = self.training_opts()
        for k, v in kw.items():
            if k == "background_color":
                o.background_color = (C.c_float * 3)(*v)
            elif k == "loss_type" and isinstance(v, str):
                o.loss_type = LOSS_TYPES[v]
            else:
                if not hasattr(o, k):
                    raise AttributeError(k)
                setattr(o, k, v)
        self._check(self.L.ngp_set_training_opts(self.h, C.byref(o)))

    def set_training_image(self, view, img):
        """img: (H, W, 4) uint8 (sRGB, straight alpha) or float32 (linear, premultiplied)."""
        if img.dtype == np.uint8:
            a, t = np.ascontiguousarray(img), IMAGE_BYTE
        else:
            a, t = np.ascontiguousarray(img, np.float32), IMAGE_FLOAT
        assert a.ndim == 3 and a.shape[2] == 4
        self._check(self.L.ngp_set_training_image(self.h, view, a.shape[1], a.shape[0], _p(a), t))

    def load_training_images(self):
        n = C.c_int32(0)
        self._check(self.L.ngp_load_training_images(self.h, C.byref(n)))
        return n.value

    def render_ground_truth(self, view, width, height, background=(0.0, 0.0, 0.0, 1.0), exposure=0.0, color_space=1, to_srgb=False, fov_axis=1, zoom=1.0):
        bg = np.asarray(background, np.float32)
        out = np.zeros((height, width, 4), np.float32)
        self._check(self.L.ngp_render_ground_truth(self.h, view, width, height, _p(bg), exposure, color_space, int(to_srgb), fov_axis, zoom, _p(out)))
        return out

    def train(self, n_steps=1, batch_size=1 << 18):
        loss = C.c_float(0)
        self._check(self.L.ngp_train(self.h, n_steps, batch_size, C.byref(loss)))
        return loss.value

    def training_state(self):
        st = TrainingState()
        self._check(self.L.ngp_get_training_state(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in TrainingState._fields_ if k != "struct_size"}

    def train_prepare_batch(self, batch_size):
        n_rays = self.training_state()["rays_per_batch"]
        counters = np.zeros(3, np.uint32)
        ray_indices = np.zeros(n_rays, np.uint32)
        numsteps = np.zeros((n_rays, 2), np.uint32)
        coords = np.zeros((batch_size, 7), np.float32)
        dloss = np.zeros((batch_size, 4), np.float16)
        loss = np.zeros(n_rays, np.float32)
        self._check(self.L.ngp_train_prepare_batch(self.h, batch_size, _p(counters), _p(ray_indices), _p(numsteps), _p(coords), _p(dloss), _p(loss)))
        return {"counters": counters, "ray_indices": ray_indices, "numsteps": numsteps, "coords": coords, "dloss": dloss, "loss": loss, "n_rays": n_rays}

    def train_gradients(self, batch_size):
        n = self.training_state()["n_params"] or self.get_model().n_params
        g = np.zeros(int(n), np.float32)
        self._check(self.L.ngp_train_gradients(self.h, batch_size, _p(g)))
        return g

    def train_apply(self):
        self._check(self.L.ngp_train_apply(self.h))

    def training_params(self):
        n = int(self.get_model().n_params)
        w, e = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self._check(self.L.ngp_get_training_params(self.h, _p(w), _p(e)))
        return w, e
