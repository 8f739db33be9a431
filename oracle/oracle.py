"""ORACLE -- TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (see oracle/orc_common.h).

ctypes binding of the CPU restatement in oracle/*.c. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

MAX_DEPTH = 16384.0
ACT_NONE, ACT_RELU, ACT_LOGISTIC, ACT_EXPONENTIAL = 0, 1, 2, 3


def build(native=False, out_dir=None):
    """Compile the C restatement (gcc). Returns the path of the shared library."""
    out = out_dir or _BUILD
    name = "liboracle_native.so" if native else "liboracle.so"
    target = ["native"] if native else ["all"]
    subprocess.run(["make", "-C", _HERE, "OUT=" + out] + target, check=True, capture_output=True)
    return os.path.join(out, name)


class NerfModel(C.Structure):
    _fields_ = [
        ("n_levels", C.c_uint32),
        ("n_features_per_level", C.c_uint32),
        ("log2_hashmap_size", C.c_uint32),
        ("base_resolution", C.c_uint32),
        ("per_level_scale", C.c_float),
        ("n_neurons", C.c_uint32),
        ("n_hidden_density", C.c_uint32),
        ("n_hidden_rgb", C.c_uint32),
        ("density_out_dims", C.c_uint32),
        ("rgb_activation", C.c_uint32),
        ("density_activation", C.c_uint32),
        ("params", C.c_void_p),
        ("n_params", C.c_uint64),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("render_aabb_min", C.c_float * 3),
        ("render_aabb_max", C.c_float * 3),
        ("render_aabb_to_local", C.c_float * 9),
        ("max_cascade", C.c_uint32),
        ("cone_angle_constant", C.c_float),
        ("density_grid_bitfield", C.c_void_p),
        ("grid_accumulate", C.c_uint32),
        ("pos_encoding", C.c_uint32), ("pos_n_frequencies", C.c_uint32),
        ("dir_encoding", C.c_uint32), ("dir_n_frequencies", C.c_uint32),
        ("mlp_alignment", C.c_uint32),
        ("mlp_accumulate", C.c_uint32),
        ("prepared", C.c_void_p),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("matrix", C.c_float * 12),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("focal_length", C.c_float * 2),
        ("screen_center", C.c_float * 2),
        ("spp_index", C.c_uint32),
        ("snap_to_pixel_centers", C.c_int32),
        ("near_distance", C.c_float),
        ("lens_mode", C.c_int32),
        ("lens_params", C.c_float * 7),
        ("aperture_size", C.c_float),
        ("focus_z", C.c_float),
        ("has_matrix1", C.c_int32),
        ("matrix1", C.c_float * 12),
        ("rolling_shutter", C.c_float * 4),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("min_transmittance", C.c_float),
        ("train_in_linear_colors", C.c_int32),
        ("depth_test", C.c_int32),
        ("capped_skip", C.c_int32),
        ("n_threads", C.c_int32),
        ("render_mode", C.c_int32),
        ("depth_scale", C.c_float),
        ("envmap", C.c_void_p),
        ("env_w", C.c_int32),
        ("env_h", C.c_int32),
    ]


class RenderStats(C.Structure):
    _fields_ = [
        ("n_rays", C.c_uint64),
        ("n_rays_alive_after_init", C.c_uint64),
        ("n_rays_hit", C.c_uint64),
        ("n_samples", C.c_uint64),
    ]


class MeshOpts(C.Structure):
    _fields_ = [
        ("sun_dir", C.c_float * 3), ("up_dir", C.c_float * 3),
        ("metallic", C.c_float), ("subsurface", C.c_float), ("specular", C.c_float), ("roughness", C.c_float),
        ("sheen", C.c_float), ("clearcoat", C.c_float), ("clearcoat_gloss", C.c_float),
        ("basecolor", C.c_float * 3), ("ambientcolor", C.c_float * 3),
        ("irradiance", C.c_void_p), ("n_theta", C.c_uint32), ("n_phi", C.c_uint32),
        ("grid_x", C.c_uint32), ("grid_y", C.c_uint32), ("probe_center", C.c_float * 3),
    ]


class ProbeGridDesc(C.Structure):
    _fields_ = [("grid_x", C.c_uint32), ("grid_y", C.c_uint32), ("n_theta", C.c_uint32), ("n_phi", C.c_uint32), ("shell_radius", C.c_float), ("center", C.c_float * 3)]


class ProbeDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_theta", C.c_uint32), ("n_phi", C.c_uint32), ("n_origin", C.c_uint32), ("origin", C.c_float * 3)]


PROBE_CENTER, PROBE_CENTER_OUTWARD, PROBE_MULTI_CENTER = 0, 1, 2


PAYLOAD_DTYPE = np.dtype(
    [("origin", "<f4", 3), ("dir", "<f4", 3), ("t", "<f4"), ("max_weight", "<f4"), ("idx", "<u4"),
     ("n_steps", "<u2"), ("alive", "u1"), ("pad", "u1")]
)
assert PAYLOAD_DTYPE.itemsize == 40


class Pcg32(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


class TrainImage(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("type", C.c_int32), ("res", C.c_int32 * 2), ("focal", C.c_float * 2), ("principal", C.c_float * 2),
                ("lens_mode", C.c_int32), ("lens_params", C.c_float * 7), ("xform", C.c_float * 12)]


class TrainOpts(C.Structure):
    _fields_ = [("n_rays", C.c_uint32), ("n_images", C.c_uint32), ("rng", Pcg32), ("snap_to_pixel_centers", C.c_int32), ("random_bg_color", C.c_int32),
                ("linear_colors", C.c_int32), ("color_space", C.c_int32), ("loss_type", C.c_int32), ("background", C.c_float * 3),
                ("near_distance", C.c_float), ("loss_scale", C.c_float), ("density_grid_mean", C.c_float)]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Thin, explicit wrapper: numpy in, numpy out."""

    def __init__(self, lib_path=None):
        if lib_path is None:
            lib_path = os.path.join(_BUILD, "liboracle.so")
            if not os.path.exists(lib_path):
                lib_path = build()
        self.lib = C.CDLL(lib_path)
        L = self.lib
        L.orc_nerf_prepare.argtypes = [C.POINTER(NerfModel)]
        L.orc_nerf_prepare.restype = C.c_int
        L.orc_nerf_release.argtypes = [C.POINTER(NerfModel)]
        L.orc_n_params.argtypes = [C.POINTER(NerfModel)]
        L.orc_n_params.restype = C.c_uint64
        L.orc_grid_layout.argtypes = [C.POINTER(NerfModel), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_grid_layout.restype = C.c_int
        L.orc_grid_encode.argtypes = [C.POINTER(NerfModel), C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_sh4_encode.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_nerf_network.argtypes = [C.POINTER(NerfModel), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_density_grid_to_bitfield.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_float)]
        L.orc_ld_random_val.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_ld_random_val.restype = C.c_float
        L.orc_ld_random_pixel_offset.argtypes = [C.c_uint32, C.c_void_p]
        L.orc_init_ray.argtypes = [C.POINTER(NerfModel), C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_advance_pos.argtypes = [C.POINTER(NerfModel), C.POINTER(Camera), C.c_void_p]
        L.orc_render_nerf.argtypes = [C.POINTER(NerfModel), C.POINTER(Camera), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p, C.POINTER(RenderStats)]
        L.orc_trace_payloads.argtypes = [C.POINTER(NerfModel), C.c_void_p, C.POINTER(RenderOpts), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RenderStats)]
        L.orc_accumulate.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_float]
        L.orc_tonemap.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p]
        L.orc_srgb_to_linear.argtypes = [C.c_float]
        L.orc_srgb_to_linear.restype = C.c_float
        L.orc_linear_to_srgb.argtypes = [C.c_float]
        L.orc_linear_to_srgb.restype = C.c_float
        self._bind_mesh()

    # ------------------------------------------------------------------ model
    def make_model(self, scene):
        """scene: dict produced by the package's synthetic/snapshot loaders (plain numpy + scalars)."""
        m = NerfModel()
        enc = scene["encoding"]
        if enc.get("otype", "HashGrid") in ("Frequency", "Identity"):  # configs/nerf/frequency.json, none.json: no grid at all
            if enc["otype"] == "Identity":
                m.pos_encoding = 2
            else:
                m.pos_encoding, m.pos_n_frequencies = 1, enc["n_frequencies"]
            m.n_levels = m.n_features_per_level = m.log2_hashmap_size = m.base_resolution = 0
            m.per_level_scale = 1.0
        else:
            m.n_levels = enc["n_levels"]
            m.n_features_per_level = enc["n_features_per_level"]
            m.log2_hashmap_size = enc["log2_hashmap_size"]
            m.base_resolution = enc["base_resolution"]
            m.per_level_scale = enc["per_level_scale"]
        de = scene.get("dir_encoding", {})
        if de.get("otype") == "Frequency":
            m.dir_encoding, m.dir_n_frequencies = 1, de["n_frequencies"]
        elif de.get("otype") == "Identity":
            m.dir_encoding = 2
        # the rgb network's alignment (nerf_network.h:83): the rgb input and output are padded to it; Frequency encodings pad the
        # position encoding to the density network's, and both networks are CutlassMLPs there
        m.mlp_alignment = 8 if scene["rgb_network"].get("otype", "FullyFusedMLP") == "CutlassMLP" else 16
        m.n_neurons = scene["network"]["n_neurons"]
        m.n_hidden_density = scene["network"]["n_hidden_layers"]
        m.n_hidden_rgb = scene["rgb_network"]["n_hidden_layers"]
        m.density_out_dims = scene["network"].get("n_output_dims", 16)
        m.rgb_activation = scene.get("rgb_activation", ACT_LOGISTIC)
        m.density_activation = scene.get("density_activation", ACT_EXPONENTIAL)
        params = np.ascontiguousarray(scene["params"], dtype=np.uint16)
        bitfield = np.ascontiguousarray(scene["density_grid_bitfield"], dtype=np.uint8)
        m.params = params.ctypes.data
        m.n_params = params.size
        for i in range(3):
            m.aabb_min[i] = scene["aabb"][0][i]
            m.aabb_max[i] = scene["aabb"][1][i]
            m.render_aabb_min[i] = scene["render_aabb"][0][i]
            m.render_aabb_max[i] = scene["render_aabb"][1][i]
        r2l = np.asarray(scene.get("render_aabb_to_local", np.eye(3)), dtype=np.float32)
        for c in range(3):
            for r in range(3):
                m.render_aabb_to_local[c * 3 + r] = r2l[r, c]
        m.max_cascade = scene["max_cascade"]
        m.cone_angle_constant = scene["cone_angle_constant"]
        m.density_grid_bitfield = bitfield.ctypes.data
        # corner sum of the grid encoding: "fma" (tvec-era tcnn, what libngp_hip.so ships) or "legacy" (oracle.h)
        m.grid_accumulate = {"legacy": 0, "fma": 1}[scene.get("grid_accumulate", "fma")]
        # how the MLPs sum: "exact" (default), "fp16_k16" (tcnn FullyFusedMLP's fp16 accumulator fragments) or "ideal" (float64 network); oracle.h
        m.mlp_accumulate = {"exact": 0, "fp16_k16": 1, "ideal": 2}[scene.get("mlp_accumulate", "exact")]
        m._keep = (params, bitfield)
        rc = self.lib.orc_nerf_prepare(C.byref(m))
        if rc != 0:
            raise RuntimeError(f"orc_nerf_prepare failed: {rc} (n_params={params.size}, need {self.lib.orc_n_params(C.byref(m))})")
        return m

    def release(self, m):
        self.lib.orc_nerf_release(C.byref(m))

    def grid_layout(self, m):
        n = m.n_levels
        off = np.zeros(n + 1, np.uint32)
        res = np.zeros(n, np.uint32)
        sc = np.zeros(n, np.float32)
        self.lib.orc_grid_layout(C.byref(m), _ptr(off), _ptr(res), _ptr(sc))
        return off, res, sc

    # ------------------------------------------------------------------ stages
    def grid_encode(self, m, pos01):
        pos01 = np.ascontiguousarray(pos01, np.float32)
        n = pos01.shape[0]
        out = np.zeros((n, m.n_levels * m.n_features_per_level), np.uint16)
        self.lib.orc_grid_encode(C.byref(m), n, _ptr(pos01), _ptr(out))
        return out.view(np.float16)

    def frequency_encode(self, x, n_frequencies):
        """tcnn FrequencyEncoding of n x d inputs: n x (d * 2 * n_frequencies) halves (unpadded)"""
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros((x.shape[0], x.shape[1] * 2 * n_frequencies), np.uint16)
        self.lib.orc_frequency_encode.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        self.lib.orc_frequency_encode.restype = None
        self.lib.orc_frequency_encode(x.shape[0], x.shape[1], n_frequencies, _ptr(x), _ptr(out))
        return out.view(np.float16)

    def density_gradient(self, m, pos01):
        pos01 = np.ascontiguousarray(pos01, np.float32)
        out = np.zeros((pos01.shape[0], 3), np.float32)
        self.lib.orc_density_gradient.argtypes = [C.POINTER(NerfModel), C.c_uint32, C.c_void_p, C.c_void_p]
        self.lib.orc_density_gradient.restype = None
        self.lib.orc_density_gradient(C.byref(m), pos01.shape[0], _ptr(pos01), _ptr(out))
        return out

    def sh4(self, dir01):
        dir01 = np.ascontiguousarray(dir01, np.float32)
        out = np.zeros((dir01.shape[0], 16), np.uint16)
        self.lib.orc_sh4_encode(dir01.shape[0], _ptr(dir01), _ptr(out))
        return out.view(np.float16)

    def network(self, m, pos01, dir01):
        pos01 = np.ascontiguousarray(pos01, np.float32)
        dir01 = np.ascontiguousarray(dir01, np.float32)
        out = np.zeros((pos01.shape[0], 4), np.uint16)
        self.lib.orc_nerf_network(C.byref(m), pos01.shape[0], _ptr(pos01), _ptr(dir01), _ptr(out))
        return out.view(np.float16)

    def density_grid_to_bitfield(self, grid, max_cascade):
        grid = np.ascontiguousarray(grid, np.float32)
        assert grid.size == 128 ** 3 * (max_cascade + 1)
        bf = np.zeros(128 ** 3 // 8 * 8, np.uint8)
        mean = C.c_float(0)
        self.lib.orc_density_grid_to_bitfield(_ptr(grid), max_cascade, _ptr(bf), C.byref(mean))
        return bf, mean.value

    def update_density_grid(self, m, grid, max_cascade, rng, ema_step, decay=0.95, n_uniform=0, n_nonuniform=0):
        """One iteration of update_density_grid_nerf on a float grid; rng = [state, inc] (mutated), returns (grid, ema_step)."""
        g = np.ascontiguousarray(grid, np.float32).copy()
        assert g.size == 128 ** 3 * (max_cascade + 1)
        r = (C.c_uint64 * 2)(rng[0], rng[1])
        step = C.c_uint32(ema_step)
        self.lib.orc_update_density_grid(C.byref(m), _ptr(g), max_cascade, r, C.byref(step), C.c_float(decay), n_uniform, n_nonuniform)
        rng[0], rng[1] = r[0], r[1]
        return g, step.value

    def grid_rng(self, seed=1337):
        """m_nerf.training.density_grid_rng after reset_network: pcg32(pcg32(seed).next_uint())"""
        a = (C.c_uint64 * 2)()
        self.lib.orc_pcg32_seed(a, C.c_uint64(seed), C.c_uint64(1))
        self.lib.orc_pcg32_next_uint.restype = C.c_uint32
        first = self.lib.orc_pcg32_next_uint(a)
        b = (C.c_uint64 * 2)()
        self.lib.orc_pcg32_seed(b, C.c_uint64(first), C.c_uint64(1))
        return [int(b[0]), int(b[1])]

    def ld_random_val(self, index, seed, dim=0):
        return self.lib.orc_ld_random_val(index & 0xFFFFFFFF, seed & 0xFFFFFFFF, dim)

    def pixel_offset(self, spp):
        out = np.zeros(2, np.float32)
        self.lib.orc_ld_random_pixel_offset(spp, _ptr(out))
        return out

    @staticmethod
    def make_camera(matrix_4x3, width, height, focal_length, screen_center=(0.5, 0.5), spp_index=0, snap=True, near=0.0, lens_mode=0, lens_params=(), aperture_size=0.0, focus_z=1.0,
                    matrix1_4x3=None, rolling_shutter=(0.0, 0.0, 0.0, 1.0)):
        """matrix_4x3: numpy (3,4) [R|t] camera-to-world in NGP convention."""
        cam = Camera()
        mat = np.asarray(matrix_4x3, np.float32)
        assert mat.shape == (3, 4)
        for c in range(4):
            for r in range(3):
                cam.matrix[c * 3 + r] = mat[r, c]
        cam.width, cam.height = width, height
        cam.focal_length[0], cam.focal_length[1] = focal_length
        cam.screen_center[0], cam.screen_center[1] = screen_center
        cam.spp_index = spp_index
        cam.snap_to_pixel_centers = 1 if snap else 0
        cam.near_distance = near
        cam.lens_mode = lens_mode
        for i, q in enumerate(lens_params):
            cam.lens_params[i] = q
        cam.aperture_size, cam.focus_z = aperture_size, focus_z
        if matrix1_4x3 is not None:
            m1 = np.asarray(matrix1_4x3, np.float32)
            cam.has_matrix1 = 1
            for c in range(4):
                for r in range(3):
                    cam.matrix1[c * 3 + r] = m1[r, c]
            for i in range(4):
                cam.rolling_shutter[i] = rolling_shutter[i]
        return cam

    @staticmethod
    def make_opts(min_transmittance=0.01, linear_colors=False, depth_test=False, capped_skip=False, n_threads=0, render_mode=0, depth_scale=1.0 / 0.33, envmap=None):
        o = RenderOpts()
        o.min_transmittance = min_transmittance
        o.train_in_linear_colors = int(linear_colors)
        o.depth_test = int(depth_test)
        o.capped_skip = int(capped_skip)
        o.n_threads = n_threads
        o.render_mode = render_mode
        o.depth_scale = depth_scale
        if envmap is not None:  # (H, W, 4) lat-long radiance behind the NeRF
            env = np.ascontiguousarray(envmap, np.float32)
            o._keep = env
            o.envmap, o.env_w, o.env_h = env.ctypes.data, env.shape[1], env.shape[0]
        return o

    def init_rays(self, m, cam, advance=True):
        n = cam.width * cam.height
        pl = np.zeros(n, PAYLOAD_DTYPE)
        for i in range(n):
            p = pl[i:i + 1]
            self.lib.orc_init_ray(C.byref(m), C.byref(cam), i % cam.width, i // cam.width, _ptr(p))
            if advance:
                self.lib.orc_advance_pos(C.byref(m), C.byref(cam), _ptr(p))
        return pl

    def render_nerf(self, m, cam, opts=None, frame_buffer=None, depth_buffer=None):
        opts = opts or self.make_opts()
        n = cam.width * cam.height
        fb = np.zeros((n, 4), np.float32) if frame_buffer is None else np.ascontiguousarray(frame_buffer, np.float32).reshape(n, 4).copy()
        db = np.zeros(n, np.float32) if depth_buffer is None else np.ascontiguousarray(depth_buffer, np.float32).reshape(n).copy()
        st = RenderStats()
        self.lib.orc_render_nerf(C.byref(m), C.byref(cam), C.byref(opts), _ptr(fb), _ptr(db), C.byref(st))
        stats = {k: getattr(st, k) for k, _ in RenderStats._fields_}
        return fb.reshape(cam.height, cam.width, 4), db.reshape(cam.height, cam.width), stats

    def trace_payloads(self, m, cam_matrix12, payloads, opts=None):
        opts = opts or self.make_opts()
        n = payloads.shape[0]
        rgba = np.zeros((n, 4), np.float32)
        depth = np.zeros(n, np.float32)
        st = RenderStats()
        cm = np.ascontiguousarray(cam_matrix12, np.float32)
        self.lib.orc_trace_payloads(C.byref(m), _ptr(cm), C.byref(opts), n, _ptr(payloads), _ptr(rgba), _ptr(depth), C.byref(st))
        return rgba, depth, {k: getattr(st, k) for k, _ in RenderStats._fields_}

    def accumulate(self, frame_buffer, accumulate_buffer, sample_count, color_space=0):
        fb = np.ascontiguousarray(frame_buffer, np.float32)
        acc = np.ascontiguousarray(accumulate_buffer, np.float32).copy()
        self.lib.orc_accumulate_cs(fb.size // 4, _ptr(fb), _ptr(acc), C.c_float(sample_count), int(color_space))
        return acc

    def tonemap(self, accumulate_buffer, background=(0, 0, 0, 1), exposure=0.0, to_srgb=False, color_space=0):
        acc = np.ascontiguousarray(accumulate_buffer, np.float32)
        bg = np.asarray(background, np.float32)
        out = np.zeros_like(acc)
        self.lib.orc_tonemap_cs(acc.size // 4, _ptr(acc), _ptr(bg), C.c_float(exposure), int(to_srgb), int(color_space), _ptr(out))
        return out

    # ------------------------------------------------------------------ mesh (orc_mesh.c)
    def _bind_mesh(self):
        L = self.lib
        L.orc_mesh_scene_create.argtypes = [C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]
        L.orc_mesh_scene_create.restype = C.c_void_p
        L.orc_mesh_scene_destroy.argtypes = [C.c_void_p]
        L.orc_mesh_scene_aabb.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_mesh_scene_n_nodes.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_mesh_scene_n_nodes.restype = C.c_uint32
        L.orc_trace_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_render_mesh.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(MeshOpts), C.c_void_p, C.c_void_p]
        L.orc_probe_n_rays.argtypes = [C.POINTER(ProbeDesc)]
        L.orc_probe_n_rays.restype = C.c_uint32
        L.orc_probe_payloads.argtypes = [C.POINTER(NerfModel), C.POINTER(ProbeDesc), C.c_void_p]
        L.orc_compute_envmap.argtypes = [C.POINTER(NerfModel), C.POINTER(ProbeDesc), C.POINTER(RenderOpts), C.c_void_p, C.POINTER(RenderStats)]
        L.orc_irradiance.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_texel_direction.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_irradiance_from.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_probe_grid_origin.argtypes = [C.POINTER(ProbeGridDesc), C.c_uint32, C.c_void_p]
        L.orc_compute_envmap_grid.argtypes = [C.POINTER(NerfModel), C.POINTER(ProbeGridDesc), C.POINTER(RenderOpts), C.c_void_p, C.POINTER(RenderStats)]
        L.orc_irradiance_grid_tabulate.argtypes = [C.POINTER(ProbeGridDesc), C.c_void_p, C.c_void_p]
        L.orc_irradiance_read.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_irradiance_grid_lookup.argtypes = [C.POINTER(ProbeGridDesc), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]

    # ------------------------------------------------------------------ probes (orc_probe.c)
    @staticmethod
    def make_probe(mode=0, n_theta=32, n_phi=16, n_origin=1, origin=(0.0, 0.0, 0.0)):
        d = ProbeDesc()
        d.mode, d.n_theta, d.n_phi, d.n_origin = mode, n_theta, n_phi, n_origin
        for i in range(3):
            d.origin[i] = origin[i]
        return d

    def probe_payloads(self, m, desc):
        pl = np.zeros(self.lib.orc_probe_n_rays(C.byref(desc)), PAYLOAD_DTYPE)
        self.lib.orc_probe_payloads(C.byref(m), C.byref(desc), _ptr(pl))
        return pl

    def compute_envmap(self, m, desc, opts=None):
        opts = opts or self.make_opts()
        env = np.zeros((desc.n_phi, desc.n_theta, 4), np.float32)
        st = RenderStats()
        self.lib.orc_compute_envmap(C.byref(m), C.byref(desc), C.byref(opts), _ptr(env), C.byref(st))
        return env, {k: getattr(st, k) for k, _ in RenderStats._fields_}

    def irradiance(self, envmap, normals, origin=None):
        """E(n) of one probe texture; origin = the shell position of an outward (K11) probe, whose texel (a, b) holds the
        radiance along -frame(normalize(origin)) * texel_direction(a, b)"""
        env = np.ascontiguousarray(envmap, np.float32)
        nrm = np.ascontiguousarray(normals, np.float32)
        out = np.zeros((nrm.shape[0], 3), np.float32)
        org = None if origin is None else np.ascontiguousarray(origin, np.float32)
        self.lib.orc_irradiance_from(env.shape[1], env.shape[0], _ptr(env), None if org is None else _ptr(org), nrm.shape[0], _ptr(nrm), _ptr(out))
        return out

    # ---- the grid of probes (Testbed::computeEnvmapGrid / ShadeGridEnvMap: definition in orc_probe.h)
    @staticmethod
    def make_probe_grid(grid_x=4, grid_y=4, n_theta=32, n_phi=16, shell_radius=1.0, center=(0.5, 0.5, 0.5)):
        d = ProbeGridDesc()
        d.grid_x, d.grid_y, d.n_theta, d.n_phi, d.shell_radius = grid_x, grid_y, n_theta, n_phi, shell_radius
        for i in range(3):
            d.center[i] = center[i]
        return d

    def probe_grid_origins(self, d):
        out = np.zeros((d.grid_x * d.grid_y, 3), np.float32)
        tmp = np.zeros(3, np.float32)
        for g in range(out.shape[0]):
            self.lib.orc_probe_grid_origin(C.byref(d), g, _ptr(tmp))
            out[g] = tmp
        return out

    def compute_envmap_grid(self, m, d, opts=None):
        """m: the model with render_aabb = the box the probe rays are traced in (in Geometry mode the inflated scene box)"""
        opts = opts or self.make_opts()
        env = np.zeros((d.grid_x * d.grid_y, d.n_phi, d.n_theta, 4), np.float32)
        st = RenderStats()
        self.lib.orc_compute_envmap_grid(C.byref(m), C.byref(d), C.byref(opts), _ptr(env), C.byref(st))
        return env, {k: getattr(st, k) for k, _ in RenderStats._fields_}

    def irradiance_grid_tabulate(self, d, envmaps):
        env = np.ascontiguousarray(envmaps, np.float32)
        out = np.zeros_like(env)
        self.lib.orc_irradiance_grid_tabulate(C.byref(d), _ptr(env), _ptr(out))
        return out

    def irradiance_read(self, table, normal):
        t = np.ascontiguousarray(table, np.float32)
        n = np.ascontiguousarray(normal, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.orc_irradiance_read(t.shape[1], t.shape[0], _ptr(t), _ptr(n), _ptr(out))
        return out

    def irradiance_grid_lookup(self, d, tables, positions, normals):
        t = np.ascontiguousarray(tables, np.float32)
        p = np.ascontiguousarray(positions, np.float32)
        n = np.ascontiguousarray(normals, np.float32)
        out = np.zeros((p.shape[0], 3), np.float32)
        self.lib.orc_irradiance_grid_lookup(C.byref(d), _ptr(t), p.shape[0], _ptr(p), _ptr(n), _ptr(out))
        return out

    def texel_directions(self, n_theta, n_phi):
        out = np.zeros((n_phi, n_theta, 3), np.float32)
        tmp = np.zeros(3, np.float32)
        for j in range(n_phi):
            for i in range(n_theta):
                self.lib.orc_texel_direction(n_theta, n_phi, i, j, _ptr(tmp))
                out[j, i] = tmp
        return out

    def mesh_scene(self, meshes):
        """meshes: list of (vertices float32 (n_tris, 3, 3) in file space, center (3,))."""
        n = len(meshes)
        verts = [np.ascontiguousarray(v, np.float32).reshape(-1) for v, _ in meshes]
        ptrs = (C.c_void_p * n)(*[v.ctypes.data for v in verts])
        ntris = np.array([v.size // 9 for v in verts], np.uint32)
        centers = np.ascontiguousarray(np.array([c for _, c in meshes], np.float32))
        h = self.lib.orc_mesh_scene_create(n, ptrs, _ptr(ntris), _ptr(centers))
        return h

    def mesh_scene_aabb(self, h):
        out = np.zeros(6, np.float32)
        self.lib.orc_mesh_scene_aabb(h, _ptr(out))
        return out[:3].copy(), out[3:].copy()

    def mesh_scene_destroy(self, h):
        self.lib.orc_mesh_scene_destroy(h)

    def trace_mesh(self, h, positions, directions):
        p = np.ascontiguousarray(positions, np.float32).copy()
        d = np.ascontiguousarray(directions, np.float32).copy()
        self.lib.orc_trace_mesh(h, p.shape[0], _ptr(p), _ptr(d))
        return p, d

    @staticmethod
    def make_mesh_opts(sun_dir=(1.0, 1.0, 1.0), up_dir=(0.0, 1.0, 0.0), metallic=0.0, subsurface=0.0, specular=1.0, roughness=0.5, sheen=0.0,
                       clearcoat=0.0, clearcoat_gloss=0.0, basecolor=(0.8, 0.8, 0.8), ambientcolor=(0.0, 0.0, 0.0), irradiance=None, grid=None,
                       probe_center=(0.5, 0.5, 0.5)):
        """irradiance: (n_phi, n_theta, 4) table of one probe (ShadeEnvMap) or, with grid = (grid_x, grid_y), the
        (grid_x * grid_y, n_phi, n_theta, 4) tables of a probe grid (ShadeGridEnvMap)"""
        o = MeshOpts()
        if irradiance is not None:
            irr = np.ascontiguousarray(irradiance, np.float32)
            assert irr.shape[-1] == 4 and irr.ndim == (4 if grid else 3)
            o._keep = irr
            o.irradiance, o.n_theta, o.n_phi = irr.ctypes.data, irr.shape[-2], irr.shape[-3]
            if grid:
                assert irr.shape[0] == grid[0] * grid[1]
                o.grid_x, o.grid_y = grid
                for i in range(3):
                    o.probe_center[i] = probe_center[i]
        for i in range(3):
            o.sun_dir[i], o.up_dir[i], o.basecolor[i], o.ambientcolor[i] = sun_dir[i], up_dir[i], basecolor[i], ambientcolor[i]
        o.metallic, o.subsurface, o.specular, o.roughness = metallic, subsurface, specular, roughness
        o.sheen, o.clearcoat, o.clearcoat_gloss = sheen, clearcoat, clearcoat_gloss
        return o

    def render_mesh(self, h, cam, opts=None):
        opts = opts or self.make_mesh_opts()
        n = cam.width * cam.height
        fb = np.zeros((n, 4), np.float32)
        db = np.zeros(n, np.float32)
        self.lib.orc_render_mesh(h, C.byref(cam), C.byref(opts), _ptr(fb), _ptr(db))
        return fb.reshape(cam.height, cam.width, 4), db.reshape(cam.height, cam.width)

    # ------------------------------------------------------------------ training step
    def train_rng(self, seed=1337, n_advances=0):
        """m_rng of training step `n_advances`: default_rng_t{seed}, one draw for the density-grid generator, one advance() per step."""
        r = Pcg32()
        self.lib.orc_pcg32_seed.argtypes = [C.POINTER(Pcg32), C.c_uint64, C.c_uint64]
        self.lib.orc_pcg32_next_uint.argtypes = [C.POINTER(Pcg32)]
        self.lib.orc_pcg32_advance.argtypes = [C.POINTER(Pcg32), C.c_uint64]
        self.lib.orc_pcg32_seed(C.byref(r), seed, 1)
        self.lib.orc_pcg32_next_uint(C.byref(r))
        for _ in range(n_advances):
            self.lib.orc_pcg32_advance(C.byref(r), 1 << 32)
        return r

    def make_train_images(self, views):
        """views: list of dicts {pixels (H, W, 4) uint8 | float32, xform (3, 4), focal (2), principal (2)}."""
        arr = (TrainImage * len(views))()
        keep = []
        for im, v in zip(arr, views):
            px = np.ascontiguousarray(v["pixels"])
            keep.append(px)
            im.pixels = px.ctypes.data
            im.type = 1 if px.dtype == np.uint8 else 3
            im.res[0], im.res[1] = px.shape[1], px.shape[0]
            im.focal[0], im.focal[1] = v["focal"]
            im.principal[0], im.principal[1] = v.get("principal", (0.5, 0.5))
            im.lens_mode = 0
            x = np.asarray(v["xform"], np.float32)
            for c in range(4):
                for r in range(3):
                    im.xform[c * 3 + r] = x[r, c]
        arr._keep = keep
        return arr

    def train_generate_samples(self, m, images, opts, max_samples):
        n = opts.n_rays
        numsteps, base = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        rays = np.zeros((n, 6), np.float32)
        coords = np.zeros((max_samples, 7), np.float32)
        self.lib.orc_train_generate_samples.restype = C.c_uint32
        self.lib.orc_train_generate_samples.argtypes = [C.POINTER(NerfModel), C.c_void_p, C.POINTER(TrainOpts), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        total = self.lib.orc_train_generate_samples(C.byref(m), C.cast(images, C.c_void_p), C.byref(opts), max_samples, _ptr(numsteps), _ptr(base), _ptr(rays), _ptr(coords))
        return {"numsteps": numsteps, "base": base, "rays": rays, "coords": coords, "total": total}

    def train_loss(self, m, images, opts, gen, network_output):
        n = opts.n_rays
        net = np.ascontiguousarray(network_output).view(np.uint16)
        compacted = np.zeros(n, np.uint32)
        loss = np.zeros(n, np.float32)
        dloss = np.zeros((gen["coords"].shape[0], 4), np.uint16)
        self.lib.orc_train_loss.restype = None
        self.lib.orc_train_loss.argtypes = [C.POINTER(NerfModel), C.c_void_p, C.POINTER(TrainOpts)] + [C.c_void_p] * 8
        self.lib.orc_train_loss(C.byref(m), C.cast(images, C.c_void_p), C.byref(opts), _ptr(gen["numsteps"]), _ptr(gen["base"]), _ptr(gen["rays"]), _ptr(gen["coords"]),
                                _ptr(net), _ptr(compacted), _ptr(loss), _ptr(dloss))
        return {"compacted_numsteps": compacted, "loss": loss, "dloss": dloss.view(np.float16)}
