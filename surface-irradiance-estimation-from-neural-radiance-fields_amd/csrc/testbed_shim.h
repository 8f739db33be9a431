// ngp::Testbed -- host-side mirror of the reference's facade (include/neural-graphics-primitives/testbed.h) for the
// inference path, written against the C ABI only (include/ngp_hip.h). Same member names, argument meaning and error
// behaviour (std::runtime_error) as the reference so that scripts/run.py-style drivers and src/python_api.cu-style
// bindings work unchanged for load -> set camera -> render. Training, GUI, VR, DLSS, SDF/image/volume are not here.
#pragma once

#include "../../include/ngp_hip.h"
#include "minijson.h"

#include <array>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace ngp {

enum class ETestbedMode : int { Nerf, Sdf, Image, Volume, Geometry, None }; // common.h:35-43
enum class ELossType : int { L2, L1, Mape, Smape, Huber, LogL1, RelativeL2 }; // common.h:84-92
enum class EColorSpace : int { Linear, SRGB, VisPosNeg }; // common.h
enum class ERenderMode : int { AO, Shade, ShadeNerf, ShadeEnvMap, ShadeGridEnvMap, Normals, Positions, Depth, Distortion, Cost, Slice, NumRenderModes, EncodingVis }; // common.h:58-72

class Testbed {
public:
	struct TrainingImageMetadata {
		std::array<int, 2> resolution;
		std::array<float, 2> focal_length;
		std::array<float, 2> principal_point;
		int32_t lens_mode = 0;             // ELensMode
		std::array<float, 7> lens_params{}; // Lens::params
	};
	struct NerfDatasetView { // testbed.nerf.training.dataset
		size_t n_images = 0;
		std::vector<TrainingImageMetadata> metadata;
		std::vector<std::array<float, 12>> xforms;
		int aabb_scale = 1;
		float scale = 1.f;
		std::array<float, 3> offset{0.f, 0.f, 0.f};
	};
	struct Nerf {
		float render_min_transmittance = 0.01f; // nerf.h:172
		float cone_angle_constant = 0.f;        // written through set_cone_angle_constant (run.py:167)
		float sharpen = 0.f;                    // training-image sharpening (nerf.h): accepted, no effect on inference
		bool render_with_lens_distortion = false; // accepted; only perspective (undistorted) lenses are rendered
		struct Training {
			NerfDatasetView dataset;
			int view = 0;
			// nerf.h:98-117 defaults; pushed to the context before every training step
			bool random_bg_color = true;
			bool linear_colors = false;
			ELossType loss_type = ELossType::Huber; // configs/nerf/base.json "loss" (reset_network, src/testbed.cu:3912)
			bool snap_to_pixel_centers = true;
			float near_distance = 0.1f;
			float density_grid_decay = 0.95f;
			int n_images_for_training = 0; // images with pixels on the device
		} training;
	} nerf;

	explicit Testbed(ETestbedMode mode = ETestbedMode::None, int device = 0) : m_testbed_mode(mode) {
		m_ctx = ngp_create(device);
		if (!m_ctx) throw std::runtime_error("Testbed: no HIP device available (the MI355X renderer has no CPU fallback)");
		reset_camera();
	}
	Testbed(ETestbedMode mode, const std::string& data_path, int device = 0) : Testbed(mode, device) { load_training_data(data_path); }
	// m_devices (testbed.h:1211-1253): several GPUs behind this Testbed -- devices[0] is the primary, a frame's camera tiles are dealt to all of them
	Testbed(ETestbedMode mode, const std::vector<int>& devices) : m_testbed_mode(mode) {
		m_ctx = devices.empty() ? ngp_create(0) : ngp_create_multi(devices.data(), (int)devices.size());
		if (!m_ctx) throw std::runtime_error("Testbed: the requested HIP devices are not available (the MI355X renderer has no CPU fallback)");
		reset_camera();
	}
	int n_devices() const { return ngp_n_devices(m_ctx); }
	~Testbed() { ngp_destroy(m_ctx); }
	Testbed(const Testbed&) = delete;
	Testbed& operator=(const Testbed&) = delete;

	// ---- loading (src/testbed.cu:125-152, 319-395, 5465-5475; src/testbed_geometry_training.cu:3101-3210)
	static ETestbedMode mode_from_scene(const std::string& scene) { // src/common_host.cu:146-166
		struct stat st;
		if (stat(scene.c_str(), &st) != 0) return ETestbedMode::None;
		auto ends = [&](const char* e) { size_t n = strlen(e); return scene.size() >= n && strcasecmp(scene.c_str() + scene.size() - n, e) == 0; };
		if (S_ISDIR(st.st_mode) || ends(".json")) {
			size_t slash = scene.find_last_of('/');
			std::string fname = slash == std::string::npos ? scene : scene.substr(slash + 1);
			return fname.find("geometry") != std::string::npos ? ETestbedMode::Geometry : ETestbedMode::Nerf;
		}
		if (ends(".obj") || ends(".stl")) return ETestbedMode::Sdf;
		if (ends(".nvdb")) return ETestbedMode::Volume;
		return ETestbedMode::Image;
	}
	void load_training_data(const std::string& path) {
		ETestbedMode scene_mode = mode_from_scene(path);
		if (scene_mode == ETestbedMode::None) throw std::runtime_error("Data path '" + path + "' does not exist.");
		if (scene_mode != ETestbedMode::Nerf && scene_mode != ETestbedMode::Geometry) throw std::runtime_error("Only NeRF and geometry scenes are supported by the MI355X renderer.");
		m_testbed_mode = scene_mode;
		m_data_path = path;
		if (scene_mode == ETestbedMode::Geometry) {
			check(ngp_load_scene(m_ctx, path.c_str()));
			sync_model_state();
		} else {
			check(ngp_load_training_data(m_ctx, path.c_str()));
			sync_dataset();
			// images (PNG, baseline JPEG) for training; a dataset whose images are absent or of another format still serves its cameras
			int32_t n_loaded = 0;
			if (ngp_load_training_images(m_ctx, &n_loaded) != 0) n_loaded = 0;
			nerf.training.n_images_for_training = n_loaded;
			m_training_data_available = n_loaded > 0;
		}
	}
	// ---- training (src/testbed.cu:3820-4210 reset_network, 4364-4470 train, 3600-3660 frame; python_api.cu:416-434)
	void reset_network(bool reset_density_grid = true) {
		(void)reset_density_grid; // a fresh network always starts from an empty occupancy grid here
		check(ngp_reset_network(m_ctx, m_log2_hashmap_size, m_seed));
		m_training_step = 0;
		m_loss = 0.f;
		sync_model_state();
	}
	void reload_network_from_file(const std::string& path = "") {
		// configs/nerf/*.json: the render path is specialised for base.json's shapes; what varies between the shipped
		// configs that it accepts is the table size (base 19, small 15, base_14 14, big 21)
		if (!path.empty()) {
			FILE* f = fopen(path.c_str(), "rb");
			if (!f) throw std::runtime_error("Network config \"" + path + "\" does not exist.");
			std::string text;
			char buf[4096];
			size_t got;
			while ((got = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
			fclose(f);
			// what the trainer builds is configs/nerf/base.json's architecture (HashGrid + FullyFusedMLP 64-wide heads, SH degree 4 over
			// an Identity remainder); the table size is what varies between the shipped configs it accepts (base 19, small 15, base_14 14,
			// big 21). Anything else is refused by name rather than trained as something it is not.
			const mj::Value cfg = mj::parse_json(text);
			auto otype = [](const mj::Value& v) { return v.is_object() && v.contains("otype") ? v.at("otype").str() : std::string(); };
			if (cfg.contains("encoding")) {
				const mj::Value& e = cfg.at("encoding");
				const std::string ot = otype(e);
				if (!ot.empty() && ot != "HashGrid" && ot != "Grid") throw std::runtime_error("network config: encoding \"" + ot + "\" cannot be trained by the MI355X path (HashGrid only)");
				if (e.contains("log2_hashmap_size")) m_log2_hashmap_size = (uint32_t)e.at("log2_hashmap_size").integer();
			}
			for (const char* net : {"network", "rgb_network"})
				if (cfg.contains(net) && !otype(cfg.at(net)).empty() && otype(cfg.at(net)) != "FullyFusedMLP")
					throw std::runtime_error(std::string("network config: ") + net + " \"" + otype(cfg.at(net)) + "\" cannot be trained by the MI355X path (FullyFusedMLP only)");
			if (cfg.contains("dir_encoding")) {
				const mj::Value& de = cfg.at("dir_encoding");
				std::string first = otype(de);
				if (first == "Composite" && de.contains("nested") && de.at("nested").is_array() && de.at("nested").size() > 0) first = otype(de.at("nested").at(0));
				if (first != "SphericalHarmonics") throw std::runtime_error("network config: dir_encoding \"" + first + "\" cannot be trained by the MI355X path (SphericalHarmonics degree 4)");
			}
		}
		reset_network();
	}
	void set_training_image(int frame_idx, int width, int height, const float* rgba_linear_premultiplied) { // Nerf::Training::set_image, python_api.cu:46-64
		check(ngp_set_training_image(m_ctx, frame_idx, width, height, rgba_linear_premultiplied, NGP_IMAGE_FLOAT));
		m_training_data_available = true;
	}
	void train(uint32_t batch_size) {
		if (!m_training_data_available) { // src/testbed.cu:4365-4369
			m_train = false;
			return;
		}
		if (m_testbed_mode == ETestbedMode::None) throw std::runtime_error("Cannot train without a mode.");
		ngp_model_desc d{};
		if (ngp_get_model(m_ctx, &d) != 0 || d.n_params == 0) reset_network(); // "Creating neural network trainer."
		ngp_training_opts o{};
		check(ngp_get_training_opts(m_ctx, &o));
		o.loss_type = (int32_t)nerf.training.loss_type;
		o.random_bg_color = nerf.training.random_bg_color;
		o.linear_colors = nerf.training.linear_colors;
		o.snap_to_pixel_centers = nerf.training.snap_to_pixel_centers;
		o.near_distance = nerf.training.near_distance;
		o.density_grid_decay = nerf.training.density_grid_decay;
		o.train_network = m_train_network;
		o.train_encoding = m_train_encoding;
		o.color_space = (int32_t)m_color_space;
		for (int i = 0; i < 3; ++i) o.background_color[i] = m_background_color[(size_t)i];
		check(ngp_set_training_opts(m_ctx, &o));
		float loss = 0.f;
		check(ngp_train(m_ctx, 1, batch_size, &loss));
		ngp_training_state st{};
		check(ngp_get_training_state(m_ctx, &st));
		m_training_step = st.training_step;
		m_loss = st.loss;
	}
	bool frame() { // headless: one training step when shall_train is set; there is no window to close
		if (m_train) train(m_training_batch_size);
		return true;
	}
	void load_snapshot(const std::string& path) {
		check(ngp_load_snapshot_file(m_ctx, path.c_str()));
		if (m_testbed_mode != ETestbedMode::Geometry) m_testbed_mode = ETestbedMode::Nerf;
		sync_model_state();
		ngp_session_state st{};
		if (ngp_get_session_state(m_ctx, &st) == 0 && st.valid) { // src/testbed.cu:5395-5418
			memcpy(m_background_color.data(), st.background_color, 16);
			m_exposure = st.exposure;
			memcpy(m_sun_dir.data(), st.sun_dir, 12);
			memcpy(m_up_dir.data(), st.up_dir, 12);
			m_scale = st.camera_scale;
			m_aperture_size = st.aperture_size;
			m_slice_plane_z = st.autofocus_depth;
		}
		float m[12], rfl[2], sc[2], zoom;
		int32_t axis;
		if (ngp_get_snapshot_camera(m_ctx, m, rfl, &axis, sc, &zoom) == 0) { // src/testbed.cu:5404-5420
			memcpy(m_camera.data(), m, sizeof(m));
			m_relative_focal_length = {rfl[0], rfl[1]};
			m_fov_axis = axis;
			m_screen_center = {sc[0], sc[1]};
			m_zoom = zoom;
		}
	}
	void save_snapshot(const std::string& path, bool include_optimizer_state = false, bool compress = true) {
		(void)include_optimizer_state; // inference state only
		ngp_session_state st{}; // src/testbed.cu:5245-5263: the session travels with the model
		st.valid = 1;
		memcpy(st.background_color, m_background_color.data(), 16);
		st.exposure = m_exposure;
		memcpy(st.sun_dir, m_sun_dir.data(), 12);
		memcpy(st.up_dir, m_up_dir.data(), 12);
		st.camera_scale = m_scale;
		st.aperture_size = m_aperture_size;
		st.autofocus_depth = m_slice_plane_z;
		check(ngp_set_session_state(m_ctx, &st, m_camera.data(), m_relative_focal_length.data(), m_fov_axis, m_screen_center.data(), m_zoom));
		check(ngp_save_snapshot_file(m_ctx, path.c_str(), compress ? 1 : 0));
	}
	// src/testbed_nerf.cu:2772 (the stream argument is the context's); n = 0 / 0 selects training_prep_nerf's schedule
	void update_density_grid_nerf(float decay, uint32_t n_uniform_density_grid_samples, uint32_t n_nonuniform_density_grid_samples) {
		check(ngp_update_density_grid(m_ctx, decay, n_uniform_density_grid_samples, n_nonuniform_density_grid_samples, 1));
	}
	void load_file(const std::string& path) { // src/testbed.cu:319-395
		auto ends = [&](const char* e) { size_t n = strlen(e); return path.size() >= n && strcasecmp(path.c_str() + path.size() - n, e) == 0; };
		if (ends(".ingp") || ends(".msgpack")) { load_snapshot(path); return; }
		load_training_data(path);
	}
	void load_mesh(const std::string& path, const std::array<float, 3>& center = {0.f, 0.f, 0.f}) {
		check(ngp_load_mesh_file(m_ctx, path.c_str(), center.data()));
		m_testbed_mode = ETestbedMode::Geometry;
	}

	// ---- camera path (include/neural-graphics-primitives/camera_path.h, src/camera_path.cu:30-160, src/testbed.cu:3724-3740)
	struct CameraKeyframe {
		std::array<float, 4> R{0.f, 0.f, 0.f, 1.f}; // quaternion x, y, z, w
		std::array<float, 3> T{0.f, 0.f, 0.f};
		float slice = 0.f, scale = 1.f, fov = 50.f, aperture_size = 0.f;
		std::array<float, 12> m() const { // to_mat3(normalize(R)) and T, column-major 4x3
			float n = std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2] + R[3] * R[3]);
			const float x = R[0] / n, y = R[1] / n, z = R[2] / n, w = R[3] / n;
			return {1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y),
			        2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x),
			        2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y),
			        T[0], T[1], T[2]};
		}
	};
	struct CameraPath {
		std::vector<CameraKeyframe> keyframes;
		bool loop = false;
		float play_time = 0.f;
		const CameraKeyframe& get_keyframe(int i) const {
			const int size = (int)keyframes.size();
			return loop ? keyframes[(size_t)((i % size + size) % size)] : keyframes[(size_t)std::min(std::max(i, 0), size - 1)];
		}
		// cubic B-spline over the keyframes, quaternions added on the same hemisphere and renormalised (camera_path.cu:58-76)
		CameraKeyframe eval_camera_path(float t) const {
			if (keyframes.empty()) return {};
			t *= (float)(loop ? keyframes.size() : keyframes.size() - 1);
			const int t1 = (int)std::floor(t);
			const float u = t - std::floor(t), uu = u * u, uuu = uu * u;
			const float w[4] = {(1.f - u) * (1.f - u) * (1.f - u) * (1.f / 6.f), (3.f * uuu - 6.f * uu + 4.f) * (1.f / 6.f), (-3.f * uuu + 3.f * uu + 3.f * u + 1.f) * (1.f / 6.f), uuu * (1.f / 6.f)};
			CameraKeyframe acc;
			acc.R = {0.f, 0.f, 0.f, 0.f};
			acc.T = {0.f, 0.f, 0.f};
			acc.slice = acc.scale = acc.fov = acc.aperture_size = 0.f;
			for (int k = 0; k < 4; ++k) {
				const CameraKeyframe& p = get_keyframe(t1 - 1 + k);
				// operator+ flips the right-hand quaternion onto the running sum's hemisphere
				const float d = k == 0 ? 1.f : acc.R[0] * p.R[0] * w[k] + acc.R[1] * p.R[1] * w[k] + acc.R[2] * p.R[2] * w[k] + acc.R[3] * p.R[3] * w[k];
				const float sgn = d < 0.f ? -1.f : 1.f;
				for (int c = 0; c < 4; ++c) acc.R[(size_t)c] += sgn * w[k] * p.R[(size_t)c];
				for (int c = 0; c < 3; ++c) acc.T[(size_t)c] += w[k] * p.T[(size_t)c];
				acc.slice += w[k] * p.slice; acc.scale += w[k] * p.scale; acc.fov += w[k] * p.fov; acc.aperture_size += w[k] * p.aperture_size;
			}
			const float n = std::sqrt(acc.R[0] * acc.R[0] + acc.R[1] * acc.R[1] + acc.R[2] * acc.R[2] + acc.R[3] * acc.R[3]);
			for (float& c : acc.R) c /= n;
			return acc;
		}
	} m_camera_path;
	bool m_camera_smoothing = false;
	// CameraPath::load (src/camera_path.cu:124-148): {"loop", "time", "path": [{"R": [x, y, z, w], "T": [..], "slice", "scale", "fov", "aperture_size" | "dof"}]}
	// (the quaternion's serialisation lives in the un-vendored tiny-cuda-nn: storage order x, y, z, w)
	void load_camera_path(const std::string& path) {
		FILE* f = fopen(path.c_str(), "rb");
		if (!f) throw std::runtime_error("Camera path " + path + " does not exist.");
		std::string text;
		char buf[4096];
		size_t got;
		while ((got = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
		fclose(f);
		const mj::Value j = mj::parse_json(text);
		m_camera_path.keyframes.clear();
		m_camera_path.loop = j.contains("loop") ? j.at("loop").boolean() : false;
		m_camera_path.play_time = j.contains("time") ? (float)j.at("time").num() : 0.f;
		if (j.contains("path")) {
			for (size_t i = 0; i < j.at("path").size(); ++i) {
				const mj::Value& el = j.at("path").at(i);
				CameraKeyframe p;
				for (size_t c = 0; c < 4; ++c) p.R[c] = (float)el.at("R").at(c).num();
				for (size_t c = 0; c < 3; ++c) p.T[c] = (float)el.at("T").at(c).num();
				p.slice = (float)el.at("slice").num();
				p.scale = (float)el.at("scale").num();
				p.fov = (float)el.at("fov").num();
				p.aperture_size = (float)(el.contains("dof") ? el.at("dof").num() : el.at("aperture_size").num());
				m_camera_path.keyframes.push_back(p);
			}
		}
	}
	void set_camera_from_time(float t) { // src/testbed.cu:3724-3740
		if (m_camera_path.keyframes.empty()) return;
		const CameraKeyframe k = m_camera_path.eval_camera_path(t);
		m_camera = k.m();
		m_slice_plane_z = k.slice;
		m_scale = k.scale;
		set_fov(k.fov);
		m_aperture_size = k.aperture_size;
	}

	// ---- camera (src/testbed.cu:425-427, 481-496, 541-566, 3750-3764, 4474-4481)
	void reset_camera() {
		m_fov_axis = 1;
		m_zoom = 1.0f;
		m_screen_center = {0.5f, 0.5f};
		set_fov(50.625f);
		m_scale = 1.5f;
		m_camera = {1.f, 0.f, 0.f, 0.f, -1.f, 0.f, 0.f, 0.f, -1.f, 0.5f, 0.5f, 0.5f};
		for (int i = 0; i < 3; ++i) m_camera[9 + i] -= m_scale * m_camera[6 + i];
	}
	void set_nerf_camera_matrix(const std::array<float, 12>& cam /* column-major 4x3, NeRF convention */) {
		std::array<float, 12> r = cam; // NerfDataset::nerf_matrix_to_ngp, nerf_loader.h:101-120
		const auto& ds = nerf.training.dataset;
		for (int i = 0; i < 3; ++i) {
			r[3 + i] *= -1.f;
			r[6 + i] *= -1.f;
			r[9 + i] = r[9 + i] * ds.scale + ds.offset[i];
		}
		for (int c = 0; c < 4; ++c) {
			float t = r[c * 3 + 0];
			r[c * 3 + 0] = r[c * 3 + 1];
			r[c * 3 + 1] = r[c * 3 + 2];
			r[c * 3 + 2] = t;
		}
		m_camera = r;
	}
	void set_camera_to_training_view(int trainview) {
		const auto& ds = nerf.training.dataset;
		if (trainview < 0 || (size_t)trainview >= ds.n_images) throw std::runtime_error("Invalid training view.");
		m_camera = ds.xforms[(size_t)trainview];
		const auto& md = ds.metadata[(size_t)trainview];
		for (int i = 0; i < 2; ++i) m_relative_focal_length[i] = md.focal_length[i] / (float)md.resolution[m_fov_axis];
		m_screen_center = {1.0f - md.principal_point[0], 1.0f - md.principal_point[1]};
		nerf.render_with_lens_distortion = true; // src/testbed.cu:486-487
		m_render_lens_mode = md.lens_mode;
		m_render_lens_params = md.lens_params;
		nerf.training.view = trainview;
	}
	float fov() const { return 2.0f * 180.0f / 3.14159265358979323846f * std::atan(1.0f / (m_relative_focal_length[m_fov_axis] * 2.0f)); }
	void set_fov(float degrees) {
		float f = 0.5f / std::tan(0.5f * degrees * 3.14159265358979323846f / 180.0f);
		m_relative_focal_length = {f, f};
	}

	// ---- render: Testbed::render_to_cpu (src/python_api.cu:124-202). out: height*width*4 floats.
	void render_to_cpu(float* out, int width, int height, int spp, bool linear, float start_time = -1.f, float end_time = -1.f, float fps = 30.f, float shutter_fraction = 1.0f, float* depth_out = nullptr) {
		(void)fps;
		if (start_time >= 0.f) {
			// Testbed::render_to_cpu along the camera path (src/python_api.cu:124-202): EVERY sample of the frame is rendered between
			// the camera at the start of the shutter interval (camera_matrix0) and the camera at its end (camera_matrix1 =
			// camera_log_lerp(start, end, shutter_fraction)) with rolling_shutter (0, 0, 0, 1), i.e. each pixel of each sample takes
			// a low-discrepancy time in the interval -- motion blur. Here camera1 is the path itself evaluated at
			// start + shutter_fraction (end - start) rather than the matrix-logarithm blend of the two end points (tcnn's
			// mat_log / mat_exp are not in the reference mount); for a shutter fraction of 1 the two coincide.
			if (m_camera_smoothing) throw std::runtime_error("camera_smoothing is not supported by the MI355X renderer");
			if (end_time < 0.f) end_time = start_time;
			set_camera_from_time(start_time + (end_time - start_time) * shutter_fraction);
			const std::array<float, 12> end_cam = m_camera;
			set_camera_from_time(start_time);
			m_camera_end = end_cam;
			m_has_camera_end = true;
			try {
				render_to_cpu(out, width, height, spp, linear, -1.f, -1.f, fps, shutter_fraction, depth_out);
			} catch (...) {
				m_has_camera_end = false;
				throw;
			}
			m_has_camera_end = false;
			// (the reference leaves m_camera at the middle of the last sample's slice, python_api.cu:169-171)
			set_camera_from_time(start_time + (end_time - start_time) * (((float)spp - 0.5f) / (float)spp * shutter_fraction));
			return;
		}
		const bool gbuffer = m_render_mode == ERenderMode::AO || m_render_mode == ERenderMode::Positions || m_render_mode == ERenderMode::Depth || m_render_mode == ERenderMode::Cost ||
		                     m_render_mode == ERenderMode::Normals;
		const bool shade_family = m_render_mode == ERenderMode::Shade || m_render_mode == ERenderMode::ShadeEnvMap || m_render_mode == ERenderMode::ShadeGridEnvMap;
		if (!shade_family && !gbuffer) throw std::runtime_error("render modes supported: Shade, ShadeEnvMap, ShadeGridEnvMap, AO, Normals, Positions, Depth, Cost");
		// pre computation of the envmap: src/main.cu:184-188 runs it before the first frame; a Python session has no such hook
		// (python_api.cu binds neither function), so the first Geometry-mode render in these modes runs it with the defaults
		if (m_testbed_mode == ETestbedMode::Geometry && ngp_n_meshes(m_ctx) > 0) {
			if (m_render_mode == ERenderMode::ShadeGridEnvMap && !m_envmap_grid_ready) computeEnvmapGrid();
			if (m_render_mode == ERenderMode::ShadeEnvMap && !m_envmap_ready) computeEnvmapMultipleMain();
		}
		ngp_camera cam{};
		memcpy(cam.matrix, m_camera.data(), sizeof(cam.matrix));
		cam.width = width;
		cam.height = height;
		const float res_axis = (float)(m_fov_axis == 0 ? width : height);
		cam.focal_length[0] = m_relative_focal_length[0] * res_axis * m_zoom; // calc_focal_length
		cam.focal_length[1] = m_relative_focal_length[1] * res_axis * m_zoom;
		cam.screen_center[0] = (0.5f - m_screen_center[0]) * m_zoom + 0.5f;   // render_screen_center
		cam.screen_center[1] = (0.5f - m_screen_center[1]) * m_zoom + 0.5f;
		cam.spp_index = 0;
		cam.snap_to_pixel_centers = m_snap_to_pixel_centers ? 1 : 0;
		cam.near_distance = m_render_near_distance;
		cam.aperture_size = m_aperture_size; // src/testbed_nerf.cu:2342, 2380
		cam.focus_z = m_slice_plane_z + m_scale;
		if (m_has_camera_end) { // camera_matrix1 + m_rolling_shutter of render_frame
			cam.has_matrix1 = 1;
			memcpy(cam.matrix1, m_camera_end.data(), sizeof(cam.matrix1));
			memcpy(cam.rolling_shutter, m_rolling_shutter.data(), sizeof(cam.rolling_shutter));
		}
		if (nerf.render_with_lens_distortion) { // m_nerf.render_lens, src/testbed_nerf.cu render_nerf
			cam.lens_mode = m_render_lens_mode;
			memcpy(cam.lens_params, m_render_lens_params.data(), sizeof(cam.lens_params));
		}
		ngp_render_opts o{};
		o.render_mode = m_render_mode == ERenderMode::ShadeEnvMap ? NGP_RENDER_SHADE_ENVMAP : m_render_mode == ERenderMode::ShadeGridEnvMap ? NGP_RENDER_SHADE_GRID_ENVMAP : m_render_mode == ERenderMode::AO ? NGP_RENDER_AO
		              : m_render_mode == ERenderMode::Normals ? NGP_RENDER_NORMALS : m_render_mode == ERenderMode::Positions ? NGP_RENDER_POSITIONS : m_render_mode == ERenderMode::Depth ? NGP_RENDER_DEPTH : m_render_mode == ERenderMode::Cost ? NGP_RENDER_COST : NGP_RENDER_SHADE;
		o.min_transmittance = nerf.render_min_transmittance;
		memcpy(o.background, m_background_color.data(), sizeof(o.background));
		o.exposure = m_exposure;
		o.to_srgb = linear ? 0 : 1;
		o.spp = spp;
		o.shard_index = 0;
		o.shard_count = 1;
		o.testbed_mode = m_testbed_mode == ETestbedMode::Geometry ? NGP_MODE_GEOMETRY : NGP_MODE_NERF;
		o.packed_output = 0;
		if (m_color_space == EColorSpace::VisPosNeg) throw std::runtime_error("color space VisPosNeg is not supported");
		o.color_space = m_color_space == EColorSpace::SRGB ? 1 : 0;
		if (nerf.cone_angle_constant != m_pushed_cone_angle) {
			check(ngp_set_cone_angle_constant(m_ctx, nerf.cone_angle_constant));
			m_pushed_cone_angle = nerf.cone_angle_constant;
		}
		if (m_render_ground_truth) { // the training image of nerf.training.view over the frame at alpha 1 (src/testbed.cu:4979-4994)
			check(ngp_render_ground_truth(m_ctx, nerf.training.view, width, height, m_background_color.data(), m_exposure, o.color_space, linear ? 0 : 1, m_fov_axis, m_zoom, out));
			return;
		}
		ngp_geometry_opts g{};
		memcpy(g.sun_dir, m_sun_dir.data(), 12);
		memcpy(g.up_dir, m_up_dir.data(), 12);
		g.metallic = brdf.metallic; g.subsurface = brdf.subsurface; g.specular = brdf.specular; g.roughness = brdf.roughness;
		g.sheen = brdf.sheen; g.clearcoat = brdf.clearcoat; g.clearcoat_gloss = brdf.clearcoat_gloss;
		memcpy(g.basecolor, brdf.basecolor.data(), 12);
		memcpy(g.ambientcolor, brdf.ambientcolor.data(), 12);
		ngp_set_geometry_opts(m_ctx, &g);
		check(ngp_render(m_ctx, &cam, &o, out, depth_out));
	}
	// irradiance probe pre-pass: what src/main.cu:185-188 calls before the render loop in ShadeEnvMap mode
	void computeEnvmapMultipleMain(uint32_t n_theta = 256, uint32_t n_phi = 128, uint32_t n_origin = 1) {
		ngp_probe_desc d{};
		d.mode = n_origin > 1 ? NGP_PROBE_MULTI_CENTER : NGP_PROBE_CENTER;
		d.n_theta = n_theta; d.n_phi = n_phi; d.n_origin = n_origin;
		d.min_transmittance = nerf.render_min_transmittance;
		check(ngp_compute_envmap(m_ctx, &d, nullptr));
		m_envmap_ready = true;
		m_envmap_grid_ready = false;
	}
	// Testbed::computeEnvmapGrid (testbed.h:743; src/main.cu:187-188 in ShadeGridEnvMap mode): gridSize probes on a shell around the NeRF
	void computeEnvmapGrid(uint32_t grid_x = 8, uint32_t grid_y = 8, uint32_t n_theta = 64, uint32_t n_phi = 32, float shell_radius = 1.0f) {
		ngp_probe_grid_desc d{};
		d.grid_x = grid_x; d.grid_y = grid_y; d.n_theta = n_theta; d.n_phi = n_phi;
		d.shell_radius = shell_radius;
		d.min_transmittance = nerf.render_min_transmittance;
		check(ngp_compute_envmap_grid(m_ctx, &d, nullptr));
		m_envmap_grid_ready = true;
		m_envmap_ready = false;
	}
	bool m_envmap_ready = false, m_envmap_grid_ready = false;
	std::array<float, 12> m_camera_end{};                       // camera_matrix1 of the frame being rendered along a path
	bool m_has_camera_end = false;
	std::array<float, 4> m_rolling_shutter{0.f, 0.f, 0.f, 1.f}; // what Testbed::render_to_cpu passes (src/python_api.cu:183)

	struct BRDFParams { // common.h:167-177
		float metallic = 0.f, subsurface = 0.f, specular = 1.f, roughness = 0.5f, sheen = 0.f, clearcoat = 0.f, clearcoat_gloss = 0.f;
		std::array<float, 3> basecolor{0.8f, 0.8f, 0.8f}, ambientcolor{0.f, 0.f, 0.f};
	} brdf;

	ngp_ctx* ctx() { return m_ctx; }

	// public state, names as in testbed.h
	ETestbedMode m_testbed_mode;
	std::array<float, 12> m_camera{};
	std::array<float, 2> m_relative_focal_length{1.f, 1.f};
	uint32_t m_fov_axis = 1;
	float m_zoom = 1.f;
	std::array<float, 2> m_screen_center{0.5f, 0.5f};
	float m_scale = 1.0f;
	std::array<float, 4> m_background_color{0.f, 0.f, 0.f, 1.f};
	std::array<float, 3> m_up_dir{0.f, 1.f, 0.f};
	std::array<float, 3> m_sun_dir{0.57735026f, 0.57735026f, 0.57735026f};
	float m_exposure = 0.f;
	// testbed.h:880. In Nerf mode the fork's default renders like Shade WITHOUT the sRGB -> linear step of shade_kernel_nerf
	// (src/testbed_nerf.cu:1392-1395 tests for Shade / Slice only); every BASELINE run pins Shade.
	ERenderMode m_render_mode = ERenderMode::ShadeGridEnvMap;
	EColorSpace m_color_space = EColorSpace::Linear; // testbed.color_space (run.py:160)
	float m_aperture_size = 0.f, m_slice_plane_z = 0.f; // testbed.aperture_size, testbed.slice_plane_z (focus = slice_plane_z + scale)
	int32_t m_render_lens_mode = 0;                  // m_nerf.render_lens
	std::array<float, 7> m_render_lens_params{};
	bool m_render_ground_truth = false;
	void set_render_aabb(const std::array<float, 6>& bb) { // testbed.render_aabb = ...
		check(ngp_set_render_aabb(m_ctx, bb.data(), bb.data() + 3, nullptr));
		m_render_aabb = bb;
	}
	float m_pushed_cone_angle = 0.f; // what the context holds; testbed.nerf.cone_angle_constant is pushed at the next render
	bool m_snap_to_pixel_centers = false;
	float m_render_near_distance = 0.f;
	bool m_train = false;
	bool m_train_encoding = true, m_train_network = true;
	bool m_training_data_available = false;
	uint32_t m_training_batch_size = 1u << 18;
	uint32_t m_seed = 1337;
	uint32_t m_log2_hashmap_size = 19;
	std::string m_data_path, m_root_dir;
	std::array<float, 6> m_aabb{0, 0, 0, 1, 1, 1}, m_render_aabb{0, 0, 0, 1, 1, 1};
	uint32_t m_training_step = 0;
	float m_loss = 0.f;

private:
	void check(int rc) {
		if (rc != 0) throw std::runtime_error(ngp_last_error(m_ctx));
	}
	void sync_dataset() {
		auto& ds = nerf.training.dataset;
		int n = ngp_n_training_views(m_ctx);
		ds.n_images = n > 0 ? (size_t)n : 0;
		ds.metadata.resize(ds.n_images);
		ds.xforms.resize(ds.n_images);
		for (size_t i = 0; i < ds.n_images; ++i) {
			int32_t res[2];
			ngp_get_training_view(m_ctx, (int)i, ds.xforms[i].data(), res, ds.metadata[i].focal_length.data(), ds.metadata[i].principal_point.data());
			ngp_get_training_view_lens(m_ctx, (int)i, &ds.metadata[i].lens_mode, ds.metadata[i].lens_params.data());
			ds.metadata[i].resolution = {res[0], res[1]};
		}
		int32_t aabb_scale = 1, is_hdr = 0;
		ngp_get_dataset_info(m_ctx, &aabb_scale, &ds.scale, ds.offset.data(), &is_hdr);
		ds.aabb_scale = aabb_scale;
	}
	void sync_model_state() {
		ngp_model_desc d;
		if (ngp_get_model(m_ctx, &d) == 0) {
			memcpy(m_aabb.data(), d.aabb_min, 12); memcpy(m_aabb.data() + 3, d.aabb_max, 12);
			memcpy(m_render_aabb.data(), d.render_aabb_min, 12); memcpy(m_render_aabb.data() + 3, d.render_aabb_max, 12);
			nerf.cone_angle_constant = m_pushed_cone_angle = d.cone_angle_constant;
		}
		sync_dataset();
	}
	ngp_ctx* m_ctx = nullptr;
};

} // namespace ngp
