// pybind11 module `pyngp`: the subset of the reference's src/python_api.cu that scripts/run.py's training, evaluation,
// screenshot and camera-path rendering paths use (Testbed, TestbedMode, RenderMode, LossType; load_*, train / frame,
// render, camera and render-state properties), bound to the C-ABI-backed ngp::Testbed shim. Same names and defaults
// as python_api.cu:263-733.
#include "testbed_shim.h"

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

namespace py = pybind11;
using namespace ngp;

static std::array<float, 12> to_colmajor(const py::array_t<float, py::array::c_style | py::array::forcecast>& a) {
	if (a.ndim() != 2 || a.shape(0) != 3 || a.shape(1) != 4) throw std::runtime_error("expected a 3x4 matrix");
	std::array<float, 12> m;
	auto r = a.unchecked<2>();
	for (int c = 0; c < 4; ++c)
		for (int row = 0; row < 3; ++row) m[c * 3 + row] = r(row, c);
	return m;
}
static py::array_t<float> from_colmajor(const std::array<float, 12>& m) {
	py::array_t<float> a({3, 4});
	auto w = a.mutable_unchecked<2>();
	for (int c = 0; c < 4; ++c)
		for (int row = 0; row < 3; ++row) w(row, c) = m[c * 3 + row];
	return a;
}

PYBIND11_MODULE(pyngp, m) {
	m.doc() = "MI355X-native NeRF renderer and trainer behind the instant-ngp Testbed API";
	py::enum_<ETestbedMode>(m, "TestbedMode")
		.value("Nerf", ETestbedMode::Nerf).value("Sdf", ETestbedMode::Sdf).value("Image", ETestbedMode::Image)
		.value("Volume", ETestbedMode::Volume).value("Geometry", ETestbedMode::Geometry).value("None", ETestbedMode::None)
		.export_values();
	// the reference registers the name "Shade" twice (python_api.cu:284-294), which pybind11 rejects at import;
	// ShadeNerf keeps its own name here
	py::enum_<EColorSpace>(m, "ColorSpace").value("Linear", EColorSpace::Linear).value("SRGB", EColorSpace::SRGB).value("VisPosNeg", EColorSpace::VisPosNeg).export_values();
	py::enum_<ELossType>(m, "LossType")
		.value("L2", ELossType::L2).value("L1", ELossType::L1).value("Mape", ELossType::Mape).value("Smape", ELossType::Smape)
		.value("Huber", ELossType::Huber).value("LogL1", ELossType::LogL1).value("RelativeL2", ELossType::RelativeL2)
		.export_values();
	py::enum_<ERenderMode>(m, "RenderMode")
		.value("AO", ERenderMode::AO).value("Shade", ERenderMode::Shade).value("Normals", ERenderMode::Normals)
		.value("Positions", ERenderMode::Positions).value("Depth", ERenderMode::Depth).value("Distortion", ERenderMode::Distortion)
		.value("Cost", ERenderMode::Cost).value("Slice", ERenderMode::Slice).value("ShadeNerf", ERenderMode::ShadeNerf)
		.value("ShadeEnvMap", ERenderMode::ShadeEnvMap).value("ShadeGridEnvMap", ERenderMode::ShadeGridEnvMap)
		.export_values();

	// nested under Testbed like python_api.cu does (py::class_<Testbed::Nerf> nerf(testbed, "Nerf")): the exported enum
	// value TestbedMode.Nerf already owns the module-level name "Nerf"
	py::class_<Testbed> testbed(m, "Testbed");
	py::class_<Testbed::TrainingImageMetadata>(testbed, "TrainingImageMetadata")
		.def_readonly("resolution", &Testbed::TrainingImageMetadata::resolution)
		.def_readonly("focal_length", &Testbed::TrainingImageMetadata::focal_length)
		.def_readonly("principal_point", &Testbed::TrainingImageMetadata::principal_point);
	py::class_<Testbed::NerfDatasetView>(testbed, "NerfDataset")
		.def_readonly("n_images", &Testbed::NerfDatasetView::n_images)
		.def_readonly("metadata", &Testbed::NerfDatasetView::metadata)
		.def_readonly("aabb_scale", &Testbed::NerfDatasetView::aabb_scale)
		.def_readonly("scale", &Testbed::NerfDatasetView::scale)
		.def_readonly("offset", &Testbed::NerfDatasetView::offset);
	py::class_<Testbed::Nerf::Training>(testbed, "NerfTraining")
		.def_readonly("dataset", &Testbed::Nerf::Training::dataset)
		.def_readwrite("view", &Testbed::Nerf::Training::view)
		.def_readwrite("random_bg_color", &Testbed::Nerf::Training::random_bg_color)
		.def_readwrite("linear_colors", &Testbed::Nerf::Training::linear_colors)
		.def_readwrite("loss_type", &Testbed::Nerf::Training::loss_type)
		.def_readwrite("snap_to_pixel_centers", &Testbed::Nerf::Training::snap_to_pixel_centers)
		.def_readwrite("density_grid_decay", &Testbed::Nerf::Training::density_grid_decay)
		.def_readonly("n_images_for_training", &Testbed::Nerf::Training::n_images_for_training)
		.def_readwrite("near_distance", &Testbed::Nerf::Training::near_distance);
	py::class_<Testbed::Nerf>(testbed, "Nerf")
		.def_readwrite("render_min_transmittance", &Testbed::Nerf::render_min_transmittance)
		.def_readwrite("cone_angle_constant", &Testbed::Nerf::cone_angle_constant)
		.def_readwrite("sharpen", &Testbed::Nerf::sharpen)
		.def_readwrite("render_with_lens_distortion", &Testbed::Nerf::render_with_lens_distortion)
		.def_readonly("training", &Testbed::Nerf::training);
	py::class_<Testbed::BRDFParams>(testbed, "BRDFParams")
		.def_readwrite("metallic", &Testbed::BRDFParams::metallic).def_readwrite("subsurface", &Testbed::BRDFParams::subsurface)
		.def_readwrite("specular", &Testbed::BRDFParams::specular).def_readwrite("roughness", &Testbed::BRDFParams::roughness)
		.def_readwrite("sheen", &Testbed::BRDFParams::sheen).def_readwrite("clearcoat", &Testbed::BRDFParams::clearcoat)
		.def_readwrite("clearcoat_gloss", &Testbed::BRDFParams::clearcoat_gloss)
		.def_readwrite("basecolor", &Testbed::BRDFParams::basecolor).def_readwrite("ambientcolor", &Testbed::BRDFParams::ambientcolor);

	testbed
		.def(py::init<ETestbedMode, int>(), py::arg("mode") = ETestbedMode::None, py::arg("device") = 0)
		.def(py::init<ETestbedMode, const std::string&, int>(), py::arg("mode"), py::arg("data_path"), py::arg("device") = 0)
		.def(py::init<ETestbedMode, const std::vector<int>&>(), py::arg("mode"), py::arg("devices"), "Several GPUs behind one Testbed: the camera's tiles are dealt to all of them (devices[0] assembles the frame)")
		.def_property_readonly("n_devices", &Testbed::n_devices)
		.def("load_training_data", &Testbed::load_training_data, py::call_guard<py::gil_scoped_release>(), "Load training data from a given path.")
		.def("load_snapshot", &Testbed::load_snapshot, py::arg("path"), "Load a previously saved snapshot")
		.def("save_snapshot", &Testbed::save_snapshot, py::arg("path"), py::arg("include_optimizer_state") = false, py::arg("compress") = true)
		.def("load_file", &Testbed::load_file, py::arg("path"))
		.def("load_mesh", &Testbed::load_mesh, py::arg("path"), py::arg("center") = std::array<float, 3>{0.f, 0.f, 0.f})
		.def("reset_camera", &Testbed::reset_camera)
		.def("set_nerf_camera_matrix", [](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& a) { t.set_nerf_camera_matrix(to_colmajor(a)); })
		.def("set_camera_to_training_view", &Testbed::set_camera_to_training_view)
		.def("compute_envmap", &Testbed::computeEnvmapMultipleMain, py::arg("n_theta") = 256, py::arg("n_phi") = 128, py::arg("n_origin") = 1)
		.def("compute_envmap_grid", &Testbed::computeEnvmapGrid, py::arg("grid_x") = 8, py::arg("grid_y") = 8, py::arg("n_theta") = 64, py::arg("n_phi") = 32, py::arg("shell_radius") = 1.0f)
		.def("render", [](Testbed& t, int width, int height, int spp, bool linear, float start_t, float end_t, float fps, float shutter_fraction) {
				// a fresh array per call, like python_api.cu:124-202 -- whose memory is page-locked and pooled (ngp_host_alloc), so the
				// device-to-host copy is a single DMA; the array owns its buffer and returns it to the pool when collected
				const size_t bytes = (size_t)height * width * 4 * sizeof(float);
				float* mem = (float*)ngp_host_alloc(bytes);
				if (!mem) throw std::runtime_error("out of host memory");
				py::capsule owner(mem, [](void* p) { ngp_host_free(p); });
				py::array_t<float> result({(py::ssize_t)height, (py::ssize_t)width, (py::ssize_t)4}, mem, owner);
				{
					py::gil_scoped_release release;
					t.render_to_cpu(result.mutable_data(), width, height, spp, linear, start_t, end_t, fps, shutter_fraction);
				}
				return result;
			}, "Renders an image at the requested resolution. Does not require a window.",
			py::arg("width") = 1920, py::arg("height") = 1080, py::arg("spp") = 1, py::arg("linear") = true, py::arg("start_t") = -1.f,
			py::arg("end_t") = -1.f, py::arg("fps") = 30.f, py::arg("shutter_fraction") = 1.0f)
		.def_property("camera_matrix", [](Testbed& t) { return from_colmajor(t.m_camera); },
			[](Testbed& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& a) { t.m_camera = to_colmajor(a); })
		.def_property("fov", &Testbed::fov, &Testbed::set_fov)
		.def_readwrite("fov_axis", &Testbed::m_fov_axis)
		.def_readwrite("relative_focal_length", &Testbed::m_relative_focal_length)
		.def_readwrite("screen_center", &Testbed::m_screen_center)
		.def_readwrite("zoom", &Testbed::m_zoom)
		.def_readwrite("scale", &Testbed::m_scale)
		.def_readwrite("background_color", &Testbed::m_background_color)
		.def_readwrite("snap_to_pixel_centers", &Testbed::m_snap_to_pixel_centers)
		.def_readwrite("exposure", &Testbed::m_exposure)
		.def_readwrite("render_mode", &Testbed::m_render_mode)
		.def_readwrite("color_space", &Testbed::m_color_space)
		.def_readwrite("aperture_size", &Testbed::m_aperture_size)
		.def_readwrite("slice_plane_z", &Testbed::m_slice_plane_z)
		.def_readwrite("render_ground_truth", &Testbed::m_render_ground_truth)
		.def_readwrite("render_near_distance", &Testbed::m_render_near_distance)
		.def_readwrite("sun_dir", &Testbed::m_sun_dir)
		.def_readwrite("up_dir", &Testbed::m_up_dir)
		.def_readwrite("root_dir", &Testbed::m_root_dir)
		.def_readonly("data_path", &Testbed::m_data_path)
		.def_readonly("aabb", &Testbed::m_aabb)
		.def_property("render_aabb", [](Testbed& t) { return t.m_render_aabb; }, &Testbed::set_render_aabb, "crop box of the render: [min xyz, max xyz] in ngp space")
		.def_readonly("mode", &Testbed::m_testbed_mode)
		.def_readonly("training_step", &Testbed::m_training_step)
		.def_readonly("loss", &Testbed::m_loss)
		.def_readwrite("brdf", &Testbed::brdf)
		.def_readwrite("shall_train", &Testbed::m_train)
		.def_readwrite("shall_train_encoding", &Testbed::m_train_encoding)
		.def_readwrite("shall_train_network", &Testbed::m_train_network)
		.def_readwrite("training_batch_size", &Testbed::m_training_batch_size)
		.def_readwrite("seed", &Testbed::m_seed)
		.def("want_repl", [](Testbed&) { return false; }, "scripts/run.py polls this inside its training loop (the GUI's console key); headless: never")
		.def("init_window", [](Testbed&, int, int, bool, bool) { throw std::runtime_error("this build is headless: render() / frame() work without a window"); },
			py::arg("width"), py::arg("height"), py::arg("hidden") = false, py::arg("second_window") = false)
		.def("init_vr", [](Testbed&) { throw std::runtime_error("this build is headless: no VR"); })
		.def("load_camera_path", &Testbed::load_camera_path, py::arg("path"), "Load a camera path")
		.def("set_camera_from_time", &Testbed::set_camera_from_time, py::arg("t"), "place the camera on the loaded path, t in [0, 1]")
		.def_readwrite("camera_smoothing", &Testbed::m_camera_smoothing)
		.def("compute_and_save_marching_cubes_mesh", [](Testbed&, py::args, py::kwargs) { throw std::runtime_error("marching cubes is outside the MI355X renderer's scope (SURVEY section 2)"); })
		.def("frame", &Testbed::frame, py::call_guard<py::gil_scoped_release>(), "Process a single frame: one training step when shall_train is set (headless, nothing is drawn).")
		.def("train", &Testbed::train, py::call_guard<py::gil_scoped_release>(), "Perform a single training step with a specified batch size.")
		.def("reset", &Testbed::reset_network, py::arg("reset_density_grid") = true, "Reset training.")
		.def("reload_network_from_file", &Testbed::reload_network_from_file, py::arg("path") = "", "Reload the network from a config file.")
		.def("set_training_image", [](Testbed& t, int frame_idx, py::array_t<float, py::array::c_style | py::array::forcecast> img) {
				py::buffer_info b = img.request();
				if (b.ndim != 3 || b.shape[2] != 4) throw std::runtime_error("image should be (H,W,C) where C=4");
				t.set_training_image(frame_idx, (int)b.shape[1], (int)b.shape[0], (const float*)b.ptr);
			}, py::arg("frame_idx"), py::arg("img"),
			"nerf.training.set_image of the reference (python_api.cu:691-697): a float (H,W,4) image, linear colour space, premultiplied alpha")
		.def_readonly("nerf", &Testbed::nerf);
}
