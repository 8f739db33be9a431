// Headless command line of the MI355X renderer: the flags of the reference's src/main.cu:29-199 (--scene, --snapshot,
// --width, --height, --no-gui, --no-train, --version, positional files) on the C-ABI-backed ngp::Testbed, plus what a
// run without a window needs to leave a result behind: --screenshot (one PNG from the snapshot's / default camera) and
// --screenshot_transforms / --screenshot_dir (one PNG per frame of a transforms.json, as scripts/run.py:276-299).
// There is no training and no GUI in this build: --no-gui / --no-train are accepted and implied.
#include "minijson.h"
#include "testbed_shim.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace {

void put_be32(std::string& s, uint32_t v) {
	for (int i = 3; i >= 0; --i) s.push_back((char)((v >> (8 * i)) & 0xff));
}
void png_chunk(std::string& out, const char* type, const std::string& data) {
	put_be32(out, (uint32_t)data.size());
	std::string body(type, 4);
	body += data;
	out += body;
	put_be32(out, (uint32_t)crc32(0L, (const Bytef*)body.data(), (uInt)body.size()));
}
// 8-bit RGBA PNG (filter 0 on every row); the input is linear premultiplied RGBA like Testbed::render returns with linear=true
void write_png(const std::string& path, const std::vector<float>& rgba, int w, int h, float exposure) {
	auto to_srgb = [](float l) { return l < 0.0031308f ? 12.92f * l : 1.055f * std::pow(l, 0.41666f) - 0.055f; };
	std::string raw;
	raw.reserve((size_t)h * ((size_t)w * 4 + 1));
	const float scale = std::pow(2.0f, exposure);
	for (int y = 0; y < h; ++y) {
		raw.push_back(0);
		for (int x = 0; x < w; ++x) {
			const float* p = &rgba[((size_t)y * w + x) * 4];
			const float a = std::min(std::max(p[3], 0.0f), 1.0f);
			for (int c = 0; c < 3; ++c) { // un-premultiply, sRGB-encode (scripts/common.py write_image)
				float v = a > 0.f ? p[c] / a : 0.f;
				v = to_srgb(std::min(std::max(v * scale, 0.0f), 1.0f));
				raw.push_back((char)(unsigned char)std::lround(std::min(std::max(v, 0.0f), 1.0f) * 255.0f));
			}
			raw.push_back((char)(unsigned char)std::lround(a * 255.0f));
		}
	}
	uLongf n = compressBound((uLong)raw.size());
	std::string z(n, '\0');
	if (compress2((Bytef*)z.data(), &n, (const Bytef*)raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("png: deflate failed");
	z.resize(n);
	std::string out("\x89PNG\r\n\x1a\n", 8), ihdr;
	put_be32(ihdr, (uint32_t)w);
	put_be32(ihdr, (uint32_t)h);
	ihdr += std::string("\x08\x06\x00\x00\x00", 5);
	png_chunk(out, "IHDR", ihdr);
	png_chunk(out, "IDAT", z);
	png_chunk(out, "IEND", "");
	std::ofstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("cannot write " + path);
	f.write(out.data(), (std::streamsize)out.size());
}

std::string read_text(const std::string& path) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("cannot read " + path);
	std::stringstream ss;
	ss << f.rdbuf();
	return ss.str();
}

void usage() {
	std::cout << "ngp_hip_main [files...] [--scene PATH] [--snapshot|--load_snapshot PATH] [--width W] [--height H] [--spp N]\n"
	             "             [--screenshot OUT.png] [--screenshot_transforms T.json --screenshot_dir DIR] [--render_mode Shade|ShadeEnvMap|ShadeGridEnvMap|AO|Positions|Depth]\n"
	             "             [--exposure E] [--n_steps N] [--save_snapshot OUT.ingp] [--network CONFIG.json] [--no-gui] [--no-train] [--version]\n"
	             "             [--video_camera_path PATH.json --video_output DIR/%04d.png [--video_n_seconds S] [--video_fps F] [--video_spp N]]\n";
}

} // namespace

int main(int argc, char** argv) {
	try {
		std::vector<std::string> files;
		std::string scene, snapshot, screenshot, shot_transforms, shot_dir, render_mode = "Shade", save_snapshot, network, video_path, video_output = "video_%04d.png";
		int width = 1920, height = 1080, spp = 1, n_steps = -1, video_seconds = 1, video_fps = 60, video_spp = 8;
		bool no_train = false;
		float exposure = 0.f;
		for (int i = 1; i < argc; ++i) {
			std::string a = argv[i];
			auto val = [&]() -> std::string {
				if (i + 1 >= argc) throw std::runtime_error("missing value after " + a);
				return argv[++i];
			};
			if (a == "-h" || a == "--help") { usage(); return 0; }
			else if (a == "-v" || a == "--version") { std::cout << ngp_version() << "\n"; return 0; }
			else if (a == "--scene" || a == "--training_data" || a == "-s") scene = val();
			else if (a == "--snapshot" || a == "--load_snapshot") snapshot = val();
			else if (a == "--width") width = std::atoi(val().c_str());
			else if (a == "--height") height = std::atoi(val().c_str());
			else if (a == "--spp" || a == "--screenshot_spp") spp = std::atoi(val().c_str());
			else if (a == "--screenshot") screenshot = val();
			else if (a == "--screenshot_transforms") shot_transforms = val();
			else if (a == "--screenshot_dir") shot_dir = val();
			else if (a == "--render_mode") render_mode = val();
			else if (a == "--exposure") exposure = (float)std::atof(val().c_str());
			else if (a == "--n_steps") n_steps = std::atoi(val().c_str()); // scripts/run.py:66
			else if (a == "--save_snapshot") save_snapshot = val();       // scripts/run.py:37
			else if (a == "--network" || a == "--config" || a == "-n" || a == "-c") network = val(); // src/main.cu:96-101
			else if (a == "--video_camera_path") video_path = val();   // scripts/run.py:46-54, 304-337 (frames only: no ffmpeg here)
			else if (a == "--video_output") video_output = val();
			else if (a == "--video_n_seconds") video_seconds = std::atoi(val().c_str());
			else if (a == "--video_fps") video_fps = std::atoi(val().c_str());
			else if (a == "--video_spp") video_spp = std::atoi(val().c_str());
			else if (a == "--no-train") no_train = true;
			else if (a == "--no-gui" || a == "--vr") { /* headless build */ }
			else if (a == "--mode" || a == "-m") { (void)val(); std::cerr << "warning: " << a << " has no effect in this build\n"; }
			else if (!a.empty() && a[0] == '-') throw std::runtime_error("unknown flag " + a);
			else files.push_back(a);
		}
		ngp::Testbed testbed;
		for (auto& f : files) {
			std::cerr << "Loading file " << f << "\n";
			testbed.load_file(f);
		}
		if (!scene.empty()) {
			std::cerr << "Loading training data " << scene << "\n";
			testbed.load_training_data(scene);
		}
		if (!snapshot.empty()) testbed.load_snapshot(snapshot);
		// training (scripts/run.py:172-208): n_steps < 0 with a scene and no snapshot trains run.py's default of 35000 steps
		if (!network.empty()) testbed.reload_network_from_file(network);
		if (n_steps < 0 && !scene.empty() && snapshot.empty() && !no_train && testbed.m_training_data_available && (!save_snapshot.empty() || !screenshot.empty() || !shot_transforms.empty())) n_steps = 35000;
		if (n_steps > 0 && !no_train) {
			if (!testbed.m_training_data_available) throw std::runtime_error("No training data available (the dataset's images must be PNG or baseline JPEG files).");
			testbed.m_train = true;
			while ((int)testbed.m_training_step < n_steps && testbed.frame()) {
				if (testbed.m_training_step % 1000 == 0 || (int)testbed.m_training_step == n_steps) std::cerr << "step " << testbed.m_training_step << " loss " << testbed.m_loss << "\n";
				if (!testbed.m_train) break;
			}
			testbed.m_train = false;
		}
		if (!save_snapshot.empty()) {
			testbed.save_snapshot(save_snapshot, false);
			std::cerr << "wrote " << save_snapshot << "\n";
		}
		if (render_mode == "Shade") testbed.m_render_mode = ngp::ERenderMode::Shade;
		else if (render_mode == "ShadeGridEnvMap") testbed.m_render_mode = ngp::ERenderMode::ShadeGridEnvMap;
		else if (render_mode == "ShadeEnvMap") testbed.m_render_mode = ngp::ERenderMode::ShadeEnvMap;
		else if (render_mode == "AO") testbed.m_render_mode = ngp::ERenderMode::AO;
		else if (render_mode == "Normals") testbed.m_render_mode = ngp::ERenderMode::Normals;
		else if (render_mode == "Positions") testbed.m_render_mode = ngp::ERenderMode::Positions;
		else if (render_mode == "Depth") testbed.m_render_mode = ngp::ERenderMode::Depth;
		else throw std::runtime_error("unknown render mode " + render_mode);
		// pre computation of the envmap, src/main.cu:184-188
		if (testbed.m_render_mode == ngp::ERenderMode::ShadeEnvMap) testbed.computeEnvmapMultipleMain();
		else if (testbed.m_render_mode == ngp::ERenderMode::ShadeGridEnvMap && testbed.m_testbed_mode == ngp::ETestbedMode::Geometry) testbed.computeEnvmapGrid();
		testbed.m_background_color = {0.f, 0.f, 0.f, 0.f};
		std::vector<float> img((size_t)width * height * 4);
		if (!video_path.empty()) { // scripts/run.py:304-337: one PNG per frame along the camera path
			if (video_output.find('%') == std::string::npos) throw std::runtime_error("--video_output needs a printf pattern such as frames/%04d.png (frames are not encoded into a video here)");
			testbed.load_camera_path(video_path);
			const int n_frames = video_seconds * video_fps;
			for (int i = 0; i < n_frames; ++i) {
				testbed.render_to_cpu(img.data(), width, height, video_spp, true, (float)i / (float)n_frames, (float)(i + 1) / (float)n_frames, (float)video_fps, 0.5f);
				char name[1024];
				snprintf(name, sizeof(name), video_output.c_str(), i);
				write_png(name, img, width, height, exposure);
			}
			std::cerr << "wrote " << n_frames << " frames\n";
		}
		if (!shot_transforms.empty()) { // scripts/run.py:276-299
			mj::Value t = mj::parse_json(read_text(shot_transforms));
			testbed.m_fov_axis = 0;
			testbed.set_fov((float)(t.at("camera_angle_x").num() * 180.0 / 3.14159265358979323846));
			const mj::Value& frames = t.at("frames");
			for (size_t k = 0; k < frames.size(); ++k) {
				const mj::Value& fr = frames.at(k);
				const mj::Value& m = fr.contains("transform_matrix") ? fr.at("transform_matrix") : fr.at("transform_matrix_start");
				std::array<float, 12> cam;
				for (int r = 0; r < 3; ++r)
					for (int c = 0; c < 4; ++c) cam[(size_t)c * 3 + r] = (float)m.at((size_t)r).at((size_t)c).num();
				testbed.set_nerf_camera_matrix(cam);
				std::string name = fr.at("file_path").str();
				size_t slash = name.find_last_of('/');
				if (slash != std::string::npos) name = name.substr(slash + 1);
				if (name.find('.') == std::string::npos) name += ".png";
				const std::string out = (shot_dir.empty() ? std::string(".") : shot_dir) + "/" + name;
				std::cerr << "rendering " << out << "\n";
				testbed.render_to_cpu(img.data(), width, height, spp, true);
				write_png(out, img, width, height, exposure);
			}
		} else if (!screenshot.empty()) {
			testbed.render_to_cpu(img.data(), width, height, spp, true);
			write_png(screenshot, img, width, height, exposure);
			std::cerr << "wrote " << screenshot << "\n";
		} else {
			if (save_snapshot.empty() && video_path.empty()) std::cerr << "nothing to do: give --screenshot, --screenshot_transforms or --n_steps with --save_snapshot (this build has no window)\n";
		}
		return 0;
	} catch (const std::exception& e) {
		std::cerr << "error: " << e.what() << "\n";
		return 1;
	}
}
