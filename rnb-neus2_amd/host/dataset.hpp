// dataset.hpp — scene loader of the testbed host: transform*.json + 16-bit PNG normal/albedo maps -> rnb_view[] and RGBA16 pixels.
// Follows src/nerf_loader.cu:225-764 and include/neural-graphics-primitives/nerf_loader.h:180-201 (nerf_matrix_to_ngp).
#pragma once
#include "../../include/rnb_neus2.h"
#include "json_min.hpp"
#include "png16.hpp"

#include <sys/stat.h>
#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace hostio {

inline bool path_exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
inline bool is_dir(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
inline void make_dir(const std::string& p) { if (!path_exists(p)) { std::printf("create_directory:%s\n", p.c_str()); ::mkdir(p.c_str(), 0775); } }
inline std::string parent_path(const std::string& p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? "." : (k == 0 ? "/" : p.substr(0, k)); }
inline std::string lower(std::string s) { for (auto& c : s) c = (char)std::tolower(c); return s; }

// ---- dataset (src/nerf_loader.cu:225-764; include/neural-graphics-primitives/nerf_loader.h:180-201) ----
struct Dataset {
	std::vector<rnb_view> views;
	std::vector<png16::Image> normals, albedos;
	float scale = 0.33f;                       // NERF_SCALE, nerf_loader.h:31
	float offset[3] = {0.5f, 0.5f, 0.5f};
	float n2w_s = 1.f, n2w_t[3] = {0.f, 0.f, 0.f};
	int aabb_scale = 1;
	bool from_na = false, from_mitsuba = false;
	std::string json_path;
};

inline std::string find_transform_json(const std::string& scene) { // src/testbed_nerf.cu:3081-3118: first transform*.json alphabetically
	if (!is_dir(scene)) {
		if (lower(scene).size() > 5 && lower(scene).substr(scene.size() - 5) == ".json") return scene;
		throw std::runtime_error("NeRF data path must either be a json file or a directory containing json files.");
	}
	std::vector<std::string> found;
	if (DIR* d = ::opendir(scene.c_str())) {
		while (dirent* e = ::readdir(d)) {
			std::string n = e->d_name;
			if (n.size() > 5 && lower(n.substr(n.size() - 5)) == ".json" && n.find("transform") != std::string::npos) found.push_back(n);
		}
		::closedir(d);
	}
	if (found.empty()) throw std::runtime_error("No transform*.json found in " + scene);
	std::sort(found.begin(), found.end());
	for (const auto& f : found) std::printf("founded json file: %s/%s\n", scene.c_str(), f.c_str());
	std::printf("total frame: %d\n", (int)found.size());
	return scene + "/" + found.front();
}

inline Dataset load_dataset(const std::string& scene) {
	Dataset ds;
	ds.json_path = find_transform_json(scene);
	const jsonmin::Value j = jsonmin::parse_file(ds.json_path);
	if (!j.contains("frames") || !j["frames"].is_array() || j["frames"].size() == 0) throw std::invalid_argument("No training images were found for NeRF training!");
	const std::string base = parent_path(ds.json_path);
	if (j.contains("normal_mts_args")) ds.from_mitsuba = true;
	if (j.contains("from_na")) ds.from_na = true;                                   // presence, not value (nerf_loader.cu:392-394)
	if (ds.from_mitsuba) { ds.scale = 0.66f; for (float& o : ds.offset) o = 0.25f * ds.scale; }
	if (j.contains("scale")) ds.scale = j["scale"].as_float();
	if (j.contains("aabb_scale")) ds.aabb_scale = (int)j["aabb_scale"].as_number();
	if (j.contains("offset")) {
		const auto& o = j["offset"];
		for (int k = 0; k < 3; ++k) ds.offset[k] = o.is_array() ? o[(size_t)k].as_float() : o.as_float();
	}
	if (j.contains("aabb")) { // nerf_loader.cu:513-519
		const auto& ab = j["aabb"];
		float len = 0.000001f;
		for (size_t k = 0; k < 3; ++k) len = std::max(len, std::fabs(ab[(size_t)1][k].as_float() - ab[(size_t)0][k].as_float()));
		ds.scale = 1.f / len;
		for (size_t k = 0; k < 3; ++k) ds.offset[k] = ((ab[(size_t)1][k].as_float() + ab[(size_t)0][k].as_float()) * 0.5f) * -ds.scale + 0.5f;
	}
	if (j.contains("n2w")) { // nerf_loader.cu:574-579
		for (size_t m = 0; m < 3; ++m) ds.n2w_t[m] = j["n2w"][m][(size_t)3].as_float();
		ds.n2w_s = j["n2w"][(size_t)0][(size_t)0].as_float();
	}
	const float w = j.value("w", 0.f), h = j.value("h", 0.f);
	const auto& frames = j["frames"];
	size_t n = frames.size();
	if (j.contains("n_frames")) n = std::min(n, (size_t)j["n_frames"].as_number());
	ds.views.resize(n); ds.normals.resize(n); ds.albedos.resize(n);
	for (size_t i = 0; i < n; ++i) {
		const auto& fr = frames[i];
		const auto& M = fr.contains("transform_matrix_start") ? fr["transform_matrix_start"] : fr["transform_matrix"];
		float x[3][4];
		for (size_t r = 0; r < 3; ++r) for (size_t c = 0; c < 4; ++c) x[r][c] = M[r][c].as_float();
		// nerf_matrix_to_ngp (nerf_loader.h:180-201)
		for (int r = 0; r < 3; ++r) { x[r][1] *= -1; x[r][2] *= -1; x[r][3] = x[r][3] * ds.scale + ds.offset[r]; }
		if (ds.from_na) { for (int r = 0; r < 3; ++r) { x[r][1] *= -1; x[r][2] *= -1; } }
		else if (ds.from_mitsuba) { for (int r = 0; r < 3; ++r) { x[r][0] *= -1; x[r][2] *= -1; } }
		else { float t[4]; std::memcpy(t, x[0], 16); std::memcpy(x[0], x[1], 16); std::memcpy(x[1], x[2], 16); std::memcpy(x[2], t, 16); } // cycle axes xyz <- yzx
		rnb_view& v = ds.views[i];
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) v.xform[r * 4 + c] = x[r][c];
		auto load_image = [&](const char* key) {
			std::string p = fr[key].as_string();
			std::replace(p.begin(), p.end(), '\\', '/');
			std::string full = base + "/" + p;
			const size_t slash = full.find_last_of('/'), dot = full.find_last_of('.');
			if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) full += ".png";
			return png16::load(full);
		};
		ds.normals[i] = load_image("normal_path");
		ds.albedos[i] = load_image("albedo_path");
		if (ds.albedos[i].width != ds.normals[i].width || ds.albedos[i].height != ds.normals[i].height) throw std::runtime_error("normal / albedo image sizes differ in frame " + std::to_string(i));
		v.width = ds.normals[i].width; v.height = ds.normals[i].height;
		const auto& K = fr["intrinsic_matrix"]; // nerf_loader.cu:679-689
		v.focal_length[0] = K[(size_t)0][(size_t)0].as_float();
		v.focal_length[1] = K[(size_t)1][(size_t)1].as_float();
		v.principal_point[0] = K[(size_t)0][(size_t)2].as_float() / w;
		v.principal_point[1] = K[(size_t)1][(size_t)2].as_float() / h;
	}
	return ds;
}

} // namespace hostio
