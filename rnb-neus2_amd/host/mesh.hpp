// mesh.hpp — host side of `--save-mesh` (src/marching_cubes.cu:794-982; src/testbed_nerf.cu:4218-4350): the case table in the form the device
// kernels take it (mc_table.hpp: the public Bourke / PyMCubes table the reference uses, src/marching_cubes.cu:401-659), area-weighted normals and
// the OBJ writer. The extraction itself runs on the device (rnb_marching_cubes, csrc/kernels_mesh.cuh): vertices on lattice edges where the
// reference's gen_vertices puts them, vertices and triangles numbered in lattice order (the reference numbers them with atomic counters, i.e. in
// no particular order). Until round 4 this file also held a host loop of the extraction; its only caller was the CPU checker, which has its own
// statement now (oracle/orc_mesh.h). Normals and face order follow save_mesh (src/marching_cubes.cu:354-356, 930-975).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "mc_table.hpp"

namespace mesh {

struct Vec3 { float x, y, z; };

// corner c -> (x, y, z) offsets; edges e -> (corner a, corner b). Numbering as in src/marching_cubes.cu:261-275.
static const int CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
static const int EDGE[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

struct Tables {
	// tri[mask] = list of edge ids, 3 per triangle, -1 terminated (at most 5 triangles; the array keeps the length the device copy uses)
	std::array<std::array<int8_t, 40>, 256> tri;
	Tables() {
		for (int mask = 0; mask < 256; ++mask) {
			int n = 0;
			for (const char* c = MC_TRIANGLES[mask]; *c; ++c) tri[mask][n++] = (int8_t)(*c <= '9' ? *c - '0' : *c - 'a' + 10);
			for (; n < 40; ++n) tri[mask][n] = -1;
		}
	}
};

struct Mesh {
	std::vector<Vec3> verts, normals, colors;
	std::vector<uint32_t> indices;
};

// area-weighted vertex normals exactly as accumulate_1ring forms them (marching_cubes.cu:354-356): n = (pb - pa) x (pa - pc), i.e. the
// NEGATIVE of the counter-clockwise normal of (a, b, c). With the table's winding and "corner bit = sdf > 0" that is the direction of
// decreasing SDF (into the object); the reference writes these to the OBJ as they are, and so does save_obj.
inline void compute_normals(Mesh& m) {
	m.normals.assign(m.verts.size(), {0, 0, 0});
	for (size_t i = 0; i + 2 < m.indices.size(); i += 3) {
		const uint32_t ia = m.indices[i], ib = m.indices[i + 1], ic = m.indices[i + 2];
		const Vec3 a = m.verts[ia], b = m.verts[ib], c = m.verts[ic];
		const Vec3 u = {b.x - a.x, b.y - a.y, b.z - a.z}, v = {a.x - c.x, a.y - c.y, a.z - c.z};
		const Vec3 n = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
		for (uint32_t q : {ia, ib, ic}) { m.normals[q].x += n.x; m.normals[q].y += n.y; m.normals[q].z += n.z; }
	}
}

// OBJ with per-vertex colours and normals (save_mesh, marching_cubes.cu:922-981): v = n2w_s * ((p - offset) / scale) + n2w_t.
// invert_normals = the dataset's from_na flag (src/testbed.cu:376-379): faces are written (a, b, c) when set -- counter-clockwise seen
// from the sdf > 0 side -- and (c, b, a) otherwise (marching_cubes.cu:966-975); the vn records are not touched by it.
inline void save_obj(const std::string& path, const Mesh& m, float nerf_scale, const float nerf_offset[3], float n2w_s, const float n2w_t[3], bool invert_normals = true) {
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) throw std::runtime_error("cannot write " + path);
	for (size_t i = 0; i < m.verts.size(); ++i) {
		const Vec3 v = m.verts[i];
		const float p[3] = {n2w_s * ((v.x - nerf_offset[0]) / nerf_scale) + n2w_t[0], n2w_s * ((v.y - nerf_offset[1]) / nerf_scale) + n2w_t[1], n2w_s * ((v.z - nerf_offset[2]) / nerf_scale) + n2w_t[2]};
		const Vec3 c = i < m.colors.size() ? m.colors[i] : Vec3{1.f, 1.f, 1.f};
		auto cl = [](float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); };
		std::fprintf(f, "v %0.5f %0.5f %0.5f %0.3f %0.3f %0.3f\n", p[0], p[1], p[2], cl(c.x), cl(c.y), cl(c.z));
	}
	for (const Vec3& n0 : m.normals) {
		Vec3 n = {n2w_s * n0.x, n2w_s * n0.y, n2w_s * n0.z};
		const float l = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
		if (l > 0.f) { n.x /= l; n.y /= l; n.z /= l; }
		std::fprintf(f, "vn %0.5f %0.5f %0.5f\n", n.x, n.y, n.z);
	}
	for (size_t i = 0; i + 2 < m.indices.size(); i += 3) {
		const uint32_t a = m.indices[invert_normals ? i : i + 2] + 1, b = m.indices[i + 1] + 1, c = m.indices[invert_normals ? i + 2 : i] + 1;
		std::fprintf(f, "f %u//%u %u//%u %u//%u\n", a, a, b, b, c, c);
	}
	std::fclose(f);
}

} // namespace mesh
