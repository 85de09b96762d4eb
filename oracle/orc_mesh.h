// orc_mesh.h -- the checker's own statement of the reference's marching cubes (src/marching_cubes.cu: gen_vertices 276-327, gen_faces 329-399,
// marching_cubes_gpu 794-822). TEST INFRASTRUCTURE ONLY; shares no source with the product's host / device mesh code (round 4: it used to call
// rnb-neus2_amd/host/mesh.hpp). The reference numbers vertices and triangles with atomic counters (no particular order); here -- as in the product,
// whose buffers the tests compare with these bit for bit -- they are numbered by prefix sums in lattice order: vertex slots (lattice point, axis) with
// the axis fastest, triangles cell by cell in the table's order.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "orc_mc_table.h"

namespace orc_mesh {

// edge e of a cell: the lattice point it starts at (offset from the cell's corner 0) and the axis it runs along. Corners (0,0,0) (1,0,0) (1,1,0) (0,1,0)
// (0,0,1) (1,0,1) (1,1,1) (0,1,1); edges 0-3 bottom face 0-1 1-2 2-3 3-0, 4-7 top face 4-5 5-6 6-7 7-4, 8-11 verticals (marching_cubes.cu:684-705)
static const int EDGE_AT[12][4] = {{0, 0, 0, 0}, {1, 0, 0, 1}, {0, 1, 0, 0}, {0, 0, 0, 1}, {0, 0, 1, 0}, {1, 0, 1, 1}, {0, 1, 1, 0}, {0, 0, 1, 1}, {0, 0, 0, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {0, 1, 0, 2}};
static const int CORNER_AT[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

// returns false if a triangle refers to an edge without a vertex (cannot happen on a consistent lattice)
inline bool marching_cubes(const float* density, const uint32_t res[3], const float aabb_min[3], const float aabb_max[3], const float thresh,
                           std::vector<float>& verts, std::vector<uint32_t>& indices) {
	const int64_t rx = res[0], ry = res[1], rz = res[2], n = rx * ry * rz;
	const int64_t step[3] = {1, rx, rx * ry}, lim[3] = {rx - 1, ry - 1, rz - 1};
	const float scale[3] = {(aabb_max[0] - aabb_min[0]) / rx, (aabb_max[1] - aabb_min[1]) / ry, (aabb_max[2] - aabb_min[2]) / rz};
	// pass 1 (gen_vertices): which (lattice point, axis) slots carry a vertex -- the two ends of the edge on different sides of the threshold
	std::vector<uint32_t> slot((size_t)n * 3 + 1, 0);
	for (int64_t i = 0; i < n; ++i) {
		const int64_t p[3] = {i % rx, (i / rx) % ry, i / (rx * ry)};
		for (int a = 0; a < 3; ++a)
			if (p[a] < lim[a] && (density[i] > thresh) != (density[i + step[a]] > thresh)) slot[(size_t)i * 3 + a] = 1;
	}
	uint32_t run = 0;
	for (size_t k = 0; k < slot.size(); ++k) { const uint32_t f = slot[k]; slot[k] = f ? run : 0xffffffffu; run += f; }
	verts.assign((size_t)run * 3, 0.f);
	for (int64_t i = 0; i < n; ++i)
		for (int a = 0; a < 3; ++a) {
			const uint32_t v = slot[(size_t)i * 3 + a];
			if (v == 0xffffffffu) continue;
			const float f0 = density[i], f1 = density[i + step[a]];
			float q[3] = {(float)(i % rx), (float)((i / rx) % ry), (float)(i / (rx * ry))};
			q[a] += (thresh - f0) / (f1 - f0); // marching_cubes.cu:300-321: linear interpolation along the edge
			for (int d = 0; d < 3; ++d) verts[(size_t)v * 3 + d] = q[d] * scale[d] + aabb_min[d];
		}
	// pass 2 (gen_faces): cells in lattice order, triangles in the table's order
	indices.clear();
	for (int64_t z = 0; z + 1 < rz; ++z)
		for (int64_t y = 0; y + 1 < ry; ++y)
			for (int64_t x = 0; x + 1 < rx; ++x) {
				const int64_t i = x + y * rx + z * rx * ry;
				int mask = 0;
				for (int c = 0; c < 8; ++c)
					if (density[i + CORNER_AT[c][0] + CORNER_AT[c][1] * step[1] + CORNER_AT[c][2] * step[2]] > thresh) mask |= 1 << c;
				for (const signed char* t = ORC_MC_TRIANGLES[mask]; *t >= 0; ++t) {
					const int* e = EDGE_AT[(int)*t];
					const uint32_t v = slot[(size_t)(i + e[0] + e[1] * step[1] + e[2] * step[2]) * 3 + e[3]];
					if (v == 0xffffffffu) return false;
					indices.push_back(v);
				}
			}
	return true;
}

} // namespace orc_mesh
