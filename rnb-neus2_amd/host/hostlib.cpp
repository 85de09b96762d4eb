// hostlib.cpp — librnb_host.so: PNG I/O and triangle-mesh ray casting behind the C-ABI of include/rnb_host.h.
#include "../../include/rnb_host.h"
#include "png16.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
template <typename F> int guarded(F&& f) {
	try { f(); return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown error"; return -1; }
}

struct Node { float lo[3], hi[3]; uint32_t left, count; }; // leaf: count > 0, left = first triangle slot; inner: children left, left+1

} // namespace

struct rnb_bvh {
	std::vector<double> v0, e1, e2; // per triangle slot (reordered): vertex 0 and the two edges
	std::vector<uint32_t> tri_id;
	std::vector<Node> nodes;

	void build(const float* V, const uint32_t* T, uint32_t n) {
		std::vector<float> lo((size_t)n * 3), hi((size_t)n * 3), ce((size_t)n * 3);
		for (uint32_t t = 0; t < n; ++t) for (int k = 0; k < 3; ++k) {
			const float a = V[(size_t)T[3 * t] * 3 + k], b = V[(size_t)T[3 * t + 1] * 3 + k], c = V[(size_t)T[3 * t + 2] * 3 + k];
			lo[3 * t + k] = std::min(a, std::min(b, c)); hi[3 * t + k] = std::max(a, std::max(b, c)); ce[3 * t + k] = (lo[3 * t + k] + hi[3 * t + k]) * 0.5f;
		}
		std::vector<uint32_t> order(n);
		std::iota(order.begin(), order.end(), 0u);
		nodes.clear(); nodes.reserve(2 * (size_t)n / 2 + 2);
		nodes.push_back(Node{});
		struct Job { uint32_t node, begin, end; };
		std::vector<Job> stack{{0, 0, n}};
		while (!stack.empty()) {
			const Job j = stack.back(); stack.pop_back();
			Node nd;
			for (int k = 0; k < 3; ++k) { nd.lo[k] = std::numeric_limits<float>::max(); nd.hi[k] = -std::numeric_limits<float>::max(); }
			float clo[3] = {nd.lo[0], nd.lo[1], nd.lo[2]}, chi[3] = {nd.hi[0], nd.hi[1], nd.hi[2]};
			for (uint32_t i = j.begin; i < j.end; ++i) for (int k = 0; k < 3; ++k) {
				const uint32_t t = order[i];
				nd.lo[k] = std::min(nd.lo[k], lo[3 * t + k]); nd.hi[k] = std::max(nd.hi[k], hi[3 * t + k]);
				clo[k] = std::min(clo[k], ce[3 * t + k]); chi[k] = std::max(chi[k], ce[3 * t + k]);
			}
			const uint32_t cnt = j.end - j.begin;
			int axis = 0;
			for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
			if (cnt <= 4 || !(chi[axis] > clo[axis])) { nd.left = j.begin; nd.count = cnt; nodes[j.node] = nd; continue; }
			const uint32_t mid = j.begin + cnt / 2;
			std::nth_element(order.begin() + j.begin, order.begin() + mid, order.begin() + j.end, [&](uint32_t a, uint32_t b) { return ce[3 * a + axis] < ce[3 * b + axis]; });
			nd.left = (uint32_t)nodes.size(); nd.count = 0;
			nodes[j.node] = nd;
			nodes.push_back(Node{}); nodes.push_back(Node{});
			stack.push_back({nd.left, j.begin, mid});
			stack.push_back({nd.left + 1, mid, j.end});
		}
		tri_id = order;
		v0.resize((size_t)n * 3); e1.resize((size_t)n * 3); e2.resize((size_t)n * 3);
		for (uint32_t s = 0; s < n; ++s) for (int k = 0; k < 3; ++k) {
			const uint32_t t = order[s];
			const double a = V[(size_t)T[3 * t] * 3 + k];
			v0[3 * s + k] = a; e1[3 * s + k] = (double)V[(size_t)T[3 * t + 1] * 3 + k] - a; e2[3 * s + k] = (double)V[(size_t)T[3 * t + 2] * 3 + k] - a;
		}
	}

	// Moeller-Trumbore, two-sided; returns t or -1
	inline double hit_tri(uint32_t s, const double* o, const double* d) const {
		const double *a = &v0[3 * s], *u = &e1[3 * s], *v = &e2[3 * s];
		const double p[3] = {d[1] * v[2] - d[2] * v[1], d[2] * v[0] - d[0] * v[2], d[0] * v[1] - d[1] * v[0]};
		const double det = u[0] * p[0] + u[1] * p[1] + u[2] * p[2];
		if (std::fabs(det) < 1e-300) return -1.0;
		const double inv = 1.0 / det;
		const double s0[3] = {o[0] - a[0], o[1] - a[1], o[2] - a[2]};
		const double bu = (s0[0] * p[0] + s0[1] * p[1] + s0[2] * p[2]) * inv;
		if (bu < -1e-12 || bu > 1.0 + 1e-12) return -1.0;
		const double q[3] = {s0[1] * u[2] - s0[2] * u[1], s0[2] * u[0] - s0[0] * u[2], s0[0] * u[1] - s0[1] * u[0]};
		const double bv = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
		if (bv < -1e-12 || bu + bv > 1.0 + 1e-12) return -1.0;
		const double t = (v[0] * q[0] + v[1] * q[1] + v[2] * q[2]) * inv;
		return t > 0.0 ? t : -1.0;
	}

	template <bool ANY> void trace(const double* o, const double* d, double tmax, double& t_best, int32_t& tri_best) const {
		t_best = tmax; tri_best = -1;
		if (nodes.empty()) return;
		double inv[3];
		for (int k = 0; k < 3; ++k) inv[k] = 1.0 / (d[k] != 0.0 ? d[k] : 1e-300);
		uint32_t stack[64]; int sp = 0;
		stack[sp++] = 0;
		while (sp) {
			const Node& nd = nodes[stack[--sp]];
			double t0 = 0.0, t1 = t_best;
			for (int k = 0; k < 3; ++k) {
				double a = ((double)nd.lo[k] - 1e-6 - o[k]) * inv[k], b = ((double)nd.hi[k] + 1e-6 - o[k]) * inv[k];
				if (a > b) std::swap(a, b);
				t0 = std::max(t0, a); t1 = std::min(t1, b);
			}
			if (t0 > t1) continue;
			if (nd.count) {
				for (uint32_t s = nd.left; s < nd.left + nd.count; ++s) {
					const double t = hit_tri(s, o, d);
					if (t > 0.0 && t < t_best) { t_best = t; tri_best = (int32_t)tri_id[s]; if (ANY) return; }
				}
			} else if (sp + 2 <= 64) { stack[sp++] = nd.left; stack[sp++] = nd.left + 1; }
		}
	}
};

extern "C" {

const char* rnb_host_last_error(void) { return g_err.c_str(); }

int rnb_png_info(const char* path, uint32_t* width, uint32_t* height, int32_t* channels, int32_t* depth) {
	return guarded([&] { const png16::Info i = png16::probe(path); *width = i.width; *height = i.height; *channels = i.channels; *depth = i.depth; });
}

int rnb_png_read_rgba16(const char* path, uint16_t* out) {
	return guarded([&] { const png16::Image im = png16::load(path); std::memcpy(out, im.rgba.data(), im.rgba.size() * 2); });
}

int rnb_png_write(const char* path, const void* data, uint32_t width, uint32_t height, int32_t channels, int32_t depth, int32_t level) {
	return guarded([&] { png16::save(path, data, width, height, channels, depth, level); });
}

int rnb_bvh_create(const float* vertices, uint32_t n_vertices, const uint32_t* triangles, uint32_t n_triangles, rnb_bvh** out) {
	return guarded([&] {
		for (size_t k = 0; k < (size_t)n_triangles * 3; ++k) if (triangles[k] >= n_vertices) throw std::runtime_error("bvh: triangle index out of range");
		rnb_bvh* b = new rnb_bvh();
		b->build(vertices, triangles, n_triangles);
		*out = b;
	});
}

int rnb_bvh_destroy(rnb_bvh* bvh) { delete bvh; return 0; }

int rnb_bvh_first_hit(const rnb_bvh* bvh, const double* origins, const double* directions, uint32_t n_rays, double* t_out, int32_t* tri_out) {
	return guarded([&] {
		#pragma omp parallel for schedule(dynamic, 256)
		for (int64_t i = 0; i < (int64_t)n_rays; ++i) {
			double t; int32_t tri;
			bvh->trace<false>(origins + 3 * i, directions + 3 * i, std::numeric_limits<double>::infinity(), t, tri);
			t_out[i] = tri >= 0 ? t : std::numeric_limits<double>::infinity(); tri_out[i] = tri;
		}
	});
}

int rnb_bvh_occluded(const rnb_bvh* bvh, const double* origins, const double* directions, const double* t_max, uint32_t n_rays, uint8_t* occluded_out) {
	return guarded([&] {
		#pragma omp parallel for schedule(dynamic, 256)
		for (int64_t i = 0; i < (int64_t)n_rays; ++i) {
			double t; int32_t tri;
			bvh->trace<true>(origins + 3 * i, directions + 3 * i, t_max[i], t, tri);
			occluded_out[i] = tri >= 0 ? 1 : 0;
		}
	});
}

} // extern "C"
