// kernels_mesh.cuh — iso-surface extraction on the device:
//   k_lattice_positions + k_half_to_float   get_density_on_grid / generate_grid_samples_nerf_uniform   src/testbed_nerf.cu:541-553, 4218-4269
//   k_mc_verts<WRITE>                       gen_vertices                                               src/marching_cubes.cu:276-327
//   k_mc_faces<WRITE>                       gen_faces                                                  src/marching_cubes.cu:377-430, 676-717
// The reference numbers vertices and triangles with atomicAdd counters (any order). Here both are numbered by prefix sums in
// lattice order -- point index ascending, then axis x, y, z; cell index ascending, then table order -- which is the order of the
// host loop in host/mesh.hpp: the two produce identical buffers (tests/test_gpu_mesh.py), and a mesh is reproducible from run to
// run. The case table is the one host/mesh.hpp generates (same polygon loops as the published table in all 256 cases).
#pragma once
#include "common.cuh"

namespace rnb {

struct McArgs {
	const float* density; // [rx * ry * rz], x fastest
	uint32_t rx, ry, rz;
	float thresh;
	float sc[3], mn[3]; // lattice point p sits at mn + p * sc
};

constexpr uint32_t MC_WG = 256;

// Exclusive prefix sums over blocks of 1024 values, in place; block totals to `sums` (one per workgroup).
__global__ __launch_bounds__(1024) void k_scan_blocks(uint32_t* __restrict__ data, const uint64_t n, uint32_t* __restrict__ sums) {
	__shared__ uint32_t wsum[16];
	const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t c = i < n ? data[i] : 0u;
	uint32_t v = c;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if ((int)lane >= off) v += t; }
	if (lane == 63) wsum[wave] = v;
	__syncthreads();
	uint32_t before = 0, total = 0;
#pragma unroll
	for (uint32_t q = 0; q < 16; ++q) { if (q < wave) before += wsum[q]; total += wsum[q]; }
	if (i < n) data[i] = before + v - c;
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_scan_add(uint32_t* __restrict__ data, const uint64_t n, const uint32_t* __restrict__ block_offsets) {
	const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
	if (i < n) data[i] += block_offsets[blockIdx.x];
}

// Workgroup-local exclusive sum of one small count per thread (256 threads); returns the thread's offset, *total = workgroup sum.
__device__ __forceinline__ uint32_t wg_exclusive_256(const uint32_t c, uint32_t* total) {
	__shared__ uint32_t wsum[4];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t v = c;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if ((int)lane >= off) v += t; }
	__syncthreads(); // wsum may still be read from a previous call
	if (lane == 63) wsum[wave] = v;
	__syncthreads();
	uint32_t before = 0, tot = 0;
#pragma unroll
	for (uint32_t q = 0; q < 4; ++q) { if (q < wave) before += wsum[q]; tot += wsum[q]; }
	*total = tot;
	return before + v - c;
}

// gen_vertices: one vertex per lattice edge whose ends lie on different sides of the threshold, at the linear interpolation
// of the two values. WRITE = false: per-workgroup counts; WRITE = true: positions + the edge -> vertex index grid [3][res^3]
// (-1: no vertex), numbered from wg_offset.
template <bool WRITE>
__global__ __launch_bounds__(MC_WG) void k_mc_verts(const McArgs a, uint32_t* __restrict__ wg_count, const uint32_t* __restrict__ wg_offset, float* __restrict__ verts, int32_t* __restrict__ vidx) {
	const uint64_t res2 = (uint64_t)a.rx * a.ry, res3 = res2 * a.rz;
	const uint64_t idx = (uint64_t)blockIdx.x * MC_WG + threadIdx.x;
	uint32_t cross = 0; // bit a: the edge from this point along axis a carries a vertex
	float f0 = 0.f, f1[3] = {0.f, 0.f, 0.f};
	uint32_t p[3] = {0, 0, 0};
	if (idx < res3) {
		p[0] = (uint32_t)(idx % a.rx); p[1] = (uint32_t)((idx / a.rx) % a.ry); p[2] = (uint32_t)(idx / res2);
		f0 = a.density[idx];
		const bool in0 = f0 > a.thresh;
		const uint64_t step[3] = {1, a.rx, res2};
		const uint32_t lim[3] = {a.rx - 1, a.ry - 1, a.rz - 1};
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			if (p[d] >= lim[d]) continue;
			f1[d] = a.density[idx + step[d]];
			if (in0 != (f1[d] > a.thresh)) cross |= 1u << d;
		}
	}
	uint32_t total;
	const uint32_t local = wg_exclusive_256(__popc(cross), &total);
	if (!WRITE) {
		if (threadIdx.x == 0) wg_count[blockIdx.x] = total;
		return;
	}
	if (idx >= res3) return;
	uint32_t id = wg_offset[blockIdx.x] + local;
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		int32_t out = -1;
		if (cross & (1u << d)) {
			const float dt = (a.thresh - f0) / (f1[d] - f0);
			float q[3] = {(float)p[0], (float)p[1], (float)p[2]};
			q[d] += dt;
			verts[(size_t)id * 3 + 0] = q[0] * a.sc[0] + a.mn[0];
			verts[(size_t)id * 3 + 1] = q[1] * a.sc[1] + a.mn[1];
			verts[(size_t)id * 3 + 2] = q[2] * a.sc[2] + a.mn[2];
			out = (int32_t)id++;
		}
		vidx[idx + res3 * d] = out;
	}
}

struct McTable { int8_t tri[256][40]; uint8_t n[256]; }; // edge ids, 3 per triangle; n = number of indices

// gen_faces: the cell whose lowest corner is this lattice point.
template <bool WRITE>
__global__ __launch_bounds__(MC_WG) void k_mc_faces(const McArgs a, const McTable* __restrict__ T, uint32_t* __restrict__ wg_count, const uint32_t* __restrict__ wg_offset,
                                                    const int32_t* __restrict__ vidx, uint32_t* __restrict__ indices) {
	const uint64_t res2 = (uint64_t)a.rx * a.ry, res3 = res2 * a.rz;
	const uint64_t idx = (uint64_t)blockIdx.x * MC_WG + threadIdx.x;
	uint32_t mask = 0;
	if (idx < res3) {
		const uint32_t x = (uint32_t)(idx % a.rx), y = (uint32_t)((idx / a.rx) % a.ry), z = (uint32_t)(idx / res2);
		if (x + 1 < a.rx && y + 1 < a.ry && z + 1 < a.rz) {
			// corner numbering of src/marching_cubes.cu:261-275: 0 (0,0,0) 1 (1,0,0) 2 (1,1,0) 3 (0,1,0), 4..7 the same at z + 1
			const uint64_t o[8] = {0, 1, 1 + (uint64_t)a.rx, a.rx, res2, res2 + 1, res2 + 1 + a.rx, res2 + a.rx};
#pragma unroll
			for (int c = 0; c < 8; ++c) if (a.density[idx + o[c]] > a.thresh) mask |= 1u << c;
			if (mask == 255u) mask = 0;
		}
	}
	const uint32_t n = mask ? T->n[mask] : 0u;
	uint32_t total;
	const uint32_t local = wg_exclusive_256(n, &total);
	if (!WRITE) {
		if (threadIdx.x == 0) wg_count[blockIdx.x] = total;
		return;
	}
	if (!n) return;
	uint32_t* dst = indices + (size_t)wg_offset[blockIdx.x] + local;
	// edge e of the cell -> (lattice point carrying it, axis): edges 0-3 in the z plane, 4-7 in the z + 1 plane, 8-11 along z
	const uint64_t eo[12] = {0, 1, a.rx, 0, res2, res2 + 1, res2 + a.rx, res2, 0, 1, 1 + (uint64_t)a.rx, a.rx};
	const uint32_t ea[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
	for (uint32_t k = 0; k < n; ++k) {
		const int e = T->tri[mask][k];
		dst[k] = (uint32_t)vidx[idx + eo[e] + res3 * ea[e]];
	}
}

// generate_grid_samples_nerf_uniform (src/testbed_nerf.cu:541-553) for lattice points [first, first + n): warped positions.
__global__ void k_lattice_positions(const uint64_t first, const uint32_t n, const uint32_t rx, const uint32_t ry, const uint32_t rz, const float lat_min, const float lat_size,
                                    const float aabb_min, const float aabb_diag, float* __restrict__ out) {
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q >= n) return;
	const uint64_t i = first + q;
	const uint32_t p[3] = {(uint32_t)(i % rx), (uint32_t)((i / rx) % ry), (uint32_t)(i / ((uint64_t)rx * ry))};
	const uint32_t r[3] = {rx, ry, rz};
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const float inv = 1.f / (float)r[k];
		const float w = (float)p[k] * inv * lat_size + lat_min;
		out[(size_t)q * 3 + k] = (w - aabb_min) / aabb_diag; // warp_position
	}
}
__global__ void k_half_to_float(const half_t* __restrict__ src, float* __restrict__ dst, const uint32_t n) {
	const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
	if (q < n) dst[q] = h2f(src[q]);
}

} // namespace rnb
