// probe_level_major.hip — VERDICT round 2, item 4 / SURVEY.md section 7 "hard parts": is the hash-grid encode faster LEVEL-MAJOR with every XCD
// working on one level at a time (a hashed level's table is 2 MB, an XCD's L2 4 MB), than SAMPLE-MAJOR with all 14 levels (20 MB) live in
// every wavefront? The encode of k_point_query_chained in isolation -- the product's own encode_level_core (csrc/common.cuh), the real
// table geometry, 2^20 uniformly random points (an occupancy update's worst case) -- in three forms, each writing feat[level][point]:
//   sample-major      one thread per point walks the 14 levels (what every kernel of the product does)
//   level-major       levels striped over ALL workgroups: every XCD still touches every table, one level at a time per workgroup
//   level-major, XCD  the workgroups of XCD x (hardware XCC_ID) take levels x and x + 8 one after the other: an XCD's L2 serves one table at a time
// Prints microseconds per pass and the rate of gather lane-lines (points x 14 levels x 8 corners / time). Counters: run under
//   rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum ... / --pmc TCC_HIT_sum TCC_MISS_sum
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I rnb-neus2_amd/csrc tools/probe_level_major.hip -o build/probe_level_major
#include "common.cuh"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rnb;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7u; }

__global__ __launch_bounds__(256) void k_sample_major(const GridMeta G, const uint32_t* __restrict__ grid, const float* __restrict__ xyz, const uint32_t n, uint32_t* __restrict__ feat) {
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
#pragma unroll 1
		for (uint32_t l = 0; l < G.n_levels; ++l) {
			half_t f0, f1; float d0[3], d1[3];
			encode_level<false>(G, grid, l, x, y, z, f0, f1, d0, d1);
			feat[(size_t)l * n + i] = pack_h2(f0, f1);
		}
	}
}

// level_of[slot]: the levels a workgroup class works through, -1 = none. XCD = true: class = XCC_ID, index inside the class from a ticket counter.
template <bool XCD>
__global__ __launch_bounds__(256) void k_level_major(const GridMeta G, const uint32_t* __restrict__ grid, const float* __restrict__ xyz, const uint32_t n, uint32_t* __restrict__ feat,
                                                     uint32_t* __restrict__ tickets, const uint32_t wgs_per_class) {
	__shared__ uint32_t s_idx;
	const uint32_t cls = XCD ? xcc_id() : blockIdx.x % 8u;
	if (threadIdx.x == 0) s_idx = atomicAdd(tickets + cls, 1u);
	__syncthreads();
	const uint32_t idx = s_idx; // this workgroup's index among those of its class
	if (idx >= wgs_per_class) return; // (a class that received more workgroups than planned: the surplus has nothing to do)
	for (uint32_t pass = 0; pass < 2; ++pass) {
		const uint32_t l = cls + 8 * pass;
		if (l >= G.n_levels) break;
		for (uint32_t i = idx * blockDim.x + threadIdx.x; i < n; i += wgs_per_class * blockDim.x) {
			half_t f0, f1; float d0[3], d1[3];
			encode_level<false>(G, grid, l, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], f0, f1, d0, d1);
			feat[(size_t)l * n + i] = pack_h2(f0, f1);
		}
	}
}

int main() {
	GridMeta G{};
	G.n_levels = 14; G.valid_level = 14;
	const float pls = std::exp(std::log(2048.0f / 16.0f) / 13.0f);
	uint32_t off = 0;
	for (uint32_t i = 0; i < 14; ++i) { // grid.h:977-1012 as in rnb_neus2_hip.hip build_grid_tables
		const float scale = std::exp2f(i * std::log2(pls)) * 16 - 1.0f;
		const uint32_t res = (uint32_t)std::ceil(scale) + 1;
		uint32_t p = res * res * res; p = (p + 7) / 8 * 8; p = std::min(p, 1u << 19);
		G.scale[i] = (float)(res - 1); G.resolution[i] = res; G.offsets[i] = off; off += p;
	}
	for (uint32_t i = 14; i <= RNB_MAX_LEVELS; ++i) G.offsets[i] = off;
	const uint32_t n = 1u << 20;
	std::vector<uint32_t> hg(off + 8);
	for (size_t i = 0; i < hg.size(); ++i) hg[i] = (uint32_t)(i * 2654435761u) & 0x33ff33ffu; // small finite halfs
	std::vector<float> hx((size_t)n * 3);
	uint32_t s = 12345;
	for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f); }
	uint32_t *dg, *df, *df2, *dt; float* dx;
	CHECK(hipMalloc(&dg, hg.size() * 4)); CHECK(hipMalloc(&df, (size_t)14 * n * 4)); CHECK(hipMalloc(&df2, (size_t)14 * n * 4)); CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dt, 32));
	CHECK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const double lane_lines = (double)n * 14 * 8;
	auto time_it = [&](const char* name, auto launch, uint32_t* out) {
		float best = 1e30f, sum = 0.f;
		for (int rep = 0; rep < 12; ++rep) {
			CHECK(hipMemset(dt, 0, 32));
			CHECK(hipEventRecord(e0, 0));
			launch(out);
			CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (rep >= 2) { best = std::min(best, ms); sum += ms; }
		}
		std::printf("%-44s %8.1f us (best %8.1f)   %6.1f G corner gathers/s\n", name, 1e3 * sum / 10, 1e3 * best, lane_lines / (sum / 10 * 1e-3) / 1e9);
	};
	for (uint32_t wg_cu : {4u, 8u}) {
		const uint32_t wgs = 256 * wg_cu;
		std::printf("-- %u workgroups of 256 threads per CU --\n", wg_cu);
		time_it("sample-major (14 levels per thread)", [&](uint32_t* o) { hipLaunchKernelGGL(k_sample_major, dim3(wgs), dim3(256), 0, 0, G, dg, dx, n, o); }, df);
		time_it("level-major, levels striped over all XCDs", [&](uint32_t* o) { hipLaunchKernelGGL(k_level_major<false>, dim3(wgs), dim3(256), 0, 0, G, dg, dx, n, o, dt, wgs / 8); }, df2);
		time_it("level-major, one level at a time per XCD", [&](uint32_t* o) { hipLaunchKernelGGL(k_level_major<true>, dim3(wgs), dim3(256), 0, 0, G, dg, dx, n, o, dt, wgs / 8); }, df2);
	}
	std::vector<uint32_t> a((size_t)14 * n), b((size_t)14 * n);
	CHECK(hipMemcpy(a.data(), df, a.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), df2, b.size() * 4, hipMemcpyDeviceToHost));
	size_t diff = 0;
	for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
	std::printf("features of the level-major form that differ from the sample-major ones: %zu of %zu\n", diff, a.size());
	return 0;
}
