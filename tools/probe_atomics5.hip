// Probe (round 2, second part): is the 21 G lines/s of global fp32 atomics (probe_atomics4) a limit of each XCD's L2 atomic unit, or of
// lines travelling between the eight L2s / to the memory side?
//   (a) table size 256 KB ... 256 MB, every workgroup over the whole table
//   (b) only the workgroups of ONE XCD active (hardware XCC_ID), whole table
//   (c) XCD-partitioned: a workgroup only touches the slice of the table that belongs to its XCD (XCC_ID), agent and workgroup scope
//   hipcc --offload-arch=gfx950 -O3 tools/probe_atomics5.hip -o tools/probe_atomics5 && tools/probe_atomics5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7u; }

enum Mode { ALL = 0, ONE_XCD, PARTITION_AGENT, PARTITION_WG, PARTITION_STORE, PARTITION_LOAD };

template <int MODE, int G>
__global__ void k(float* tab, uint32_t n_lines, uint32_t per_thread, uint32_t seed, float* sink, uint32_t* xcd_hist) {
	const uint32_t x = xcc_id();
	if (threadIdx.x == 0 && xcd_hist) atomicAdd(&xcd_hist[x], 1u);
	if (MODE == ONE_XCD && x != 0) return;
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t grp = gid / G, sub = gid % G;
	float acc = 0.f;
	for (uint32_t it = 0; it < per_thread; ++it) {
		uint32_t line = mix(grp * 977u + it * 131071u + seed);
		if (MODE >= PARTITION_AGENT) line = (line % (n_lines / 8)) + x * (n_lines / 8);
		else line = line % n_lines;
		float* p = tab + (size_t)line * 16 + sub;
		if (MODE == PARTITION_WG) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else if (MODE == PARTITION_STORE) *p = 1.0f;
		else if (MODE == PARTITION_LOAD) acc += *p;
		else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (acc == 123456.f) *sink = acc;
}

template <int MODE, int G> void run(const char* name, float* tab, size_t bytes, uint32_t blocks, float* sink, uint32_t* hist) {
	const uint32_t threads = 256;
	const uint32_t n_lines = (uint32_t)(bytes / 64);
	const uint32_t per_thread = (uint32_t)((1ull << 25) / ((uint64_t)blocks * threads));
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipMemset(hist, 0, 32);
	k<MODE, G><<<blocks, threads>>>(tab, n_lines, per_thread, 1, sink, hist);
	hipDeviceSynchronize();
	uint32_t h[8]; hipMemcpy(h, hist, 32, hipMemcpyDeviceToHost);
	hipEventRecord(a);
	for (int r = 0; r < 3; ++r) k<MODE, G><<<blocks, threads>>>(tab, n_lines, per_thread, 7 + r, sink, nullptr);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	double lane_ops = 3.0 * blocks * threads * per_thread;
	if (MODE == ONE_XCD) lane_ops *= (double)h[0] / blocks;
	printf("%-22s G=%2d table %8.2f MB blocks %5u: %8.3f ms  %7.1f G lines/s   (workgroups per XCD: %u %u %u %u %u %u %u %u)\n", name, G, bytes / 1048576.0, blocks, ms / 3,
	       lane_ops / G / (ms * 1e6), h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main() {
	float* sink; hipMalloc(&sink, 4);
	uint32_t* hist; hipMalloc(&hist, 32);
	const size_t max_bytes = 256ull << 20;
	float* tab; hipMalloc(&tab, max_bytes); hipMemset(tab, 0, max_bytes);
	for (size_t bytes : {256ull << 10, 1ull << 20, 4ull << 20, 16ull << 20, 32ull << 20, 256ull << 20}) {
		run<ALL, 1>("all, agent scope", tab, bytes, 4096, sink, hist);
		run<ALL, 4>("all, agent scope", tab, bytes, 4096, sink, hist);
	}
	for (size_t bytes : {1ull << 20, 32ull << 20}) {
		run<ONE_XCD, 1>("one XCD active", tab, bytes, 4096, sink, hist);
		run<ONE_XCD, 4>("one XCD active", tab, bytes, 4096, sink, hist);
	}
	for (size_t bytes : {2ull << 20, 8ull << 20, 32ull << 20, 256ull << 20}) {
		run<PARTITION_AGENT, 1>("partitioned, agent", tab, bytes, 4096, sink, hist);
		run<PARTITION_AGENT, 4>("partitioned, agent", tab, bytes, 4096, sink, hist);
		run<PARTITION_WG, 1>("partitioned, workgroup", tab, bytes, 4096, sink, hist);
		run<PARTITION_WG, 4>("partitioned, workgroup", tab, bytes, 4096, sink, hist);
		run<PARTITION_STORE, 1>("partitioned, store", tab, bytes, 4096, sink, hist);
		run<PARTITION_LOAD, 1>("partitioned, load", tab, bytes, 4096, sink, hist);
	}
	return 0;
}
