// Probe (round 2): what bounds global fp32 / packed-f16 atomics on gfx950 -- lane operations issued by a CU, or requests at the memory side?
//   (a) group size G: G adjacent lanes hit G contiguous dwords of one random line (1, 2, 4, 8, 16)
//   (b) number of workgroups (64 ... 4096): does the rate scale with the CUs in use, or saturate?
//   (c) random 4-byte gathers / stores at the same patterns for comparison, and LDS atomics
//   hipcc --offload-arch=gfx950 -O3 tools/probe_atomics4.hip -o tools/probe_atomics4 && tools/probe_atomics4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

enum Op { ATOM_F32 = 0, ATOM_PK16, LOAD4, STORE4, LOAD16 };

template <int OP, int G>
__global__ void k(float* tab, uint32_t n_dwords, uint32_t per_thread, uint32_t seed, float* sink) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	float acc = 0.f;
	for (uint32_t it = 0; it < per_thread; ++it) {
		const uint32_t grp = gid / G, sub = gid % G;
		const uint32_t e = (mix(grp * 977u + it * 131071u + seed) % (n_dwords / 16)) * 16 + sub; // an aligned 64-byte line, dword `sub` of it
		if (OP == ATOM_F32) atomicAdd(&tab[e], 1.0f);
		else if (OP == ATOM_PK16) { h2 hv = {(_Float16)1.f, (_Float16)1.f}; __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(reinterpret_cast<h2*>(tab) + e), hv); }
		else if (OP == LOAD4) acc += tab[e];
		else if (OP == STORE4) tab[e] = 1.0f;
		else if (OP == LOAD16) { const float4 v = reinterpret_cast<const float4*>(tab)[e / 4]; acc += v.x + v.w; }
	}
	if (acc == 123456.f) *sink = acc;
}

template <int OP, int G> void run(const char* name, float* tab, uint32_t n_dwords, uint32_t blocks, float* sink) {
	const uint32_t threads = 256;
	const uint32_t per_thread = (uint32_t)((1ull << 25) / ((uint64_t)blocks * threads)); // 32 M lane operations per launch
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	k<OP, G><<<blocks, threads>>>(tab, n_dwords, per_thread, 1, sink);
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int r = 0; r < 3; ++r) k<OP, G><<<blocks, threads>>>(tab, n_dwords, per_thread, 7 + r, sink);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	const double lane_ops = 3.0 * blocks * threads * per_thread;
	printf("%-10s G=%2d blocks %5u: %8.3f ms  %7.1f G lane-ops/s  %7.1f G lines/s\n", name, G, blocks, ms / 3, lane_ops / (ms * 1e6), lane_ops / G / (ms * 1e6));
}

__global__ void k_lds(float* out, uint32_t per_thread, uint32_t seed, int same_bank) {
	extern __shared__ float tab[];
	for (uint32_t q = threadIdx.x; q < 32768; q += blockDim.x) tab[q] = 0.f;
	__syncthreads();
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	for (uint32_t it = 0; it < per_thread; ++it) {
		uint32_t e = mix(gid * 977u + it * 131071u + seed) & 32767u;
		if (same_bank) e = (e & ~63u) | (threadIdx.x & 63u); // conflict-free: lane i -> bank i
		atomicAdd(&tab[e], 1.0f);
	}
	__syncthreads();
	if (tab[threadIdx.x] == 123456.f) out[0] = 1.f;
}

int main() {
	float* sink; hipMalloc(&sink, 4);
	const uint32_t n_dwords = 8u << 20; // 32 MB: the eight fine levels' fp32 gradient tables
	float* tab; hipMalloc(&tab, (size_t)n_dwords * 4); hipMemset(tab, 0, (size_t)n_dwords * 4);
	for (uint32_t blocks : {64u, 256u, 1024u, 4096u}) {
		run<ATOM_F32, 1>("atom f32", tab, n_dwords, blocks, sink);
		run<ATOM_F32, 4>("atom f32", tab, n_dwords, blocks, sink);
		run<ATOM_F32, 16>("atom f32", tab, n_dwords, blocks, sink);
	}
	run<ATOM_F32, 2>("atom f32", tab, n_dwords, 4096, sink);
	run<ATOM_F32, 8>("atom f32", tab, n_dwords, 4096, sink);
	run<ATOM_PK16, 1>("atom pk16", tab, n_dwords, 4096, sink);
	run<ATOM_PK16, 2>("atom pk16", tab, n_dwords, 4096, sink);
	run<ATOM_PK16, 4>("atom pk16", tab, n_dwords, 4096, sink);
	for (uint32_t blocks : {256u, 4096u}) {
		run<LOAD4, 1>("load 4B", tab, n_dwords, blocks, sink);
		run<LOAD4, 4>("load 4B", tab, n_dwords, blocks, sink);
		run<LOAD16, 1>("load 16B", tab, n_dwords, blocks, sink);
		run<STORE4, 1>("store 4B", tab, n_dwords, blocks, sink);
		run<STORE4, 4>("store 4B", tab, n_dwords, blocks, sink);
	}
	// LDS atomics: 256 workgroups x 1024 threads, 128 KB table each
	hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	for (int same_bank = 0; same_bank < 2; ++same_bank) {
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		k_lds<<<256, 1024, 131072>>>(sink, 128, 1, same_bank); hipDeviceSynchronize();
		hipEventRecord(a);
		for (int r = 0; r < 3; ++r) k_lds<<<256, 1024, 131072>>>(sink, 128, 7 + r, same_bank);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("lds atomic f32 %s: %8.3f ms  %7.1f G lane-ops/s\n", same_bank ? "conflict-free" : "random       ", ms / 3, 3.0 * 256 * 1024 * 128 / (ms * 1e6));
	}
	return 0;
}
