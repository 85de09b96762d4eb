// Probe 2: what unit the atomic rate is counted in. G lanes share one random, G*4-byte aligned block; SAME = all G lanes hit one word.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int G, int KIND> // KIND 0: contiguous words; 1: same word; 2: lanes spread 16 B apart inside one G*16-byte region (same line, different 16B blocks)
__global__ void k(float* tab, uint32_t n_words, uint32_t per_thread, uint32_t seed) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	for (uint32_t it = 0; it < per_thread; ++it) {
		const uint32_t r = mix((gid / G) * 977u + it * 131071u + seed);
		uint32_t w;
		if (KIND == 0) w = (r % (n_words / G)) * G + (gid % G);
		else if (KIND == 1) w = (r % n_words);
		else w = (r % (n_words / (G * 4))) * (G * 4) + (gid % G) * 4;
		atomicAdd(&tab[w], 1.0f);
	}
}
template <int G, int KIND> void run(const char* name, float* tab, uint32_t n_words, uint32_t blocks) {
	const uint32_t threads = 256, per_thread = 32;
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	k<G, KIND><<<blocks, threads>>>(tab, n_words, per_thread, 1);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	for (int r = 0; r < 5; ++r) k<G, KIND><<<blocks, threads>>>(tab, n_words, per_thread, 7 + r);
	(void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b);
	const double lane_ops = 5.0 * blocks * threads * per_thread;
	printf("%-22s blocks %5u: %7.3f ms  %8.1f G lane-ops/s  %7.1f G groups/s\n", name, blocks, ms / 5, lane_ops / (ms * 1e6), lane_ops / G / (ms * 1e6));
}
int main() {
	const uint32_t n_words = 9u << 20; // 37.7 MB
	float* tab; (void)hipMalloc(&tab, (size_t)n_words * 4); (void)hipMemset(tab, 0, (size_t)n_words * 4);
	for (uint32_t blocks : {4096u, 512u, 256u}) {
		run<1, 0>("1 lane / block", tab, n_words, blocks);
		run<4, 0>("4 lanes contiguous", tab, n_words, blocks);
		run<8, 0>("8 lanes contiguous", tab, n_words, blocks);
		run<16, 0>("16 lanes contiguous", tab, n_words, blocks);
		run<32, 0>("32 lanes contiguous", tab, n_words, blocks);
		run<64, 0>("64 lanes contiguous", tab, n_words, blocks);
		run<4, 1>("4 lanes same word", tab, n_words, blocks);
		run<16, 1>("16 lanes same word", tab, n_words, blocks);
		run<64, 1>("64 lanes same word", tab, n_words, blocks);
		run<2, 2>("2 lanes 16B apart", tab, n_words, blocks);
		run<4, 2>("4 lanes 16B apart", tab, n_words, blocks);
		run<8, 2>("8 lanes 16B apart", tab, n_words, blocks);
	}
	return 0;
}
