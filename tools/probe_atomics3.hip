// Probe 3: does any cache-scope variant of global_atomic_add_f32 run faster (i.e. execute in the XCD-local L2)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int V>
__global__ void k(float* tab, uint32_t n_words, uint32_t per_thread, uint32_t seed, uint32_t per_xcd_words) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	float* base = tab;
	if (per_xcd_words) { // private copy per XCD
		uint32_t xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		base = tab + (size_t)(xcc & 7u) * per_xcd_words;
	}
	for (uint32_t it = 0; it < per_thread; ++it) {
		const uint32_t r = mix((gid / 4) * 977u + it * 131071u + seed);
		float* p = base + (size_t)(r % (n_words / 4)) * 4 + (gid % 4);
		float v = 1.0f;
		if (V == 0) asm volatile("global_atomic_add_f32 %0, %1, off" :: "v"(p), "v"(v) : "memory");
		if (V == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory"); // note: sc0 on an atomic = return value on gfx940
		if (V == 2) asm volatile("global_atomic_add_f32 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
		if (V == 3) asm volatile("global_atomic_add_f32 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
		if (V == 4) asm volatile("global_atomic_add_f32 %0, %1, off nt sc1" :: "v"(p), "v"(v) : "memory");
		if (V == 5) atomicAdd(p, v);
		if (V == 6) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		if (V == 7) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	asm volatile("s_waitcnt vmcnt(0)");
}
template <int V> void run(const char* name, float* tab, uint32_t n_words, uint32_t per_xcd_words) {
	const uint32_t blocks = 4096, threads = 256, per_thread = 32;
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
	k<V><<<blocks, threads>>>(tab, n_words, per_thread, 1, per_xcd_words);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	for (int r = 0; r < 5; ++r) k<V><<<blocks, threads>>>(tab, n_words, per_thread, 7 + r, per_xcd_words);
	(void)hipEventRecord(b); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b);
	const double lane_ops = 5.0 * blocks * threads * per_thread;
	printf("%-28s xcd-private %d: %7.3f ms  %8.1f G lane-ops/s  %6.1f G quads/s  (err %s)\n", name, per_xcd_words ? 1 : 0, ms / 5, lane_ops / (ms * 1e6), lane_ops / 4 / (ms * 1e6), hipGetErrorString(hipGetLastError()));
}
int main() {
	const uint32_t n_words = 1u << 20; // 4 MB table (one hashed level, fp32 x 2)
	float* tab; (void)hipMalloc(&tab, (size_t)n_words * 4 * 8); (void)hipMemset(tab, 0, (size_t)n_words * 4 * 8);
	for (uint32_t px : {0u, n_words}) {
		run<0>("asm plain", tab, n_words, px);
		run<2>("asm sc1", tab, n_words, px);
		run<3>("asm nt", tab, n_words, px);
		run<4>("asm nt sc1", tab, n_words, px);
		run<5>("atomicAdd", tab, n_words, px);
		run<6>("hip_atomic wavefront scope", tab, n_words, px);
		run<7>("hip_atomic system scope", tab, n_words, px);
	}
	return 0;
}
