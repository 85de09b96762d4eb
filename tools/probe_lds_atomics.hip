// Probe: rate of LDS atomics by type on gfx950 (256 workgroups x 1024 threads, random addresses in a 128 KB table).
//   hipcc --offload-arch=gfx950 -O3 -w tools/probe_lds_atomics.hip -o tools/probe_lds_atomics && tools/probe_lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, uint32_t per_thread, uint32_t seed) {
	extern __shared__ __attribute__((aligned(16))) char raw[];
	uint32_t* t32 = reinterpret_cast<uint32_t*>(raw);
	for (uint32_t q = threadIdx.x; q < 32768; q += blockDim.x) t32[q] = 0;
	__syncthreads();
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t h = mix(gid * 977u + seed);
	for (uint32_t it = 0; it < per_thread; ++it) {
		h = h * 1664525u + 1013904223u;
		const uint32_t e = (h >> 8) & 32767u;
		if (MODE == 0) atomicAdd(reinterpret_cast<float*>(raw) + e, 1.0f);
		else if (MODE == 1) atomicAdd(t32 + e, 1u);
		else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(raw) + (e & 16383u), 1ull);
		else if (MODE == 3) t32[e] = it;            // plain store
		else if (MODE == 4) { float* p = reinterpret_cast<float*>(raw) + e; *p = *p + 1.0f; } // non-atomic read-modify-write
		else if (MODE == 5) atomicAdd(reinterpret_cast<double*>(raw) + (e & 16383u), 1.0);
		else if (MODE == 6) { typedef _Float16 h2v __attribute__((ext_vector_type(2))); const h2v one = {(_Float16)1.0f, (_Float16)1.0f};
			(void)__builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2v*)(t32 + e), one); }      // ds_pk_add_f16
		else if (MODE == 7) unsafeAtomicAdd(reinterpret_cast<float*>(raw) + e, 1.0f);
		else if (MODE == 8) atomicAdd(reinterpret_cast<double*>(raw) + ((e & 16383u) & ~63u) + (threadIdx.x & 63u), 1.0); // f64, one lane per bank pair: no conflicts
		else if (MODE == 9) atomicAdd(reinterpret_cast<double*>(raw) + (e & 15u), 1.0); // f64, 16 hot entries (the coarse levels' few hundred cells)
		else if (MODE == 10) atomicAdd(reinterpret_cast<float*>(raw) + (e & 15u), 1.0f); // f32, 16 hot entries
	}
	__syncthreads();
	if (t32[threadIdx.x] == 0x7fffffffu) out[0] = 1.f;
}
template <int MODE> void run(const char* name, float* sink) {
	hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	k<MODE><<<256, 1024, 131072>>>(sink, 128, 1); hipDeviceSynchronize();
	hipEventRecord(a);
	for (int r = 0; r < 3; ++r) k<MODE><<<256, 1024, 131072>>>(sink, 128, 7 + r);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	printf("%-28s %8.3f ms  %7.1f G lane-ops/s  (%.2f cycles per lane-op per CU at 2.4 GHz)\n", name, ms / 3, 3.0 * 256 * 1024 * 128 / (ms * 1e6), (ms / 3 * 1e-3 * 2.4e9) / (1024.0 * 128));
}
int main() {
	float* sink; hipMalloc(&sink, 4);
	run<0>("ds atomic add f32", sink);
	run<1>("ds atomic add u32", sink);
	run<2>("ds atomic add u64", sink);
	run<5>("ds atomic add f64", sink);
	run<6>("ds pk add f16", sink);
	run<7>("ds unsafe atomic add f32", sink);
	run<8>("ds atomic add f64 no conflicts", sink);
	run<9>("ds atomic add f64 16 hot", sink);
	run<10>("ds atomic add f32 16 hot", sink);
	run<3>("ds store b32", sink);
	run<4>("ds load + add + store", sink);
	return 0;
}
