// Probe: throughput of the atomic flavours the grid-gradient scatter could use, on hashed (random) and clustered addresses.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_atomics.hip -o tools/probe_atomics && tools/probe_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

enum Mode { F32_1 = 0, F32_PAIR, F32_QUAD, PK16, PK16_PAIR, F32_1_WG, F32_QUAD_WG, PK16_WG, F32_QUAD_RET, STORE4 };

template <int MODE>
__global__ void k(float* tab, uint32_t n_entries /* 8-byte entries */, uint32_t per_thread, uint32_t seed) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	for (uint32_t it = 0; it < per_thread; ++it) {
		float v = 1.0f;
		if (MODE == F32_1 || MODE == F32_1_WG) {
			const uint32_t e = mix(gid * 977u + it * 131071u + seed) % (n_entries * 2);
			if (MODE == F32_1) atomicAdd(&tab[e], v);
			else __hip_atomic_fetch_add(&tab[e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else if (MODE == F32_PAIR) { // 2 lanes share an 8-byte entry
			const uint32_t e = mix((gid >> 1) * 977u + it * 131071u + seed) % n_entries;
			atomicAdd(&tab[e * 2 + (gid & 1)], v);
		} else if (MODE == F32_QUAD || MODE == F32_QUAD_WG || MODE == F32_QUAD_RET) { // 4 lanes share an aligned 16-byte block
			const uint32_t e = mix((gid >> 2) * 977u + it * 131071u + seed) % (n_entries / 2);
			if (MODE == F32_QUAD) atomicAdd(&tab[e * 4 + (gid & 3)], v);
			else if (MODE == F32_QUAD_WG) __hip_atomic_fetch_add(&tab[e * 4 + (gid & 3)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else { float r = atomicAdd(&tab[e * 4 + (gid & 3)], v); if (r == 123456.f) tab[0] = r; }
		} else if (MODE == PK16 || MODE == PK16_WG) { // one lane = one half2 entry (4 bytes)
			const uint32_t e = mix(gid * 977u + it * 131071u + seed) % (n_entries * 2);
			typedef _Float16 h2 __attribute__((ext_vector_type(2)));
			h2 hv = {(_Float16)1.f, (_Float16)1.f};
			h2* p = reinterpret_cast<h2*>(tab) + e;
			if (MODE == PK16) __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)p, hv);
			else __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)p, hv);
		} else if (MODE == PK16_PAIR) { // 2 lanes = adjacent half2 entries (x, x+1)
			const uint32_t e = mix((gid >> 1) * 977u + it * 131071u + seed) % n_entries;
			typedef _Float16 h2 __attribute__((ext_vector_type(2)));
			h2 hv = {(_Float16)1.f, (_Float16)1.f};
			h2* p = reinterpret_cast<h2*>(tab) + e * 2 + (gid & 1);
			__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)p, hv);
		} else if (MODE == STORE4) {
			const uint32_t e = mix(gid * 977u + it * 131071u + seed) % (n_entries * 2);
			tab[e] = v;
		}
	}
}

template <int MODE> void run(const char* name, float* tab, uint32_t n_entries, double lanes_per_op) {
	const uint32_t blocks = 4096, threads = 256, per_thread = 32;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	k<MODE><<<blocks, threads>>>(tab, n_entries, per_thread, 1);
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int r = 0; r < 5; ++r) k<MODE><<<blocks, threads>>>(tab, n_entries, per_thread, 7 + r);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	const double lane_ops = 5.0 * blocks * threads * per_thread;
	printf("%-14s table %6.1f MB: %7.3f ms  %7.1f G lane-ops/s  %7.1f G entries(8B)/s\n", name, n_entries * 8.0 / 1e6, ms / 5, lane_ops / (ms * 1e6), lane_ops / lanes_per_op / (ms * 1e6));
}

int main() {
	for (uint32_t n_entries : {1u << 19, 9u << 19}) { // one hashed level (4 MB fp32 pairs) / all nine (38 MB)
		float* tab; hipMalloc(&tab, (size_t)n_entries * 8); hipMemset(tab, 0, (size_t)n_entries * 8);
		run<F32_1>("f32 x1", tab, n_entries, 2);
		run<F32_PAIR>("f32 pair", tab, n_entries, 2);
		run<F32_QUAD>("f32 quad", tab, n_entries, 2);
		run<F32_QUAD_RET>("f32 quad ret", tab, n_entries, 2);
		run<PK16>("pk_f16 x1", tab, n_entries, 1);
		run<PK16_PAIR>("pk_f16 pair", tab, n_entries, 1);
		run<F32_1_WG>("f32 x1 wg", tab, n_entries, 2);
		run<F32_QUAD_WG>("f32 quad wg", tab, n_entries, 2);
		run<STORE4>("store 4B", tab, n_entries, 2);
		hipFree(tab);
	}
	return 0;
}
