// probe_counters.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_ATOMIC report on gfx950 for THIS path's access patterns?
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is exactly half of the bytes of a wide coalesced streaming read on gfx950; "other access widths and WRITE_SIZE are
// uncalibrated: calibrate on a known byte count in your own access pattern".) Every kernel below moves a KNOWN number of bytes / lines, far beyond the 256 MiB
// Infinity Cache where the pattern allows it; tools/calibrate_counters.sh runs this binary under `rocprofv3 --pmc <one counter>` and divides.
//   p_stream_read16     16 B per lane, coalesced, 2 GiB read once                                  (the guide's calibration case)
//   p_stream_write16    16 B per lane, coalesced, 2 GiB written once
//   p_gather8_far       2^26 8-byte gathers, each from its own 64-byte line of a 4 GiB table        (a hash-grid gather that misses every cache)
//   p_gather8_grid      2^26 8-byte gathers at random entries of a 21 MB table                     (the hash grid itself: L2 / Infinity Cache absorb most)
//   p_record64_rw       2^25 records of 64 bytes: a lane reads its record as 4 x 16 B and writes it back   (k_adam_ema's optimizer records)
//   p_half4_rw          2^27 x 8 bytes per lane read and written back, coalesced                    (k_adam_ema's fp16 weights / EMA, 8 B per lane)
//   p_atomic_far        2^24 float atomics without return, each to its own 64-byte line of a 1 GiB array
//   p_atomic_grid       2^26 float atomics at random entries of a 42 MB table                       (the gradient scatter's table)
// The binary prints one JSON line with the true bytes / operations and the wall time of every kernel (the time bounds what a request can have carried:
// requests/s x bytes per request cannot exceed what HBM delivers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void p_stream_read16(const f4* __restrict__ src, size_t n, float* __restrict__ sink) {
	f4 acc = {0, 0, 0, 0};
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
	if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = 1.f;
}
__global__ void p_stream_write16(f4* __restrict__ dst, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = f4{1.f, 2.f, 3.f, (float)i};
}
// line q of the table is visited exactly once: q = (i * odd) mod n_lines is a permutation of the lines for a power-of-two line count
__global__ void p_gather8_far(const uint2* __restrict__ table, uint32_t n_lines_log2, size_t n, float* __restrict__ sink) {
	uint32_t acc = 0;
	const uint32_t mask = (1u << n_lines_log2) - 1u;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const uint32_t line = ((uint32_t)i * 0x9E3779B1u) & mask;
		const uint2 v = table[(size_t)line * 8 + (mix((uint32_t)i) & 7u)];
		acc += v.x ^ v.y;
	}
	if (acc == 0x12345678u) sink[0] = 1.f;
}
__global__ void p_gather8_grid(const uint2* __restrict__ table, uint32_t n_entries, size_t n, float* __restrict__ sink) {
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const uint2 v = table[mix((uint32_t)i * 2654435761u + 17u) % n_entries];
		acc += v.x ^ v.y;
	}
	if (acc == 0x12345678u) sink[0] = 1.f;
}
__global__ void p_record64_rw(f4* __restrict__ rec, size_t n_rec) {
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_rec; q += (size_t)gridDim.x * blockDim.x) {
		f4* r = rec + q * 4;
		f4 a = r[0], b = r[1], c = r[2], d = r[3];
		a += 1.f; b = b * 0.9f + a; c = c * 0.99f + a * a; d += 1.f;
		r[0] = a; r[1] = b; r[2] = c; r[3] = d;
	}
}
__global__ void p_half4_rw(h4* __restrict__ w, size_t n) {
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) { h4 v = w[q]; v += (_Float16)1.f; w[q] = v; }
}
__global__ void p_atomic_far(float* __restrict__ a, uint32_t n_lines_log2, size_t n) {
	const uint32_t mask = (1u << n_lines_log2) - 1u;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const uint32_t line = ((uint32_t)i * 0x9E3779B1u) & mask;
		atomicAdd(a + (size_t)line * 16 + (mix((uint32_t)i) & 15u), 1.0f);
	}
}
__global__ void p_atomic_grid(float* __restrict__ a, uint32_t n_entries, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(a + mix((uint32_t)i * 2654435761u + 17u) % n_entries, 1.0f);
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
	const size_t GiB = 1ull << 30;
	char* big = nullptr;
	float* sink = nullptr;
	CK(hipMalloc((void**)&big, 4 * GiB));
	CK(hipMalloc((void**)&sink, 64));
	CK(hipMemset(big, 0, 4 * GiB));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	struct Row { std::string name; double read_bytes, write_bytes, ops, ms; };
	std::vector<Row> rows;
	auto timed = [&](const char* name, double rb, double wb, double ops, auto&& launch) {
		launch(); // warm (page tables)
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0, 0);
		launch();
		(void)hipEventRecord(e1, 0);
		(void)hipEventSynchronize(e1);
		float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
		rows.push_back({name, rb, wb, ops, ms});
	};
	const dim3 grid(256 * 16), block(256);
	timed("p_stream_read16", 2.0 * GiB, 0, 2.0 * GiB / 16, [&] { hipLaunchKernelGGL(p_stream_read16, grid, block, 0, 0, (const f4*)big, 2 * GiB / 16, sink); });
	timed("p_stream_write16", 0, 2.0 * GiB, 2.0 * GiB / 16, [&] { hipLaunchKernelGGL(p_stream_write16, grid, block, 0, 0, (f4*)big, 2 * GiB / 16); });
	timed("p_gather8_far", (double)(1ull << 26) * 64, 0, (double)(1ull << 26), [&] { hipLaunchKernelGGL(p_gather8_far, grid, block, 0, 0, (const uint2*)big, 26u, (size_t)1 << 26, sink); });
	timed("p_gather8_grid", (double)(1ull << 26) * 8, 0, (double)(1ull << 26), [&] { hipLaunchKernelGGL(p_gather8_grid, grid, block, 0, 0, (const uint2*)big, 2637032u, (size_t)1 << 26, sink); });
	timed("p_record64_rw", 2.0 * GiB, 2.0 * GiB, (double)(1ull << 25), [&] { hipLaunchKernelGGL(p_record64_rw, grid, block, 0, 0, (f4*)big, (size_t)1 << 25); });
	timed("p_half4_rw", 1.0 * GiB, 1.0 * GiB, (double)(1ull << 27), [&] { hipLaunchKernelGGL(p_half4_rw, grid, block, 0, 0, (h4*)big, (size_t)1 << 27); });
	timed("p_atomic_far", (double)(1ull << 24) * 64, (double)(1ull << 24) * 64, (double)(1ull << 24), [&] { hipLaunchKernelGGL(p_atomic_far, grid, block, 0, 0, (float*)big, 24u, (size_t)1 << 24); });
	timed("p_atomic_grid", 0, 0, (double)(1ull << 26), [&] { hipLaunchKernelGGL(p_atomic_grid, grid, block, 0, 0, (float*)big, 10548128u, (size_t)1 << 26); });
	std::printf("{\"kernels\": {");
	for (size_t i = 0; i < rows.size(); ++i)
		std::printf("%s\"%s\": {\"true_read_bytes\": %.0f, \"true_write_bytes\": %.0f, \"ops\": %.0f, \"ms\": %.4f}", i ? ", " : "", rows[i].name.c_str(), rows[i].read_bytes, rows[i].write_bytes, rows[i].ops, rows[i].ms);
	std::printf("}}\n");
	return 0;
}
