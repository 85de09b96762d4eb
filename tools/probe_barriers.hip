// probe_barriers.hip -- what the stream-ordering primitives of the step's critical stream cost on MI355X / ROCm 7.2 (round 4).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probe_barriers tools/probe_barriers.hip && tools/probe_barriers
// Stream 1 runs a chain of N short kernels (~20 us each); between two of them we put
//   (a) nothing
//   (b) a completion event on the first kernel (hipExtLaunchKernelGGL stop event) -- the "signal-carrying kernel" of the step
//   (c) k hipStreamWaitEvent on events of stream 2 that completed long ago                 (satisfied barrier packets)
//   (d) one hipStreamWaitEvent on an event of stream 2 that completes ~at the same time     (cross-queue hand-over)
//   (e) a hipEventRecord (marker packet) between them
// and report the wall time per chain link from a device-side timestamp written by each kernel (wall_clock64), so that the host's
// launch latency is not part of the number (everything is queued before the first kernel starts: a gate kernel spins on a host flag).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_gate(volatile int* flag) { while (*flag == 0) __builtin_amdgcn_s_sleep(10); }
__global__ void k_work(unsigned long long* stamps, int slot, int spin) {
	if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = wall_clock64();
	unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < (unsigned long long)spin) { }
	__syncthreads();
	if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}

int main() {
	hipStream_t s1, s2;
	CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	int* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
	int* flag_dev; CK(hipHostGetDevicePointer((void**)&flag_dev, flag, 0));
	unsigned long long* stamps; CK(hipHostMalloc((void**)&stamps, 4096 * 16, hipHostMallocMapped));
	unsigned long long* stamps_dev; CK(hipHostGetDevicePointer((void**)&stamps_dev, stamps, 0));
	const unsigned dev_flags = hipEventDisableTiming | (unsigned)hipEventDisableSystemFence;
	std::vector<hipEvent_t> ev(64);
	for (auto& e : ev) CK(hipEventCreateWithFlags(&e, dev_flags));
	int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
	const double us = 1e3 / (double)rate_khz; // one tick in us
	const int spin = (int)(20.0 / us);         // ~20 us of work per kernel
	const int blocks = 256, threads = 256;
	auto gap = [&](const char* name, auto&& between) -> int {
		std::vector<double> gaps;
		for (int rep = 0; rep < 30; ++rep) {
			*flag = 0;
			hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s1, flag_dev);
			hipLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s1, stamps_dev, 0, spin);
			between(rep);
			hipLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s1, stamps_dev, 1, spin);
			CK(hipGetLastError());
			*flag = 1;
			CK(hipDeviceSynchronize());
			gaps.push_back((double)(stamps[2] - stamps[1]) * us);
		}
		std::sort(gaps.begin(), gaps.end());
		std::printf("%-72s gap p50 %6.2f us  min %6.2f  max %6.2f\n", name, gaps[gaps.size() / 2], gaps.front(), gaps.back());
		return 0;
	};
	std::printf("wall clock %d kHz; ~20 us kernels of %d x %d threads\n", rate_khz, blocks, threads);
	gap("(a) two kernels back to back", [&](int) {});
	// (b) completion signal on the first kernel: re-issue the first kernel with a stop event (the lambda runs between the two launches, so emulate by an extra kernel)
	gap("(b) + a kernel with a completion event in between (gap includes its 20 us)", [&](int) { hipExtLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s1, nullptr, ev[0], 0, stamps_dev, 2, spin); });
	gap("(b0) + a plain kernel in between (gap includes its 20 us)", [&](int) { hipLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s1, stamps_dev, 2, spin); });
	gap("(e) hipEventRecord between", [&](int) { (void)hipEventRecord(ev[1], s1); });
	// (b2) the event of the kernel in between has a waiter on stream 2 (as every event of the step's critical stream has)
	gap("(b2) + a kernel with a completion event in between AND stream 2 waiting for it (incl. 20 us)", [&](int) {
		hipExtLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s1, nullptr, ev[30], 0, stamps_dev, 2, spin);
		(void)hipStreamWaitEvent(s2, ev[30], 0);
		hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, stamps_dev, 13, 1);
	});
	// (b3) the same, the event recorded with hipEventRecord (marker packet) instead of riding on the kernel
	gap("(b3) + a kernel, hipEventRecord, stream 2 waiting for it (incl. 20 us)", [&](int) {
		hipLaunchKernelGGL(k_work, dim3(blocks), dim3(threads), 0, s1, stamps_dev, 2, spin);
		(void)hipEventRecord(ev[31], s1);
		(void)hipStreamWaitEvent(s2, ev[31], 0);
		hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, stamps_dev, 13, 1);
	});
	// (b4) a kernel that writes 64 MB in between (dirty L2 at its end), with and without a completion event
	{
		static float* big = nullptr;
		if (!big) (void)hipMalloc((void**)&big, 64u << 20);
		gap("(b4) + hipMemsetAsync of 64 MB in between", [&](int) { (void)hipMemsetAsync(big, 1, 64u << 20, s1); });
	}
	// satisfied waits: events of stream 2 recorded and completed before the gate opens
	for (int k = 1; k <= 3; ++k) {
		char name[96]; std::snprintf(name, sizeof(name), "(c) %d hipStreamWaitEvent on events that are complete at enqueue", k);
		gap(name, [&](int) {
			for (int q = 0; q < k; ++q) { hipExtLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, nullptr, ev[2 + q], 0, stamps_dev, 8 + q, 1); }
			(void)hipStreamSynchronize(s2);
			for (int q = 0; q < k; ++q) (void)hipStreamWaitEvent(s1, ev[2 + q], 0);
		});
	}
	// waits on events that are NOT complete at enqueue but complete well before the first kernel of stream 1 ends: stream 2's kernels sit behind their own gate
	for (int k = 1; k <= 3; ++k) {
		char name[96]; std::snprintf(name, sizeof(name), "(c') %d hipStreamWaitEvent on events pending at enqueue, complete ~15 us before needed", k);
		gap(name, [&](int) {
			hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s2, flag_dev);
			for (int q = 0; q < k; ++q) { hipExtLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, nullptr, ev[8 + q], 0, stamps_dev, 8 + q, 1); }
			for (int q = 0; q < k; ++q) (void)hipStreamWaitEvent(s1, ev[8 + q], 0);
		});
	}
	// (d) hand-over: stream 2's kernel ends ~5 us AFTER stream 1's first kernel; the gap minus 5 us is the signal-to-start latency
	gap("(d) wait on a stream-2 kernel that ends ~5 us after stream 1's (gap includes those 5 us)", [&](int) {
		hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s2, flag_dev);
		hipExtLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, nullptr, ev[20], 0, stamps_dev, 12, spin + (int)(5.0 / us));
		(void)hipStreamWaitEvent(s1, ev[20], 0);
	});
	gap("(d') the same through hipEventRecord on stream 2", [&](int) {
		hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s2, flag_dev);
		hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, stamps_dev, 12, spin + (int)(5.0 / us));
		(void)hipEventRecord(ev[21], s2);
		(void)hipStreamWaitEvent(s1, ev[21], 0);
	});
	return 0;
}
