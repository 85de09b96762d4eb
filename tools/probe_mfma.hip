// Probe: verifies the lane->element maps of v_mfma_f32_16x16x32_f16 that csrc/ relies on.
//   A: lane l holds A[m = l&15][k = 8*(l>>4) + j], j = 0..7
//   B: lane l holds B[k = 8*(l>>4) + j][n = l&15]
//   D: lane l, reg r holds D[row = 4*(l>>4) + r][col = l&15]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16* A /*16x32 row-major*/, const _Float16* B /*32x16 row-major*/, float* D /*16x16*/) {
	int l = threadIdx.x;
	h8 a, b;
	for (int j = 0; j < 8; ++j) {
		a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
		b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
	}
	f4 acc = {0, 0, 0, 0};
	acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
	for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
int main() {
	std::vector<_Float16> A(16 * 32), B(32 * 16);
	for (int i = 0; i < 16 * 32; ++i) A[i] = (_Float16)((rand() % 17) - 8);
	for (int i = 0; i < 32 * 16; ++i) B[i] = (_Float16)((rand() % 13) - 6);
	_Float16 *dA, *dB; float* dD;
	hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
	hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
	hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
	k<<<1, 64>>>(dA, dB, dD);
	std::vector<float> D(256);
	hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
		float ref = 0;
		for (int kk = 0; kk < 32; ++kk) ref += (float)A[m * 32 + kk] * (float)B[kk * 16 + n];
		if (ref != D[m * 16 + n]) ++bad;
	}
	printf("mfma_16x16x32_f16 layout probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
	return bad ? 1 : 0;
}
