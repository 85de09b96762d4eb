// Probe (GPU box): WHICH ARITHMETIC does v_mfma_f32_16x16x16_f16 (and _16x16x32_f16) perform inside one instruction?  D = C + sum_k A[m][k] B[k][n] -- the products of two
// halfs are exact in fp32 (22-bit significands); what the ISA documents does not say is in which order and with how many roundings the K products and C are added.
// The half mode's k-step (mlp.cuh: mfma_emul16_k16) is one such instruction, and the oracle's model of it (dot_h) has to sum the same way if HIP and model are to agree on
// more than "a neighbouring half now and then". Random operands over several exponent spreads; every candidate model is evaluated on the host in __float128 (exact for
// these sums: 2^-48 .. 2^32 x 22 bits < 113 bits) and compared bit for bit with the instruction's result.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result tools/probe_mfma_arith.hip -o /tmp/probe_mfma_arith && /tmp/probe_mfma_arith
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ void k_mfma(const _Float16* A /*[t][16][K]*/, const _Float16* B /*[t][K][16]*/, const float* C /*[t][16][16]*/, float* D, int n_tiles) {
	const int t = blockIdx.x, l = threadIdx.x;
	if (t >= n_tiles) return;
	A += (size_t)t * 16 * K; B += (size_t)t * K * 16; C += (size_t)t * 256; D += (size_t)t * 256;
	f4 acc;
	for (int r = 0; r < 4; ++r) acc[r] = C[(4 * (l >> 4) + r) * 16 + (l & 15)];
	if (K == 16) {
		h4 a, b;
		for (int j = 0; j < 4; ++j) { a[j] = A[(l & 15) * 16 + 4 * (l >> 4) + j]; b[j] = B[(4 * (l >> 4) + j) * 16 + (l & 15)]; }
		acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
	} else {
		h8 a, b;
		for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
		acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
	}
	for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

typedef __float128 q_t;
static float rne(q_t x) { return (float)x; }
static float rtz(q_t x) { float g = (float)x; if (g == 0.f || (q_t)g == x) return g; if ((x > 0 && (q_t)g > x) || (x < 0 && (q_t)g < x)) g = nextafterf(g, 0.f); return g; }

struct Model { const char* name; float (*f)(const float* p, int K, float c, int G); int G; };
// p[k] = exact products in the instruction's k order; G = group size
static float m_seq_c_first(const float* p, int K, float c, int) { float a = c; for (int k = 0; k < K; ++k) a = a + p[k]; return a; }
static float m_seq_then_c(const float* p, int K, float c, int) { float a = 0.f; for (int k = 0; k < K; ++k) a = a + p[k]; return c + a; }
static float m_exact(const float* p, int K, float c, int) { q_t s = c; for (int k = 0; k < K; ++k) s += p[k]; return rne(s); }
static float m_exact_rtz(const float* p, int K, float c, int) { q_t s = c; for (int k = 0; k < K; ++k) s += p[k]; return rtz(s); }
static float m_groups_fused(const float* p, int K, float c, int G) { float a = c; for (int g = 0; g < K; g += G) { q_t s = a; for (int k = g; k < g + G; ++k) s += p[k]; a = rne(s); } return a; }
static float m_groups_fused_rtz(const float* p, int K, float c, int G) { float a = c; for (int g = 0; g < K; g += G) { q_t s = a; for (int k = g; k < g + G; ++k) s += p[k]; a = rtz(s); } return a; }
static float m_groups_round(const float* p, int K, float c, int G) { float a = c; for (int g = 0; g < K; g += G) { q_t s = 0; for (int k = g; k < g + G; ++k) s += p[k]; a = a + rne(s); } return a; }
static float m_groups_then_c(const float* p, int K, float c, int G) { q_t t = 0; float a = 0.f; for (int g = 0; g < K; g += G) { q_t s = a; for (int k = g; k < g + G; ++k) s += p[k]; a = rne(s); } (void)t; return c + a; }
// interleaved k order: lanes' quarter index q = k / (K/4) -- the instruction may walk j (the element inside a lane) in its outer loop
static float m_groups_fused_strided(const float* p, int K, float c, int G) { const int per = K / 4; float a = c; for (int j = 0; j < per; j += G / 4 ? G / 4 : 1) { q_t s = a; for (int jj = j; jj < j + (G / 4 ? G / 4 : 1); ++jj) for (int q = 0; q < 4; ++q) s += p[q * per + jj]; a = rne(s); } return a; }

static FILE* g_dump = nullptr; static int g_dumped = 0;
template <int K>
static void run(int spread, bool c_half_valued, bool c_zero, std::mt19937& gen) {
	const int T = 512;
	std::vector<_Float16> A((size_t)T * 16 * K), B((size_t)T * K * 16);
	std::vector<float> C((size_t)T * 256), D((size_t)T * 256);
	std::uniform_real_distribution<float> u(-1.f, 1.f);
	std::uniform_int_distribution<int> e(-spread, spread);
	for (auto& v : A) v = (_Float16)(u(gen) * std::ldexp(1.f, e(gen) / 2));
	for (auto& v : B) v = (_Float16)(u(gen) * std::ldexp(1.f, e(gen) / 2));
	for (auto& v : C) { float x = c_zero ? 0.f : u(gen) * std::ldexp(1.f, e(gen)); v = c_half_valued ? (float)(_Float16)x : x; }
	_Float16 *dA, *dB; float *dC, *dD;
	hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
	hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
	k_mfma<K><<<T, 64>>>(dA, dB, dC, dD, T);
	hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
	hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
	std::vector<Model> models = {{"sequential fp32, C first", m_seq_c_first, 1}, {"sequential fp32 from 0, then + C (the oracle's dot_h)", m_seq_then_c, 1}, {"exact sum of C and all K products, one rounding (RNE)", m_exact, K},
	                             {"exact sum, one rounding toward zero", m_exact_rtz, K}};
	for (int G : {2, 4, 8, 16}) if (G < K) {
		models.push_back({"groups: acc = RNE(acc + exact sum of G products)", m_groups_fused, G});
		models.push_back({"groups: acc = RTZ(acc + exact sum of G products)", m_groups_fused_rtz, G});
		models.push_back({"groups: acc = acc + RNE(exact sum of G products)", m_groups_round, G});
		models.push_back({"groups from 0 (fused), then + C", m_groups_then_c, G});
	}
	for (int G : {4, 8, 16}) if (G < K) models.push_back({"groups fused, k interleaved over the four lane quarters", m_groups_fused_strided, G});
	std::vector<long> hit(models.size(), 0);
	long n = 0, differ_from_exact = 0;
	for (int t = 0; t < T; ++t) for (int m = 0; m < 16; ++m) for (int nn = 0; nn < 16; ++nn) {
		float p[32];
		for (int k = 0; k < K; ++k) p[k] = (float)A[((size_t)t * 16 + m) * K + k] * (float)B[((size_t)t * K + k) * 16 + nn];
		const float c = C[(size_t)t * 256 + m * 16 + nn], d = D[(size_t)t * 256 + m * 16 + nn];
		++n;
		for (size_t q = 0; q < models.size(); ++q) { const float r = models[q].f(p, K, c, models[q].G); if (std::memcmp(&r, &d, 4) == 0 || (r == 0.f && d == 0.f)) ++hit[q]; }
		const float ex = m_exact(p, K, c, K); if (std::memcmp(&ex, &d, 4) != 0) ++differ_from_exact;
		if (g_dump && g_dumped < 4000) { // every case, for offline analysis: K products (hex), C, D
			std::fprintf(g_dump, "%d %d", K, spread);
			for (int k = 0; k < K; ++k) { uint32_t u; std::memcpy(&u, &p[k], 4); std::fprintf(g_dump, " %08x", u); }
			uint32_t uc, ud; std::memcpy(&uc, &c, 4); std::memcpy(&ud, &d, 4);
			std::fprintf(g_dump, " %08x %08x\n", uc, ud);
			++g_dumped;
		}
	}
	std::printf("K = %d, exponent spread +-%d, C %s: %ld dot products\n", K, spread, c_zero ? "= 0" : c_half_valued ? "half-valued" : "any fp32", n);
	for (size_t q = 0; q < models.size(); ++q) std::printf("  %-70s G=%2d  %8.4f %%%s\n", models[q].name, models[q].G, 100.0 * hit[q] / n, hit[q] == n ? "   <== ALL" : "");
}

int main(int argc, char** argv) {
	std::mt19937 gen(12345);
	if (argc > 1) { g_dump = std::fopen(argv[1], "w"); run<16>(2, true, false, gen); g_dumped = 0; run<16>(8, true, false, gen); g_dumped = 0; run<16>(2, true, true, gen); std::fclose(g_dump); return 0; }
	for (int spread : {2, 8, 14}) { run<16>(spread, true, false, gen); }
	run<16>(14, false, false, gen);
	run<16>(14, true, true, gen);
	for (int spread : {2, 14}) run<32>(spread, true, false, gen);
	run<32>(14, false, false, gen);
	return 0;
}
