// repro_sidestream_hazard.hip — stand-alone attempt to reproduce the side-stream hazard of DESIGN.md section 6 (round 1/2: rays of wavefront
// lanes 48-63 marched with a wrong direction in 2-3 % of the launches of the 16-lanes-per-ray march kernel WHILE k_fwd_bwd ran beside it on
// another stream; never on an idle GPU; gone without packed fp32 instructions and without ballot masks spilled to VGPR lanes).
//
// Two kernels on two streams:
//   k_mfma_hog   the neighbour: register-chained v_mfma_f32_16x16x32_f16 layers + LDS traffic, 240+ VGPRs, two workgroups per CU (k_fwd_bwd's shape)
//   k_victim     the march's ray set-up in miniature: per lane a 3x4 camera matrix times a pixel direction (the component pairs are written so
//                that the compiler emits v_pk_mul_f32 / v_pk_fma_f32 when packed fp32 is enabled), a normalisation, then a loop that carries
//                FOUR 64-bit ballot masks across iterations together with ~90 live SGPRs (so that the compiler spills SGPRs through
//                v_writelane / v_readlane), and finally the SAME product recomputed: a lane whose two results differ is reported.
// Modes (argv[1]): "idle" = victim alone, "beside" = victim launched while the hog runs. Build twice:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/repro_sidestream_hazard.hip -o repro_pk        (packed fp32 on: the default)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops ... -o repro_nopk
// Output: launches with at least one deviating lane, and a histogram of the deviating lanes by 16-lane row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

__global__ __launch_bounds__(256, 2) void k_mfma_hog(const _Float16* __restrict__ w, float* __restrict__ out, const int iters) {
	extern __shared__ _Float16 lds[];
	for (int i = threadIdx.x; i < 64 * 72 * 4; i += 256) lds[i] = w[i & 4095];
	__syncthreads();
	const int lane = threadIdx.x & 63, r16 = lane & 15, hq = lane >> 4;
	f4 acc[4][4];
	for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
	h8 bf[4][2];
	for (int nt = 0; nt < 4; ++nt) for (int ks = 0; ks < 2; ++ks) bf[nt][ks] = *reinterpret_cast<const h8*>(lds + (16 * nt + r16) * 72 + 32 * ks + 8 * hq);
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int mt = 0; mt < 4; ++mt)
#pragma unroll
			for (int ks = 0; ks < 2; ++ks) {
				const h8 af = *reinterpret_cast<const h8*>(lds + ((16 * mt + r16 + it) & 63) * 72 + 32 * ks + 8 * hq);
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[nt][ks], acc[mt][nt], 0, 0, 0);
			}
#pragma unroll
		for (int nt = 0; nt < 4; ++nt)
#pragma unroll
			for (int ks = 0; ks < 2; ++ks)
#pragma unroll
				for (int j = 0; j < 8; ++j) bf[nt][ks][j] = (_Float16)(acc[2 * ks + (j >> 2)][nt][j & 3] * 1e-3f);
	}
	float s = 0.f;
	for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

struct Cam { float m[12]; };

__device__ __forceinline__ void ray_dir(const Cam& c, const float px, const float py, float d[3]) {
	// pairs (x, y) go through f2 arithmetic: v_pk_mul_f32 / v_pk_fma_f32 with packed fp32 enabled
	const f2 cx = {c.m[0], c.m[4]}, cy = {c.m[1], c.m[5]}, cz = {c.m[2], c.m[6]};
	f2 xy = cx * f2{px, px};
	xy = xy + cy * f2{py, py};
	xy = xy + cz;
	const float z = c.m[8] * px + c.m[9] * py + c.m[10];
	const float n = sqrtf(xy[0] * xy[0] + xy[1] * xy[1] + z * z);
	d[0] = xy[0] / n; d[1] = xy[1] / n; d[2] = z / n;
}

struct Uniforms { float k[40]; }; // kernel-argument scalars that stay live in the loop (the march's MarchArgs): SGPR pressure

__global__ __launch_bounds__(256) void k_victim(const Cam* __restrict__ cams, const uint32_t n_rays, const uint32_t rounds, const uint32_t* __restrict__ bits, uint32_t* __restrict__ bad, float* __restrict__ sink, const Uniforms U) {
	const uint32_t ray = blockIdx.x * 16 + threadIdx.x / 16; // 16 lanes per ray, as the march
	const int lane = threadIdx.x & 63, g = lane & 15, gb = lane & ~15;
	const Cam c = cams[ray % 64];
	const float px = (float)((ray * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f, py = (float)((ray * 805459861u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
	float d[3];
	ray_dir(c, px, py, d);
	// carry four ballot masks + a lot of scalar state through a loop (SGPR pressure -> spills through VGPR lanes)
	unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
	float t = 0.f, acc = 0.f;
	uint32_t j = 0;
	for (uint32_t r = 0; r < rounds; ++r) {
		// 16 running positions with loop-invariant lane predicates (m > g): the compiler keeps the 15 masks in SGPR pairs, as it did in the march
		float T[17];
		T[0] = t;
		float my_t = t;
#pragma unroll
		for (int m = 0; m < 16; ++m) { T[m + 1] = T[m] + U.k[m & 7] * 1e-3f + 0.0016914f; if (m + 1 == g) my_t = T[m + 1]; }
		uint32_t nxt = 16;
		const float tgt = my_t + U.k[8 + (r & 7)] * 1e-2f + 0.004f;
#pragma unroll
		for (int m = 15; m >= 1; --m) if (m > g && T[m] >= tgt) nxt = (uint32_t)m;
		float usum = 0.f;
#pragma unroll
		for (int q = 16; q < 40; ++q) usum += U.k[q] * (float)((r + q) & 3);
		const float x = d[0] * my_t + c.m[3] + usum * 1e-9f, y = d[1] * my_t + c.m[7], z = d[2] * my_t + c.m[11] + (float)nxt * 1e-9f;
		const uint32_t w = bits[(ray * 31u + r * 17u + (uint32_t)g) & 4095u];
		const bool occ = (w >> (g & 31)) & 1u, in = fabsf(x) < 4.f && fabsf(y) < 4.f && fabsf(z) < 4.f;
		m0 = __ballot(occ); m1 = __ballot(in); m2 = __ballot(occ && in); m3 = __ballot(!occ && in && (g & 1));
		const unsigned long long o16 = (m0 >> gb) & 0xffffull, i16 = (m1 >> gb) & 0xffffull, b16 = (m2 >> gb) & 0xffffull, e16 = (m3 >> gb) & 0xffffull;
		int cur = 0;
		while (cur < 16) {
			if (!((i16 >> cur) & 1ull)) break;
			if ((o16 >> cur) & 1ull) { const int run = __builtin_ctzll(~(b16 >> cur)); j += (uint32_t)run; cur += run > 0 ? run : 1; }
			else cur += 1 + (int)((e16 >> cur) & 3ull);
		}
		t += 0.0016914f * (float)(16 - (cur & 3));
		acc += x * 1e-3f + (float)j * 1e-6f;
	}
	float d2[3];
	ray_dir(c, px, py, d2); // the same expression again: equal bits unless something went wrong in between
	const bool dev = __float_as_uint(d[0]) != __float_as_uint(d2[0]) || __float_as_uint(d[1]) != __float_as_uint(d2[1]) || __float_as_uint(d[2]) != __float_as_uint(d2[2]);
	if (dev) atomicAdd(bad + (lane >> 4), 1u);
	if (ray < n_rays && g == 0) sink[ray] = acc + d[0] + d2[1];
}

int main(int argc, char** argv) {
	const bool beside = argc > 1 && !std::strcmp(argv[1], "beside");
	const int launches = argc > 2 ? std::atoi(argv[2]) : 400;
	hipStream_t s1, s2;
	CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	std::vector<_Float16> hw(4096);
	for (int i = 0; i < 4096; ++i) hw[i] = (_Float16)(((i * 37) % 101 - 50) * 0.01f);
	std::vector<Cam> hc(64);
	for (int v = 0; v < 64; ++v) for (int k = 0; k < 12; ++k) hc[v].m[k] = std::sin(0.37f * (float)(v * 12 + k)) * (k % 4 == 3 ? 1.5f : 1.0f);
	std::vector<uint32_t> hb(4096);
	for (int i = 0; i < 4096; ++i) hb[i] = (uint32_t)i * 2654435761u ^ 0x9e3779b9u;
	_Float16* dw; float *dout, *dsink; Cam* dc; uint32_t *dbits, *dbad;
	const uint32_t n_rays = 12800;
	CHECK(hipMalloc(&dw, 4096 * 2)); CHECK(hipMalloc(&dout, 512 * 256 * 4)); CHECK(hipMalloc(&dc, 64 * sizeof(Cam))); CHECK(hipMalloc(&dbits, 4096 * 4));
	CHECK(hipMalloc(&dbad, 16)); CHECK(hipMalloc(&dsink, n_rays * 4));
	CHECK(hipMemcpy(dw, hw.data(), 4096 * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, hc.data(), 64 * sizeof(Cam), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dbits, hb.data(), 4096 * 4, hipMemcpyHostToDevice));
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 72 * 4 * 2));
	uint32_t total[4] = {0, 0, 0, 0}, bad_launches = 0;
	for (int l = 0; l < launches; ++l) {
		CHECK(hipMemsetAsync(dbad, 0, 16, s2));
		if (beside) hipLaunchKernelGGL(k_mfma_hog, dim3(512), dim3(256), 64 * 72 * 4 * 2, s1, dw, dout, 3000);
		Uniforms U;
		for (int q = 0; q < 40; ++q) U.k[q] = 0.01f * (float)(q + 1);
		hipLaunchKernelGGL(k_victim, dim3((n_rays + 15) / 16), dim3(256), 0, s2, dc, n_rays, 64u, dbits, dbad, dsink, U);
		uint32_t hb4[4];
		CHECK(hipMemcpyAsync(hb4, dbad, 16, hipMemcpyDeviceToHost, s2));
		CHECK(hipStreamSynchronize(s2));
		CHECK(hipStreamSynchronize(s1));
		bool any = false;
		for (int q = 0; q < 4; ++q) { total[q] += hb4[q]; any = any || hb4[q]; }
		bad_launches += any;
	}
	std::printf("%s: %u of %d launches with deviating lanes; deviating lanes by row (lanes 0-15, 16-31, 32-47, 48-63): %u %u %u %u\n", beside ? "beside the MFMA kernel" : "idle GPU", bad_launches, launches,
	            total[0], total[1], total[2], total[3]);
	return 0;
}
