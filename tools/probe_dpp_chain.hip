// Checks rnb-neus2_amd/csrc/chain.cuh (the compositing recurrence through DPP wavefront shifts) against the plain sequential
// loop, bit for bit, over random inputs and every count 1..64, and times both.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I rnb-neus2_amd/csrc tools/probe_dpp_chain.hip -o tools/probe_dpp_chain && tools/probe_dpp_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include "chain.cuh"
using namespace rnb;

struct In { float alpha, shading, albedo[4], ekterm; };
struct Out { float T, w, ws, rgb[4], ek; };

template <bool NO_ALBEDO>
__global__ void k_chain(const In* in, const int* cnts, Out* out, int reps) {
	const int ray = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
	const int cnt = cnts[ray];
	In v = in[ray * 64 + lane];
	if (lane >= cnt) { v.alpha = 0.f; v.ekterm = 0.f; v.shading = 0.f; }
	const float rgb_in[4] = {0.125f, 0.25f, 0.375f, 0.0625f};
	ChainState s;
	for (int r = 0; r < reps; ++r) s = replay_chain<NO_ALBEDO>(cnt, v.alpha, v.shading, v.albedo, v.ekterm, 0.875f, 0.03125f, rgb_in, 0.5f);
	Out o; o.T = s.T; o.w = s.w; o.ws = s.ws; o.ek = s.ek;
	for (int k = 0; k < 4; ++k) o.rgb[k] = s.rgb[k];
	out[ray * 64 + lane] = o;
}

template <bool NO_ALBEDO>
void host_chain(const In* in, int cnt, Out* out) {
	float T = 0.875f, ws = 0.03125f, rgb[4] = {0.125f, 0.25f, 0.375f, 0.0625f}, ek = 0.5f;
	for (int q = 0; q < cnt; ++q) {
		const float al = in[q].alpha, om = 1.f - al;
		volatile float weight = al * T;
		if (NO_ALBEDO) { volatile float c = weight * in[q].shading; rgb[0] = rgb[0] + c; }
		else for (int k = 0; k < 4; ++k) { volatile float t = weight * in[q].albedo[k]; volatile float c = t * in[q].shading; rgb[k] = rgb[k] + c; }
		ws = ws + weight; T = T * om; ek = ek + in[q].ekterm;
		out[q].T = T; out[q].w = weight; out[q].ws = ws; out[q].ek = ek;
		for (int k = 0; k < 4; ++k) out[q].rgb[k] = rgb[k];
	}
}

// four rays per wavefront (LR = 16): ray = 4 * wave + row, counts 0..16 per row
template <bool NO_ALBEDO>
__global__ void k_chain_row(const In* in, const int* cnts, Out* out) {
	const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64, lane = threadIdx.x & 63;
	const int ray = wave * 4 + (lane >> 4), q = lane & 15;
	const int cnt = cnts[ray] % 17;
	In v = in[ray * 64 + q];
	if (q >= cnt) { v.alpha = 0.f; v.ekterm = 0.f; v.shading = 0.f; }
	const float rgb_in[4] = {0.125f, 0.25f, 0.375f, 0.0625f};
	const ChainState s = replay_chain<NO_ALBEDO, 16>(cnt, v.alpha, v.shading, v.albedo, v.ekterm, 0.875f, 0.03125f, rgb_in, 0.5f);
	Out o; o.T = s.T; o.w = s.w; o.ws = s.ws; o.ek = s.ek;
	for (int k = 0; k < 4; ++k) o.rgb[k] = s.rgb[k];
	out[ray * 64 + q] = o;
}

int main() {
	const int n_rays = 64 * 64;
	std::vector<In> in(n_rays * 64);
	std::vector<int> cnts(n_rays);
	std::mt19937 rng(5);
	std::uniform_real_distribution<float> u(0.f, 1.f);
	for (auto& v : in) { v.alpha = u(rng) < 0.2f ? 0.f : u(rng) * u(rng); v.shading = u(rng) * 2 - 0.5f; for (float& a : v.albedo) a = u(rng); v.ekterm = u(rng) * 0.1f; }
	for (int r = 0; r < n_rays; ++r) cnts[r] = 1 + r % 64;
	In* d_in; int* d_c; Out* d_out;
	hipMalloc(&d_in, in.size() * sizeof(In)); hipMalloc(&d_c, n_rays * 4); hipMalloc(&d_out, in.size() * sizeof(Out));
	hipMemcpy(d_in, in.data(), in.size() * sizeof(In), hipMemcpyHostToDevice); hipMemcpy(d_c, cnts.data(), n_rays * 4, hipMemcpyHostToDevice);
	std::vector<Out> got(in.size()), want(64);
	int bad = 0;
	for (int mode = 0; mode < 2; ++mode) {
		if (mode == 0) k_chain<true><<<n_rays / 4, 256>>>(d_in, d_c, d_out, 1); else k_chain<false><<<n_rays / 4, 256>>>(d_in, d_c, d_out, 1);
		hipDeviceSynchronize();
		hipMemcpy(got.data(), d_out, got.size() * sizeof(Out), hipMemcpyDeviceToHost);
		for (int r = 0; r < n_rays; ++r) {
			if (mode == 0) host_chain<true>(&in[r * 64], cnts[r], want.data()); else host_chain<false>(&in[r * 64], cnts[r], want.data());
			for (int q = 0; q < cnts[r]; ++q) {
				const Out &a = got[r * 64 + q], &b = want[q];
				bool same = !memcmp(&a.T, &b.T, 4) && !memcmp(&a.w, &b.w, 4) && !memcmp(&a.ws, &b.ws, 4) && !memcmp(&a.ek, &b.ek, 4) && !memcmp(&a.rgb[0], &b.rgb[0], 4);
				if (mode == 1) same = same && !memcmp(a.rgb, b.rgb, 16);
				if (!same && bad++ < 10) printf("mode %d ray %d cnt %d lane %d: T %g/%g w %g/%g ws %g/%g rgb0 %g/%g ek %g/%g\n", mode, r, cnts[r], q, a.T, b.T, a.w, b.w, a.ws, b.ws, a.rgb[0], b.rgb[0], a.ek, b.ek);
			}
		}
		printf("mode %s: %d mismatching lanes so far\n", mode == 0 ? "no-albedo" : "albedo", bad);
	}
	for (int mode = 0; mode < 2; ++mode) {
		if (mode == 0) k_chain_row<true><<<n_rays / 16, 256>>>(d_in, d_c, d_out); else k_chain_row<false><<<n_rays / 16, 256>>>(d_in, d_c, d_out);
		hipDeviceSynchronize();
		hipMemcpy(got.data(), d_out, got.size() * sizeof(Out), hipMemcpyDeviceToHost);
		for (int r = 0; r < n_rays; ++r) {
			const int cnt = cnts[r] % 17;
			if (mode == 0) host_chain<true>(&in[r * 64], cnt, want.data()); else host_chain<false>(&in[r * 64], cnt, want.data());
			for (int q = 0; q < cnt; ++q) {
				const Out &a = got[r * 64 + q], &b = want[q];
				bool same = !memcmp(&a.T, &b.T, 4) && !memcmp(&a.w, &b.w, 4) && !memcmp(&a.ws, &b.ws, 4) && !memcmp(&a.ek, &b.ek, 4) && !memcmp(&a.rgb[0], &b.rgb[0], 4);
				if (mode == 1) same = same && !memcmp(a.rgb, b.rgb, 16);
				if (!same && bad++ < 20) printf("row mode %d ray %d cnt %d lane %d: T %g/%g w %g/%g ws %g/%g rgb0 %g/%g ek %g/%g\n", mode, r, cnt, q, a.T, b.T, a.w, b.w, a.ws, b.ws, a.rgb[0], b.rgb[0], a.ek, b.ek);
			}
		}
		printf("row mode %s (4 rays per wavefront): %d mismatching lanes so far\n", mode == 0 ? "no-albedo" : "albedo", bad);
	}
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int mode = 0; mode < 2; ++mode) {
		hipEventRecord(e0);
		if (mode == 0) k_chain<true><<<n_rays / 4, 256>>>(d_in, d_c, d_out, 64); else k_chain<false><<<n_rays / 4, 256>>>(d_in, d_c, d_out, 64);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("mode %d: %d rays x 64 chains (counts 1..64) %.3f ms -> %.1f ns per 64-lane chain per SIMD-slot\n", mode, n_rays, ms, ms * 1e6 / (n_rays * 64.0) * 1024);
	}
	return bad != 0;
}
