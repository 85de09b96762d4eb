// chain.cuh — the sequential compositing recurrence of one ray (testbed_nerf.cu:1653-1690, 1893-1930), one sample per lane.
//
//   weight_q = alpha_q * T_(q-1)        T_q = T_(q-1) * (1 - alpha_q)        wsum_q = wsum_(q-1) + weight_q
//   rgb_q[k] = rgb_(q-1)[k] + (weight_q * albedo_q[k]) * shading_q           ek_q = ek_(q-1) + ekterm_q
//
// The reference runs it in a per-thread loop; the order of the fp32 operations is part of the result (it decides where the
// transmittance falls below 1e-4, i.e. which samples a ray keeps). A wavefront that owns one ray used to replay it from lane
// broadcasts, ~20 wave instructions per sample, all 64 lanes computing the same numbers: half of k_loss_pass2 (measured by
// running it twice: +21 us of 45). Here every lane keeps ITS sample's values and takes the running values of the lane before it
// through a DPP wavefront shift (wave_shr:1), so a sample costs 6 instructions (15 with four colour channels). DPP writes can
// be masked per group of 4 lanes only (row_mask/bank_mask), so a group's instructions are issued 4 times: the k-th pass makes
// the k-th lane of the group final, lanes before it recompute the same value from the same inputs, lanes after it are
// overwritten in their own pass. The operations and their order per sample are the loop's, so the results are bit-identical.
// Every DPP source register was written at least 5 instructions earlier (the hardware needs 2); an s_nop opens each block
// because the compiler does not look into it.
#pragma once
#include "common.cuh"

namespace rnb {

#define RNB_CHAIN1_STEP(SH, RM, BM) \
	"v_mul_f32_dpp %[w], %[Ta], %[al] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32_dpp %[Ta], %[Ta], %[om] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ws], %[ws], %[w] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32 %[c], %[w], %[sh]\n\t" \
	"v_add_f32_dpp %[r0], %[r0], %[c] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ek], %[ek], %[ekt] " SH " row_mask:" RM " bank_mask:" BM "\n\t"
#define RNB_CHAIN1_GROUP(SH, RM, BM) \
	asm volatile("s_nop 1\n\t" RNB_CHAIN1_STEP(SH, RM, BM) RNB_CHAIN1_STEP(SH, RM, BM) RNB_CHAIN1_STEP(SH, RM, BM) RNB_CHAIN1_STEP(SH, RM, BM) \
	             : [w] "+v"(s.w), [Ta] "+v"(s.T), [ws] "+v"(s.ws), [c] "+v"(c), [r0] "+v"(s.rgb[0]), [ek] "+v"(s.ek) \
	             : [al] "v"(alpha), [om] "v"(one_minus), [sh] "v"(shading), [ekt] "v"(ekterm))

#define RNB_CHAIN4_STEP(SH, RM, BM) \
	"v_mul_f32_dpp %[w], %[Ta], %[al] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32_dpp %[Ta], %[Ta], %[om] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ws], %[ws], %[w] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32 %[c0], %[w], %[a0]\n\t" \
	"v_mul_f32 %[c1], %[w], %[a1]\n\t" \
	"v_mul_f32 %[c2], %[w], %[a2]\n\t" \
	"v_mul_f32 %[c3], %[w], %[a3]\n\t" \
	"v_mul_f32 %[c0], %[c0], %[sh]\n\t" \
	"v_mul_f32 %[c1], %[c1], %[sh]\n\t" \
	"v_mul_f32 %[c2], %[c2], %[sh]\n\t" \
	"v_mul_f32 %[c3], %[c3], %[sh]\n\t" \
	"v_add_f32_dpp %[r0], %[r0], %[c0] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[r1], %[r1], %[c1] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[r2], %[r2], %[c2] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[r3], %[r3], %[c3] " SH " row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ek], %[ek], %[ekt] " SH " row_mask:" RM " bank_mask:" BM "\n\t"
#define RNB_CHAIN4_GROUP(SH, RM, BM) \
	asm volatile("s_nop 1\n\t" RNB_CHAIN4_STEP(SH, RM, BM) RNB_CHAIN4_STEP(SH, RM, BM) RNB_CHAIN4_STEP(SH, RM, BM) RNB_CHAIN4_STEP(SH, RM, BM) \
	             : [w] "+v"(s.w), [Ta] "+v"(s.T), [ws] "+v"(s.ws), [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), \
	               [r0] "+v"(s.rgb[0]), [r1] "+v"(s.rgb[1]), [r2] "+v"(s.rgb[2]), [r3] "+v"(s.rgb[3]), [ek] "+v"(s.ek) \
	             : [al] "v"(alpha), [om] "v"(one_minus), [sh] "v"(shading), [ekt] "v"(ekterm), [a0] "v"(albedo[0]), [a1] "v"(albedo[1]), [a2] "v"(albedo[2]), [a3] "v"(albedo[3]))

// Per-lane state after the lane's sample: T (transmittance), w (the sample's weight), ws (weight sum), rgb, ek.
struct ChainState { float T, w, ws, rgb[4], ek; };

#define RNB_CHAIN_GROUPS(G) \
	G("wave_shr:1", "0x1", "0x1"); if (cnt > 4) { G("wave_shr:1", "0x1", "0x2"); } if (cnt > 8) { G("wave_shr:1", "0x1", "0x4"); } if (cnt > 12) { G("wave_shr:1", "0x1", "0x8"); } \
	if (cnt > 16) { G("wave_shr:1", "0x2", "0x1"); } if (cnt > 20) { G("wave_shr:1", "0x2", "0x2"); } if (cnt > 24) { G("wave_shr:1", "0x2", "0x4"); } if (cnt > 28) { G("wave_shr:1", "0x2", "0x8"); } \
	if (cnt > 32) { G("wave_shr:1", "0x4", "0x1"); } if (cnt > 36) { G("wave_shr:1", "0x4", "0x2"); } if (cnt > 40) { G("wave_shr:1", "0x4", "0x4"); } if (cnt > 44) { G("wave_shr:1", "0x4", "0x8"); } \
	if (cnt > 48) { G("wave_shr:1", "0x8", "0x1"); } if (cnt > 52) { G("wave_shr:1", "0x8", "0x2"); } if (cnt > 56) { G("wave_shr:1", "0x8", "0x4"); } if (cnt > 60) { G("wave_shr:1", "0x8", "0x8"); }
// Four rays per wavefront, one per row of 16 lanes: the shift stays inside the row (row_shr:1, lane 0 of a row keeps its own start
// values), all four rows are written at once, and a chunk of 16 samples takes 4 groups.
#define RNB_CHAIN_GROUPS_ROW(G) \
	G("row_shr:1", "0xf", "0x1"); if (cnt > 4) { G("row_shr:1", "0xf", "0x2"); } if (cnt > 8) { G("row_shr:1", "0xf", "0x4"); } if (cnt > 12) { G("row_shr:1", "0xf", "0x8"); }

// The recurrence over the samples of a group of LR lanes (LR = 64: the wavefront = one ray; LR = 16: a row = one ray, four rays
// per wavefront), lane q of the group holding sample q (alpha = 0, ekterm = 0 beyond the group's count), from the running values
// `in` (uniform inside the group). cnt_any_lane: LR = 64: the count; LR = 16: the group's own count (the groups are issued for
// the largest of the four). ALL 64 lanes must be active. NO_ALBEDO: albedo = (1, 1, 1, 0), only rgb[0] is formed (weight * 1.f *
// shading = weight * shading exactly). Lanes beyond the count of a started group carry the last sample's values on.
template <bool NO_ALBEDO, int LR = 64>
__device__ __forceinline__ ChainState replay_chain(const int cnt_any_lane, const float alpha, const float shading, const float (&albedo)[4], const float ekterm,
                                                   const float T_in, const float ws_in, const float (&rgb_in)[4], const float ek_in) {
	static_assert(LR == 64 || LR == 16, "lanes per ray");
	int cnt_all = cnt_any_lane;
	if (LR == 16) { cnt_all = max(cnt_all, __shfl_xor(cnt_all, 16, 64)); cnt_all = max(cnt_all, __shfl_xor(cnt_all, 32, 64)); }
	const int cnt = __builtin_amdgcn_readfirstlane(cnt_all); // scalar branches around the groups: the DPP blocks need every lane
	const float one_minus = 1.f - alpha;
	ChainState s;
	s.w = alpha * T_in;
	s.T = T_in * one_minus;
	s.ws = ws_in + s.w;
	s.ek = ek_in + ekterm;
	if (NO_ALBEDO) {
		float c = s.w * shading;
		s.rgb[0] = rgb_in[0] + c;
		s.rgb[1] = s.rgb[2] = s.rgb[3] = 0.f;
		if (LR == 64) { RNB_CHAIN_GROUPS(RNB_CHAIN1_GROUP) } else { RNB_CHAIN_GROUPS_ROW(RNB_CHAIN1_GROUP) }
	} else {
		float c0 = s.w * albedo[0] * shading, c1 = s.w * albedo[1] * shading, c2 = s.w * albedo[2] * shading, c3 = s.w * albedo[3] * shading;
		s.rgb[0] = rgb_in[0] + c0; s.rgb[1] = rgb_in[1] + c1; s.rgb[2] = rgb_in[2] + c2; s.rgb[3] = rgb_in[3] + c3;
		if (LR == 64) { RNB_CHAIN_GROUPS(RNB_CHAIN4_GROUP) } else { RNB_CHAIN_GROUPS_ROW(RNB_CHAIN4_GROUP) }
	}
	return s;
}

} // namespace rnb
