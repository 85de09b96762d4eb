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

#define RNB_CHAIN1_STEP(RM, BM) \
	"v_mul_f32_dpp %[w], %[Ta], %[al] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32_dpp %[Ta], %[Ta], %[om] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ws], %[ws], %[w] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32 %[c], %[w], %[sh]\n\t" \
	"v_add_f32_dpp %[r0], %[r0], %[c] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ek], %[ek], %[ekt] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t"
#define RNB_CHAIN1_GROUP(RM, BM) \
	asm volatile("s_nop 1\n\t" RNB_CHAIN1_STEP(RM, BM) RNB_CHAIN1_STEP(RM, BM) RNB_CHAIN1_STEP(RM, BM) RNB_CHAIN1_STEP(RM, BM) \
	             : [w] "+v"(s.w), [Ta] "+v"(s.T), [ws] "+v"(s.ws), [c] "+v"(c), [r0] "+v"(s.rgb[0]), [ek] "+v"(s.ek) \
	             : [al] "v"(alpha), [om] "v"(one_minus), [sh] "v"(shading), [ekt] "v"(ekterm))

#define RNB_CHAIN4_STEP(RM, BM) \
	"v_mul_f32_dpp %[w], %[Ta], %[al] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32_dpp %[Ta], %[Ta], %[om] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ws], %[ws], %[w] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_mul_f32 %[c0], %[w], %[a0]\n\t" \
	"v_mul_f32 %[c1], %[w], %[a1]\n\t" \
	"v_mul_f32 %[c2], %[w], %[a2]\n\t" \
	"v_mul_f32 %[c3], %[w], %[a3]\n\t" \
	"v_mul_f32 %[c0], %[c0], %[sh]\n\t" \
	"v_mul_f32 %[c1], %[c1], %[sh]\n\t" \
	"v_mul_f32 %[c2], %[c2], %[sh]\n\t" \
	"v_mul_f32 %[c3], %[c3], %[sh]\n\t" \
	"v_add_f32_dpp %[r0], %[r0], %[c0] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[r1], %[r1], %[c1] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[r2], %[r2], %[c2] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[r3], %[r3], %[c3] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t" \
	"v_add_f32_dpp %[ek], %[ek], %[ekt] wave_shr:1 row_mask:" RM " bank_mask:" BM "\n\t"
#define RNB_CHAIN4_GROUP(RM, BM) \
	asm volatile("s_nop 1\n\t" RNB_CHAIN4_STEP(RM, BM) RNB_CHAIN4_STEP(RM, BM) RNB_CHAIN4_STEP(RM, BM) RNB_CHAIN4_STEP(RM, BM) \
	             : [w] "+v"(s.w), [Ta] "+v"(s.T), [ws] "+v"(s.ws), [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), \
	               [r0] "+v"(s.rgb[0]), [r1] "+v"(s.rgb[1]), [r2] "+v"(s.rgb[2]), [r3] "+v"(s.rgb[3]), [ek] "+v"(s.ek) \
	             : [al] "v"(alpha), [om] "v"(one_minus), [sh] "v"(shading), [ekt] "v"(ekterm), [a0] "v"(albedo[0]), [a1] "v"(albedo[1]), [a2] "v"(albedo[2]), [a3] "v"(albedo[3]))

// Per-lane state after the lane's sample: T (transmittance), w (the sample's weight), ws (weight sum), rgb, ek.
struct ChainState { float T, w, ws, rgb[4], ek; };

#define RNB_CHAIN_GROUPS(G) \
	G("0x1", "0x1"); if (cnt > 4) { G("0x1", "0x2"); } if (cnt > 8) { G("0x1", "0x4"); } if (cnt > 12) { G("0x1", "0x8"); } \
	if (cnt > 16) { G("0x2", "0x1"); } if (cnt > 20) { G("0x2", "0x2"); } if (cnt > 24) { G("0x2", "0x4"); } if (cnt > 28) { G("0x2", "0x8"); } \
	if (cnt > 32) { G("0x4", "0x1"); } if (cnt > 36) { G("0x4", "0x2"); } if (cnt > 40) { G("0x4", "0x4"); } if (cnt > 44) { G("0x4", "0x8"); } \
	if (cnt > 48) { G("0x8", "0x1"); } if (cnt > 52) { G("0x8", "0x2"); } if (cnt > 56) { G("0x8", "0x4"); } if (cnt > 60) { G("0x8", "0x8"); }

// The recurrence over samples [0, cnt) of the wavefront, lane q holding sample q (alpha = 0, ekterm = 0 beyond cnt), from the
// running values `in` (wave-uniform). ALL 64 lanes must be active. NO_ALBEDO: albedo = (1, 1, 1, 0), only rgb[0] is formed
// (weight * 1.f * shading = weight * shading exactly). Lanes >= cnt of a started group carry the last sample's values on.
template <bool NO_ALBEDO>
__device__ __forceinline__ ChainState replay_chain(const int cnt_any_lane, const float alpha, const float shading, const float (&albedo)[4], const float ekterm,
                                                   const float T_in, const float ws_in, const float (&rgb_in)[4], const float ek_in) {
	const int cnt = __builtin_amdgcn_readfirstlane(cnt_any_lane); // scalar branches around the groups: the DPP blocks need every lane
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
		RNB_CHAIN_GROUPS(RNB_CHAIN1_GROUP)
	} else {
		float c0 = s.w * albedo[0] * shading, c1 = s.w * albedo[1] * shading, c2 = s.w * albedo[2] * shading, c3 = s.w * albedo[3] * shading;
		s.rgb[0] = rgb_in[0] + c0; s.rgb[1] = rgb_in[1] + c1; s.rgb[2] = rgb_in[2] + c2; s.rgb[3] = rgb_in[3] + c3;
		RNB_CHAIN_GROUPS(RNB_CHAIN4_GROUP)
	}
	return s;
}

} // namespace rnb
