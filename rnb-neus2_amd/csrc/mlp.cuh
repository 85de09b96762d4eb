// mlp.cuh — the fused SDF+color MLP of NerfNetwork (nerf_network.h:97-452) on CDNA4 matrix cores.
//
// Data layout. One wavefront owns a tile of 64 samples (lane = sample in the per-sample phases).
// Activations are staged through per-wave LDS tiles, sample-major [64][stride] halfs, so that
//   B fragment of v_mfma_f32_16x16x32_f16: lane l reads X[16*nt + (l&15)][32*ks + 8*(l>>4) .. +7]  (one ds_read_b128)
//   A fragment: lane l reads W[16*mt + (l&15)][32*ks + 8*(l>>4) .. +7] from the weight image in LDS, which is the
//               reference's own row-major [out][in] layout (fully_fused_mlp.cu:786-819) with padded rows
//   D fragment: lane l, reg r = Y[out = 16*mt + 4*(l>>4) + r][sample = 16*nt + (l&15)] -> 4 consecutive outputs of
//               one sample -> one ds_write_b64 into the next layer's sample-major tile.
// Row strides of 40 (K=32) and 72 (K=64) halfs make every ds_read_b128 of a fragment conflict-free
// (16 rows x {80,144} B land on 16 distinct 16-byte slots of the 64-bank LDS).
// fp32 accumulation in the MFMA by default (the reference's WMMA accumulates in fp16, fully_fused_mlp.cu:68: the EMU_* forms below).
#pragma once
#include "common.cuh"

namespace rnb {

constexpr int S32 = 40;  // row stride (halfs) of K=32 matrices / 32-wide tiles
constexpr int S64 = 72;  // row stride (halfs) of K=64 matrices / 64-wide tiles
constexpr int TILE = 64; // samples per wavefront tile
constexpr int ACT_TILE_HALFS = TILE * S64; // one activation tile (any width up to 64)

// Weight image in LDS (offsets in halfs). Forward part first, training-only transposes after it.
constexpr int W_S0 = 0;                    // [64][S32] sdf W0
constexpr int W_S1 = W_S0 + 64 * S32;      // [16][S64] sdf W1
constexpr int W_S0T = W_S1 + 16 * S64;     // [32][S64] sdf W0^T
constexpr int W_C0 = W_S0T + 32 * S64;     // [64][S32] rgb W0, compact columns {0..15, 32..47}
constexpr int W_C1 = W_C0 + 64 * S32;      // [64][S64] rgb W1
constexpr int W_C2 = W_C1 + 64 * S64;      // [16][S64] rgb W2
constexpr int W_FWD_END = W_C2 + 16 * S64; // 14336 halfs
constexpr int W_C1T = W_FWD_END;           // [64][S64] rgb W1^T
constexpr int W_C0T = W_C1T + 64 * S64;    // [32][S64] rgb W0^T, compact rows
constexpr int W_S1T = W_C0T + 32 * S64;    // [64][S32] sdf W1^T (K = 16 padded to 32 with zeros)
constexpr int W_TRAIN_END = W_S1T + 64 * S32; // 23808 halfs

// Cooperative load of the weight image by the whole workgroup (once per kernel).
template <bool TRAIN>
__device__ inline void load_weights(half_t* __restrict__ w, const NetW& net, int tid, int nthreads) {
	for (int i = tid; i < 64 * 32; i += nthreads) { int o = i >> 5, k = i & 31; w[W_S0 + o * S32 + k] = net.sdf_w0[i]; w[W_S0T + k * S64 + o] = net.sdf_w0[i]; }
	for (int i = tid; i < 16 * 64; i += nthreads) { int o = i >> 6, k = i & 63; w[W_S1 + o * S64 + k] = net.sdf_w1[i]; }
	for (int i = tid; i < 64 * 32; i += nthreads) { // compact input index c: 0..15 -> column c, 16..31 -> column c + 16
		int o = i >> 5, c = i & 31;
		int col = c < 16 ? c : c + 16;
		half_t v = net.rgb_w0[o * 48 + col];
		w[W_C0 + o * S32 + c] = v;
		if (TRAIN) w[W_C0T + c * S64 + o] = v;
	}
	for (int i = tid; i < 64 * 64; i += nthreads) { int o = i >> 6, k = i & 63; half_t v = net.rgb_w1[i]; w[W_C1 + o * S64 + k] = v; if (TRAIN) w[W_C1T + k * S64 + o] = v; }
	for (int i = tid; i < 16 * 64; i += nthreads) { int o = i >> 6, k = i & 63; w[W_C2 + o * S64 + k] = net.rgb_w2[i]; }
	if (TRAIN) {
		for (int i = tid; i < 64 * 32; i += nthreads) { int j = i >> 5, k = i & 31; w[W_S1T + j * S32 + k] = k < 16 ? net.sdf_w1[k * 64 + j] : (half_t)0.f; }
	}
}

// Half accumulation (the product mode rnb_config::accumulate = RNB_ACCUM_HALF): the reference's tensor-core path accumulates in HALF -- wmma 16x16x16
// fragments of __half (fully_fused_mlp.cu:59-68, 198) -- where the default mode accumulates in fp32 (deviation D1). The matrix cores of
// gfx950 have no half accumulator, so the model of the CPU checker (oracle/rnb_oracle.cpp dot_h) is followed instead: products exact, the 16 products of one
// LOGICAL k-step (the reference's k index 16 q .. 16 q + 15) summed in fp32, the running accumulator rounded to half after every k-step. One k-step is one
// v_mfma_f32_16x16x16_f16 (lane (r16, hq) supplies k = 4 hq + j, j < 4) wherever the operands' K order allows it:
//   EMU_CHAINED  chained fragments, k <-> feature 16 (2 ks + (j >> 2)) + 4 hq + (j & 3) (below): elements j < 4 of a lane's 8 ARE the first k-step's operand, j >= 4 the second's
//   EMU_NATURAL  tiles in LDS in the reference's column order: the two halves of a 32-wide row are the two k-steps, read as two 8-byte operands (mfma_layer);
//                the same order in REGISTERS (k = 8 hq + j: the colour MLP's input rows in k_rgb_fwd_bwd) takes a 32-wide MFMA twice, each time with the other
//                step's lanes (hq >= 2 / hq < 2) of the weight operand zeroed, the accumulator rounded in between (split_ksteps)
//   k_fwd_bwd_sdf's input tiles are written in an order whose halves are the k-steps (kernels_net.cuh, fbs_logical_h) and read like EMU_NATURAL.
constexpr int EMU_OFF = 0, EMU_NATURAL = 1, EMU_CHAINED = 2;
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f4 round_acc_half(f4 a) { // pairs: v_cvt_pk_f16_f32 (gfx950) + two unpacking converts = 3 instructions per 2 elements instead of 4
	const h2 lo = __builtin_convertvector((f2v{a[0], a[1]}), h2), hi = __builtin_convertvector((f2v{a[2], a[3]}), h2);
	const f2v l = __builtin_convertvector(lo, f2v), h = __builtin_convertvector(hi, f2v);
	return f4{l[0], l[1], h[0], h[1]};
}
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
// The two halves of operand b by logical k-step (an operand is masked once and may then serve several MFMAs).
template <int ORD>
__device__ __forceinline__ void split_ksteps(const h8 b, const int hq, h8& lo, h8& hi) {
	const h8 zero = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
	lo = zero; hi = zero;
	if (ORD == EMU_NATURAL) { if (hq < 2) lo = b; else hi = b; }
	else if (ORD == EMU_CHAINED) {
#pragma unroll
		for (int j = 0; j < 4; ++j) { lo[j] = b[j]; hi[4 + j] = b[4 + j]; }
	}
}
__device__ __forceinline__ f4 mfma_emul16_split(const h8 a, const h8 lo, const h8 hi, f4 acc) {
	acc = round_acc_half(__builtin_amdgcn_mfma_f32_16x16x32_f16(a, lo, acc, 0, 0, 0));
	acc = round_acc_half(__builtin_amdgcn_mfma_f32_16x16x32_f16(a, hi, acc, 0, 0, 0));
	return acc;
}
// One 32-wide product as the reference's two k-steps: two K = 16 MFMAs, the accumulator rounded to half behind each.
__device__ __forceinline__ f4 mfma_emul16_k16(const h4 a0, const h4 a1, const h4 b0, const h4 b1, f4 acc) {
	acc = round_acc_half(__builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, acc, 0, 0, 0));
	acc = round_acc_half(__builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, acc, 0, 0, 0));
	return acc;
}
template <int ORD>
__device__ __forceinline__ f4 mfma_emul16(const h8 a, const h8 b, f4 acc, const int hq) {
	if (ORD == EMU_CHAINED) {
		const h4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
		const h4 b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
		return mfma_emul16_k16(a0, a1, b0, b1, acc);
	}
	h8 lo, hi;
	split_ksteps<ORD>(b, hq, lo, hi);
	return mfma_emul16_split(a, lo, hi, acc);
}

// acc[mt][nt] += W[16mt.., :] * X[16nt.., :]^T over K = 32*K_STEPS.
template <int M_TILES, int K_STEPS, int EMU = EMU_OFF>
__device__ __forceinline__ void mfma_layer(const half_t* __restrict__ W, const int w_stride, const half_t* __restrict__ X, const int x_stride, f4 (&acc)[M_TILES][4], const int lane) {
	const int r16 = lane & 15, hq = lane >> 4;
	static_assert(EMU == EMU_OFF || EMU == EMU_NATURAL, "operands in LDS: the reference's column order");
	if (EMU == EMU_NATURAL) { // columns 32 ks .. + 15 and + 16 .. + 31 are the two k-steps: lane (r16, hq) reads slots 4 hq .. + 3 of either half as a K = 16 operand (no masks)
		h4 b0[4][K_STEPS], b1[4][K_STEPS];
#pragma unroll
		for (int nt = 0; nt < 4; ++nt)
#pragma unroll
			for (int ks = 0; ks < K_STEPS; ++ks) {
				const half_t* xr = X + (16 * nt + r16) * x_stride + 32 * ks + 4 * hq;
				b0[nt][ks] = *reinterpret_cast<const h4*>(xr); b1[nt][ks] = *reinterpret_cast<const h4*>(xr + 16);
			}
#pragma unroll
		for (int mt = 0; mt < M_TILES; ++mt)
#pragma unroll
			for (int ks = 0; ks < K_STEPS; ++ks) {
				const half_t* wr = W + (16 * mt + r16) * w_stride + 32 * ks + 4 * hq;
				const h4 a0 = *reinterpret_cast<const h4*>(wr), a1 = *reinterpret_cast<const h4*>(wr + 16);
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma_emul16_k16(a0, a1, b0[nt][ks], b1[nt][ks], acc[mt][nt]);
			}
		return;
	}
	h8 b[4][K_STEPS];
#pragma unroll
	for (int nt = 0; nt < 4; ++nt)
#pragma unroll
		for (int ks = 0; ks < K_STEPS; ++ks) b[nt][ks] = *reinterpret_cast<const h8*>(X + (16 * nt + r16) * x_stride + 32 * ks + 8 * hq);
#pragma unroll
	for (int mt = 0; mt < M_TILES; ++mt) {
#pragma unroll
		for (int ks = 0; ks < K_STEPS; ++ks) {
			const h8 a = *reinterpret_cast<const h8*>(W + (16 * mt + r16) * w_stride + 32 * ks + 8 * hq);
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[nt][ks], acc[mt][nt], 0, 0, 0);
		}
	}
}

template <int M_TILES>
__device__ __forceinline__ void zero_acc(f4 (&acc)[M_TILES][4]) {
#pragma unroll
	for (int mt = 0; mt < M_TILES; ++mt)
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
}

// D fragments -> sample-major LDS tile Y[64][y_stride] at column col0. RELU applies the forward activation
// (warp_activation, common_device.h:69-115) and returns the relu' bit mask of this lane's 16*M_TILES elements.
template <int M_TILES, bool RELU>
__device__ __forceinline__ uint64_t store_acc(const f4 (&acc)[M_TILES][4], half_t* __restrict__ Y, const int y_stride, const int col0, const int lane) {
	const int r16 = lane & 15, hq = lane >> 4;
	uint64_t mask = 0;
#pragma unroll
	for (int mt = 0; mt < M_TILES; ++mt) {
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) {
			h4 v;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				float a = acc[mt][nt][r];
				if (RELU) {
					const bool on = a > 0.f;
					if (!on) a = 0.f;
					const half_t hv = f2h(a);
					// the backward transfer tests the stored half activation (common_device.h:182 ff.)
					if (h2f(hv) > 0.f) mask |= (1ull << ((mt * 4 + nt) * 4 + r));
					v[r] = hv;
				} else {
					v[r] = f2h(a);
				}
			}
			*reinterpret_cast<h4*>(Y + (16 * nt + r16) * y_stride + col0 + 16 * mt + 4 * hq) = v;
		}
	}
	return mask;
}

// D fragments gated by a relu' mask (warp_activation_backward) -> sample-major tile.
template <int M_TILES>
__device__ __forceinline__ void store_acc_masked(const f4 (&acc)[M_TILES][4], const uint64_t mask, half_t* __restrict__ Y, const int y_stride, const int col0, const int lane) {
	const int r16 = lane & 15, hq = lane >> 4;
#pragma unroll
	for (int mt = 0; mt < M_TILES; ++mt) {
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) {
			h4 v;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const bool on = (mask >> ((mt * 4 + nt) * 4 + r)) & 1ull;
				v[r] = f2h(on ? acc[mt][nt][r] : 0.f);
			}
			*reinterpret_cast<h4*>(Y + (16 * nt + r16) * y_stride + col0 + 16 * mt + 4 * hq) = v;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Register-chained layers (k_forward). The D fragments of a layer are, up to a permutation of the K index, exactly the
// B fragments the next layer needs: lane (r16, hq) holds outputs 16*mt + 4*hq + r of sample 16*nt + r16, and wants 8
// K-values of that same sample for K-step ks. Taking (mt = 2ks, r = 0..3) and (mt = 2ks+1, r = 0..3) gives
//   physical k = 32*ks + 8*hq + j   <->   logical feature 16*(2*ks + (j >> 2)) + 4*hq + (j & 3),
// and since K is a summation index it is enough to store the next layer's weights with their columns in that order
// (load_weights_chained). Hidden activations then never touch LDS: one 5 KB exchange tile per wavefront instead of three
// 9 KB tiles, which is what lets two workgroups share a CU.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int chain_logical(int p) { return 16 * (2 * (p >> 5) + ((p & 7) >> 2)) + 4 * ((p >> 3) & 3) + (p & 3); }
// K = 32 input of the colour MLP: [sdf_out (16, chained) | x y z grad 0.. (16, from the exchange tile)] in compact indexing
__host__ __device__ constexpr int chain_logical_c0(int p) { return (p & 7) < 4 ? 4 * (p >> 3) + (p & 3) : 16 + 4 * (p >> 3) + ((p & 7) - 4); }

// Weight image for the chained forward; same offsets as load_weights<false>, columns of every chained operand permuted.
__device__ inline void load_weights_chained(half_t* __restrict__ w, const NetW& net, int tid, int nthreads) {
	for (int i = tid; i < 64 * 32; i += nthreads) { int o = i >> 5, k = i & 31; w[W_S0 + o * S32 + k] = net.sdf_w0[i]; } // input comes from LDS: natural order
	for (int i = tid; i < 16 * 64; i += nthreads) { int o = i >> 6, p = i & 63; w[W_S1 + o * S64 + p] = net.sdf_w1[o * 64 + chain_logical(p)]; }
	for (int i = tid; i < 32 * 64; i += nthreads) { int k = i >> 6, p = i & 63; w[W_S0T + k * S64 + p] = net.sdf_w0[chain_logical(p) * 32 + k]; } // W0^T[k][hidden]
	for (int i = tid; i < 64 * 32; i += nthreads) {
		int o = i >> 5, p = i & 31;
		int c = chain_logical_c0(p);          // compact input index: 0..15 -> column c, 16..31 -> column c + 16
		w[W_C0 + o * S32 + p] = net.rgb_w0[o * 48 + (c < 16 ? c : c + 16)];
	}
	for (int i = tid; i < 64 * 64; i += nthreads) { int o = i >> 6, p = i & 63; w[W_C1 + o * S64 + p] = net.rgb_w1[o * 64 + chain_logical(p)]; }
	for (int i = tid; i < 16 * 64; i += nthreads) { int o = i >> 6, p = i & 63; w[W_C2 + o * S64 + p] = net.rgb_w2[o * 64 + chain_logical(p)]; }
}

// acc[mt][nt] += W[16mt.., :] * B over K = 32*K_STEPS, B fragments in registers.
template <int M_TILES, int K_STEPS, int EMU = EMU_OFF>
__device__ __forceinline__ void mfma_layer_regs(const half_t* __restrict__ W, const int w_stride, const h8 (&b)[4][K_STEPS], f4 (&acc)[M_TILES][4], const int lane) {
	const int r16 = lane & 15, hq = lane >> 4;
#pragma unroll
	for (int mt = 0; mt < M_TILES; ++mt) {
#pragma unroll
		for (int ks = 0; ks < K_STEPS; ++ks) {
			const h8 a = *reinterpret_cast<const h8*>(W + (16 * mt + r16) * w_stride + 32 * ks + 8 * hq);
			h8 a_lo = a, a_hi = a; // (the weight operand is the one shared by the four sample tiles: masked once; a slot's product is zero whichever factor is)
			if (EMU == EMU_NATURAL) split_ksteps<EMU>(a, hq, a_lo, a_hi);
#pragma unroll
			for (int nt = 0; nt < 4; ++nt)
				acc[mt][nt] = EMU == EMU_OFF ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[nt][ks], acc[mt][nt], 0, 0, 0)
				            : EMU == EMU_CHAINED ? mfma_emul16<EMU_CHAINED>(a, b[nt][ks], acc[mt][nt], hq) : round_acc_half(__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b[nt][ks], round_acc_half(__builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b[nt][ks], acc[mt][nt], 0, 0, 0)), 0, 0, 0));
		}
	}
}

// D fragments of a 64-wide layer -> B fragments of the next one (half, optional ReLU: warp_activation, common_device.h:69-115).
template <bool RELU>
__device__ __forceinline__ void chain_pack(const f4 (&acc)[4][4], h8 (&b)[4][2]) {
#pragma unroll
	for (int nt = 0; nt < 4; ++nt)
#pragma unroll
		for (int ks = 0; ks < 2; ++ks)
#pragma unroll
			for (int j = 0; j < 8; j += 2) { // convert first, ReLU on the packed halfs (v_cvt_pk_f16_f32 + v_pk_max_f16): same values, 2 instead of 5 instructions per pair
				h2 p = {f2h(acc[2 * ks + (j >> 2)][nt][j & 3]), f2h(acc[2 * ks + (j >> 2)][nt][(j & 3) + 1])};
				if (RELU) p = __builtin_elementwise_max(p, h2{(half_t)0.f, (half_t)0.f});
				b[nt][ks][j] = p[0];
				b[nt][ks][j + 1] = p[1];
			}
}

// The same for one K-step of the next layer: acc2[h] = the D fragments of hidden-unit tiles 2 ks + h.
template <bool RELU>
__device__ __forceinline__ void chain_pack_ks(const f4 (&acc2)[2][4], h8 (&b)[4][2], const int ks) {
#pragma unroll
	for (int nt = 0; nt < 4; ++nt)
#pragma unroll
		for (int j = 0; j < 8; j += 2) {
			h2 p = {f2h(acc2[j >> 2][nt][j & 3]), f2h(acc2[j >> 2][nt][(j & 3) + 1])};
			if (RELU) p = __builtin_elementwise_max(p, h2{(half_t)0.f, (half_t)0.f});
			b[nt][ks][j] = p[0];
			b[nt][ks][j + 1] = p[1];
		}
}

// Also writes the fragments feature-major to global memory: dst[feature][n_total] at sample column s0 + ...
// (operands of the weight-gradient GEMMs, whose K dimension is the sample index).
template <int M_TILES>
__device__ __forceinline__ void store_tile_feature_major(const half_t* __restrict__ Y, const int y_stride, const int width, half_t* __restrict__ dst, const uint32_t n_total, const uint32_t s0, const int lane) {
	// lane = sample: read the row from LDS, scatter one half per feature row (64 lanes -> 128 contiguous bytes per feature)
	const half_t* row = Y + lane * y_stride;
	for (int f = 0; f < width; f += 8) {
		const h8 v = *reinterpret_cast<const h8*>(row + f);
#pragma unroll
		for (int j = 0; j < 8; ++j) dst[(size_t)(f + j) * n_total + s0 + lane] = v[j];
	}
}

} // namespace rnb
