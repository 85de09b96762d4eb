// kernels_net.cuh — network kernels of the hot path:
//   k_point_query_chained   NerfNetwork::sdf / ::density (+ splat, K2+K3)    nerf_network.h:454-537, testbed_nerf.cu:616-635
//   k_forward_chained       NerfNetwork::forward_impl, inference flavour (K7) nerf_network.h:97-253
//   k_fwd_bwd       forward + backward data path on the compacted batch (K10+K11)  nerf_network.h:97-452
//   k_dw / k_dw_finish   weight-gradient GEMMs (K = samples)            fully_fused_mlp.cu:960-1014, 1120-1131
//   k_grid_scatter  hash-grid gradient scatter, first + second order     grid.h:366-495, 556-683
//   k_adam_ema      ExponentialDecay -> Adam -> EMA                      adam.h:52-202, ema.h:63-78
#pragma once
#include "mlp.cuh"

namespace rnb {

constexpr int WG = 256;          // 4 wavefronts per workgroup
constexpr int WAVES_PER_WG = 4;

// ---------------------------------------------------------------------------------------------
// per-lane pieces
// ---------------------------------------------------------------------------------------------

// [x - 0.5 | 28 features | 0] as halfs into row `lane` of a 32-wide tile (nerf_network.h:149-155).
__device__ __forceinline__ void write_sdf_in_row(half_t* __restrict__ tile, const int lane, const float x, const float y, const float z, const half_t (&feat)[28]) {
	half_t row[32];
	row[0] = f2h(x) - (half_t)0.5f; // fill_positions_view_with_fixed_offset: half arithmetic (common_operation.cuh:187-199)
	row[1] = f2h(y) - (half_t)0.5f;
	row[2] = f2h(z) - (half_t)0.5f;
#pragma unroll
	for (int k = 0; k < 28; ++k) row[3 + k] = feat[k];
	row[31] = (half_t)0.f;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		h8 v;
#pragma unroll
		for (int j = 0; j < 8; ++j) v[j] = row[q * 8 + j];
		*reinterpret_cast<h8*>(tile + lane * S32 + q * 8) = v;
	}
}

template <bool GRAD>
__device__ __forceinline__ void encode_all_lm(const LevelMeta* __restrict__ lm, const uint32_t n_levels, const uint32_t valid_level, const uint32_t* __restrict__ grid,
                                              const float x, const float y, const float z, half_t (&feat)[28], float (&dydx)[GRAD ? 28 : 1][3]) {
#pragma unroll
	for (uint32_t level = 0; level < 14; ++level) {
		half_t f0 = (half_t)0.f, f1 = (half_t)0.f;
		float d0[3] = {0.f, 0.f, 0.f}, d1[3] = {0.f, 0.f, 0.f};
		if (level < n_levels && level <= valid_level) {
			encode_level_lm<GRAD>(lm, grid, level, x, y, z, f0, f1, d0, d1);
		}
		feat[level * 2 + 0] = f0;
		feat[level * 2 + 1] = f1;
		if (GRAD) {
#pragma unroll
			for (int d = 0; d < 3; ++d) { dydx[GRAD ? level * 2 + 0 : 0][d] = d0[d]; dydx[GRAD ? level * 2 + 1 : 0][d] = d1[d]; }
		}
	}
}

// The same encode with the gathers of DEPTH levels in flight (common.cuh: level_issue / level_consume). Levels [0, n_live) are encoded, the others give zeros
// (grid.h:192-210); n_live is wave-uniform. Fully unrolled: the in-flight values live in registers v[l % DEPTH].
template <bool GRAD, int DEPTH, int ND = 0>
__device__ __forceinline__ void encode_all_pipelined(const LevelMeta* __restrict__ lm, const uint32_t n_levels, const uint32_t valid_level, const uint32_t* __restrict__ grid,
                                                     const float x, const float y, const float z, half_t (&feat)[28], float (&dydx)[GRAD ? 28 : 1][3]) {
	const uint32_t n_live = min(n_levels, valid_level + 1u);
	uint32_t v[DEPTH][8];
	// No branch anywhere (a branch ends the scheduling region and with it the overlap of the levels' gathers): a level that is not live gathers from the last live
	// level's table instead (its values are dropped below) -- only before training step 660, when fewer than 14 levels are live.
	const uint32_t last = n_live ? n_live - 1u : 0u;
#pragma unroll
	// ND: the first ND levels are dense (host-checked): their x-pairs travel in one gather (level_issue<true>); a slot whose level is not live yet gathers from a level in FRONT of it, which is dense too
	for (uint32_t l = 0; l < (uint32_t)DEPTH; ++l) { if (l < (uint32_t)ND) level_issue<true>(lm, grid, min(l, last), x, y, z, v[l]); else level_issue<false>(lm, grid, min(l, last), x, y, z, v[l]); }
#pragma unroll
	for (uint32_t level = 0; level < 14; ++level) {
		half_t f0, f1;
		float d0[3], d1[3];
		const bool live = level < n_live;
		level_consume<GRAD>(lm, min(level, last), x, y, z, v[level % DEPTH], f0, f1, d0, d1);
		if (level + DEPTH < 14) { if (level + DEPTH < (uint32_t)ND) level_issue<true>(lm, grid, min(level + (uint32_t)DEPTH, last), x, y, z, v[level % DEPTH]); else level_issue<false>(lm, grid, min(level + (uint32_t)DEPTH, last), x, y, z, v[level % DEPTH]); }
		feat[level * 2 + 0] = live ? f0 : (half_t)0.f;
		feat[level * 2 + 1] = live ? f1 : (half_t)0.f;
		if (GRAD) {
#pragma unroll
			for (int d = 0; d < 3; ++d) { dydx[GRAD ? level * 2 + 0 : 0][d] = live ? d0[d] : 0.f; dydx[GRAD ? level * 2 + 1 : 0][d] = live ? d1[d] : 0.f; }
		}
	}
}

// sdf_to_density_variance_buffer (common_operation.cuh:311-328): half arithmetic throughout.
__device__ __forceinline__ half_t sdf_to_density(half_t sdf, half_t variance) {
	const half_t s = f2h(expf(h2f(variance * (half_t)10.0f)));
	const half_t sig = f2h(1.0f / (1.0f + expf(-h2f(sdf * s))));
	return (s * sig) * ((half_t)1.0f - sig);
}

// ---------------------------------------------------------------------------------------------
// K2 (+K3): point query
// ---------------------------------------------------------------------------------------------
struct PointArgs {
	const float* xyz;          // [n][3]
	uint32_t n;
	half_t* out;               // [n] or null
	const uint32_t* splat_idx; // [n] or null: atomicMax of the density into grid_tmp[idx]
	float* grid_tmp;
	int want_density;          // 0: sdf + bias, 1: density
	float sdf_bias;
	const uint32_t* range;     // optional (device): evaluate points range[0] .. range[1] - 1 of xyz / splat_idx instead of 0 .. n - 1 (k_shard_range; n bounds the launch)
	uint32_t xcd;              // (round 6; RNB_POINT_XCD=0 for the A/B) workgroup b (XCD b % 8) walks the b % 8-th eighth of the tiles: cell-ordered points that are neighbours in space share one L2
};

// ---------------------------------------------------------------------------------------------
// K7: inference-flavoured forward
// ---------------------------------------------------------------------------------------------
struct FwdArgs {
	const float* coords;       // [n][7]
	const uint32_t* n_ptr;     // device-side sample count (null -> n_max)
	uint32_t n_max;
	half_t* out;               // [n][16]
	float sdf_bias;
	const uint32_t* idx;       // optional: evaluate the samples idx[0 .. n) (slots into coords / out) instead of 0 .. n
	const half_t* wimg;        // optional: the LDS weight image (load_weights_chained layout) prepared once per step by k_prepare_weight_images
	half_t* cin_out;           // optional [slot][32]: the colour MLP's input row of every evaluated sample, [sdf_out 16 | x y z | grad sdf 3 | 0 x 10] (the compact
	                           // column order of W_C0), for k_rgb_fwd_bwd: the training pass does not evaluate the SDF MLP a second time to obtain it
};

// K7, register-chained flavour (mlp.cuh): per wavefront one 32-wide exchange tile X (sdf_in rows -> d sdf / d in rows -> r
// rows), 16 bytes per sample of colour-MLP side inputs (Y) and the raw sdf (Z); 6.3 KB instead of 27.6 KB, so two
// workgroups fit a CU and the encode of one hides behind the MFMA / LDS phases of the other.
// Every workgroup starts by building its LDS weight image; element-wise with permuted indices that is ~50 dependent 2-byte
// loads per thread, a visible part of a small launch (second evaluation round). k_prepare_weight_images writes the images
// once per step into global memory in LDS layout, and the kernels then copy 16 bytes per thread and iteration.
__device__ __forceinline__ void copy_weight_image(half_t* __restrict__ dst, const half_t* __restrict__ src, const int n_halfs, const int tid, const int nthreads) {
	for (int i = tid * 8; i < n_halfs; i += nthreads * 8) *reinterpret_cast<h8*>(dst + i) = *reinterpret_cast<const h8*>(src + i);
}

// K2 (+K3), register-chained flavour: the two SDF layers of k_forward_chained (same weight image, of which only W_S0 | W_S1
// are copied), level constants in LDS, no derivative bookkeeping: 9 KB of LDS and few registers, so four workgroups share a CU
// and the 112 gathers of one sample hide behind those of the others. Results are those of k_forward_chained's sdf channel.
constexpr int PQ2_WAVE_HALFS = TILE * S32 + TILE;
constexpr size_t LDS_POINT2 = (size_t)(W_S0T + WAVES_PER_WG * PQ2_WAVE_HALFS) * sizeof(half_t);
// -DRNB_POISON_LDS (tools/lds_poison_check.sh, never shipped): the MFMA kernels begin by filling their dynamic LDS with the half NaN pattern 0x7e00 (as fp32: 4.3e37). A kernel that
// reads only what it has written computes what it computes without the fill; one that reads a tile, a padding column or a weight slot it never wrote shows NaNs or another result.
__device__ __forceinline__ void poison_lds(char* smem_raw, const size_t bytes, const int tid, const int nthreads) {
#ifdef RNB_POISON_LDS
	uint32_t* w = reinterpret_cast<uint32_t*>(smem_raw);
	for (uint32_t q = (uint32_t)tid; q < (uint32_t)(bytes / 4); q += (uint32_t)nthreads) w[q] = 0x7e007e00u;
	__syncthreads();
#else
	(void)smem_raw; (void)bytes; (void)tid; (void)nthreads;
#endif
}

template <bool EMU, int PIPE_DEPTH = 0, int ND = 0>
__device__ __forceinline__ void point_query_chained_body(const GridMeta& G, const NetW& net, const PointArgs& a, const half_t* __restrict__ wimg, char* smem_raw, LevelMeta* lm) {
	poison_lds(smem_raw, LDS_POINT2, threadIdx.x, WG);
	half_t* wts = reinterpret_cast<half_t*>(smem_raw);
	fill_level_meta(lm, G, threadIdx.x);
	const uint32_t n_levels = G.n_levels, valid_level = G.valid_level;
	if (wimg) copy_weight_image(wts, wimg, W_S0T, threadIdx.x, WG);
	else {
		for (int i = threadIdx.x; i < 64 * 32; i += WG) { int o = i >> 5, k = i & 31; wts[W_S0 + o * S32 + k] = net.sdf_w0[i]; }
		for (int i = threadIdx.x; i < 16 * 64; i += WG) { int o = i >> 6, q = i & 63; wts[W_S1 + o * S64 + q] = net.sdf_w1[o * 64 + chain_logical(q)]; }
	}
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	half_t* X = wts + W_S0T + wave * PQ2_WAVE_HALFS;
	half_t* Z = X + TILE * S32;
	const half_t variance = net.variance[0];
	const half_t bias = f2h(a.sdf_bias);
	uint32_t s_first = 0, s_end = a.n;
	if (a.range) { s_first = a.range[0]; s_end = min(a.range[1], a.n); }
	const uint32_t n_tiles = (s_end - s_first + TILE - 1) / TILE;
	const int r16 = lane & 15, hq = lane >> 4;
	const bool by_xcd = a.xcd != 0u && (gridDim.x & 7u) == 0u;
	const uint32_t t8 = by_xcd ? (n_tiles + 7u) / 8u : n_tiles, t_base = by_xcd ? (blockIdx.x & 7u) * t8 : 0u, t_stop = min(n_tiles, t_base + t8);
	const uint32_t t_stride = (by_xcd ? gridDim.x / 8u : gridDim.x) * WAVES_PER_WG;
	for (uint32_t tile = t_base + (by_xcd ? blockIdx.x / 8u : blockIdx.x) * WAVES_PER_WG + wave; tile < t_stop; tile += t_stride) {
		const uint32_t s = s_first + tile * TILE + lane;
		const bool valid = s < s_end;
		float x = 0.5f, y = 0.5f, z = 0.5f;
		if (valid) { x = a.xyz[(size_t)s * 3 + 0]; y = a.xyz[(size_t)s * 3 + 1]; z = a.xyz[(size_t)s * 3 + 2]; }
		uint32_t cell = 0;
		if (valid && a.splat_idx) cell = a.splat_idx[s];
		half_t feat[28];
		float dummy[1][3];
		if (PIPE_DEPTH) encode_all_pipelined<false, PIPE_DEPTH ? PIPE_DEPTH : 1, ND>(lm, n_levels, valid_level, net.grid, x, y, z, feat, dummy);
		else encode_all_lm<false>(lm, n_levels, valid_level, net.grid, x, y, z, feat, dummy);
		write_sdf_in_row(X, lane, x, y, z, feat);
		wave_lds_sync();
		h8 bz[4][2];
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 1, EMU ? EMU_NATURAL : EMU_OFF>(wts + W_S0, S32, X, S32, acc, lane);
			chain_pack<true>(acc, bz);
		}
		f4 acc_so[1][4];
		zero_acc<1>(acc_so);
		mfma_layer_regs<1, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + W_S1, S64, bz, acc_so, lane);
		if (hq == 0) { // D layout: row 0 (the sdf) of sample 16 nt + r16
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) Z[16 * nt + r16] = f2h(acc_so[0][nt][0]);
		}
		wave_lds_sync();
		half_t v = Z[lane] + bias; // sdf_add_bias (common_operation.cuh:299-309)
		if (a.want_density) v = sdf_to_density(v, variance);
		if (valid) {
			if (a.out) a.out[s] = v;
			if (a.splat_idx) atomicMax(reinterpret_cast<uint32_t*>(a.grid_tmp) + cell, __float_as_uint(h2f(v))); // testbed_nerf.cu:634
		}
		wave_lds_sync(); // X, Z are rewritten by the next tile
	}
}
__global__ __launch_bounds__(WG, 4) void k_point_query_chained(const GridMeta G, const NetW net, const PointArgs a, const half_t* __restrict__ wimg) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	point_query_chained_body<false>(G, net, a, wimg, smem_raw, lm);
}
template <int DEPTH, int ND = 0>
__global__ __launch_bounds__(WG, 3) void k_point_query_chained_pipe(const GridMeta G, const NetW net, const PointArgs a, const half_t* __restrict__ wimg) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	point_query_chained_body<false, DEPTH, ND>(G, net, a, wimg, smem_raw, lm);
}
// rnb_config::accumulate = RNB_ACCUM_HALF: the reference's half accumulators (mlp.cuh, mfma_emul16)
__global__ __launch_bounds__(WG, 2) void k_point_query_chained_emul(const GridMeta G, const NetW net, const PointArgs a, const half_t* __restrict__ wimg) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	point_query_chained_body<true>(G, net, a, wimg, smem_raw, lm);
}
template <int DEPTH, int ND = 0>
__global__ __launch_bounds__(WG, 2) void k_point_query_chained_emul_pipe(const GridMeta G, const NetW net, const PointArgs a, const half_t* __restrict__ wimg) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	point_query_chained_body<true, DEPTH, ND>(G, net, a, wimg, smem_raw, lm);
}

constexpr int FWD2_WAVE_HALFS = TILE * S32 + TILE * 8 + TILE;
constexpr size_t LDS_FWD2 = (size_t)(W_FWD_END + WAVES_PER_WG * FWD2_WAVE_HALFS) * sizeof(half_t);

template <bool EMU, int PIPE_DEPTH = 0, int ND = 0>
__device__ __forceinline__ void forward_chained_body(const GridMeta& G, const NetW& net, const FwdArgs& a, char* smem_raw, LevelMeta* lm) {
	poison_lds(smem_raw, LDS_FWD2, threadIdx.x, WG);
	half_t* wts = reinterpret_cast<half_t*>(smem_raw);
	fill_level_meta(lm, G, threadIdx.x);
	const uint32_t n_levels = G.n_levels, valid_level = G.valid_level;
	if (a.wimg) copy_weight_image(wts, a.wimg, W_FWD_END, threadIdx.x, WG);
	else load_weights_chained(wts, net, threadIdx.x, WG);
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	half_t* X = wts + W_FWD_END + wave * FWD2_WAVE_HALFS;
	half_t* Y = X + TILE * S32;
	half_t* Z = Y + TILE * 8;
	const half_t variance = net.variance[0];
	const half_t bias = f2h(a.sdf_bias);
	uint32_t n = a.n_max;
	if (a.n_ptr) n = min(*a.n_ptr, a.n_max);
	const uint32_t n_tiles = (n + TILE - 1) / TILE;
	const int r16 = lane & 15, hq = lane >> 4;
	for (uint32_t tile = blockIdx.x * WAVES_PER_WG + wave; tile < n_tiles; tile += gridDim.x * WAVES_PER_WG) {
		const uint32_t q_in = tile * TILE + lane;
		const bool valid = q_in < n;
		uint32_t s = q_in;
		if (a.idx && valid) s = a.idx[q_in];
		float c[7] = {0.5f, 0.5f, 0.5f, 0.f, 0.f, 0.f, 0.f};
		if (valid) {
#pragma unroll
			for (int q = 0; q < 7; ++q) c[q] = a.coords[(size_t)s * 7 + q];
		}
		half_t feat[28];
		float dydx[28][3];
		if (PIPE_DEPTH) encode_all_pipelined<true, PIPE_DEPTH ? PIPE_DEPTH : 1, ND>(lm, n_levels, valid_level, net.grid, c[0], c[1], c[2], feat, dydx);
		else encode_all_lm<true>(lm, n_levels, valid_level, net.grid, c[0], c[1], c[2], feat, dydx);
		write_sdf_in_row(X, lane, c[0], c[1], c[2], feat);
		wave_lds_sync();
		// z1 = relu(W0 sdf_in) (registers) ; dz1 = W1[0,:] (.) relu'(z1) (registers)   (nerf_network.h:159-176)
		h8 bz[4][2];
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 1, EMU ? EMU_NATURAL : EMU_OFF>(wts + W_S0, S32, X, S32, acc, lane);
			chain_pack<true>(acc, bz);
		}
		f4 acc_so[1][4];
		zero_acc<1>(acc_so);
		mfma_layer_regs<1, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + W_S1, S64, bz, acc_so, lane); // sdf_out = W1 z1
		{
			// the backward transfer tests the stored half activation (common_device.h:182 ff.)
#pragma unroll
			for (int ks = 0; ks < 2; ++ks) {
				const h8 w1 = *reinterpret_cast<const h8*>(wts + W_S1 + 0 * S64 + 32 * ks + 8 * hq);
#pragma unroll
				for (int nt = 0; nt < 4; ++nt)
#pragma unroll
					for (int j = 0; j < 8; ++j) bz[nt][ks][j] = (bz[nt][ks][j] > (half_t)0.f) ? w1[j] : (half_t)0.f;
			}
		}
		wave_lds_sync(); // X (sdf_in) fully consumed by the first layer's MFMAs
		{
			f4 acc[2][4];
			zero_acc<2>(acc);
			mfma_layer_regs<2, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + W_S0T, S64, bz, acc, lane); // d sdf / d in = W0^T dz1
			store_acc<2, false>(acc, X, S32, 0, lane);
		}
		wave_lds_sync();
		// per lane: grad = sum_k dsdf_din[3+k] * dy_dx[k] + dsdf_din[0..2]  (grid.h:527-554, nerf_network.h:185-189)
		float grad[3] = {0.f, 0.f, 0.f};
		{
			half_t din[32];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const h8 v = *reinterpret_cast<const h8*>(X + lane * S32 + q * 8);
#pragma unroll
				for (int j = 0; j < 8; ++j) din[q * 8 + j] = v[j];
			}
#pragma unroll
			for (int k = 0; k < 28; ++k) {
				const float dl = h2f(din[3 + k]);
#pragma unroll
				for (int d = 0; d < 3; ++d) grad[d] += dl * dydx[k][d];
			}
#pragma unroll
			for (int d = 0; d < 3; ++d) grad[d] += h2f(din[d]);
		}
		// colour-MLP side inputs [x y z | grad | 0 0] per sample -> Y ; raw sdf (D layout, hq == 0, r == 0) -> Z   (nerf_network.h:206-218)
		{
			const h8 v = {f2h(c[0]), f2h(c[1]), f2h(c[2]), f2h(grad[0]), f2h(grad[1]), f2h(grad[2]), (half_t)0.f, (half_t)0.f};
			*reinterpret_cast<h8*>(Y + lane * 8) = v;
			if (hq == 0) {
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) Z[16 * nt + r16] = f2h(acc_so[0][nt][0]);
			}
		}
		wave_lds_sync();
		const half_t sdf0 = Z[lane];
		if (a.cin_out) { // rows of the colour MLP's input (natural column order); lane (r16, hq) holds sdf_out[4 hq .. 4 hq + 3] of sample 16 nt + r16
			if (valid) {
				h8* row = reinterpret_cast<h8*>(a.cin_out + (size_t)s * 32);
				row[2] = *reinterpret_cast<const h8*>(Y + lane * 8);
				row[3] = h8{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
			}
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) {
				const uint32_t s_nt = (uint32_t)__shfl((int)s, 16 * nt + r16, 64);
				const bool v_nt = __shfl((int)valid, 16 * nt + r16, 64) != 0;
				const h4 o = {f2h(acc_so[0][nt][0]), f2h(acc_so[0][nt][1]), f2h(acc_so[0][nt][2]), f2h(acc_so[0][nt][3])};
				if (v_nt) *reinterpret_cast<h4*>(a.cin_out + (size_t)s_nt * 32 + 4 * hq) = o;
			}
		}
		h8 bh[4][2];
		{
			h8 bin[4][1];
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) {
				h4 o = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
				if (hq < 2) o = *reinterpret_cast<const h4*>(Y + (16 * nt + r16) * 8 + 4 * hq);
#pragma unroll
				for (int j = 0; j < 4; ++j) { bin[nt][0][j] = f2h(acc_so[0][nt][j]); bin[nt][0][4 + j] = o[j]; }
			}
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer_regs<4, 1, EMU ? EMU_CHAINED : EMU_OFF>(wts + W_C0, S32, bin, acc, lane);
			chain_pack<true>(acc, bh);
		}
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer_regs<4, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + W_C1, S64, bh, acc, lane);
			chain_pack<true>(acc, bh);
		}
		{
			f4 acc[1][4];
			zero_acc<1>(acc);
			mfma_layer_regs<1, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + W_C2, S64, bh, acc, lane);
			store_acc<1, false>(acc, X, S32, 0, lane); // X rows were last read before the previous sync
		}
		wave_lds_sync();
		// output packing (nerf_network.h:221-250)
		{
			h8 o0 = *reinterpret_cast<const h8*>(X + lane * S32 + 0);
			h8 o1 = *reinterpret_cast<const h8*>(X + lane * S32 + 8);
			o0[3] = sdf0 + bias;
			o0[4] = f2h(grad[0]); o0[5] = f2h(grad[1]); o0[6] = f2h(grad[2]);
			o0[7] = variance;
			o1[0] = f2h(c[4]); o1[1] = f2h(c[5]); o1[2] = f2h(c[6]);
			if (valid) {
				h8* dst = reinterpret_cast<h8*>(a.out + (size_t)s * 16);
				dst[0] = o0;
				dst[1] = o1;
			}
		}
		wave_lds_sync();
	}
}
__global__ __launch_bounds__(WG, 2) void k_forward_chained(const GridMeta G, const NetW net, const FwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	forward_chained_body<false>(G, net, a, smem_raw, lm);
}
template <int DEPTH, int ND = 0>
__global__ __launch_bounds__(WG, 2) void k_forward_chained_pipe(const GridMeta G, const NetW net, const FwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	forward_chained_body<false, DEPTH, ND>(G, net, a, smem_raw, lm);
}
// rnb_config::accumulate = RNB_ACCUM_HALF: the reference's half accumulators (mlp.cuh, mfma_emul16)
__global__ __launch_bounds__(WG, 2) void k_forward_chained_emul(const GridMeta G, const NetW net, const FwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	forward_chained_body<true>(G, net, a, smem_raw, lm);
}
template <int DEPTH, int ND = 0>
__global__ __launch_bounds__(WG, 2) void k_forward_chained_emul_pipe(const GridMeta G, const NetW net, const FwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	forward_chained_body<true, DEPTH, ND>(G, net, a, smem_raw, lm);
}


// ---------------------------------------------------------------------------------------------
// K10 + K11 data path on the compacted batch
// ---------------------------------------------------------------------------------------------
struct TrainScratch {
	// feature-major [feature][B] halfs: operands of the weight-gradient GEMMs
	half_t *h2, *h1, *cin, *z1, *sdfin, *dz1; // activations (cin: 32 compact rows)
	half_t *dr, *dh2, *dh1, *dso, *dz, *ddin, *front; // gradients (dr: 16 rows, dso: 16 rows, ddin: 32 rows)
	// per-sample inputs of the grid scatter: one 8-byte gather per (level, sample), one 32-byte record per sample
	uint32_t* g12; // [14][B][2] half2 pairs: {dL/dfeat (first order), d sdf / d feat (dL_denc_output of the double backward)}
	float* srec;   // [B][8]: x y z | dL/d(grad sdf) (3) | 0 0
	float* var_partial; // [n_waves_total] partial sums of dL_doutput[7]
};

struct TrainArgs {
	const float* coords;   // [B][7] compacted
	const half_t* dout;    // [B][16]
	uint32_t B;
	uint32_t B_global; // the batch the Eikonal term is divided by (nerf_network.h:359-365): B x world_size, the samples of the whole step
	float sdf_bias;
	uint32_t skip_rgb; // --no-albedo: dL/d(rgb logits) is identically 0 (opti_rgb = 0, testbed_nerf.cu:1954-1962), so the
	                   // colour MLP receives and propagates exact zeros: its forward/backward are skipped, not approximated
	const half_t* wimg; // optional: k_fwd_bwd_sdf's LDS weight image prepared by k_prepare_weight_images
	float *dw_w0, *dw_w0b, *dw_w1, *dw_w1b; // k_fwd_bwd_sdf: one partial per workgroup of the four SDF-MLP weight gradients (k_dw's layout)
	const half_t* dcin;    // k_fwd_bwd_sdf_full: [B][32] dL/d(colour-MLP input row) from k_rgb_fwd_bwd
	TrainScratch t;
};

__device__ __forceinline__ void fm_store(const half_t* __restrict__ tile, const int stride, const int width, half_t* __restrict__ dst, const uint32_t B, const uint32_t s) {
	const half_t* row = tile + (s & 63) * stride;
	for (int f = 0; f < width; f += 8) {
		const h8 v = *reinterpret_cast<const h8*>(row + f);
#pragma unroll
		for (int j = 0; j < 8; ++j) dst[(size_t)(f + j) * B + s] = v[j];
	}
}

__global__ __launch_bounds__(WG, 1) void k_fwd_bwd(const GridMeta G, const NetW net, const TrainArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	half_t* wts = reinterpret_cast<half_t*>(smem_raw);
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	fill_level_meta(lm, G, threadIdx.x);
	const uint32_t n_levels = G.n_levels, valid_level = G.valid_level;
	if (a.wimg) copy_weight_image(wts, a.wimg, W_TRAIN_END, threadIdx.x, WG);
	else load_weights<true>(wts, net, threadIdx.x, WG);
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	half_t* tA = wts + W_TRAIN_END + wave * 3 * ACT_TILE_HALFS;
	half_t* tB = tA + ACT_TILE_HALFS;
	half_t* tC = tB + ACT_TILE_HALFS;
	const int r16 = lane & 15, hq = lane >> 4;
	const uint32_t B = a.B;
	const uint32_t n_tiles = B / TILE;
	const TrainScratch& T = a.t;
	float var_sum = 0.f;
	for (uint32_t tile = blockIdx.x * WAVES_PER_WG + wave; tile < n_tiles; tile += gridDim.x * WAVES_PER_WG) {
		const uint32_t s = tile * TILE + lane;
		float c[7];
#pragma unroll
		for (int q = 0; q < 7; ++q) c[q] = a.coords[(size_t)s * 7 + q];
		half_t dout[16];
		{
			const h8* src = reinterpret_cast<const h8*>(a.dout + (size_t)s * 16);
			const h8 d0 = src[0], d1 = src[1];
#pragma unroll
			for (int j = 0; j < 8; ++j) { dout[j] = d0[j]; dout[8 + j] = d1[j]; }
		}
		// ---------------- forward (as k_forward, activations also exported feature-major) ----------------
		half_t feat[28];
		float dydx[28][3];
		encode_all_lm<true>(lm, n_levels, valid_level, net.grid, c[0], c[1], c[2], feat, dydx);
		write_sdf_in_row(tA, lane, c[0], c[1], c[2], feat);
		wave_lds_sync();
		fm_store(tA, S32, 32, T.sdfin, B, s);
		uint64_t m_z1;
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 1>(wts + W_S0, S32, tA, S32, acc, lane);
			m_z1 = store_acc<4, true>(acc, tB, S64, 0, lane);
		}
		wave_lds_sync();
		fm_store(tB, S64, 64, T.z1, B, s);
		// sdf_out -> tC[.][0..15]
		{
			f4 acc[1][4];
			zero_acc<1>(acc);
			mfma_layer<1, 2>(wts + W_S1, S64, tB, S64, acc, lane);
			store_acc<1, false>(acc, tC, S32, 0, lane);
		}
		wave_lds_sync(); // tA (sdf_in) and tB (z1) are dead from here on
		// dz1 -> tA (64 wide)
#pragma unroll
		for (int mt = 0; mt < 4; ++mt) {
			const h4 w1 = *reinterpret_cast<const h4*>(wts + W_S1 + 0 * S64 + 16 * mt + 4 * hq);
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) {
				h4 v;
#pragma unroll
				for (int r = 0; r < 4; ++r) v[r] = ((m_z1 >> ((mt * 4 + nt) * 4 + r)) & 1ull) ? w1[r] : (half_t)0.f;
				*reinterpret_cast<h4*>(tA + (16 * nt + r16) * S64 + 16 * mt + 4 * hq) = v;
			}
		}
		wave_lds_sync();
		fm_store(tA, S64, 64, T.dz1, B, s);
		// dsdf_din -> tB[.][0..31]
		{
			f4 acc[2][4];
			zero_acc<2>(acc);
			mfma_layer<2, 2>(wts + W_S0T, S64, tA, S64, acc, lane);
			store_acc<2, false>(acc, tB, S32, 0, lane);
		}
		wave_lds_sync();
		float grad[3] = {0.f, 0.f, 0.f};
		half_t din[32];
		{
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const h8 v = *reinterpret_cast<const h8*>(tB + lane * S32 + q * 8);
#pragma unroll
				for (int j = 0; j < 8; ++j) din[q * 8 + j] = v[j];
			}
#pragma unroll
			for (int k = 0; k < 28; ++k) {
				const float dl = h2f(din[3 + k]);
#pragma unroll
				for (int d = 0; d < 3; ++d) grad[d] += dl * dydx[k][d];
			}
#pragma unroll
			for (int d = 0; d < 3; ++d) grad[d] += h2f(din[d]);
#pragma unroll
			for (int l = 0; l < 14; ++l) T.g12[((size_t)l * B + s) * 2 + 1] = pack_h2(din[3 + 2 * l], din[3 + 2 * l + 1]);
		}
		{
			h8 v0 = {f2h(c[0]), f2h(c[1]), f2h(c[2]), f2h(grad[0]), f2h(grad[1]), f2h(grad[2]), (half_t)0.f, (half_t)0.f};
			h8 v1 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
			*reinterpret_cast<h8*>(tC + lane * S32 + 16) = v0;
			*reinterpret_cast<h8*>(tC + lane * S32 + 24) = v1;
		}
		wave_lds_sync();
		uint64_t m_h1 = 0, m_h2 = 0;
		float dn[3];
		if (!a.skip_rgb) {
		fm_store(tC, S32, 32, T.cin, B, s);
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 1>(wts + W_C0, S32, tC, S32, acc, lane);
			m_h1 = store_acc<4, true>(acc, tA, S64, 0, lane);
		}
		wave_lds_sync();
		fm_store(tA, S64, 64, T.h1, B, s);
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 2>(wts + W_C1, S64, tA, S64, acc, lane);
			m_h2 = store_acc<4, true>(acc, tB, S64, 0, lane);
		}
		wave_lds_sync();
		fm_store(tB, S64, 64, T.h2, B, s);
		// (the rgb output layer is not needed by the backward pass: no output activation, fully_fused_mlp.cu:936-940)

		// ---------------- backward (nerf_network.h:257-452) ----------------
		// dL_drgb = rows 0..2 of dL_doutput (extract_rgb); dh2 = (W2^T dr) (.) relu'(h2), 3 MACs per element
		{
#pragma unroll
			for (int q = 0; q < 16; ++q) T.dr[(size_t)q * B + s] = q < 3 ? dout[q] : (half_t)0.f;
			half_t* row = tA + lane * S64; // tA (h1) is dead: dh2 goes there, sample-major, computed per lane
			const float d0 = h2f(dout[0]), d1 = h2f(dout[1]), d2 = h2f(dout[2]);
			const half_t* hrow = tB + lane * S64;
			for (int f = 0; f < 64; f += 8) {
				const h8 hv = *reinterpret_cast<const h8*>(hrow + f);
				h8 o;
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					float acc = 0.f;
					acc += h2f(wts[W_C2 + 0 * S64 + f + j]) * d0;
					acc += h2f(wts[W_C2 + 1 * S64 + f + j]) * d1;
					acc += h2f(wts[W_C2 + 2 * S64 + f + j]) * d2;
					o[j] = (h2f(hv[j]) > 0.f) ? f2h(acc) : (half_t)0.f;
				}
				*reinterpret_cast<h8*>(row + f) = o;
			}
		}
		wave_lds_sync();
		fm_store(tA, S64, 64, T.dh2, B, s);
		// dh1 = (W1^T dh2) (.) relu'(h1) -> tB
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 2>(wts + W_C1T, S64, tA, S64, acc, lane);
			wave_lds_sync();
			store_acc_masked<4>(acc, m_h1, tB, S64, 0, lane);
		}
		wave_lds_sync();
		fm_store(tB, S64, 64, T.dh1, B, s);
		// dcin (compact 32) = W0c^T dh1 -> tA[.][0..31]
		{
			f4 acc[2][4];
			zero_acc<2>(acc);
			mfma_layer<2, 2>(wts + W_C0T, S64, tB, S64, acc, lane);
			store_acc<2, false>(acc, tA, S32, 0, lane);
		}
		} else {
			// dcin == 0
			const h8 zero = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
#pragma unroll
			for (int q = 0; q < 4; ++q) *reinterpret_cast<h8*>(tA + lane * S32 + q * 8) = zero;
		}
		wave_lds_sync();
		// dso = dcin[0:16], [0] += dL_doutput[3] (add_density_gradient); dn (nerf_network.h:343-373)
		{
			const h8 a0 = *reinterpret_cast<const h8*>(tA + lane * S32 + 0);
			const h8 a1 = *reinterpret_cast<const h8*>(tA + lane * S32 + 8);
			const h8 a2 = *reinterpret_cast<const h8*>(tA + lane * S32 + 16);
			h8 o0 = a0, o1 = a1;
			o0[0] = o0[0] + dout[3];
			const h8 zero = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
			// 32-wide tile for the K = 32 MFMA: columns 16..31 zero
			*reinterpret_cast<h8*>(tC + lane * S32 + 0) = o0;
			*reinterpret_cast<h8*>(tC + lane * S32 + 8) = o1;
			*reinterpret_cast<h8*>(tC + lane * S32 + 16) = zero;
			*reinterpret_cast<h8*>(tC + lane * S32 + 24) = zero;
#pragma unroll
			for (int j = 0; j < 8; ++j) { T.dso[(size_t)j * B + s] = o0[j]; T.dso[(size_t)(8 + j) * B + s] = o1[j]; }
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				float v = h2f(a2[3 + d]);                  // dL_drgb_network_input rows 35..37
				v += h2f(dout[4 + d]) / (float)a.B_global;          // add_positions_view_ekloss (common_operation.cuh:283-296)
				v += h2f(dout[8 + d]);                     // add_positions_view
				dn[d] = v;
			}
			reinterpret_cast<f4*>(T.srec)[(size_t)s * 2 + 0] = f4{c[0], c[1], c[2], dn[0]};
			reinterpret_cast<f4*>(T.srec)[(size_t)s * 2 + 1] = f4{dn[1], dn[2], 0.f, 0.f};
			var_sum += h2f(dout[7]);
		}
		wave_lds_sync();
		// dz = (W1^T dso) (.) relu'(z1) -> tB (64 wide)
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 1>(wts + W_S1T, S32, tC, S32, acc, lane);
			store_acc_masked<4>(acc, m_z1, tB, S64, 0, lane);
		}
		wave_lds_sync();
		fm_store(tB, S64, 64, T.dz, B, s);
		// dsin = W0^T dz -> tA[.][0..31]; its feature rows are the first-order dL/dfeat of the grid
		{
			f4 acc[2][4];
			zero_acc<2>(acc);
			mfma_layer<2, 2>(wts + W_S0T, S64, tB, S64, acc, lane);
			store_acc<2, false>(acc, tA, S32, 0, lane);
		}
		wave_lds_sync();
		{
			half_t dsin[32];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const h8 v = *reinterpret_cast<const h8*>(tA + lane * S32 + q * 8);
#pragma unroll
				for (int j = 0; j < 8; ++j) dsin[q * 8 + j] = v[j];
			}
#pragma unroll
			for (int l = 0; l < 14; ++l) T.g12[((size_t)l * B + s) * 2 + 0] = pack_h2(dsin[3 + 2 * l], dsin[3 + 2 * l + 1]);
		}
		// ddin = [half(dn) | dL/d(dL_dy) | 0] -> tC (32 wide)   (grid.h:858-883, nerf_network.h:423-433)
		{
			half_t dd[32];
#pragma unroll
			for (int d = 0; d < 3; ++d) dd[d] = f2h(dn[d]);
#pragma unroll
			for (int k = 0; k < 28; ++k) {
				float r = 0.f;
#pragma unroll
				for (int d = 0; d < 3; ++d) r += dydx[k][d] * dn[d];
				dd[3 + k] = f2h(r);
			}
			dd[31] = (half_t)0.f;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				h8 v;
#pragma unroll
				for (int j = 0; j < 8; ++j) v[j] = dd[q * 8 + j];
				*reinterpret_cast<h8*>(tC + lane * S32 + q * 8) = v;
			}
		}
		wave_lds_sync();
		fm_store(tC, S32, 32, T.ddin, B, s);
		// front = (W0 ddin) (.) relu'(z1) -> tB   (fully_fused_mlp.cu:1097-1107)
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer<4, 1>(wts + W_S0, S32, tC, S32, acc, lane);
			wave_lds_sync();
			store_acc_masked<4>(acc, m_z1, tB, S64, 0, lane);
		}
		wave_lds_sync();
		fm_store(tB, S64, 64, T.front, B, s);
		wave_lds_sync();
		(void)m_h2;
	}
	// variance gradient partial (nerf_network.h:327-340): one value per wavefront, reduced in fixed order later
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) var_sum += __shfl_down(var_sum, off, 64);
	if (lane == 0) T.var_partial[blockIdx.x * WAVES_PER_WG + wave] = var_sum;
}

// ---------------------------------------------------------------------------------------------
// K10 + K11 when the colour MLP receives and returns exact zeros (TrainArgs::skip_rgb, i.e. --no-albedo: stage 1 of the
// pipeline and the headline benchmark). What remains of nerf_network.h:97-452 is the SDF MLP with its first- and
// second-order backward: 4 GEMMs of the 12, all register-chained as in k_forward_chained (mlp.cuh), a 10 KB weight image,
// one 5 KB exchange tile per wavefront -> 2+ workgroups per CU instead of 1.
//   z1 = relu(W0 in)                      dz1 = W1[0,:] (.) relu'(z1)        d sdf/d in = W0^T dz1      (-> g2)
//   dso = e0 * dL/dsdf                    dz = (W1^T dso) (.) relu'(z1) = dL/dsdf * dz1  (one product per element: the
//                                         GEMM's other 15 K-terms are zeros)   dL/d in = W0^T dz           (-> g1)
//   ddin = [dn | dy_dx . dn]              front = (W0 ddin) (.) relu'(z1)
// Weight gradients (round 3): accumulated HERE, per wavefront, over all its tiles -- no operand export, no GEMM launches beside the
// scatter. K of those GEMMs is the sample index, and every activation of this kernel lives as "8 features of one sample per lane"; what
// the GEMMs need is "8 samples of one feature per lane". Both are obtained on the matrix cores (3 % busy on this path) instead of through
// memory: per 64-sample tile, from the two sample-major input tiles X (in) and D (ddin) and dL/dsdf in LDS,
//   z1^T  = in W0^T      MFMA with the SAMPLES as M and the hidden units as N: D puts 4 samples of one hidden unit in a lane, so two
//   fr^T  = ddin W0^T    m-tiles are the 8 K-values an A operand of the weight-gradient GEMM needs
//   dz    = relu'(z1) (.) half(w1 * dL/dsdf),   dz1 = relu'(z1) (.) w1,   front = relu'(z1) (.) half(fr)   (the roundings of k_fwd_bwd)
//   in^T, ddin^T         the B operands (input column in the lane, 8 samples in registers): MFMA against the identity (exact)
//   dW0 += dz in^T, dW0' += dz1 ddin^T (64x32 each, MFMA);   dW1[0,:] += dL/dsdf . z1,  dW1'[0,:] += sum front   (VALU)
//   (first order: fully_fused_mlp.cu:953-1030; second order: fully_fused_mlp.cu:1097-1131)
// 80 MFMAs per tile, 72 accumulator registers; the workgroup's four wavefronts are summed in LDS at the end and leave ONE partial per
// workgroup, which k_dw_finish sums in a fixed order (deterministic). Round 2 exported z1, dz1, dz, front, dso and both inputs
// feature-major (672 B per sample of 2-byte stores, 176 MB per step) to four GEMM launches that ran 190 us beside the atomic-bound scatter.
// ---------------------------------------------------------------------------------------------
constexpr int SW_S0 = 0;                    // [64][S32] sdf W0, input columns in tile order (below)
constexpr int SW_S0T = SW_S0 + 64 * S32;    // [32][S64] sdf W0^T, rows in tile order, columns in chain order
constexpr int SW_W1 = SW_S0T + 32 * S64;    // [64] sdf W1 row 0 in chain order
constexpr int SW_W1N = SW_W1 + 64;          // [64] the same row in natural order (weight-gradient block)
constexpr int SW_END = SW_W1N + 64;
constexpr int SW_END_PADDED = (SW_END + 7) / 8 * 8;
constexpr int FBS_WAVE_HALFS = 2 * TILE * S32 + TILE; // two 32-wide tiles (network input / second-order input) + one half per sample
constexpr size_t LDS_FBS = (size_t)(SW_END + WAVES_PER_WG * FBS_WAVE_HALFS) * sizeof(half_t);
// albedo mode (k_fwd_bwd_sdf_full): dL/d sdf_out has all 16 rows, so W1^T is a matrix again: [64][S32], row = hidden unit, columns = the 16 outputs + zeros
constexpr int SW_W1T = SW_END_PADDED;
constexpr int SWF_END = SW_W1T + 64 * S32;
constexpr int SO_STRIDE = 24;                                          // halfs per row of the dso tile (16 + 8: conflict-free 16-byte fragment reads)
constexpr int FBS_FULL_WAVE_HALFS = FBS_WAVE_HALFS + TILE * SO_STRIDE; // + dso rows [64][16] staged at the top of the tile
constexpr size_t LDS_FBS_FULL = (size_t)(SWF_END + WAVES_PER_WG * FBS_FULL_WAVE_HALFS) * sizeof(half_t);
// Column order of the 32-wide input tiles: the 28 hash features first (a level's pair is one aligned 4-byte LDS access),
// then x y z, then the pad -- the input index is a summation index of W0 . in, the weight images follow the same order.
__host__ __device__ constexpr int fbs_logical(int p) { return p < 28 ? 3 + p : (p < 31 ? p - 28 : 31); }
// The half mode's order of the INPUT tiles (and of W0's columns): features 0..12 | x y z | features 13..27 | pad, so that slots 0..15 are the reference's
// columns 0..15 = its first 16-wide k-step and slots 16..31 its second: lane (r16, hq) reads slots 4 hq .. 4 hq + 3 of either half as the operand of a
// K = 16 MFMA -- no masked operands (mlp.cuh). A level's pair is two 2-byte LDS stores there. W0^T's rows (the order the input GRADIENTS leave in) stay in tile order.
__host__ __device__ constexpr int fbs_logical_h(int p) { return p < 13 ? 3 + p : (p < 16 ? p - 13 : p); }
__host__ __device__ constexpr int fbs_feature_slot_h(int q) { return q < 13 ? q : q + 3; }

__device__ inline void load_weights_fbs(half_t* __restrict__ w, const NetW& net, const int tid, const int nthreads, const bool half_order = false) {
	for (int i = tid; i < 64 * 32; i += nthreads) { const int o = i >> 5, p = i & 31; w[SW_S0 + o * S32 + p] = net.sdf_w0[o * 32 + (half_order ? fbs_logical_h(p) : fbs_logical(p))]; }
	for (int i = tid; i < 32 * 64; i += nthreads) { const int q = i >> 6, p = i & 63; w[SW_S0T + q * S64 + p] = net.sdf_w0[chain_logical(p) * 32 + fbs_logical(q)]; }
	for (int i = tid; i < 64; i += nthreads) { w[SW_W1 + i] = net.sdf_w1[chain_logical(i)]; w[SW_W1N + i] = net.sdf_w1[i]; }
	for (int i = SW_END + tid; i < SW_END_PADDED; i += nthreads) w[i] = (half_t)0.f;
	// (row padding of the images is never read)
}
__device__ inline void load_weights_fbs_full(half_t* __restrict__ w, const NetW& net, const int tid, const int nthreads, const bool half_order = false) {
	load_weights_fbs(w, net, tid, nthreads, half_order);
	for (int i = tid; i < 64 * 32; i += nthreads) { const int u = i >> 5, o = i & 31; w[SW_W1T + u * S32 + o] = o < 16 ? net.sdf_w1[o * 64 + u] : (half_t)0.f; }
}

__device__ inline void load_weights_rgb(half_t* __restrict__ w, const NetW& net, const int tid, const int nthreads);
// blockIdx.y == 0: image of k_forward_chained; 1: image of k_fwd_bwd_sdf(_full); 2: image of k_fwd_bwd; 3: image of k_rgb_fwd_bwd. All from the training weights.
// gridDim.x workgroups per image (round 4: 16, one element per thread and array; one workgroup per image walked ~50 dependent 2-byte loads per thread, 15 us
// alone and 70-90 us beside the scatter, at the end of the side stream that the next step's network evaluation waits for).
constexpr uint32_t WIMG_WGS = 16;
__global__ __launch_bounds__(WG) void k_prepare_weight_images(const NetW net, half_t* __restrict__ img_fwd, half_t* __restrict__ img_fbs, half_t* __restrict__ img_train, half_t* __restrict__ img_rgb, const int half_order) {
	const int tid = blockIdx.x * WG + threadIdx.x, nthreads = gridDim.x * WG;
	if (blockIdx.y == 0) load_weights_chained(img_fwd, net, tid, nthreads);
	else if (blockIdx.y == 1) load_weights_fbs_full(img_fbs, net, tid, nthreads, half_order != 0); // (accumulate = half: k_fwd_bwd_sdf*_h's input order)
	else if (blockIdx.y == 2) load_weights<true>(img_train, net, tid, nthreads);
	else load_weights_rgb(img_rgb, net, tid, nthreads);
}

// dst[idx] with a wave-uniform base and a 32-bit element index (scalar base + vector byte offset addressing)
template <typename T>
__device__ __forceinline__ void st32(T* __restrict__ base, const uint32_t idx, const T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + idx * (uint32_t)sizeof(T)) = v; }

__device__ __forceinline__ void export_frags(const h8 (&b)[4][2], half_t* __restrict__ dst, const uint32_t B, const uint32_t tile, const int lane) {
	const uint32_t r16 = lane & 15, hq = lane >> 4;
	const uint32_t lane_off = 4u * hq * B + tile * TILE + r16 * 4u; // per-lane part; the row part below is wave-uniform
#pragma unroll
	for (uint32_t ks = 0; ks < 2; ++ks)
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) {
			const uint32_t row_off = (16u * (2u * ks + (j >> 2)) + (j & 3u)) * B;
			const h4 v = {b[0][ks][j], b[1][ks][j], b[2][ks][j], b[3][ks][j]};
			*reinterpret_cast<h4*>(reinterpret_cast<char*>(dst) + (row_off + lane_off) * 2u) = v;
		}
}

// FULL (albedo mode, part 2 of 2 behind k_rgb_fwd_bwd): dL/d sdf_out = rows 0..15 of dL/d(colour input) (+ dL/dsdf on row 0) and
// dL/d(grad sdf) gains the colour MLP's rows 19..21 -- both known before the encode, so everything else is the kernel above with
//   dz = (W1^T dso) (.) relu'(z1)  as a K = 16 MFMA (both orientations) instead of one product per element, and
//   dW1 += dso z1^T                as a 16 x 64 MFMA (dso^T by an identity MFMA) instead of row 0 on the VALU.
// EMU (rnb_config::accumulate = RNB_ACCUM_HALF): every dot product over features rounds its accumulator to half after each of the reference's 16-wide k-steps (mlp.cuh);
// products with K = 16 (W1^T dso, the identity transposes) are one k-step and need nothing. The weight gradients (K = samples): SLICED (below) hands their operands to k_dw_sliced,
// which sums them in the reference's split-K order; without it (the albedo mode's FULL kernel, RNB_DW_SLICED=0) they keep fp32 accumulators in this kernel's tiling (deviation D1').
// SLICED (half mode, round 6): the weight gradients are NOT accumulated here: the tile's GEMM operands leave feature-major (TrainScratch: dz, dz1, z1, front as the fragments they are formed in,
// the two input tiles from LDS, dL/dsdf) for k_dw_sliced, which sums them in the reference's split-K order -- 642 B per sample of stores instead of 80 MFMAs per tile.
template <bool FULL, bool EMU = false, bool SLICED = false>
__device__ __forceinline__ void fwd_bwd_sdf_body(const GridMeta& G, const NetW& net, const TrainArgs& a, char* smem_raw, LevelMeta* lm) {
	poison_lds(smem_raw, FULL ? LDS_FBS_FULL : LDS_FBS, threadIdx.x, WG);
	half_t* wts = reinterpret_cast<half_t*>(smem_raw);
	constexpr int W_END = FULL ? SWF_END : SW_END;
	fill_level_meta(lm, G, threadIdx.x);
	const uint32_t n_live = min(G.n_levels, G.valid_level + 1u); // levels [0, n_live) are encoded, the others are zeros (grid.h:192-210)
	if (a.wimg) copy_weight_image(wts, a.wimg, W_END, threadIdx.x, WG);
	else if (FULL) load_weights_fbs_full(wts, net, threadIdx.x, WG, EMU);
	else load_weights_fbs(wts, net, threadIdx.x, WG, EMU);
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr int WAVE_HALFS = FULL ? FBS_FULL_WAVE_HALFS : FBS_WAVE_HALFS;
	half_t* X = wts + W_END + wave * WAVE_HALFS; // network input rows, later d sdf / d in, later dL / d in
	half_t* D = X + TILE * S32;                        // second-order input rows (ddin)
	half_t* Z = D + TILE * S32;
	half_t* SO = Z + TILE; // FULL: dso rows
	const int r16 = lane & 15, hq = lane >> 4;
	const uint32_t B = a.B;
	const uint32_t n_tiles = B / TILE;
	const TrainScratch& T = a.t;
	float var_sum = 0.f;
	f4 acc_w0[4][2], acc_w0b[4][2]; // weight-gradient accumulators of this wavefront (all its tiles)
	float acc_w1[4] = {0.f, 0.f, 0.f, 0.f}, acc_w1b[4] = {0.f, 0.f, 0.f, 0.f};
	f4 acc_w1f[FULL ? 4 : 1]; // FULL: dW1[o = 4 hq + r][u = 16 nt + r16]
#pragma unroll
	for (int nt = 0; nt < (FULL ? 4 : 1); ++nt) acc_w1f[nt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int mo = 0; mo < 4; ++mo)
#pragma unroll
		for (int ni = 0; ni < 2; ++ni) { acc_w0[mo][ni] = f4{0.f, 0.f, 0.f, 0.f}; acc_w0b[mo][ni] = f4{0.f, 0.f, 0.f, 0.f}; }
	for (uint32_t tile = blockIdx.x * WAVES_PER_WG + wave; tile < n_tiles; tile += gridDim.x * WAVES_PER_WG) {
		const uint32_t s = tile * TILE + lane;
		float c[3];
#pragma unroll
		for (int q = 0; q < 3; ++q) c[q] = a.coords[(size_t)s * 7 + q];
		half_t dout[16];
		{
			const h8* src = reinterpret_cast<const h8*>(a.dout + (size_t)s * 16);
			const h8 d0 = src[0], d1 = src[1];
#pragma unroll
			for (int j = 0; j < 8; ++j) { dout[j] = d0[j]; dout[8 + j] = d1[j]; }
		}
		// dn = dL/d(grad sdf) depends on the loss gradient only here (the colour MLP returns zeros), so the second-order input
		// ddin = [half(dn) | dy_dx . dn | 0] (grid.h:858-883, nerf_network.h:423-433) is formed level by level with the encode:
		// nothing per level stays in registers, the level loop is not unrolled, and occupancy rather than unrolling hides
		// the gather latency.
		float dn[3];
		{
			h8 a2 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
			if (FULL) {
				const h8* row = reinterpret_cast<const h8*>(a.dcin + (size_t)s * 32);
				a2 = row[2]; // dL_drgb_network_input rows 35..37 = compact columns 19..21
				h8 o0 = row[0];
				o0[0] = o0[0] + dout[3]; // dso: row 0 += dL/dsdf (add_density_gradient)
				*reinterpret_cast<h8*>(SO + lane * SO_STRIDE) = o0; // (read as fragments behind the encode's wave_lds_sync; requested here, beside the gathers)
				*reinterpret_cast<h8*>(SO + lane * SO_STRIDE + 8) = row[1];
			}
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				float v = FULL ? h2f(a2[3 + d]) : 0.f;     // (zero without the colour MLP)
				v += h2f(dout[4 + d]) / (float)a.B_global;          // add_positions_view_ekloss (common_operation.cuh:283-296)
				v += h2f(dout[8 + d]);                     // add_positions_view (nerf_network.h:343-373)
				dn[d] = v;
			}
		}
		reinterpret_cast<f4*>(T.srec)[(size_t)s * 2 + 0] = f4{c[0], c[1], c[2], dn[0]};
		reinterpret_cast<f4*>(T.srec)[(size_t)s * 2 + 1] = f4{dn[1], dn[2], 0.f, 0.f};
		uint32_t* Xrow = reinterpret_cast<uint32_t*>(X + lane * S32);
		uint32_t* Drow = reinterpret_cast<uint32_t*>(D + lane * S32);
#pragma unroll 1
		for (uint32_t level = 0; level < 14; ++level) {
			half_t f0 = (half_t)0.f, f1 = (half_t)0.f;
			float d0[3] = {0.f, 0.f, 0.f}, d1[3] = {0.f, 0.f, 0.f};
			if (level < n_live) encode_level_lm<true>(lm, net.grid, level, c[0], c[1], c[2], f0, f1, d0, d1);
			float r0 = 0.f, r1 = 0.f;
#pragma unroll
			for (int d = 0; d < 3; ++d) { r0 += d0[d] * dn[d]; r1 += d1[d] * dn[d]; }
			const half_t e0 = f2h(r0), e1 = f2h(r1);
			if (EMU) { // fbs_feature_slot_h
				const uint32_t q0 = 2u * level, q1 = q0 + 1u, p0 = q0 < 13u ? q0 : q0 + 3u, p1 = q1 < 13u ? q1 : q1 + 3u;
				X[lane * S32 + p0] = f0; X[lane * S32 + p1] = f1;
				D[lane * S32 + p0] = e0; D[lane * S32 + p1] = e1;
			} else {
				Xrow[level] = pack_h2(f0, f1);
				Drow[level] = pack_h2(e0, e1);
			}
		}
		{ // x y z (fill_positions_view_with_fixed_offset: half arithmetic, common_operation.cuh:187-199) | pad ; dn | pad
			const half_t px = f2h(c[0]) - (half_t)0.5f, py = f2h(c[1]) - (half_t)0.5f, pz = f2h(c[2]) - (half_t)0.5f;
			const half_t n0 = f2h(dn[0]), n1 = f2h(dn[1]), n2 = f2h(dn[2]);
			if (EMU) {
				half_t* xr = X + lane * S32; half_t* dr = D + lane * S32;
				xr[13] = px; xr[14] = py; xr[15] = pz; xr[31] = (half_t)0.f;
				dr[13] = n0; dr[14] = n1; dr[15] = n2; dr[31] = (half_t)0.f;
			} else {
				Xrow[14] = pack_h2(px, py); Xrow[15] = pack_h2(pz, (half_t)0.f);
				Drow[14] = pack_h2(n0, n1); Drow[15] = pack_h2(n2, (half_t)0.f);
			}
		}
		Z[lane] = dout[3];
		var_sum += h2f(dout[7]); // variance gradient
		wave_lds_sync();
		// FULL: dso = dL/d(colour input)[0:16], row 0 += dL/dsdf (add_density_gradient), as fragments: lane (r16, hq) holds rows 8 hq .. 8 hq + 7 of sample
		// 16 nt + r16 (zeros beyond row 15) -- the B operand of W1^T dso and, with the roles swapped, the A operand of its transpose
		auto load_fso = [&](const int nt) { // (loaded where it is used, twice per tile: 16 registers less across the encode-free middle of the tile)
			h8 v = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
			if (hq < 2) v = *reinterpret_cast<const h8*>(SO + (16 * nt + r16) * SO_STRIDE + 8 * hq);
			return v;
		};
		if (SLICED) { // dso's row 0 (FULL: all 16 rows, from their transposed fragments below); the two input tiles leave below, as the transposed fragments of the round-3 form
			static_assert(!SLICED || EMU, "the sliced weight gradients belong to the half mode");
			if (!FULL) st32(T.dso, s, dout[3]);
		}
		{ // ---- weight gradients of this tile (see the header) ----
			h8 fso[4];
			if (FULL) {
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) fso[nt] = load_fso(nt);
			}
			h8 ain[4], add[4];
			h4 d3v[4];
#pragma unroll
			for (int mt = 0; mt < 4; ++mt) {
				ain[mt] = *reinterpret_cast<const h8*>(X + (16 * mt + r16) * S32 + 8 * hq);
				add[mt] = *reinterpret_cast<const h8*>(D + (16 * mt + r16) * S32 + 8 * hq);
				d3v[mt] = *reinterpret_cast<const h4*>(Z + 16 * mt + 4 * hq);
			}
			const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
			h8 a_so[2]; // FULL: dso^T, lane = output row o = r16, registers <-> samples (A operand of dW1 += dso z1^T)
			if (FULL) {
				h8 idf;
#pragma unroll
				for (int j = 0; j < 8; ++j) idf[j] = (8 * hq + j == r16) ? (half_t)1.f : (half_t)0.f;
#pragma unroll
				for (int ks = 0; ks < 2; ++ks)
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const f4 t = __builtin_amdgcn_mfma_f32_16x16x32_f16(fso[2 * ks + h], idf, zero4, 0, 0, 0);
#pragma unroll
						for (int r = 0; r < 4; ++r) a_so[ks][4 * h + r] = f2h(t[r]);
					}
				if (SLICED) { // dso^T: lane = output row r16, register 4 h + r <-> sample 32 ks + 16 h + 4 hq + r
#pragma unroll
					for (int ks = 0; ks < 2; ++ks)
#pragma unroll
						for (int h = 0; h < 2; ++h) {
							const uint32_t off = (uint32_t)r16 * B + tile * TILE + 16u * (2 * ks + h) + 4u * hq;
							st32(reinterpret_cast<h4*>(T.dso), off / 4u, h4{a_so[ks][4 * h], a_so[ks][4 * h + 1], a_so[ks][4 * h + 2], a_so[ks][4 * h + 3]});
						}
				}
			}
			// B operands: lane = input column (tile order), registers j = 4 h + r <-> sample 32 ks + 16 h + 4 hq + r
			h8 b_in[2][2], b_dd[2][2];
#pragma unroll
			for (int n2 = 0; n2 < 2; ++n2) {
				h8 idf;
#pragma unroll
				for (int j = 0; j < 8; ++j) idf[j] = (8 * hq + j == 16 * n2 + r16) ? (half_t)1.f : (half_t)0.f;
#pragma unroll
				for (int ks = 0; ks < 2; ++ks)
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const f4 ti = __builtin_amdgcn_mfma_f32_16x16x32_f16(ain[2 * ks + h], idf, zero4, 0, 0, 0);
						const f4 td = __builtin_amdgcn_mfma_f32_16x16x32_f16(add[2 * ks + h], idf, zero4, 0, 0, 0);
#pragma unroll
						for (int r = 0; r < 4; ++r) { b_in[n2][ks][4 * h + r] = f2h(ti[r]); b_dd[n2][ks][4 * h + r] = f2h(td[r]); }
					}
			}
			if (SLICED) { // the two input tiles, feature-major in the reference's column order: lane = slot 16 n2 + r16 of the tile, four consecutive samples per 8-byte store
#pragma unroll
				for (int n2 = 0; n2 < 2; ++n2) {
					const uint32_t row = (uint32_t)fbs_logical_h(16 * n2 + r16);
#pragma unroll
					for (int ks = 0; ks < 2; ++ks)
#pragma unroll
						for (int h = 0; h < 2; ++h) {
							const uint32_t off = row * B + tile * TILE + 16u * (2 * ks + h) + 4u * hq;
							st32(reinterpret_cast<h4*>(T.sdfin), off / 4u, h4{b_in[n2][ks][4 * h], b_in[n2][ks][4 * h + 1], b_in[n2][ks][4 * h + 2], b_in[n2][ks][4 * h + 3]});
							st32(reinterpret_cast<h4*>(T.ddin), off / 4u, h4{b_dd[n2][ks][4 * h], b_dd[n2][ks][4 * h + 1], b_dd[n2][ks][4 * h + 2], b_dd[n2][ks][4 * h + 3]});
						}
				}
			}
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) { // hidden units 16 nt + r16
				const h8 wbn = *reinterpret_cast<const h8*>(wts + SW_S0 + (16 * nt + r16) * S32 + 8 * hq);
				// EMU: the two k-steps as K = 16 operands (slots 4 hq .. + 3 of either half of the row, fbs_logical_h)
				const h4 wbn0 = *reinterpret_cast<const h4*>(wts + SW_S0 + (16 * nt + r16) * S32 + 4 * hq), wbn1 = *reinterpret_cast<const h4*>(wts + SW_S0 + (16 * nt + r16) * S32 + 16 + 4 * hq);
				const half_t w1h = wts[SW_W1N + 16 * nt + r16];
				const float w1f = h2f(w1h);
				h8 wb1t;
				if (FULL) wb1t = *reinterpret_cast<const h8*>(wts + SW_W1T + (16 * nt + r16) * S32 + 8 * hq);
				h8 a_dz[2], a_dz1[2], b_z1[2], a_fr[2];
#pragma unroll
				for (int ks = 0; ks < 2; ++ks)
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const int mt = 2 * ks + h;
						f4 z, fr;
						if (EMU) {
							const half_t* xr = X + (16 * mt + r16) * S32 + 4 * hq; const half_t* dr = D + (16 * mt + r16) * S32 + 4 * hq;
							z = mfma_emul16_k16(*reinterpret_cast<const h4*>(xr), *reinterpret_cast<const h4*>(xr + 16), wbn0, wbn1, zero4);
							fr = mfma_emul16_k16(*reinterpret_cast<const h4*>(dr), *reinterpret_cast<const h4*>(dr + 16), wbn0, wbn1, zero4);
						} else {
							z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ain[mt], wbn, zero4, 0, 0, 0);
							fr = __builtin_amdgcn_mfma_f32_16x16x32_f16(add[mt], wbn, zero4, 0, 0, 0);
						}
						f4 dzt = zero4;
						if (FULL) dzt = __builtin_amdgcn_mfma_f32_16x16x32_f16(fso[mt], wb1t, zero4, 0, 0, 0); // (W1^T dso)^T: lane = hidden unit, registers = samples
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const half_t zh = f2h(z[r]);
							const bool on = zh > (half_t)0.f; // relu' tests the stored half activation (common_device.h:182 ff.)
							const float d = h2f(d3v[mt][r]);
							if (!FULL && !SLICED) acc_w1[nt] += on ? d * h2f(zh) : 0.f;
							if (!SLICED) acc_w1b[nt] += on ? h2f(f2h(fr[r])) : 0.f;
							a_dz[ks][4 * h + r] = on ? (FULL ? f2h(dzt[r]) : f2h(w1f * d)) : (half_t)0.f;
							a_dz1[ks][4 * h + r] = on ? w1h : (half_t)0.f;
							if (FULL || SLICED) b_z1[ks][4 * h + r] = on ? zh : (half_t)0.f;
							if (SLICED) a_fr[ks][4 * h + r] = on ? f2h(fr[r]) : (half_t)0.f;
						}
						if (SLICED) { // row = hidden unit 16 nt + r16, four consecutive samples 16 mt + 4 hq .. + 3 of the tile per lane
							const uint32_t off = (16u * nt + r16) * B + tile * TILE + 16u * mt + 4u * hq;
							st32(reinterpret_cast<h4*>(T.dz), off / 4u, h4{a_dz[ks][4 * h], a_dz[ks][4 * h + 1], a_dz[ks][4 * h + 2], a_dz[ks][4 * h + 3]});
							st32(reinterpret_cast<h4*>(T.dz1), off / 4u, h4{a_dz1[ks][4 * h], a_dz1[ks][4 * h + 1], a_dz1[ks][4 * h + 2], a_dz1[ks][4 * h + 3]});
							st32(reinterpret_cast<h4*>(T.z1), off / 4u, h4{b_z1[ks][4 * h], b_z1[ks][4 * h + 1], b_z1[ks][4 * h + 2], b_z1[ks][4 * h + 3]});
							st32(reinterpret_cast<h4*>(T.front), off / 4u, h4{a_fr[ks][4 * h], a_fr[ks][4 * h + 1], a_fr[ks][4 * h + 2], a_fr[ks][4 * h + 3]});
						}
					}
				if (FULL && !SLICED) {
#pragma unroll
					for (int ks = 0; ks < 2; ++ks) acc_w1f[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_so[ks], b_z1[ks], acc_w1f[nt], 0, 0, 0);
				}
#pragma unroll
				for (int ks = 0; ks < (SLICED ? 0 : 2); ++ks)
#pragma unroll
					for (int ni = 0; ni < 2; ++ni) {
						acc_w0[nt][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_dz[ks], b_in[ni][ks], acc_w0[nt][ni], 0, 0, 0);
						acc_w0b[nt][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_dz1[ks], b_dd[ni][ks], acc_w0b[nt][ni], 0, 0, 0);
					}
			}
		}
		// z1 = relu(W0 in), kept as next-layer fragments; its relu' mask in fragment order
		h8 bz[4][2];
		{
			if (EMU) { // the reference's two k-steps as K = 16 MFMAs on the two halves of the rows (fbs_logical_h)
				h4 b0[4], b1[4];
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) { const half_t* xr = X + (16 * nt + r16) * S32 + 4 * hq; b0[nt] = *reinterpret_cast<const h4*>(xr); b1[nt] = *reinterpret_cast<const h4*>(xr + 16); }
#pragma unroll
				for (int ks = 0; ks < 2; ++ks) {
					f4 acc2[2][4];
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const half_t* wr = wts + SW_S0 + (16 * (2 * ks + h) + r16) * S32 + 4 * hq;
						const h4 wa0 = *reinterpret_cast<const h4*>(wr), wa1 = *reinterpret_cast<const h4*>(wr + 16);
#pragma unroll
						for (int nt = 0; nt < 4; ++nt) acc2[h][nt] = mfma_emul16_k16(wa0, wa1, b0[nt], b1[nt], f4{0.f, 0.f, 0.f, 0.f});
					}
					chain_pack_ks<true>(acc2, bz, ks);
				}
			} else {
				f4 acc[4][4];
				zero_acc<4>(acc);
				mfma_layer<4, 1, EMU_OFF>(wts + SW_S0, S32, X, S32, acc, lane);
				chain_pack<true>(acc, bz);
			}
		}
		half_t d3[4];
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) d3[nt] = Z[16 * nt + r16];
		// dz1 = W1[0,:] (.) relu'(z1) (the stored half activation is tested, common_device.h:182 ff.); dz = dL/dsdf * dz1
		h8 bdz[4][2];
		if (FULL) { // W1^T dso (K = 16) sample tile by sample tile, masked below
			h8 w1t[4];
#pragma unroll
			for (int mt = 0; mt < 4; ++mt) w1t[mt] = *reinterpret_cast<const h8*>(wts + SW_W1T + (16 * mt + r16) * S32 + 8 * hq);
			const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int nt = 0; nt < 4; ++nt) {
				const h8 v = load_fso(nt);
				f4 ac[4];
#pragma unroll
				for (int mt = 0; mt < 4; ++mt) ac[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1t[mt], v, zero4, 0, 0, 0);
#pragma unroll
				for (int ks = 0; ks < 2; ++ks)
#pragma unroll
					for (int j = 0; j < 8; ++j) bdz[nt][ks][j] = f2h(ac[2 * ks + (j >> 2)][j & 3]); // chain_pack's mapping
			}
		}
#pragma unroll
		for (int ks = 0; ks < 2; ++ks) {
			const h8 w1 = *reinterpret_cast<const h8*>(wts + SW_W1 + 32 * ks + 8 * hq);
#pragma unroll
			for (int nt = 0; nt < 4; ++nt)
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const bool on = bz[nt][ks][j] > (half_t)0.f;
					bz[nt][ks][j] = on ? w1[j] : (half_t)0.f;                                   // bz now holds dz1
					bdz[nt][ks][j] = on ? (FULL ? bdz[nt][ks][j] : f2h(h2f(w1[j]) * h2f(d3[nt]))) : (half_t)0.f;
				}
		}
		wave_lds_sync(); // X (network input) consumed by the first GEMM
		{ // d sdf / d in = W0^T dz1 -> rows of X
			f4 acc[2][4];
			zero_acc<2>(acc);
			mfma_layer_regs<2, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + SW_S0T, S64, bz, acc, lane);
			store_acc<2, false>(acc, X, S32, 0, lane);
		}
		wave_lds_sync();
#pragma unroll
		for (uint32_t l = 0; l < 14; ++l) st32(T.g12, (l * B + s) * 2u + 1u, Xrow[l]); // d sdf / d feat of that level (dL_denc_output of the double backward)
		wave_lds_sync(); // rows of X read
		{ // dL/d in = W0^T dz -> rows of X; its feature columns are the first-order dL/dfeat of the grid
			f4 acc[2][4];
			zero_acc<2>(acc);
			mfma_layer_regs<2, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + SW_S0T, S64, bdz, acc, lane);
			store_acc<2, false>(acc, X, S32, 0, lane);
		}
		wave_lds_sync();
#pragma unroll
		for (uint32_t l = 0; l < 14; ++l) st32(T.g12, (l * B + s) * 2u + 0u, Xrow[l]);
		wave_lds_sync(); // (front = (W0 ddin) (.) relu'(z1), fully_fused_mlp.cu:1097-1107, is rebuilt by k_dw_sdf from the exported rows)
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) var_sum += __shfl_down(var_sum, off, 64);
	if (lane == 0) T.var_partial[blockIdx.x * WAVES_PER_WG + wave] = var_sum;
	// The four wavefronts' weight gradients, summed in a fixed order, leave as ONE partial per workgroup in k_dw's layout. The per-wave
	// tiles are dead: their LDS is the staging area. D layout: lane = input column 16 ni + r16 (tile order), register r = hidden unit 16 mo + 4 hq + r.
	if (SLICED) return; // (k_dw_sliced forms the weight gradients from the exported operands)
	float* red = reinterpret_cast<float*>(wts + W_END);
	static_assert((size_t)WAVES_PER_WG * FBS_WAVE_HALFS * sizeof(half_t) >= (size_t)WAVES_PER_WG * 64 * 32 * sizeof(float), "staging area of the weight-gradient partials");
	constexpr int N = 64 * 32;
	for (int pass = 0; pass < 2; ++pass) {
		__syncthreads();
#pragma unroll
		for (int mo = 0; mo < 4; ++mo)
#pragma unroll
			for (int ni = 0; ni < 2; ++ni)
#pragma unroll
				for (int r = 0; r < 4; ++r) red[wave * N + (16 * mo + 4 * hq + r) * 32 + (EMU ? fbs_logical_h(16 * ni + r16) : fbs_logical(16 * ni + r16))] = pass ? acc_w0b[mo][ni][r] : acc_w0[mo][ni][r];
		__syncthreads();
		float* dst = (pass ? a.dw_w0b : a.dw_w0) + (size_t)blockIdx.x * N;
		for (int q = threadIdx.x; q < N; q += WG) dst[q] = ((red[q] + red[N + q]) + red[2 * N + q]) + red[3 * N + q];
	}
	if (FULL) { // dW1 (first order), all 16 rows: D layout lane = hidden unit 16 nt + r16, register r = row 4 hq + r
		__syncthreads();
#pragma unroll
		for (int nt = 0; nt < 4; ++nt)
#pragma unroll
			for (int r = 0; r < 4; ++r) red[wave * (16 * 64) + (4 * hq + r) * 64 + 16 * nt + r16] = acc_w1f[nt][r];
		__syncthreads();
		float* dst = a.dw_w1 + (size_t)blockIdx.x * (16 * 64);
		for (int q = threadIdx.x; q < 16 * 64; q += WG) dst[q] = ((red[q] + red[16 * 64 + q]) + red[2 * 16 * 64 + q]) + red[3 * 16 * 64 + q];
	}
	// row 0 of the two 16x64 gradients of W1: per lane the sum over its samples; 16 lane groups (wave, hq) per hidden unit
	for (int pass = FULL ? 1 : 0; pass < 2; ++pass) {
		__syncthreads();
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) red[(wave * 4 + hq) * 64 + 16 * nt + r16] = pass ? acc_w1b[nt] : acc_w1[nt];
		__syncthreads();
		float* dst = (pass ? a.dw_w1b : a.dw_w1) + (size_t)blockIdx.x * (16 * 64);
		for (int q = threadIdx.x; q < 16 * 64; q += WG) {
			float v = 0.f;
			if (q < 64) for (int g = 0; g < 16; ++g) v += red[g * 64 + q];
			dst[q] = v; // rows 1..15 of dso are exact zeros here (TrainArgs::skip_rgb); the second-order gradient of W1 has row 0 only in either mode
		}
	}
}

__global__ __launch_bounds__(WG, 2) void k_fwd_bwd_sdf(const GridMeta G, const NetW net, const TrainArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	fwd_bwd_sdf_body<false>(G, net, a, smem_raw, lm);
}
__global__ __launch_bounds__(WG, 2) void k_fwd_bwd_sdf_full(const GridMeta G, const NetW net, const TrainArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	fwd_bwd_sdf_body<true>(G, net, a, smem_raw, lm);
}
// rnb_config::accumulate = RNB_ACCUM_HALF
__global__ __launch_bounds__(WG, 2) void k_fwd_bwd_sdf_h(const GridMeta G, const NetW net, const TrainArgs a) { // (one workgroup per CU, 352 VGPRs without spills: 137 vs 119 us)
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	fwd_bwd_sdf_body<false, true>(G, net, a, smem_raw, lm);
}
__global__ __launch_bounds__(WG, 2) void k_fwd_bwd_sdf_hs(const GridMeta G, const NetW net, const TrainArgs a) { // half mode, weight-gradient operands exported for k_dw_sliced
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	fwd_bwd_sdf_body<false, true, true>(G, net, a, smem_raw, lm);
}
__global__ __launch_bounds__(WG, 2) void k_fwd_bwd_sdf_full_hs(const GridMeta G, const NetW net, const TrainArgs a) { // albedo mode's part 2 in the half mode, operands exported for k_dw_sliced
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	fwd_bwd_sdf_body<true, true, true>(G, net, a, smem_raw, lm);
}
__global__ __launch_bounds__(WG, 2) void k_fwd_bwd_sdf_full_h(const GridMeta G, const NetW net, const TrainArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	fwd_bwd_sdf_body<true, true>(G, net, a, smem_raw, lm);
}

// ---------------------------------------------------------------------------------------------
// K10 + K11 with the colour MLP (albedo mode, nerf_network.h:257-452), part 1 of 2: the colour MLP alone -- forward, backward to its
// input, and its three weight gradients -- on the compacted batch. Its input rows [sdf_out 16 | x y z | grad sdf | 0] were written by the
// network evaluation of this step (FwdArgs::cin_out: same weights, so the training pass need not evaluate the SDF MLP twice, and --
// the point -- need not keep 84 dy/dx values per sample alive across the colour MLP: part 2, k_fwd_bwd_sdf_full, receives
// dL/d(input row) from here and consumes dy/dx level by level like the --no-albedo kernel). No hash-grid access, no LDS tiles:
//   h1 = relu(W0 cin)   h2 = relu(W1 h1)   (rgb = W2 h2 is not needed: no output activation, fully_fused_mlp.cu:936-940)
//   dh2 = (W2^T dr) (.) relu'(h2), dr = rows 0..2 of dL/doutput     dh1 = (W1^T dh2) (.) relu'(h1)     dcin = W0^T dh1
//   dW2 += dr h2^T   dW1 += dh2 h1^T   dW0 += dh1 cin^T            (fully_fused_mlp.cu:953-1030)
// The data path is register-chained (mlp.cuh: "8 features of one sample per lane"). The weight-gradient GEMMs have K = samples and take
// "8 samples of one feature per lane": every operand they need is the TRANSPOSE of a chained fragment, made by an MFMA against the identity
// (exact: one non-zero product per output) -- the matrix pipe is 3 % busy on this path, the vector ALU is what the kernel shares with the
// march of the next step that runs beside it, so everything that can be a matrix instruction is one (the first version recomputed the
// transposed copies with swapped operands and repeated the relu / mask / W2^T dr arithmetic on the VALU: 3500 vector instructions per
// tile, 211 us beside the march; this one: see DESIGN.md). A transposed fragment holds the features in CHAIN order (lane r16 of fragment q =
// chained position 16 q + r16), so the gradients come out with rows / columns permuted by chain_logical, undone when the partials are stored.
// One partial per workgroup in k_dw's layout for k_dw_finish.
// ---------------------------------------------------------------------------------------------
constexpr int RW_C0 = 0;                      // [64][S32] rgb W0, compact columns in natural order
constexpr int RW_C1 = RW_C0 + 64 * S32;       // [64][S64] rgb W1, columns in chain order
constexpr int RW_C1T = RW_C1 + 64 * S64;      // [64][S64] rgb W1^T (row = h1 unit), columns = h2 units in chain order
constexpr int RW_C0T = RW_C1T + 64 * S64;     // [32][S64] rgb W0^T (compact rows), columns = h1 units in chain order
constexpr int RW_C2T = RW_C0T + 32 * S64;     // [64][S32] rgb W2^T (row = h2 unit), columns = the 16 outputs + zeros
constexpr int RW_END = RW_C2T + 64 * S32;     // 16 640 halfs
static_assert(RW_END % 8 == 0, "16-byte copies of the image");
constexpr size_t LDS_RGB = std::max((size_t)RW_END * sizeof(half_t), (size_t)WAVES_PER_WG * 64 * 64 * sizeof(float));

__device__ inline void load_weights_rgb(half_t* __restrict__ w, const NetW& net, const int tid, const int nthreads) {
	auto col = [](int c) { return c < 16 ? c : c + 16; }; // compact input index -> column of the 48-wide W0 (the 16 direction columns receive zeros)
	for (int i = tid; i < 64 * 32; i += nthreads) {
		const int o = i >> 5, c = i & 31;
		w[RW_C0 + o * S32 + c] = net.rgb_w0[o * 48 + col(c)];
		w[RW_C2T + o * S32 + c] = c < 16 ? net.rgb_w2[c * 64 + o] : (half_t)0.f;
	}
	for (int i = tid; i < 64 * 64; i += nthreads) {
		const int o = i >> 6, p = i & 63;
		w[RW_C1 + o * S64 + p] = net.rgb_w1[o * 64 + chain_logical(p)];
		w[RW_C1T + o * S64 + p] = net.rgb_w1[chain_logical(p) * 64 + o];
	}
	for (int i = tid; i < 32 * 64; i += nthreads) { const int c = i >> 6, p = i & 63; w[RW_C0T + c * S64 + p] = net.rgb_w0[chain_logical(p) * 48 + col(c)]; }
	// (row padding of the images is never read)
}

struct RgbArgs {
	const half_t* cin;        // [slot][32] FwdArgs::cin_out
	const uint32_t* src_slot; // [B] slot of every compacted sample (null: sample s is slot s)
	const half_t* dout;       // [B][16]
	half_t* dcin;             // [B][32] dL/d(input row): columns 0..15 -> dL/d sdf_out, 19..21 -> dL/d(grad sdf)
	uint32_t B;
	const half_t* wimg;       // optional: load_weights_rgb's image prepared by k_prepare_weight_images
	float *dw_c0, *dw_c1, *dw_c2; // one partial per workgroup: [64][32] compact, [64][64], [16][64]
	TrainScratch t;               // SLICED: h2, dr, h1, dh2, dh1, cin leave feature-major for k_dw_sliced
};

// Transpose of chained fragments: in[ms][ks] (lane = sample 16 ms + r16, K slot 8 hq + j = chained position 32 ks + 8 hq + j) ->
// out[q][ks2] (lane = chained position 16 q + r16, register 4 h + r <-> sample 32 ks2 + 16 h + 4 hq + r), KS k-steps of 32 positions.
template <int KS>
__device__ __forceinline__ void transpose_frags(const h8 (&in)[4][KS], h8 (&out)[2 * KS][2], const int r16, const int hq) {
	const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int n2 = 0; n2 < 2; ++n2) {
		h8 idf;
#pragma unroll
		for (int j = 0; j < 8; ++j) idf[j] = (8 * hq + j == 16 * n2 + r16) ? (half_t)1.f : (half_t)0.f;
#pragma unroll
		for (int ks = 0; ks < KS; ++ks)
#pragma unroll
			for (int ms = 0; ms < 4; ++ms) {
				const f4 t = __builtin_amdgcn_mfma_f32_16x16x32_f16(in[ms][ks], idf, zero4, 0, 0, 0);
#pragma unroll
				for (int r = 0; r < 4; r += 2) {
					out[2 * ks + n2][ms >> 1][4 * (ms & 1) + r] = f2h(t[r]);
					out[2 * ks + n2][ms >> 1][4 * (ms & 1) + r + 1] = f2h(t[r + 1]);
				}
			}
	}
}

// SLICED: a transposed fragment set (lane = row 16 q + r16 -- a chained position or a natural index --, register 4 h + r <-> sample 32 ks + 16 h + 4 hq + r of the tile) leaves
// feature-major for k_dw_sliced: four consecutive samples of one row per 8-byte store.
template <int NQ>
__device__ __forceinline__ void export_transposed(const h8 (&t)[NQ][2], const int n_q, half_t* __restrict__ dst, const uint32_t B, const uint32_t tile, const int r16, const int hq, const bool chained) {
#pragma unroll
	for (int q = 0; q < NQ; ++q) {
		if (q >= n_q) break;
		const uint32_t row = chained ? (uint32_t)chain_logical(16 * q + r16) : (uint32_t)(16 * q + r16);
#pragma unroll
		for (int ks = 0; ks < 2; ++ks)
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const uint32_t off = row * B + tile * TILE + 16u * (2 * ks + h) + 4u * hq;
				st32(reinterpret_cast<h4*>(dst), off / 4u, h4{t[q][ks][4 * h], t[q][ks][4 * h + 1], t[q][ks][4 * h + 2], t[q][ks][4 * h + 3]});
			}
	}
}

template <bool EMU, bool SLICED = false>
__device__ __forceinline__ void rgb_fwd_bwd_body(const NetW& net, const RgbArgs& a, char* smem_raw) {
	half_t* wts = reinterpret_cast<half_t*>(smem_raw);
	if (a.wimg) copy_weight_image(wts, a.wimg, RW_END, threadIdx.x, WG);
	else load_weights_rgb(wts, net, threadIdx.x, WG);
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int r16 = lane & 15, hq = lane >> 4;
	const uint32_t n_tiles = a.B / TILE;
	const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
	const h8 zero8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
	f4 acc_w1[4][4], acc_w0[4][2], acc_w2[4]; // rows / columns in chain order (see the header), dW2: rows = outputs 4 hq + r
#pragma unroll
	for (int q = 0; q < 4; ++q) {
#pragma unroll
		for (int ni = 0; ni < 4; ++ni) acc_w1[q][ni] = zero4;
		acc_w0[q][0] = zero4; acc_w0[q][1] = zero4; acc_w2[q] = zero4;
	}
	// this tile's rows: input fragments (lane (r16, hq): columns 8 hq .. + 7 of sample 16 nt + r16) and dr fragments (rows 0..2 of dL/doutput in K slots
	// 0..2 of the hq == 0 lanes, zeros elsewhere); the next tile's are requested before this tile's arithmetic (one wavefront per SIMD hides nothing)
	auto load_tile = [&](const uint32_t tile, h8 (&cf)[4][1], h8 (&drf)[4][1]) {
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) {
			const uint32_t sidx = tile * TILE + 16 * nt + r16;
			const uint32_t slot = a.src_slot ? a.src_slot[sidx] : sidx;
			cf[nt][0] = *reinterpret_cast<const h8*>(a.cin + (size_t)slot * 32 + 8 * hq);
			h8 d = zero8;
			if (hq == 0) { const h4 v = *reinterpret_cast<const h4*>(a.dout + (size_t)sidx * 16); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
			drf[nt][0] = d;
		}
	};
	const uint32_t tile0 = blockIdx.x * WAVES_PER_WG + wave, tstride = gridDim.x * WAVES_PER_WG;
	h8 cf_n[4][1], drf_n[4][1];
	if (tile0 < n_tiles) load_tile(tile0, cf_n, drf_n);
	for (uint32_t tile = tile0; tile < n_tiles; tile += tstride) {
		const uint32_t s0 = tile * TILE;
		h8 cf[4][1], drf[4][1];
#pragma unroll
		for (int nt = 0; nt < 4; ++nt) { cf[nt][0] = cf_n[nt][0]; drf[nt][0] = drf_n[nt][0]; }
		if (tile + tstride < n_tiles) load_tile(tile + tstride, cf_n, drf_n);
		// ---- forward
		h8 bh1[4][2], bh2[4][2];
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer_regs<4, 1, EMU ? EMU_NATURAL : EMU_OFF>(wts + RW_C0, S32, cf, acc, lane); // (the input rows are in the reference's column order)
			chain_pack<true>(acc, bh1);
		}
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer_regs<4, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + RW_C1, S64, bh1, acc, lane);
			chain_pack<true>(acc, bh2);
		}
		// ---- dW2 += dr h2^T (the transposes of h2 and dr)
		{
			h8 th2[4][2], tdr[2][2];
			transpose_frags<2>(bh2, th2, r16, hq);
			transpose_frags<1>(drf, tdr, r16, hq); // tdr[0]: lane = output row r16; tdr[1] (columns 16..31) is zero
			if (SLICED) { export_transposed<4>(th2, 4, a.t.h2, a.B, tile, r16, hq, true); export_transposed<2>(tdr, 1, a.t.dr, a.B, tile, r16, hq, false); }
#pragma unroll
			for (int qi = 0; qi < (SLICED ? 0 : 4); ++qi)
#pragma unroll
				for (int ks = 0; ks < 2; ++ks) acc_w2[qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tdr[0][ks], th2[qi][ks], acc_w2[qi], 0, 0, 0);
		}
		// ---- dh2 = (W2^T dr) (.) relu'(h2) (a K = 16 MFMA whose rows 3..15 meet zeros), in place of h2
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer_regs<4, 1>(wts + RW_C2T, S32, drf, acc, lane);
			h8 d[4][2];
			chain_pack<false>(acc, d);
#pragma unroll
			for (int nt = 0; nt < 4; ++nt)
#pragma unroll
				for (int ks = 0; ks < 2; ++ks)
#pragma unroll
					for (int j = 0; j < 8; ++j) bh2[nt][ks][j] = (bh2[nt][ks][j] > (half_t)0.f) ? d[nt][ks][j] : (half_t)0.f;
		}
		// ---- dW1 += dh2 h1^T
		{
			h8 th1[4][2], tdh2[4][2];
			transpose_frags<2>(bh1, th1, r16, hq);
			transpose_frags<2>(bh2, tdh2, r16, hq);
			if (SLICED) { export_transposed<4>(th1, 4, a.t.h1, a.B, tile, r16, hq, true); export_transposed<4>(tdh2, 4, a.t.dh2, a.B, tile, r16, hq, true); }
#pragma unroll
			for (int qo = 0; qo < (SLICED ? 0 : 4); ++qo)
#pragma unroll
				for (int qi = 0; qi < 4; ++qi)
#pragma unroll
					for (int ks = 0; ks < 2; ++ks) acc_w1[qo][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tdh2[qo][ks], th1[qi][ks], acc_w1[qo][qi], 0, 0, 0);
		}
		// ---- dh1 = (W1^T dh2) (.) relu'(h1), in place of h1
		{
			f4 acc[4][4];
			zero_acc<4>(acc);
			mfma_layer_regs<4, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + RW_C1T, S64, bh2, acc, lane);
			h8 d[4][2];
			chain_pack<false>(acc, d);
#pragma unroll
			for (int nt = 0; nt < 4; ++nt)
#pragma unroll
				for (int ks = 0; ks < 2; ++ks)
#pragma unroll
					for (int j = 0; j < 8; ++j) bh1[nt][ks][j] = (bh1[nt][ks][j] > (half_t)0.f) ? d[nt][ks][j] : (half_t)0.f;
		}
		// ---- dcin = W0^T dh1 -> memory (rows of 32 halfs; this lane: columns 16 mt + 4 hq .. + 3 of sample 16 nt + r16)
		{
			f4 acc[2][4];
			zero_acc<2>(acc);
			mfma_layer_regs<2, 2, EMU ? EMU_CHAINED : EMU_OFF>(wts + RW_C0T, S64, bh1, acc, lane);
#pragma unroll
			for (int mt = 0; mt < 2; ++mt)
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) {
					const h4 v = {f2h(acc[mt][nt][0]), f2h(acc[mt][nt][1]), f2h(acc[mt][nt][2]), f2h(acc[mt][nt][3])};
					*reinterpret_cast<h4*>(a.dcin + (size_t)(s0 + 16 * nt + r16) * 32 + 16 * mt + 4 * hq) = v;
				}
		}
		// ---- dW0 += dh1 cin^T (cin is in natural column order, so its transpose is too)
		{
			h8 tdh1[4][2], tc[2][2];
			transpose_frags<2>(bh1, tdh1, r16, hq);
			transpose_frags<1>(cf, tc, r16, hq);
			if (SLICED) { export_transposed<4>(tdh1, 4, a.t.dh1, a.B, tile, r16, hq, true); export_transposed<2>(tc, 2, a.t.cin, a.B, tile, r16, hq, false); }
#pragma unroll
			for (int qo = 0; qo < (SLICED ? 0 : 4); ++qo)
#pragma unroll
				for (int ni = 0; ni < 2; ++ni)
#pragma unroll
					for (int ks = 0; ks < 2; ++ks) acc_w0[qo][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tdh1[qo][ks], tc[ni][ks], acc_w0[qo][ni], 0, 0, 0);
		}
	}
	if (SLICED) return; // (k_dw_sliced forms the weight gradients from the exported operands)
	// The four wavefronts' weight gradients, summed in a fixed order, leave as ONE partial per workgroup in k_dw's layout; the weight image is dead.
	// D layout of the accumulators: lane r16 = column of the B operand, register r = row 4 hq + r of the A operand, both possibly in chain order.
	float* red = reinterpret_cast<float*>(smem_raw);
	{
		constexpr int N = 64 * 64;
		__syncthreads();
#pragma unroll
		for (int qo = 0; qo < 4; ++qo)
#pragma unroll
			for (int qi = 0; qi < 4; ++qi)
#pragma unroll
				for (int r = 0; r < 4; ++r) red[wave * N + chain_logical(16 * qo + 4 * hq + r) * 64 + chain_logical(16 * qi + r16)] = acc_w1[qo][qi][r];
		__syncthreads();
		float* dst = a.dw_c1 + (size_t)blockIdx.x * N;
		for (int q = threadIdx.x; q < N; q += WG) dst[q] = ((red[q] + red[N + q]) + red[2 * N + q]) + red[3 * N + q];
	}
	{
		constexpr int N = 64 * 32;
		__syncthreads();
#pragma unroll
		for (int qo = 0; qo < 4; ++qo)
#pragma unroll
			for (int ni = 0; ni < 2; ++ni)
#pragma unroll
				for (int r = 0; r < 4; ++r) red[wave * N + chain_logical(16 * qo + 4 * hq + r) * 32 + 16 * ni + r16] = acc_w0[qo][ni][r];
		__syncthreads();
		float* dst = a.dw_c0 + (size_t)blockIdx.x * N;
		for (int q = threadIdx.x; q < N; q += WG) dst[q] = ((red[q] + red[N + q]) + red[2 * N + q]) + red[3 * N + q];
	}
	{ // dW2 [16][64]: rows 3..15 are sums of exact zeros
		constexpr int N = 16 * 64;
		__syncthreads();
#pragma unroll
		for (int qi = 0; qi < 4; ++qi)
#pragma unroll
			for (int r = 0; r < 4; ++r) red[wave * N + (4 * hq + r) * 64 + chain_logical(16 * qi + r16)] = acc_w2[qi][r];
		__syncthreads();
		float* dst = a.dw_c2 + (size_t)blockIdx.x * N;
		for (int q = threadIdx.x; q < N; q += WG) dst[q] = ((red[q] + red[N + q]) + red[2 * N + q]) + red[3 * N + q];
	}
}
__global__ __launch_bounds__(WG, 1) void k_rgb_fwd_bwd(const NetW net, const RgbArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	rgb_fwd_bwd_body<false>(net, a, smem_raw);
}
__global__ __launch_bounds__(WG, 1) void k_rgb_fwd_bwd_h(const NetW net, const RgbArgs a) { // rnb_config::accumulate = RNB_ACCUM_HALF
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	rgb_fwd_bwd_body<true>(net, a, smem_raw);
}
__global__ __launch_bounds__(WG, 1) void k_rgb_fwd_bwd_hs(const NetW net, const RgbArgs a) { // half mode, weight-gradient operands exported for k_dw_sliced
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	rgb_fwd_bwd_body<true, true>(net, a, smem_raw);
}

// ---------------------------------------------------------------------------------------------
// Weight-gradient GEMM: dW[o][i] = sum_s Y[o][s] X[i][s], operands feature-major in global memory, so both MFMA
// fragments are 16-byte contiguous loads (no LDS). Each wavefront owns K-steps of 32 samples; per-wave partial
// results go to `partial` and are summed in a fixed order by k_dw_finish (deterministic).
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, bool ONES>
__device__ __forceinline__ void dw_body(const half_t* __restrict__ YT, const half_t* __restrict__ XT, const uint32_t B, const uint32_t chunk, float* __restrict__ partial,
                                        const uint32_t wg, float* __restrict__ red) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int r16 = lane & 15, hq = lane >> 4;
	f4 acc[MT][NT];
#pragma unroll
	for (int mt = 0; mt < MT; ++mt)
#pragma unroll
		for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f4{0.f, 0.f, 0.f, 0.f};
	const uint32_t s_begin = wg * chunk, s_end = s_begin + chunk;
	for (uint32_t s0 = s_begin + wave * 32; s0 < s_end; s0 += WAVES_PER_WG * 32) {
		h8 bfr[NT];
#pragma unroll
		for (int nt = 0; nt < NT; ++nt) bfr[nt] = *reinterpret_cast<const h8*>(XT + (size_t)(16 * nt + r16) * B + s0 + 8 * hq);
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) {
			h8 afr;
			if (ONES) {
				const half_t one = (r16 == 0 && mt == 0) ? (half_t)1.f : (half_t)0.f;
				afr = h8{one, one, one, one, one, one, one, one};
			} else {
				afr = *reinterpret_cast<const h8*>(YT + (size_t)(16 * mt + r16) * B + s0 + 8 * hq);
			}
#pragma unroll
			for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr, bfr[nt], acc[mt][nt], 0, 0, 0);
		}
	}
	// the four waves' partials are summed in LDS (fixed order) so that k_dw_finish reads one slab per workgroup, not per wave. (A one-image staging
	// -- 16 instead of 64 KB, so that the LDS-privatised coarse-level scatter fits beside a GEMM workgroup -- changed nothing measurable in the albedo mode
	// (0.809 vs 0.796 - 0.806 ms/step over the window on three boxes: round 3.)
	constexpr int N = MT * 16 * NT * 16;
#pragma unroll
	for (int mt = 0; mt < MT; ++mt)
#pragma unroll
		for (int nt = 0; nt < NT; ++nt)
#pragma unroll
			for (int r = 0; r < 4; ++r) red[wave * N + (16 * mt + 4 * hq + r) * (NT * 16) + 16 * nt + r16] = acc[mt][nt][r];
	__syncthreads();
	float* dst = partial + (size_t)wg * N;
	for (int q = threadIdx.x; q < N; q += WG) dst[q] = ((red[q] + red[N + q]) + red[2 * N + q]) + red[3 * N + q];
}

template <int MT, int NT, bool ONES>
__global__ __launch_bounds__(WG, 1) void k_dw(const half_t* __restrict__ YT, const half_t* __restrict__ XT, const uint32_t B, const uint32_t chunk, float* __restrict__ partial) {
	__shared__ float red[WAVES_PER_WG * MT * 16 * NT * 16];
	dw_body<MT, NT, ONES>(YT, XT, B, chunk, partial, blockIdx.x, red);
}

// All weight-gradient GEMMs of a step in ONE launch (the albedo mode; without the colour MLP the four GEMMs stay separate launches:
// together they take more from the scatter beside them than they gain, 0.709 vs 0.701 ms/step over the window). Workgroup blockIdx.x works on GEMM blockIdx.x / nwg. Launched one after the
// other (7 kernels with --has-albedo) each of these bandwidth streams ran alone beside the atomic-bound scatter and the chain
// became the long pole of the albedo mode (346 us for 65 us of work); together they overlap each other's latency. Two workgroups
// per CU (the largest instance: 200 registers, 64 KB of LDS).
enum DwKind : uint32_t { DW_1x4 = 0, DW_4x4, DW_4x2, DW_1x4_ONES };
struct DwAllArgs { const half_t* YT[7]; const half_t* XT[7]; float* partial[7]; uint32_t kind[7]; uint32_t n, nwg, B, chunk; };
__global__ __launch_bounds__(WG, 2) void k_dw_all(const DwAllArgs a) {
	__shared__ float red[WAVES_PER_WG * 64 * 64];
	const uint32_t g = blockIdx.x / a.nwg, wg = blockIdx.x - g * a.nwg;
	switch (a.kind[g]) {
		case DW_1x4: dw_body<1, 4, false>(a.YT[g], a.XT[g], a.B, a.chunk, a.partial[g], wg, red); break;
		case DW_4x4: dw_body<4, 4, false>(a.YT[g], a.XT[g], a.B, a.chunk, a.partial[g], wg, red); break;
		case DW_4x2: dw_body<4, 2, false>(a.YT[g], a.XT[g], a.B, a.chunk, a.partial[g], wg, red); break;
		default: dw_body<1, 4, true>(a.YT[g], a.XT[g], a.B, a.chunk, a.partial[g], wg, red); break;
	}
}

// rnb_config::accumulate = RNB_ACCUM_HALF (round 6; removes deviation D1'): the weight-gradient GEMMs in the REFERENCE's summation order. tcnn computes them with CUTLASS split-K
// (cutlass_matmul.h:83, 315-322: slices of 4096 samples, half accumulators, the slices' results reduced in half); the model of it is oracle/rnb_oracle.cpp emulated_dw: per output element
// and slice, 256 dependent k-steps -- the 16 products of a k-step (exact in fp32) added one after the other in fp32, the half accumulator + that sum rounded to half -- and the slices' results
// added in half in slice order. This kernel IS that statement: one thread per 4 output elements (o, i0 .. i0 + 3), the samples of its slice in order, the operands read from the feature-major
// arrays the training kernel exported (TrainScratch; L1 serves a lane's 64-byte line for 32 samples). Plain VALU fp32 adds in the stated order: bit-identical to the model on the same operands
// (RNB_PRIM_DW_SLICED); an MFMA would add a k-step's products in an order of its own (tools/probe_mfma_arith.hip). Every chain is 4096 samples long whatever the launch: ~90 us alone, ~235 us beside the scatter (profiles/r06_half_mode_sliced.txt).
// out[g][slice][n_out * n_in] floats holding half values; k_dw_finish (sliced) adds the slices in half.
constexpr uint32_t DW_SLICE = 4096; // cutlass_matmul.h:83
struct DwSlicedArgs {
	const half_t* YT[7]; const half_t* XT[7]; float* out[7];
	uint32_t n_out[7], n_out_live[7], n_in[7], ones[7]; // rows >= n_out_live are exact zeros (not computed: a zero operand row gives +0 at every step); ones: Y is the constant 1 (row 0)
	uint32_t first_wg[8];                                // workgroups [first_wg[g], first_wg[g + 1]) work on GEMM g: n_slices x blocks(g)
	uint32_t n, B, n_slices;
};
constexpr uint32_t DWS_CHUNK = 128, DWS_ROW = DWS_CHUNK + 8, DWS_MAX_ROWS = 80; // samples staged per step; halfs per staged row (+16 bytes: the 16-byte reads of different rows fall into different banks); Y rows + X rows of a workgroup
__global__ __launch_bounds__(256) void k_dw_sliced(const DwSlicedArgs a) {
	// The workgroup's operand rows travel through LDS in chunks of 128 samples, the next chunk's 16-byte pieces in flight (registers) while this one is summed: a lane that read its rows
	// itself waited one L2 round trip per k-step -- 256 dependent round trips per slice, 0.5 ms beside the scatter (profiles/r06_half_mode_sliced.txt).
	__shared__ __attribute__((aligned(16))) half_t stage[2][DWS_MAX_ROWS * DWS_ROW];
	uint32_t g = 0;
	while (g + 1 < a.n && blockIdx.x >= a.first_wg[g + 1]) ++g;
	const uint32_t n_in = a.n_in[g], per_row = n_in / 4u, n_threads = a.n_out[g] * per_row, blocks = (n_threads + 255u) / 256u;
	const uint32_t wl = blockIdx.x - a.first_wg[g], slice = wl / blocks, t0 = (wl - slice * blocks) * 256u, t = t0 + threadIdx.x;
	const uint32_t o = t / per_row, i0 = 4u * (t - o * per_row), B = a.B, n_live = a.n_out_live[g];
	const bool ones = a.ones[g] != 0u, live = t < n_threads && o < n_live;
	const uint32_t o0 = t0 / per_row;                                                    // first output row of this workgroup
	const uint32_t y_rows = (ones || o0 >= n_live) ? 0u : min(n_live - o0, (255u + per_row) / per_row); // staged Y rows: o0 .. o0 + y_rows - 1
	const uint32_t rows = (o0 >= n_live) ? 0u : y_rows + n_in, pieces = rows * (DWS_CHUNK / 8u);          // (a workgroup without live rows stages nothing)
	const uint32_t s_begin = slice * DW_SLICE, s_end = min(B, s_begin + DW_SLICE);
	const half_t* YT = a.YT[g]; const half_t* XT = a.XT[g];
	if (o0 >= n_live) { // a workgroup of rows that are exact zeros (wave-uniform: no barrier is left waiting)
		if (t < n_threads) { float* dst = a.out[g] + (size_t)slice * (a.n_out[g] * n_in) + o * n_in + i0; dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f; dst[3] = 0.f; }
		return;
	}
	h8 in_flight[(DWS_MAX_ROWS * (DWS_CHUNK / 8u) + 255u) / 256u] = {};
	auto request = [&](const uint32_t s) {
#pragma unroll
		for (uint32_t k = 0; k < sizeof(in_flight) / sizeof(h8); ++k) {
			const uint32_t idx = threadIdx.x + 256u * k, row = idx / (DWS_CHUNK / 8u), piece = idx % (DWS_CHUNK / 8u);
			if (idx < pieces && s + 8u * piece < s_end) in_flight[k] = *reinterpret_cast<const h8*>((row < y_rows ? YT + (size_t)(o0 + row) * B : XT + (size_t)(row - y_rows) * B) + s + 8u * piece); // (nothing beyond the slice: a row's end may be the array's)
		}
	};
	auto deposit = [&](half_t* buf) {
#pragma unroll
		for (uint32_t k = 0; k < sizeof(in_flight) / sizeof(h8); ++k) {
			const uint32_t idx = threadIdx.x + 256u * k, row = idx / (DWS_CHUNK / 8u), piece = idx % (DWS_CHUNK / 8u);
			if (idx < pieces) *reinterpret_cast<h8*>(buf + row * DWS_ROW + 8u * piece) = in_flight[k];
		}
	};
	float acc[4] = {0.f, 0.f, 0.f, 0.f}; // half values
	const h8 one8 = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
	const uint32_t ry = ones ? 0u : (o - o0) * DWS_ROW, rx = (y_rows + i0) * DWS_ROW;
	request(s_begin);
	deposit(stage[0]);
	__syncthreads();
	uint32_t cur = 0;
	for (uint32_t s = s_begin; s < s_end; s += DWS_CHUNK) { // (B is a multiple of 64, a slice of 4096: chunks of 128 or one of 64 at the end -- whole k-steps)
		const bool more = s + DWS_CHUNK < s_end;
		if (more) request(s + DWS_CHUNK);
		if (live) {
			const half_t* buf = stage[cur];
			const uint32_t n_k = min(DWS_CHUNK, s_end - s) / 16u;
			for (uint32_t k = 0; k < n_k; ++k) {
				float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const h8 yy = ones ? one8 : *reinterpret_cast<const h8*>(buf + ry + 16u * k + 8 * h);
					h8 xx[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) xx[q] = *reinterpret_cast<const h8*>(buf + rx + q * DWS_ROW + 16u * k + 8 * h);
#pragma unroll
					for (int j = 0; j < 8; ++j)
#pragma unroll
						for (int q = 0; q < 4; ++q) part[q] = __builtin_fmaf((float)yy[j], (float)xx[q][j], part[q]); // v_fma_mix_f32: the product of two halfs is exact in fp32, so fused = multiplied and added
				}
#pragma unroll
				for (int q = 0; q < 4; ++q) acc[q] = rh(acc[q] + part[q]);
			}
		}
		if (more) deposit(stage[cur ^ 1u]);
		__syncthreads();
		cur ^= 1u;
	}
	if (t < n_threads) {
		float* dst = a.out[g] + (size_t)slice * (a.n_out[g] * n_in) + o * n_in + i0;
#pragma unroll
		for (int q = 0; q < 4; ++q) dst[q] = live ? acc[q] : 0.f;
	}
}
// the slices' results added in half, in slice order (GemmSplitKParallel's reduction, cutlass_matmul.h:315-322; emulated_dw's `total`)
__device__ __forceinline__ float dw_sum_sliced(const float* __restrict__ base, const uint32_t stride, const uint32_t idx, const uint32_t n_slices) {
	float t = 0.f;
	for (uint32_t p = 0; p < n_slices; ++p) t = rh(t + base[(size_t)p * stride + idx]);
	return t;
}
__global__ void k_prim_dw_sliced_total(const float* __restrict__ slices, const uint32_t n, const uint32_t n_slices, float* __restrict__ out) { // RNB_PRIM_DW_SLICED
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = dw_sum_sliced(slices, n, i, n_slices);
}

// Offsets (floats) of the seven partial blocks inside one wave's slab are given by the host.
struct DwFinishArgs {
	const float* partial[7]; // rgb2[16x64], rgb1[64x64], rgb0c[64x32], sdf1[16x64], sdf0[64x32], sdf0_2nd[64x32], sdf1_2nd[16x64]
	uint32_t n_partials;     // waves that produced a partial
	const float* var_partial;
	uint32_t n_var_partials;
	float* grads;            // GRADS_FP32
	half_t* grads16;         // rnb_config::accumulate = RNB_ACCUM_HALF: GRADS_FP16 instead (the values below are half-rounded already: exact)
	uint32_t off_sdf, off_rgb, off_var;
	uint32_t skip_rgb;       // colour-MLP gradients are exactly zero (see TrainArgs::skip_rgb): their accumulators are clear already and no workgroup is launched for them
	uint32_t sliced;         // the partials are k_dw_sliced's per-slice results (half values): added in half in slice order instead of in fp32
};

// DWF_PARAMS parameters per workgroup x DWF_SLICES slices of the partial list (fixed summation order -> deterministic). Mirrors the
// reference's precision staging: each GEMM result is narrowed to half (beta = 0), the second-order GEMMs accumulate onto
// it (beta = 1) and narrow again (fully_fused_mlp.cu:948, 966, 1001, 1012, 1127).
// Workgroups of 256 threads = 16 parameters x 16 slices (round 4; 64 x 16 = 1024 threads until then): the kernel runs on the side stream beside the gradient
// scatter, whose short workgroups hold every CU's wave slots -- a 16-wavefront workgroup waits for 16 slots to fall free at once (21 us of work took 130-160 us,
// profiles/r04_timeline_*), 4 are found at once. (64 parameters x 4 slices in 256 threads: 198 us -- a quarter of the threads, each walking 128 partials.)
constexpr uint32_t DWF_SLICES = 16, DWF_PARAMS = 16, DWF_WG = DWF_PARAMS * DWF_SLICES;
__device__ __forceinline__ float dw_sum(const float* __restrict__ base, const uint32_t stride, const uint32_t idx, const uint32_t n_partials, const uint32_t slice, float (*sh)[DWF_PARAMS], const uint32_t e) {
	float s = 0.f;
	if (n_partials & 0x80000000u) { // sliced: the slices' half results are fetched by the 16 slice-threads at once (64 at a time), then added in half in slice order by one of them
		__shared__ float all[64][DWF_PARAMS];
		const uint32_t n = n_partials & 0x7fffffffu;
		for (uint32_t p0 = 0; p0 < n; p0 += 64u) {
			__syncthreads();
#pragma unroll
			for (uint32_t k = 0; k < 64u / DWF_SLICES; ++k) { const uint32_t p = p0 + slice + DWF_SLICES * k; if (p < n) all[slice + DWF_SLICES * k][e] = base[(size_t)p * stride + idx]; }
			__syncthreads();
			if (slice == 0) for (uint32_t p = 0; p < min(64u, n - p0); ++p) s = rh(s + all[p][e]); // (the exchange below then adds fifteen zeros to it -- exact)
		}
	} else
	for (uint32_t p = slice; p < n_partials; p += DWF_SLICES) s += base[(size_t)p * stride + idx];
	__syncthreads();
	sh[slice][e] = s;
	__syncthreads();
	float t = 0.f;
#pragma unroll
	for (uint32_t q = 0; q < DWF_SLICES; ++q) t += sh[q][e];
	return t;
}

__global__ __launch_bounds__(DWF_WG) void k_dw_finish(const DwFinishArgs a) {
	__shared__ float sh[DWF_SLICES][DWF_PARAMS];
	const uint32_t e = threadIdx.x % DWF_PARAMS, slice = threadIdx.x / DWF_PARAMS;
	const uint32_t n_mlp = a.skip_rgb ? RNB_N_SDF_MLP_PARAMS : RNB_N_SDF_MLP_PARAMS + RNB_N_RGB_MLP_PARAMS;
	const uint32_t i = blockIdx.x * DWF_PARAMS + e;
	const uint32_t n_part = a.sliced ? (a.n_partials | 0x80000000u) : a.n_partials;
	if (blockIdx.x * DWF_PARAMS >= n_mlp) { // last workgroup: the variance gradient (nerf_network.h:327-340)
		float v = 0.f;
		for (uint32_t p = threadIdx.x; p < a.n_var_partials; p += DWF_WG) v += a.var_partial[p];
		__shared__ float shv[DWF_WG];
		shv[threadIdx.x] = v;
		__syncthreads();
		for (int off = DWF_WG / 2; off > 0; off >>= 1) {
			if ((int)threadIdx.x < off) shv[threadIdx.x] += shv[threadIdx.x + off];
			__syncthreads();
		}
		if (threadIdx.x == 0) {
			if (a.grads16) { a.grads16[a.off_var + 0] = f2h(shv[0]); a.grads16[a.off_var + 1] = (half_t)0.f; a.grads16[a.off_var + 2] = (half_t)0.f; a.grads16[a.off_var + 3] = (half_t)0.f; } // nerf_network.h:338-339: the fp32 sum narrowed to half
			else { a.grads[a.off_var + 0] = shv[0]; a.grads[a.off_var + 1] = 0.f; a.grads[a.off_var + 2] = 0.f; a.grads[a.off_var + 3] = 0.f; }
		}
		return;
	}
	// all parameters of a workgroup lie in the same matrix and, for the colour MLP's first matrix, in the same row (matrix sizes and 48 are multiples of 16)
	float g;
	if (i < 64 * 32) { // sdf W0
		g = rh(dw_sum(a.partial[4], 64 * 32, i, n_part, slice, sh, e));
		g = rh(dw_sum(a.partial[5], 64 * 32, i, n_part, slice, sh, e) + g);
		if (slice == 0) { if (a.grads16) a.grads16[a.off_sdf + i] = f2h(g); else a.grads[a.off_sdf + i] = g; }
	} else if (i < RNB_N_SDF_MLP_PARAMS) { // sdf W1
		const uint32_t j = i - 64 * 32;
		g = rh(dw_sum(a.partial[3], 16 * 64, j, n_part, slice, sh, e));
		g = rh(dw_sum(a.partial[6], 16 * 64, j, n_part, slice, sh, e) + g);
		if (slice == 0) { if (a.grads16) a.grads16[a.off_sdf + i] = f2h(g); else a.grads[a.off_sdf + i] = g; }
	} else {
		const uint32_t j = i - RNB_N_SDF_MLP_PARAMS;
		if (j < 64 * 48) { // rgb W0: compact column c <-> original column (c < 16 ? c : c + 16); the others receive zero input
			const uint32_t o = j / 48, col = j % 48;
			const uint32_t cc = col < 16 ? col : (col >= 32 ? col - 16 : 0);
			const float v = rh(dw_sum(a.partial[2], 64 * 32, o * 32 + cc, n_part, slice, sh, e));
			g = (col < 16 || col >= 32) ? v : 0.f;
		} else if (j < 64 * 48 + 64 * 64) {
			g = rh(dw_sum(a.partial[1], 64 * 64, j - 64 * 48, n_part, slice, sh, e));
		} else {
			g = rh(dw_sum(a.partial[0], 16 * 64, j - 64 * 48 - 64 * 64, n_part, slice, sh, e));
		}
		if (slice == 0) { if (a.grads16) a.grads16[a.off_rgb + j] = f2h(g); else a.grads[a.off_rgb + j] = g; }
	}
}

// ---------------------------------------------------------------------------------------------
// Hash-grid gradient scatter: first-order (grid.h:366-495) and second-order (grid.h:556-683) addends of a corner, each
// narrowed to half first as the reference does (grid.h:415-416), summed in fp32. Three mechanisms by level:
//   coarse  (tables that fit in LDS together)      k_grid_scatter_lds        private LDS copy per workgroup
//   middle  (a cell spans several march steps)     k_grid_scatter_quad_rl    run-length merge in registers, L2 atomics
//   fine    (about one sample per cell)            k_grid_scatter_quad       one L2 atomic per corner
// What bounds the two atomic kernels (tools/probe_atomics4.hip, MI355X): the device retires ~21 G atomic LINES per second -- one
// 64-byte line of one instruction, whether 1 or 16 of its lanes fall into it -- independent of the number of CUs issuing them
// (64 workgroups reach the same rate as 4096), of scope and of data type. Middle + fine levels put ~2.9 M lines on that path per
// step: 0.14 ms. A variant without global atomics for the fine levels (pairs binned by 4096-entry table chunk, accumulated in an
// fp64 LDS image per chunk, written with plain stores) was built and measured in round 2: its accumulate pass is bound by the CU
// side (8.4 M records x 3 gather lines + ~160 VALU instructions each: 0.10 ms for 8 levels, not better than the atomics' 0.13 ms)
// and its binning passes (8.4 M scattered 4-byte record stores) contend with k_fwd_bwd; dropped, see DESIGN.md.
// ---------------------------------------------------------------------------------------------
struct ScatterArgs {
	const uint32_t* g12; // [14][B][2]  TrainScratch::g12
	const float* srec;   // [B][8]      TrainScratch::srec
	uint32_t B;
	float* grid_grad;    // GRADS_FP32 + off_grid
	uint32_t* grid_grad16; // rnb_config::accumulate = RNB_ACCUM_HALF: GRADS_FP16 + off_grid, one half2 per table entry (the reference's gradient vector, trainer.h:78-84)
	unsigned long long* grid_fixed; // rnb_config::deterministic: [n_grid_params] 64-bit fixed-point accumulators (scale 2^24), narrowed into the gradient vector by k_fixed_narrow
	uint32_t prio;                  // RNB_SCATTER_PRIO (A/B, round 6): the scatter's wavefronts raise their issue priority (s_setprio) above the side streams' kernels that share their SIMDs
};
__device__ __forceinline__ void scatter_prio(const uint32_t prio) {
	if (prio == 1u) __builtin_amdgcn_s_setprio(1);
	else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
	else if (prio >= 3u) __builtin_amdgcn_s_setprio(3);
}

// rnb_config::deterministic. An addend of the scatter is a half value (grid.h:415-416: the reference narrows every addend to half before its atomicAdd), i.e. an integer
// multiple of 2^-24 below 2^16 in magnitude: times 2^24 it is an integer below 2^40, exact in fp32 and in a 64-bit integer. Integer additions commute, so a sum of such
// addends is EXACT and the same bits whatever order atomics retire in (a table entry overflows after 2^23 addends of the largest half; a batch holds 2^21 corners).
enum { SCATTER_FP32 = 0, SCATTER_HALF = 1, SCATTER_FIXED = 2 };
__device__ __forceinline__ long long fixed24(const float half_valued) { return (long long)(half_valued * 16777216.0f); }
__device__ __forceinline__ void atomic_add_fixed(unsigned long long* __restrict__ p, const long long v) { (void)atomicAdd(p, (unsigned long long)v); }

// atomicAdd(__half2) of grid.h:416 / 476-494: one packed L2 atomic carries both features of a table entry (global_atomic_pk_add_f16, no return value)
__device__ __forceinline__ void atomic_add_h2(uint32_t* __restrict__ entry, const float v0, const float v1) {
	const h2 v = {f2h(v0), f2h(v1)};
	(void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(entry), v);
}

struct ScatterSample { float x, y, z, dn[3]; };
__device__ __forceinline__ ScatterSample load_srec(const float* __restrict__ srec, const uint32_t s) {
	const f4 a = reinterpret_cast<const f4*>(srec)[(size_t)s * 2 + 0];
	const f4 b = reinterpret_cast<const f4*>(srec)[(size_t)s * 2 + 1];
	return ScatterSample{a[0], a[1], a[2], {a[3], b[0], b[1]}};
}

// The addend of corner c = (cx, cy, cz) for one feature: first order weight = 1 * wx * wy * wz in the reference's multiplication
// order (grid.h:478-490); second order (grid.h:655-681): for each gradient dimension the corner is a 'left' (-) or 'right' (+) end.
__device__ __forceinline__ float corner_addend(const float g1, const float g2, const float scale, const float (&dn)[3], const float (&pos)[3], const uint32_t (&c)[3]) {
	float w[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) w[d] = c[d] ? pos[d] : 1 - pos[d];
	float weight = 1;
	weight *= w[0]; weight *= w[1]; weight *= w[2];
	float add = rh(g1 * weight);
#pragma unroll
	for (uint32_t gd = 0; gd < 3; ++gd) {
		float w2 = scale * dn[gd] * 1.0f;
#pragma unroll
		for (uint32_t ngd = 0; ngd < 2; ++ngd) {
			const uint32_t d = ngd >= gd ? (ngd + 1) : ngd;
			w2 *= w[d];
		}
		add += rh(g2 * (c[gd] ? w2 : -w2));
	}
	return add;
}

// The same four addends one by one (the reference issues each as an atomic of its own: one from kernel_grid_backward, grid.h:410-430, three from the
// second-order kernel, grid.h:655-681).
__device__ __forceinline__ void corner_addends(const float g1, const float g2, const float scale, const float (&dn)[3], const float (&pos)[3], const uint32_t (&c)[3], float (&out)[4]) {
	float w[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) w[d] = c[d] ? pos[d] : 1 - pos[d];
	float weight = 1;
	weight *= w[0]; weight *= w[1]; weight *= w[2];
	out[0] = rh(g1 * weight);
#pragma unroll
	for (uint32_t gd = 0; gd < 3; ++gd) {
		float w2 = scale * dn[gd] * 1.0f;
#pragma unroll
		for (uint32_t ngd = 0; ngd < 2; ++ngd) {
			const uint32_t d = ngd >= gd ? (ngd + 1) : ngd;
			w2 *= w[d];
		}
		out[1 + gd] = rh(g2 * (c[gd] ? w2 : -w2));
	}
}

// rnb_config::deterministic: the same four half-valued addends as integers
__device__ __forceinline__ long long corner_addend_fixed(const float g1, const float g2, const float scale, const float (&dn)[3], const float (&pos)[3], const uint32_t (&c)[3]) {
	float t[4];
	corner_addends(g1, g2, scale, dn, pos, c, t);
	return (fixed24(t[0]) + fixed24(t[1])) + (fixed24(t[2]) + fixed24(t[3]));
}

// Coarsest dense levels (table <= 13 824 entries): every ray of the batch lands in the same few hundred surface cells, so
// global atomics pile up on a handful of lines. Each workgroup accumulates its slice of the batch into a private copy of
// the levels' gradient tables in LDS, then flushes the non-zero entries once. ds_add_f32 retires at ~3 cycles per LANE per CU
// (tools/probe_lds_atomics.hip; integer LDS atomics are 15x, ds_add_f64 6.6x faster), so it uses the quad walk of
// k_grid_scatter_quad_rl below -- four lanes = (dx, feature) own K = 16 consecutive samples of the ray-ordered batch and keep the
// four (dy, dz) corner sums in registers while the cell (37 / 27 march steps wide on levels 0 / 1) does not change -- which issues a
// third of the LDS atomics of the earlier thread-per-4-samples form (43 -> 32 us). Round 5 rebuilt it on double accumulators (one lane
// per 4 samples, all 8 corners, one feature per workgroup): 33 -> 30 us alone (walk 13.6, flush of the private tables into the same
// ~5 k global addresses 9.2, launch + zeroing 6.6) and no change of the step, whose tail is three chains of equal length (this kernel +
// the last optimizer chunk, the optimizer chunk of group B, the next march's k_march_write); not kept (profiles/r05_ab_scatter_lds_f64.txt).
struct ScatterLdsArgs { ScatterArgs a; uint32_t n_levels; uint32_t samples_per_wg; }; // levels [0, n_levels)

// LDS layout: level l's table at float offset 2 * G.offsets[l]. The per-sample loads of a walk are issued four samples ahead.
template <bool HALF>
__device__ __forceinline__ void grid_scatter_lds_body(const GridMeta& G, const ScatterLdsArgs& p) {
	scatter_prio(p.a.prio);
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	float* tab = reinterpret_cast<float*>(smem_raw);
	const ScatterArgs& a = p.a;
	const uint32_t NL = min(p.n_levels, G.valid_level + 1u);
	if (NL == 0) return;
	const uint32_t n_tab = G.offsets[NL] * 2;
	for (uint32_t q = threadIdx.x; q < n_tab; q += blockDim.x) tab[q] = 0.f;
	__syncthreads();
	constexpr uint32_t K = 16, C = 4;
	const uint32_t wg_begin = blockIdx.x * p.samples_per_wg;
	const uint32_t wg_end = min(wg_begin + p.samples_per_wg, a.B);
	const uint32_t quad = threadIdx.x >> 2, n_quads = blockDim.x >> 2;
	const uint32_t dx = (threadIdx.x >> 1) & 1u, f = threadIdx.x & 1u;
#pragma unroll 1
	for (uint32_t level = 0; level < NL; ++level) {
		float* lt = tab + (size_t)G.offsets[level] * 2;
		const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
		const float scale = G.scale[level];
		const uint32_t res = G.resolution[level];
		const uint2* g12 = reinterpret_cast<const uint2*>(a.g12) + (size_t)level * a.B;
#pragma unroll 1
		for (uint32_t s0 = wg_begin + quad * K; s0 < wg_end; s0 += n_quads * K) {
			const uint32_t s_end = min(s0 + K, wg_end);
			float acc[4] = {0.f, 0.f, 0.f, 0.f};
			uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
			auto flush = [&]() {
#pragma unroll
				for (uint32_t yz = 0; yz < 4; ++yz) {
					if (acc[yz] != 0.f) {
						const uint32_t e = grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + (yz & 1u), cur[2] + (yz >> 1));
						atomicAdd(lt + e * 2 + f, acc[yz]);
						acc[yz] = 0.f;
					}
				}
			};
#pragma unroll 1
			for (uint32_t sc = s0; sc < s_end; sc += C) {
				ScatterSample sm[C];
				uint2 q12[C];
#pragma unroll
				for (uint32_t j = 0; j < C; ++j) { const uint32_t s = min(sc + j, s_end - 1); sm[j] = load_srec(a.srec, s); q12[j] = g12[s]; }
#pragma unroll
				for (uint32_t j = 0; j < C; ++j) {
					if (sc + j >= s_end) break;
					float pos[3];
					uint32_t pg[3];
					pos_fract(sm[j].x, scale, &pos[0], &pg[0]);
					pos_fract(sm[j].y, scale, &pos[1], &pg[1]);
					pos_fract(sm[j].z, scale, &pos[2], &pg[2]);
					if (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2]) {
						if (cur[0] != 0xffffffffu) flush();
						cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
					}
					const float g1 = h2f(unpack_h2(q12[j].x)[f]);
					const float g2 = h2f(unpack_h2(q12[j].y)[f]);
#pragma unroll
					for (uint32_t yz = 0; yz < 4; ++yz) {
						const uint32_t c[3] = {dx, yz & 1u, yz >> 1};
						acc[yz] += corner_addend(g1, g2, scale, sm[j].dn, pos, c);
					}
				}
			}
			flush();
		}
	}
	__syncthreads();
	if (HALF) { // the workgroup's private sums leave as packed half atomics, one per table entry
		for (uint32_t e = threadIdx.x; e < n_tab / 2; e += blockDim.x) {
			const float v0 = tab[2 * e], v1 = tab[2 * e + 1];
			if (v0 != 0.f || v1 != 0.f) atomic_add_h2(a.grid_grad16 + e, v0, v1);
		}
		return;
	}
	float* gg = a.grid_grad;
	for (uint32_t q = threadIdx.x; q < n_tab; q += blockDim.x) {
		const float v = tab[q];
		if (v != 0.f) atomicAdd(gg + q, v);
	}
}
__global__ __launch_bounds__(512) void k_grid_scatter_lds(const GridMeta G, const ScatterLdsArgs p) { grid_scatter_lds_body<false>(G, p); }
__global__ __launch_bounds__(512) void k_grid_scatter_lds_h(const GridMeta G, const ScatterLdsArgs p) { grid_scatter_lds_body<true>(G, p); }

// One L2 atomic per corner. Four adjacent lanes own one (sample, level): lane&3 = (dx << 1) | feature. The x-neighbour of a
// cell is the next table entry on dense levels and for even x on hashed ones (hash prime 1), so the four lanes' atomics mostly
// fall on 16 contiguous bytes of one cache line and travel as one request; each lane issues the 4 (dy, dz) corners.
// Both atomic kernels run as a capped number of workgroups that walk "virtual" workgroups (blockIdx.x, + gridDim.x, ...): the atomic unit
// is saturated by a few workgroups per CU (tools/probe_atomics4.hip: 64 workgroups reach the rate of 4096), and every further resident
// wavefront only takes a slot from the kernels of the side streams (march, optimizer chunks), which were starved beside the scatter
// (k_march_write 48 -> 186 us beside the one-shot grid of 65 k wavefronts, profiles/r03_*).
__global__ __launch_bounds__(256) void k_grid_scatter_quad(const GridMeta G, const ScatterArgs a, const uint32_t level0, const uint32_t n_vblocks) {
	scatter_prio(a.prio);
	const uint32_t level = blockIdx.y + level0;
	if (level > G.valid_level) return;
	float* gg = a.grid_grad + (size_t)G.offsets[level] * 2;
	const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
	const float scale = G.scale[level];
	const uint32_t res = G.resolution[level];
#pragma unroll 1
	for (uint32_t vb = blockIdx.x; vb < n_vblocks; vb += gridDim.x) {
		const uint32_t t = vb * blockDim.x + threadIdx.x;
		const uint32_t s = t >> 2;
		if (s >= a.B) continue;
		const uint32_t dx = (t >> 1) & 1u, f = t & 1u;
		const ScatterSample sm = load_srec(a.srec, s);
		float pos[3];
		uint32_t pg[3];
		pos_fract(sm.x, scale, &pos[0], &pg[0]);
		pos_fract(sm.y, scale, &pos[1], &pg[1]);
		pos_fract(sm.z, scale, &pos[2], &pg[2]);
		const uint2 q12 = reinterpret_cast<const uint2*>(a.g12)[(size_t)level * a.B + s];
		const float g1 = h2f(unpack_h2(q12.x)[f]);
		const float g2 = h2f(unpack_h2(q12.y)[f]);
#pragma unroll
		for (uint32_t yz = 0; yz < 4; ++yz) {
			const uint32_t c[3] = {dx, yz & 1u, yz >> 1};
			const float add = corner_addend(g1, g2, scale, sm.dn, pos, c);
			if (add != 0.f) {
				const uint32_t e = grid_entry(hashmap_size, res, pg[0] + c[0], pg[1] + c[1], pg[2] + c[2]);
				atomicAdd(gg + (size_t)e * 2 + f, add);
			}
		}
	}
}

// rnb_config::accumulate = RNB_ACCUM_HALF: the gradient table is the reference's half2 per entry and one packed atomic carries both features, so the four lanes of a
// (sample, level) are (dx, dy) and each issues the 2 dz corners: half the atomic instructions, the same cache lines (an x-pair's 8 bytes travel together).
// PER_ADDEND (RNB_SCATTER_PLAIN=1, tests): the four addends of a corner as four atomics, the reference's own sequence of half additions -- on the coarse levels, where
// thousands of addends meet one entry, the sequential half sum rounds small addends away, and only this form reproduces that.
template <bool PER_ADDEND>
__device__ __forceinline__ void grid_scatter_quad_h_body(const GridMeta& G, const ScatterArgs& a, const uint32_t level0, const uint32_t n_vblocks) {
	const uint32_t level = blockIdx.y + level0;
	if (level > G.valid_level) return;
	uint32_t* gg = a.grid_grad16 + G.offsets[level];
	const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
	const float scale = G.scale[level];
	const uint32_t res = G.resolution[level];
#pragma unroll 1
	for (uint32_t vb = blockIdx.x; vb < n_vblocks; vb += gridDim.x) {
		const uint32_t t = vb * blockDim.x + threadIdx.x;
		const uint32_t s = t >> 2;
		if (s >= a.B) continue;
		const uint32_t dx = (t >> 1) & 1u, dy = t & 1u;
		const ScatterSample sm = load_srec(a.srec, s);
		float pos[3];
		uint32_t pg[3];
		pos_fract(sm.x, scale, &pos[0], &pg[0]);
		pos_fract(sm.y, scale, &pos[1], &pg[1]);
		pos_fract(sm.z, scale, &pos[2], &pg[2]);
		const uint2 q12 = reinterpret_cast<const uint2*>(a.g12)[(size_t)level * a.B + s];
		const h2 g1 = unpack_h2(q12.x), g2 = unpack_h2(q12.y);
#pragma unroll
		for (uint32_t dz = 0; dz < 2; ++dz) {
			const uint32_t c[3] = {dx, dy, dz};
			uint32_t* entry = gg + grid_entry(hashmap_size, res, pg[0] + c[0], pg[1] + c[1], pg[2] + c[2]);
			if (PER_ADDEND) {
				float t0[4], t1[4];
				corner_addends(h2f(g1[0]), h2f(g2[0]), scale, sm.dn, pos, c, t0);
				corner_addends(h2f(g1[1]), h2f(g2[1]), scale, sm.dn, pos, c, t1);
#pragma unroll
				for (int q = 0; q < 4; ++q) if (t0[q] != 0.f || t1[q] != 0.f) atomic_add_h2(entry, t0[q], t1[q]);
				continue;
			}
			const float add0 = corner_addend(h2f(g1[0]), h2f(g2[0]), scale, sm.dn, pos, c);
			const float add1 = corner_addend(h2f(g1[1]), h2f(g2[1]), scale, sm.dn, pos, c);
			if (add0 != 0.f || add1 != 0.f) atomic_add_h2(entry, add0, add1);
		}
	}
}
__global__ __launch_bounds__(256) void k_grid_scatter_quad_h(const GridMeta G, const ScatterArgs a, const uint32_t level0, const uint32_t n_vblocks) { grid_scatter_quad_h_body<false>(G, a, level0, n_vblocks); }
__global__ __launch_bounds__(256) void k_grid_scatter_quad_h_per_addend(const GridMeta G, const ScatterArgs a, const uint32_t level0, const uint32_t n_vblocks) { grid_scatter_quad_h_body<true>(G, a, level0, n_vblocks); }

// Round 6: FACE SHARING in the run-length walks (fp32 accumulate mode). A quad keeps the four (dy, dz) corner sums of its (dx, feature) while the cell does not change; when the
// ray steps into the cell BEHIND A y- OR z-FACE (one coordinate by +-1, the usual transition on the levels whose cells are wider than a march step), the two corners on that face
// belong to both cells: their sums stay in registers (moved to the other side of the pair), only the two corners left behind are flushed -- two atomic requests per cell
// instead of four. (An x-step shares the x-pair, which travels in ONE request already: nothing to save, it flushes all four like any other transition.) Same addends per table
// entry; a different grouping of the fp32 partial sums, as with any run length.
template <typename Flush2>
__device__ __forceinline__ bool share_face(float (&acc)[4], const uint32_t (&cur)[3], const uint32_t (&pg)[3], Flush2&& flush_pair) {
	const int dy = (int)(pg[1] - cur[1]), dz = (int)(pg[2] - cur[2]);
	if (pg[0] != cur[0]) return false;
	if (dz == 0 && (dy == 1 || dy == -1)) {
		if (dy == 1) { flush_pair(0, 2); acc[0] = acc[1]; acc[2] = acc[3]; acc[1] = 0.f; acc[3] = 0.f; }
		else { flush_pair(1, 3); acc[1] = acc[0]; acc[3] = acc[2]; acc[0] = 0.f; acc[2] = 0.f; }
		return true;
	}
	if (dy == 0 && (dz == 1 || dz == -1)) {
		if (dz == 1) { flush_pair(0, 1); acc[0] = acc[2]; acc[1] = acc[3]; acc[2] = 0.f; acc[3] = 0.f; }
		else { flush_pair(2, 3); acc[2] = acc[0]; acc[3] = acc[1]; acc[0] = 0.f; acc[1] = 0.f; }
		return true;
	}
	return false;
}

// Middle levels (cell a few march steps wide): the quad layout above, but each quad walks K consecutive samples of the
// ray-ordered batch and keeps the four (dy, dz) corner sums of its (dx, feature) in registers while the cell does not
// change. Same-address lanes of one atomic instruction are serialised by the memory system (one request each), so merging
// a cell run in registers divides the request count by the run length.
// One launch covers levels level0 .. level0 + n - 1 with a 1-D grid: level level0 + i owns workgroups [wg_start[i], wg_start[i+1])
// (exactly as many as its run length needs); k_log2 holds log2(K) of level0 + i in bits [4i, 4i+4).
struct ScatterRlPlan { uint32_t n; uint32_t wg_start[17]; uint64_t k_log2; };

// HALF (rnb_config::accumulate = RNB_ACCUM_HALF): lanes (dx, dy), registers (dz, feature), packed half atomics (see k_grid_scatter_quad_h); a run is still summed in fp32.
template <bool HALF, bool SHARE = false>
__device__ __forceinline__ void grid_scatter_quad_rl_body(const GridMeta& G, const ScatterArgs& a, const uint32_t level0, const ScatterRlPlan& plan) {
	scatter_prio(a.prio);
#pragma unroll 1
	for (uint32_t vb = blockIdx.x; vb < plan.wg_start[plan.n]; vb += gridDim.x) { // virtual workgroups (see k_grid_scatter_quad)
		uint32_t li = 0;
#pragma unroll 1
		for (uint32_t q = 1; q < plan.n; ++q) if (vb >= plan.wg_start[q]) li = q;
		const uint32_t level = level0 + li;
		const uint32_t K = 1u << ((plan.k_log2 >> (4 * li)) & 15u);
		if (level > G.valid_level) continue;
		const uint32_t t = (vb - plan.wg_start[li]) * blockDim.x + threadIdx.x;
		const uint32_t s0 = (t >> 2) * K;
		if (s0 >= a.B) continue;
		const uint32_t dx = (t >> 1) & 1u, f = t & 1u; // HALF: f is dy
		float* gg = a.grid_grad + (size_t)G.offsets[level] * 2;
		uint32_t* gg16 = a.grid_grad16 + G.offsets[level];
		const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
		const float scale = G.scale[level];
		const uint32_t res = G.resolution[level];
		float acc[4] = {0.f, 0.f, 0.f, 0.f}; // fp32: the four (dy, dz) corners of (dx, feature); HALF: [2 dz + feature] of the corner (dx, dy)
		uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
		auto flush = [&]() {
			if (HALF) {
#pragma unroll
				for (uint32_t dz = 0; dz < 2; ++dz) {
					if (acc[2 * dz] != 0.f || acc[2 * dz + 1] != 0.f) {
						atomic_add_h2(gg16 + grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + f, cur[2] + dz), acc[2 * dz], acc[2 * dz + 1]);
						acc[2 * dz] = 0.f; acc[2 * dz + 1] = 0.f;
					}
				}
				return;
			}
#pragma unroll
			for (uint32_t yz = 0; yz < 4; ++yz) {
				if (acc[yz] != 0.f) {
					const uint32_t e = grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + (yz & 1u), cur[2] + (yz >> 1));
					atomicAdd(gg + (size_t)e * 2 + f, acc[yz]);
					acc[yz] = 0.f;
				}
			}
		};
		const uint32_t s_end = min(s0 + K, a.B);
		const uint2* g12l = reinterpret_cast<const uint2*>(a.g12) + (size_t)level * a.B;
		// the loads of a walk are issued four samples ahead (a dependent load per sample made the walk latency-bound: 107 us for a 68 us floor)
		constexpr uint32_t C = 4;
#pragma unroll 1
		for (uint32_t sc = s0; sc < s_end; sc += C) {
			ScatterSample sm[C];
			uint2 q12[C];
#pragma unroll
			for (uint32_t j = 0; j < C; ++j) { const uint32_t s = min(sc + j, s_end - 1); sm[j] = load_srec(a.srec, s); q12[j] = g12l[s]; }
#pragma unroll
			for (uint32_t j = 0; j < C; ++j) {
				if (sc + j >= s_end) break;
				float pos[3];
				uint32_t pg[3];
				pos_fract(sm[j].x, scale, &pos[0], &pg[0]);
				pos_fract(sm[j].y, scale, &pos[1], &pg[1]);
				pos_fract(sm[j].z, scale, &pos[2], &pg[2]);
				if (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2]) {
					if (cur[0] != 0xffffffffu) {
						bool shared = false;
						if (SHARE && !HALF) shared = share_face(acc, cur, pg, [&](const uint32_t y0, const uint32_t y1) {
							const uint32_t two[2] = {y0, y1};
#pragma unroll
							for (int q = 0; q < 2; ++q) {
								const uint32_t yz = two[q];
								if (acc[yz] != 0.f) atomicAdd(gg + (size_t)grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + (yz & 1u), cur[2] + (yz >> 1)) * 2 + f, acc[yz]);
							}
						});
						if (!shared) flush();
					}
					cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
				}
				if (HALF) {
					const h2 g1 = unpack_h2(q12[j].x), g2 = unpack_h2(q12[j].y);
#pragma unroll
					for (uint32_t dz = 0; dz < 2; ++dz) {
						const uint32_t c[3] = {dx, f, dz};
						acc[2 * dz] += corner_addend(h2f(g1[0]), h2f(g2[0]), scale, sm[j].dn, pos, c);
						acc[2 * dz + 1] += corner_addend(h2f(g1[1]), h2f(g2[1]), scale, sm[j].dn, pos, c);
					}
					continue;
				}
				const float g1 = h2f(unpack_h2(q12[j].x)[f]);
				const float g2 = h2f(unpack_h2(q12[j].y)[f]);
#pragma unroll
				for (uint32_t yz = 0; yz < 4; ++yz) {
					const uint32_t c[3] = {dx, yz & 1u, yz >> 1};
					acc[yz] += corner_addend(g1, g2, scale, sm[j].dn, pos, c);
				}
			}
		}
		flush();
	}
}
// Round 5: the same walk with its operands STAGED IN LDS. In the form above a quad loads its K samples in chunks of four, and every chunk's first wait is
// `s_waitcnt vmcnt(n)`: on gfx9 that counter retires in order across loads AND atomics, so the loads of chunk c + 1 wait for the acknowledgement of the atomics
// flushed in chunk c -- which execute memory-side (tools/probe_counters.hip: one write request each, no L2 fetch) and take microseconds to come back under load.
// The walk was a chain of K / 4 such round trips (the kernel sat at 0.75 of the atomic-line rate where the plain kernel reaches 0.97). Here the workgroup first
// copies the records of its 64 K samples into LDS with coalesced loads (each record once instead of once per lane of its quad: a quarter of the load traffic),
// and the walk then reads LDS only (lgkmcnt) and issues its atomics without ever waiting for one. Sample j of quad q sits at slot j * 64 + q, so the 16 quads of a
// wavefront read consecutive 32-byte records at every step of the walk (no bank conflicts; the four lanes of a quad read the same address: a broadcast).
constexpr uint32_t RL_MAX_K = 16;
constexpr size_t LDS_SCATTER_RL = (size_t)RL_MAX_K * 64 * 32; // 16 B {x y z dn0} + 8 B {dn1 dn2} + 8 B g12 per sample
template <bool HALF, bool SHARE = false>
__device__ __forceinline__ void grid_scatter_quad_rl_staged_body(const GridMeta& G, const ScatterArgs& a, const uint32_t level0, const ScatterRlPlan& plan) {
	scatter_prio(a.prio);
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	f4* sA = reinterpret_cast<f4*>(smem_raw);                                  // [K * 64] x y z dn0
	float2* sB = reinterpret_cast<float2*>(smem_raw + (size_t)RL_MAX_K * 64 * 16); // [K * 64] dn1 dn2
	uint2* sG = reinterpret_cast<uint2*>(smem_raw + (size_t)RL_MAX_K * 64 * 24);   // [K * 64] g12 of this level
#pragma unroll 1
	for (uint32_t vb = blockIdx.x; vb < plan.wg_start[plan.n]; vb += gridDim.x) { // virtual workgroups (see k_grid_scatter_quad)
		uint32_t li = 0;
#pragma unroll 1
		for (uint32_t q = 1; q < plan.n; ++q) if (vb >= plan.wg_start[q]) li = q;
		const uint32_t level = level0 + li;
		const uint32_t k_log2 = (uint32_t)(plan.k_log2 >> (4 * li)) & 15u, K = 1u << k_log2;
		if (level > G.valid_level) continue; // (workgroup-uniform)
		const uint32_t wg_first = (vb - plan.wg_start[li]) * 64u * K; // first sample of this workgroup's 64 quads
		const uint32_t wg_n = min(64u * K, a.B - min(a.B, wg_first));
		const uint2* g12l = reinterpret_cast<const uint2*>(a.g12) + (size_t)level * a.B;
		if (vb != blockIdx.x) __syncthreads(); // the previous virtual workgroup's walk has read the stage
		for (uint32_t i = threadIdx.x; i < wg_n; i += blockDim.x) {
			const uint32_t s = wg_first + i;
			const f4 r0 = reinterpret_cast<const f4*>(a.srec)[(size_t)s * 2 + 0];
			const float2 r1 = *reinterpret_cast<const float2*>(a.srec + (size_t)s * 8 + 4);
			const uint2 g = g12l[s];
			const uint32_t slot = (i & (K - 1u)) * 64u + (i >> k_log2);
			sA[slot] = r0; sB[slot] = r1; sG[slot] = g;
		}
		__syncthreads();
		const uint32_t quad = threadIdx.x >> 2;
		const uint32_t n_mine = min(K, wg_n - min(wg_n, quad * K)); // samples of this quad (0 beyond the batch)
		const uint32_t dx = (threadIdx.x >> 1) & 1u, f = threadIdx.x & 1u; // HALF: f is dy
		float* gg = a.grid_grad + (size_t)G.offsets[level] * 2;
		uint32_t* gg16 = a.grid_grad16 + G.offsets[level];
		const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
		const float scale = G.scale[level];
		const uint32_t res = G.resolution[level];
		float acc[4] = {0.f, 0.f, 0.f, 0.f}; // fp32: the four (dy, dz) corners of (dx, feature); HALF: [2 dz + feature] of the corner (dx, dy)
		uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
		auto flush = [&]() {
			if (HALF) {
#pragma unroll
				for (uint32_t dz = 0; dz < 2; ++dz) {
					if (acc[2 * dz] != 0.f || acc[2 * dz + 1] != 0.f) {
						atomic_add_h2(gg16 + grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + f, cur[2] + dz), acc[2 * dz], acc[2 * dz + 1]);
						acc[2 * dz] = 0.f; acc[2 * dz + 1] = 0.f;
					}
				}
				return;
			}
#pragma unroll
			for (uint32_t yz = 0; yz < 4; ++yz) {
				if (acc[yz] != 0.f) {
					const uint32_t e = grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + (yz & 1u), cur[2] + (yz >> 1));
					atomicAdd(gg + (size_t)e * 2 + f, acc[yz]);
					acc[yz] = 0.f;
				}
			}
		};
#pragma unroll 1
		for (uint32_t j = 0; j < n_mine; ++j) {
			const uint32_t slot = j * 64u + quad;
			const f4 r0 = sA[slot];
			const float2 r1 = sB[slot];
			const uint2 q12 = sG[slot];
			const float dn[3] = {r0[3], r1.x, r1.y};
			float pos[3];
			uint32_t pg[3];
			pos_fract(r0[0], scale, &pos[0], &pg[0]);
			pos_fract(r0[1], scale, &pos[1], &pg[1]);
			pos_fract(r0[2], scale, &pos[2], &pg[2]);
			if (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2]) {
				if (cur[0] != 0xffffffffu) {
					bool shared = false;
					if (SHARE && !HALF) shared = share_face(acc, cur, pg, [&](const uint32_t y0, const uint32_t y1) {
						const uint32_t two[2] = {y0, y1};
#pragma unroll
						for (int q = 0; q < 2; ++q) {
							const uint32_t yz = two[q];
							if (acc[yz] != 0.f) atomicAdd(gg + (size_t)grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + (yz & 1u), cur[2] + (yz >> 1)) * 2 + f, acc[yz]);
						}
					});
					if (!shared) flush();
				}
				cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
			}
			if (HALF) {
				const h2 g1 = unpack_h2(q12.x), g2 = unpack_h2(q12.y);
#pragma unroll
				for (uint32_t dz = 0; dz < 2; ++dz) {
					const uint32_t c[3] = {dx, f, dz};
					acc[2 * dz] += corner_addend(h2f(g1[0]), h2f(g2[0]), scale, dn, pos, c);
					acc[2 * dz + 1] += corner_addend(h2f(g1[1]), h2f(g2[1]), scale, dn, pos, c);
				}
				continue;
			}
			const float g1 = h2f(unpack_h2(q12.x)[f]);
			const float g2 = h2f(unpack_h2(q12.y)[f]);
#pragma unroll
			for (uint32_t yz = 0; yz < 4; ++yz) {
				const uint32_t c[3] = {dx, yz & 1u, yz >> 1};
				acc[yz] += corner_addend(g1, g2, scale, dn, pos, c);
			}
		}
		if (n_mine) flush();
	}
}
__global__ __launch_bounds__(256) void k_grid_scatter_quad_rl(const GridMeta G, const ScatterArgs a, const uint32_t level0, const ScatterRlPlan plan) { grid_scatter_quad_rl_staged_body<false>(G, a, level0, plan); }
__global__ __launch_bounds__(256) void k_grid_scatter_quad_rl_h(const GridMeta G, const ScatterArgs a, const uint32_t level0, const ScatterRlPlan plan) { grid_scatter_quad_rl_staged_body<true>(G, a, level0, plan); }
// RNB_SCATTER_RL_STAGED=0 (A/B): the walk with its operands loaded from global memory four samples ahead (rounds 2-4)
__global__ __launch_bounds__(256) void k_grid_scatter_quad_rl_direct(const GridMeta G, const ScatterArgs a, const uint32_t level0, const ScatterRlPlan plan) { grid_scatter_quad_rl_body<false>(G, a, level0, plan); }
__global__ __launch_bounds__(256) void k_grid_scatter_quad_rl_direct_h(const GridMeta G, const ScatterArgs a, const uint32_t level0, const ScatterRlPlan plan) { grid_scatter_quad_rl_body<true>(G, a, level0, plan); }
// face sharing (round 6, RNB_SCATTER_SHARE)
__global__ __launch_bounds__(256) void k_grid_scatter_quad_rl_share(const GridMeta G, const ScatterArgs a, const uint32_t level0, const ScatterRlPlan plan) { grid_scatter_quad_rl_staged_body<false, true>(G, a, level0, plan); }
__global__ __launch_bounds__(256) void k_grid_scatter_quad_rl_direct_share(const GridMeta G, const ScatterArgs a, const uint32_t level0, const ScatterRlPlan plan) { grid_scatter_quad_rl_body<false, true>(G, a, level0, plan); }

// ---------------------------------------------------------------------------------------------
// rnb_config::deterministic: the three scatter mechanisms above on 64-bit fixed-point accumulators (ScatterArgs::grid_fixed; fixed24 / corner_addend_fixed). The same
// samples meet the same table entries through the same walks; what changes is the arithmetic of a sum -- every half-valued addend enters as an integer, registers, LDS
// tables (ds_add_u64) and global accumulators (global_atomic_add_x2) hold integers, and nothing is rounded before k_fixed_narrow: the result does not depend on K, on the
// workgroup slicing, on which kernel a level goes through or on the order anything retires in.
// ---------------------------------------------------------------------------------------------
// coarse levels: one FEATURE per workgroup (blockIdx.y), so that a level's private table of 8-byte sums takes the LDS the fp32 form takes for both features; the four
// lanes of a quad are (dx, dy), each keeps its two dz corners.
__global__ __launch_bounds__(512) void k_grid_scatter_lds_fixed(const GridMeta G, const ScatterLdsArgs p) {
	extern __shared__ __attribute__((aligned(16))) char smem_raw[];
	unsigned long long* tab = reinterpret_cast<unsigned long long*>(smem_raw);
	const ScatterArgs& a = p.a;
	const uint32_t NL = min(p.n_levels, G.valid_level + 1u);
	if (NL == 0) return;
	const uint32_t n_tab = G.offsets[NL]; // entries (one feature)
	for (uint32_t q = threadIdx.x; q < n_tab; q += blockDim.x) tab[q] = 0ull;
	__syncthreads();
	constexpr uint32_t K = 16, C = 4;
	const uint32_t f = blockIdx.y;
	const uint32_t wg_begin = blockIdx.x * p.samples_per_wg;
	const uint32_t wg_end = min(wg_begin + p.samples_per_wg, a.B);
	const uint32_t quad = threadIdx.x >> 2, n_quads = blockDim.x >> 2;
	const uint32_t dx = (threadIdx.x >> 1) & 1u, dy = threadIdx.x & 1u;
#pragma unroll 1
	for (uint32_t level = 0; level < NL; ++level) {
		unsigned long long* lt = tab + G.offsets[level];
		const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
		const float scale = G.scale[level];
		const uint32_t res = G.resolution[level];
		const uint2* g12 = reinterpret_cast<const uint2*>(a.g12) + (size_t)level * a.B;
#pragma unroll 1
		for (uint32_t s0 = wg_begin + quad * K; s0 < wg_end; s0 += n_quads * K) {
			const uint32_t s_end = min(s0 + K, wg_end);
			long long acc[2] = {0, 0};
			uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
			auto flush = [&]() {
#pragma unroll
				for (uint32_t dz = 0; dz < 2; ++dz) {
					if (acc[dz] != 0) {
						(void)atomicAdd(lt + grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + dy, cur[2] + dz), (unsigned long long)acc[dz]);
						acc[dz] = 0;
					}
				}
			};
#pragma unroll 1
			for (uint32_t sc = s0; sc < s_end; sc += C) {
				ScatterSample sm[C];
				uint2 q12[C];
#pragma unroll
				for (uint32_t j = 0; j < C; ++j) { const uint32_t s = min(sc + j, s_end - 1); sm[j] = load_srec(a.srec, s); q12[j] = g12[s]; }
#pragma unroll
				for (uint32_t j = 0; j < C; ++j) {
					if (sc + j >= s_end) break;
					float pos[3];
					uint32_t pg[3];
					pos_fract(sm[j].x, scale, &pos[0], &pg[0]);
					pos_fract(sm[j].y, scale, &pos[1], &pg[1]);
					pos_fract(sm[j].z, scale, &pos[2], &pg[2]);
					if (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2]) {
						if (cur[0] != 0xffffffffu) flush();
						cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
					}
					const float g1 = h2f(unpack_h2(q12[j].x)[f]);
					const float g2 = h2f(unpack_h2(q12[j].y)[f]);
#pragma unroll
					for (uint32_t dz = 0; dz < 2; ++dz) {
						const uint32_t c[3] = {dx, dy, dz};
						acc[dz] += corner_addend_fixed(g1, g2, scale, sm[j].dn, pos, c);
					}
				}
			}
			flush();
		}
	}
	__syncthreads();
	for (uint32_t e = threadIdx.x; e < n_tab; e += blockDim.x) {
		const unsigned long long v = tab[e];
		if (v != 0ull) (void)atomicAdd(a.grid_fixed + (size_t)e * 2 + f, v);
	}
}

// fine levels: one integer atomic per corner; lanes (dx, feature) as in k_grid_scatter_quad -- a quad's four 8-byte addends of one (dy, dz) fall on 32 contiguous bytes
__global__ __launch_bounds__(256) void k_grid_scatter_quad_fixed(const GridMeta G, const ScatterArgs a, const uint32_t level0, const uint32_t n_vblocks) {
	const uint32_t level = blockIdx.y + level0;
	if (level > G.valid_level) return;
	unsigned long long* gg = a.grid_fixed + (size_t)G.offsets[level] * 2;
	const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
	const float scale = G.scale[level];
	const uint32_t res = G.resolution[level];
#pragma unroll 1
	for (uint32_t vb = blockIdx.x; vb < n_vblocks; vb += gridDim.x) {
		const uint32_t t = vb * blockDim.x + threadIdx.x;
		const uint32_t s = t >> 2;
		if (s >= a.B) continue;
		const uint32_t dx = (t >> 1) & 1u, f = t & 1u;
		const ScatterSample sm = load_srec(a.srec, s);
		float pos[3];
		uint32_t pg[3];
		pos_fract(sm.x, scale, &pos[0], &pg[0]);
		pos_fract(sm.y, scale, &pos[1], &pg[1]);
		pos_fract(sm.z, scale, &pos[2], &pg[2]);
		const uint2 q12 = reinterpret_cast<const uint2*>(a.g12)[(size_t)level * a.B + s];
		const float g1 = h2f(unpack_h2(q12.x)[f]);
		const float g2 = h2f(unpack_h2(q12.y)[f]);
#pragma unroll
		for (uint32_t yz = 0; yz < 4; ++yz) {
			const uint32_t c[3] = {dx, yz & 1u, yz >> 1};
			const long long add = corner_addend_fixed(g1, g2, scale, sm.dn, pos, c);
			if (add != 0) atomic_add_fixed(gg + (size_t)grid_entry(hashmap_size, res, pg[0] + c[0], pg[1] + c[1], pg[2] + c[2]) * 2 + f, add);
		}
	}
}

// middle levels: the run-length walk of grid_scatter_quad_rl_body (operands loaded four samples ahead), integer run sums
__global__ __launch_bounds__(256) void k_grid_scatter_quad_rl_fixed(const GridMeta G, const ScatterArgs a, const uint32_t level0, const ScatterRlPlan plan) {
#pragma unroll 1
	for (uint32_t vb = blockIdx.x; vb < plan.wg_start[plan.n]; vb += gridDim.x) {
		uint32_t li = 0;
#pragma unroll 1
		for (uint32_t q = 1; q < plan.n; ++q) if (vb >= plan.wg_start[q]) li = q;
		const uint32_t level = level0 + li;
		const uint32_t K = 1u << ((plan.k_log2 >> (4 * li)) & 15u);
		if (level > G.valid_level) continue;
		const uint32_t t = (vb - plan.wg_start[li]) * blockDim.x + threadIdx.x;
		const uint32_t s0 = (t >> 2) * K;
		if (s0 >= a.B) continue;
		const uint32_t dx = (t >> 1) & 1u, f = t & 1u;
		unsigned long long* gg = a.grid_fixed + (size_t)G.offsets[level] * 2;
		const uint32_t hashmap_size = G.offsets[level + 1] - G.offsets[level];
		const float scale = G.scale[level];
		const uint32_t res = G.resolution[level];
		long long acc[4] = {0, 0, 0, 0};
		uint32_t cur[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
		auto flush = [&]() {
#pragma unroll
			for (uint32_t yz = 0; yz < 4; ++yz) {
				if (acc[yz] != 0) {
					atomic_add_fixed(gg + (size_t)grid_entry(hashmap_size, res, cur[0] + dx, cur[1] + (yz & 1u), cur[2] + (yz >> 1)) * 2 + f, acc[yz]);
					acc[yz] = 0;
				}
			}
		};
		const uint32_t s_end = min(s0 + K, a.B);
		const uint2* g12l = reinterpret_cast<const uint2*>(a.g12) + (size_t)level * a.B;
		constexpr uint32_t C = 4;
#pragma unroll 1
		for (uint32_t sc = s0; sc < s_end; sc += C) {
			ScatterSample sm[C];
			uint2 q12[C];
#pragma unroll
			for (uint32_t j = 0; j < C; ++j) { const uint32_t s = min(sc + j, s_end - 1); sm[j] = load_srec(a.srec, s); q12[j] = g12l[s]; }
#pragma unroll
			for (uint32_t j = 0; j < C; ++j) {
				if (sc + j >= s_end) break;
				float pos[3];
				uint32_t pg[3];
				pos_fract(sm[j].x, scale, &pos[0], &pg[0]);
				pos_fract(sm[j].y, scale, &pos[1], &pg[1]);
				pos_fract(sm[j].z, scale, &pos[2], &pg[2]);
				if (pg[0] != cur[0] || pg[1] != cur[1] || pg[2] != cur[2]) {
					if (cur[0] != 0xffffffffu) flush();
					cur[0] = pg[0]; cur[1] = pg[1]; cur[2] = pg[2];
				}
				const float g1 = h2f(unpack_h2(q12[j].x)[f]);
				const float g2 = h2f(unpack_h2(q12[j].y)[f]);
#pragma unroll
				for (uint32_t yz = 0; yz < 4; ++yz) {
					const uint32_t c[3] = {dx, yz & 1u, yz >> 1};
					acc[yz] += corner_addend_fixed(g1, g2, scale, sm[j].dn, pos, c);
				}
			}
		}
		flush();
	}
}

// The one rounding of the deterministic mode: exact integer sum -> the gradient vector of the accumulate mode (fp32 accumulator, or half), entries [lo, hi) of the grid's
// parameters; the accumulators are left clear for the next step. An entry nothing was added to is neither written (the optimizer left the vector clear) nor cleared.
__device__ __forceinline__ float fixed24_to_float(const long long v) { return (float)v * 5.9604644775390625e-08f; } // (float)int64: round to nearest even; x 2^-24: exact
__global__ __launch_bounds__(256) void k_fixed_narrow(unsigned long long* __restrict__ fixed, float* __restrict__ grads, half_t* __restrict__ grads16, const uint64_t lo, const uint64_t hi) {
	// two entries (both features of a table entry) per thread: lo and hi are even
	for (uint64_t i = lo + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < hi; i += (uint64_t)gridDim.x * blockDim.x * 2) {
		const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(fixed + i);
		if (v.x == 0ull && v.y == 0ull) continue;
		*reinterpret_cast<ulonglong2*>(fixed + i) = ulonglong2{0ull, 0ull};
		const float g0 = fixed24_to_float((long long)v.x), g1 = fixed24_to_float((long long)v.y);
		if (grads16) { grads16[i] = f2h(g0); grads16[i + 1] = f2h(g1); }
		else { grads[i] = g0; grads[i + 1] = g1; }
	}
}

// ---------------------------------------------------------------------------------------------
// K12: Adam (adam.h:52-202) + EMA (ema.h:63-78), one pass; consumes and clears the gradient accumulators.
// ---------------------------------------------------------------------------------------------
struct AdamArgs {
	uint64_t n;
	uint64_t n_matrix;
	float* rec;                // optimizer records, one 64-byte record per 4-parameter group: {fp32 weight x4 | m x4 | v x4 | step count x4} (see k_adam_ema)
	half_t* w16; half_t* ema;
	float* grads;
	half_t* grads16;           // rnb_config::accumulate = RNB_ACCUM_HALF: the gradient vector is half (trainer.h:78-84) -- 2 bytes read and 2 cleared per parameter instead of 4 + 4
	float base_lr, beta1, beta2, epsilon, l2_reg;
	float ema_decay, ema_debias_old, ema_debias_new;
	uint64_t begin, end;       // parameter range of this launch (multiples of 4)
	uint64_t skip_lo, skip_hi; // only_sdf_training: parameters [skip_lo, skip_hi) (the colour MLP) get no Adam update (adam.h:121-165)
	const float* lr_table;     // [lr_table_n] bias-correction factor by per-parameter step count (k_adam_lr_table); entry 0 unused
	uint32_t lr_table_n;
};

// The bias correction of adam.h:182-183, sqrt(1 - beta2^t) / (1 - beta1^t), depends on the parameter's own step count t only. Evaluated per
// parameter it was most of k_adam_ema: two powf per live parameter made the kernel VALU-bound (3.6 k wave instructions per wavefront, 110 us
// of issue time for a 240 MB stream, profiles/r02_pmc_sq.json). The table holds the same expression, evaluated by the same device functions,
// for t < lr_table_n; a step count beyond the table is computed in place.
__device__ __noinline__ float adam_bias_correction(const float beta1, const float beta2, const uint32_t t) {
	return sqrtf(1 - powf(beta2, (float)t)) / (1 - powf(beta1, (float)t));
}
__global__ void k_adam_lr_table(float* __restrict__ table, const uint32_t n, const float beta1, const float beta2) {
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) table[t] = t ? adam_bias_correction(beta1, beta2, t) : 0.f;
}

// Four parameters per thread (n_params, n_matrix are multiples of 4): 16-byte fp32 / 8-byte fp16 accesses. Entries of
// the hash grid whose gradient is zero only take the EMA path (adam.h:111-114), i.e. 10 B of traffic per parameter.
// Optimizer state layout (round 4): the fp32 master weight, both moments and the per-parameter step count (trainer.h:78-84, adam.h:100-110:
// four arrays in the reference, and here until round 3) of a 4-parameter group are ONE 64-byte record. About a third of the hash grid's
// groups are live in a step, at random: as four arrays a live group pulled a quarter of four 64-byte lines, and 83 % of all lines held a
// live group (~400 MB moved for 249 algorithmic MB); as a record it moves its own line. The plain arrays of rnb_buffer
// (PARAMS_FP32 / ADAM_M / ADAM_V / ADAM_STEPS) are staging views packed into / unpacked from the records on demand (k_opt_records_pack / _unpack).
__global__ __launch_bounds__(256) void k_adam_ema(const AdamArgs a) {
	const uint64_t q_end = a.end / 4;
	for (uint64_t q = a.begin / 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < q_end; q += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t i0 = q * 4;
		f4 graw;
		if (a.grads16) {
			const h4 g16 = reinterpret_cast<const h4*>(a.grads16)[q];
			graw = f4{h2f(g16[0]), h2f(g16[1]), h2f(g16[2]), h2f(g16[3])};
			if (graw[0] != 0.f || graw[1] != 0.f || graw[2] != 0.f || graw[3] != 0.f) reinterpret_cast<h4*>(a.grads16)[q] = h4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
		} else {
			graw = reinterpret_cast<const f4*>(a.grads)[q];
			if (graw[0] != 0.f || graw[1] != 0.f || graw[2] != 0.f || graw[3] != 0.f) reinterpret_cast<f4*>(a.grads)[q] = f4{0.f, 0.f, 0.f, 0.f}; // leave the accumulators clear
		}
		h4 w16 = reinterpret_cast<const h4*>(a.w16)[q];
		const h4 ema = reinterpret_cast<const h4*>(a.ema)[q];
		const bool is_matrix = i0 < a.n_matrix;
		bool any = is_matrix;
		float gradient[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) { gradient[k] = rh(graw[k]) * (1.0f / LOSS_SCALE); any = any || gradient[k] != 0.f; } // the reference's gradient vector is half (trainer.h:78-84)
		if (i0 >= a.skip_lo && i0 < a.skip_hi) any = false;
		if (any) {
			f4* rec = reinterpret_cast<f4*>(a.rec) + q * 4;
			f4 w32 = rec[0];
			f4 m = rec[1];
			f4 v = rec[2];
			const f4 st = rec[3];
			uint32_t stp[4] = {__float_as_uint(st[0]), __float_as_uint(st[1]), __float_as_uint(st[2]), __float_as_uint(st[3])};
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				if (!(is_matrix || gradient[k] != 0.f)) continue; // adam.h:111-114
				const float weight_fp = w32[k];
				float g = gradient[k];
				if (is_matrix) g += a.l2_reg * weight_fp;
				const float gradient_sq = g * g;
				const float first_moment = m[k] = a.beta1 * m[k] + (1 - a.beta1) * g;
				const float second_moment = v[k] = a.beta2 * v[k] + (1 - a.beta2) * gradient_sq;
				float learning_rate = a.base_lr;
				const uint32_t cs = ++stp[k];
				float correction;
				if (__builtin_expect(cs < a.lr_table_n, 1)) correction = a.lr_table[cs];
				else correction = adam_bias_correction(a.beta1, a.beta2, cs); // (a call: the two powf expansions stay out of the loop body)
				learning_rate *= correction;
				const float effective_learning_rate = fminf(fmaxf(learning_rate / (sqrtf(second_moment) + a.epsilon), 0.f), 3.402823466e+38f);
				const float decayed_weight = (1 - 0.f * learning_rate) * weight_fp - copysignf(0.f * learning_rate, weight_fp);
				const float new_weight = decayed_weight - effective_learning_rate * first_moment;
				w32[k] = new_weight;
				w16[k] = f2h(new_weight);
			}
			rec[0] = w32;
			rec[1] = m;
			rec[2] = v;
			rec[3] = f4{__uint_as_float(stp[0]), __uint_as_float(stp[1]), __uint_as_float(stp[2]), __uint_as_float(stp[3])};
			reinterpret_cast<h4*>(a.w16)[q] = w16;
		}
		h4 e;
#pragma unroll
		for (int k = 0; k < 4; ++k) e[k] = f2h((h2f(ema[k]) * a.ema_decay * a.ema_debias_old + h2f(w16[k]) * (1 - a.ema_decay)) * a.ema_debias_new);
		reinterpret_cast<h4*>(a.ema)[q] = e;
	}
}
// (Round 4 measured the hash grid's EMA as a launch of its own behind the step's last Adam chunk -- nothing of the next step reads the EMA weights, so its
// 63 MB need not share the memory system with the scatter: 0.629 vs 0.627 ms/step over the window with it; dropped, profiles/r04_ab_sync.txt.)

// The staging views of rnb_buffer <-> the records: group q = parameters 4 q .. 4 q + 3. Bit copies (the step counts travel as raw words).
__global__ __launch_bounds__(256) void k_opt_records_pack(const uint64_t n_groups, const float* __restrict__ w32, const float* __restrict__ m, const float* __restrict__ v,
                                                          const uint32_t* __restrict__ steps, float* __restrict__ rec) {
	for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_groups; q += (uint64_t)gridDim.x * blockDim.x) {
		f4* r = reinterpret_cast<f4*>(rec) + q * 4;
		r[0] = reinterpret_cast<const f4*>(w32)[q];
		r[1] = reinterpret_cast<const f4*>(m)[q];
		r[2] = reinterpret_cast<const f4*>(v)[q];
		r[3] = reinterpret_cast<const f4*>(steps)[q];
	}
}
__global__ __launch_bounds__(256) void k_opt_records_unpack(const uint64_t n_groups, const float* __restrict__ rec, float* __restrict__ w32, float* __restrict__ m, float* __restrict__ v,
                                                            uint32_t* __restrict__ steps) {
	for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_groups; q += (uint64_t)gridDim.x * blockDim.x) {
		const f4* r = reinterpret_cast<const f4*>(rec) + q * 4;
		reinterpret_cast<f4*>(w32)[q] = r[0];
		reinterpret_cast<f4*>(m)[q] = r[1];
		reinterpret_cast<f4*>(v)[q] = r[2];
		reinterpret_cast<f4*>(steps)[q] = r[3];
	}
}

__global__ void k_fp32_to_half(const float* __restrict__ src, half_t* __restrict__ dst, uint64_t n) {
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = f2h(src[i]);
}

// random.h:67-93 generate_random_uniform<float>: thread i draws elements i + n_threads*j (j < 4) from stream offset 4*i.
__global__ void k_random_uniform(Pcg32 rng, uint64_t n, uint64_t n_threads_total, float lower, float upper, float* __restrict__ out) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_threads_total) return;
	rng.advance((int64_t)i * 4);
	for (uint64_t j = 0; j < 4; ++j) {
		const uint64_t idx = i + n_threads_total * j;
		if (idx >= n) return;
		const float val = rng.next_float();
		out[idx] = val * (upper - lower) + lower;
	}
}

} // namespace rnb
