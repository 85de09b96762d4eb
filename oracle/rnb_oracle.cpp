/*
 * oracle/rnb_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the RNb-NeuS2 training hot path,
 * exported with the C-ABI of include/rnb_neus2.h under the prefix orc_ and host
 * pointers everywhere. It follows the reference function by function; every
 * block cites the file:line it restates (paths relative to /root/reference).
 *
 * PARITY: the reference has no CPU path, no tests and no golden vectors for this path and cannot be compiled in this image as a whole
 * (nvcuda::wmma, CUTLASS, cuRAND). What it is pinned by is the reference's OWN CODE compiled for the host piece by piece (tests/golden/README.md): its functions
 * (tests/golden/int_fixtures.json) and, since round 5, the bodies of its kernels behind their index lines or as runs of their own lines (float_fixtures.json) -- sampler
 * (pinhole ray, both march loops), kernel_grid, the loss kernel (per-ray targets, ray loss terms, one iteration of the backward loop), adam_step + EMA, the occupancy
 * update's sample generation and bitfield kernels, sdf -> density, the ray-batch controller: bit for bit. PARITY UNPINNED for what that cannot reach: the two fully fused
 * MLPs (wmma fragments, CUTLASS) and the addends of the grid backward kernels (their host-visible branch needs CUDA's atomicAdd), and for everything only a CUDA
 * binary decides (FMA contraction, the device's libm). Documented deviations from the reference:
 *   D1 MLP dot products accumulate in fp32 and round to half once per neuron
 *      (reference: WMMA fp16 accumulators, fully_fused_mlp.cu:68,198).
 *   D2 hash-grid and weight gradients accumulate in fp32 and are rounded to half
 *      once (reference: order-dependent __half2 atomics, grid.h:410-430; the
 *      addends are still rounded to half first, as the reference does).
 *   D3 sample slots are assigned in ray order by prefix sums (reference: atomicAdd
 *      order, testbed_nerf.cu:1352,1359,1722 — any order is a legal outcome).
 *   D4 the per-ray light index is PCG32 draw #7 of the ray's stream, mod 3
 *      (reference: curand_init(clock64(), ...) — irreproducible, testbed_nerf.cu:1557-1561).
 *   D5 density-grid mean is summed in fp64 (reference: fp32 tree reduce_sum).
 * rnb_config::accumulate = RNB_ACCUM_HALF switches D1 and D2 to a MODEL of the reference's half accumulation (dot_h, emulated_dw,
 * half atomics in sample order) -- what the HIP library's mode of the same name is compared with; ORC_EMULATE_FP16_ACCUM=1 /
 * ORC_EMULATE_HALF_ATOMICS=1 in the environment of orc_create switch the two parts on one by one so that their sizes can be measured
 * (tools/oracle_deviation_report.py).
 *
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may load
 * this library.
 */
#include "orc_common.h"

// Export the rnb_neus2.h signatures under the orc_ prefix.
#include "orc_prefix.h"
#include "../include/rnb_neus2.h"
#include "orc_mesh.h" // the checker's own marching cubes (no source shared with the product's mesh code)

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

constexpr uint32_t GRIDSIZE = RNB_GRIDSIZE;
constexpr uint32_t GRID_CELLS = GRIDSIZE * GRIDSIZE * GRIDSIZE;
constexpr uint32_t N_CASCADES = RNB_CASCADES;
constexpr float LOSS_SCALE = 128.f;                 // testbed.h:237
constexpr uint32_t N_MAX_RANDOM_SAMPLES_PER_RAY = 8; // testbed_nerf.cu:60
constexpr float SQRT3 = 1.73205080757f;             // testbed_nerf.cu:52
constexpr float STEPSIZE = SQRT3 / 1024;            // testbed_nerf.cu:53
constexpr float MIN_CONE_STEPSIZE = STEPSIZE;
constexpr float MAX_CONE_STEPSIZE = STEPSIZE * (1 << (N_CASCADES - 1)) * 1024 / GRIDSIZE; // testbed_nerf.cu:56
constexpr float MIN_OPTICAL_THICKNESS = 0.1f;       // testbed_nerf.cu:66

struct Vec3 { float x, y, z; };
static inline Vec3 v3(float x, float y, float z) { return {x, y, z}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
// Eigen's reduction of a fixed-size vector splits the range in halves (Redux.h, redux_novec_unroller; no SIMD under a GPU compiler): three terms are x0 + (x1 + x2), four
// (x0 + x1) + (x2 + x3) -- what .dot(), .norm(), .normalized() and the fixed-size matrix products of the reference's kernels evaluate (pinned by tests/golden/float_fixtures.json,
// which runs the kernel's own statements through the reference's vendored Eigen). Sums the reference spells out term by term stay left to right.
static inline float esum3(float x0, float x1, float x2) { return x0 + (x1 + x2); }
static inline float esum4(float x0, float x1, float x2, float x3) { return (x0 + x1) + (x2 + x3); }
static inline float dot(Vec3 a, Vec3 b) { return esum3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float norm(Vec3 a) { return sqrtf(dot(a, a)); }
static inline Vec3 normalized(Vec3 a) { float n = norm(a); return {a.x / n, a.y / n, a.z / n}; }

struct View {
	rnb_view meta;
	std::vector<uint16_t> normal, albedo;
};

// Per-sample forward activations kept for the backward pass (NerfNetwork::ForwardContext, nerf_network.h:1104-1124).
struct FwdCtx {
	float x[3];
	float dir[3];
	half_t feat[28];
	float dy_dx[28][3];
	half_t sdf_in[32];
	half_t z1[64];
	half_t sdf_out[16];
	half_t dz1[64];       // W1[0,:] masked by relu'(z1): SDF-MLP backward of e0
	half_t dsdf_din[32];  // d sdf_out[0] / d sdf_in
	float grad[3];        // ∇sdf (dSDF_dPos)
	half_t c_in[48];
	half_t h1[64];
	half_t h2[64];
	half_t r[16];
};

} // namespace

struct orc_ctx_s {
	rnb_config cfg;
	// grid tables (grid.h:977-1012)
	uint32_t offsets[RNB_MAX_LEVELS + 1];
	uint32_t resolution[RNB_MAX_LEVELS];
	float scale[RNB_MAX_LEVELS];
	uint64_t n_grid_params = 0;
	uint64_t n_params = 0;
	uint64_t off_sdf = 0, off_rgb = 0, off_grid = 0, off_var = 0;
	uint32_t max_cascade = 0;
	float aabb_min = 0.f, aabb_max = 1.f;
	float cone_angle = 0.f;

	std::vector<float> params_fp32, grads, adam_m, adam_v;
	std::vector<half_t> params_fp16, params_ema;
	std::vector<uint32_t> adam_steps;
	std::vector<float> density_grid, density_grid_tmp;
	std::vector<uint8_t> bitfield;
	float density_mean = 0.f;
	std::vector<float> grid_sample_pos;
	std::vector<uint32_t> grid_sample_idx;

	std::vector<View> views;
	float light_dirs[9]; // row-major 3x3 (testbed_nerf.cu:1537-1554)

	// step scratch (train_nerf_step, testbed_nerf.cu:3850-3889)
	std::vector<uint32_t> ray_indices, numsteps;
	std::vector<float> rays, coords, coords_compacted, loss, ek_loss, mask_loss;
	std::vector<half_t> mlp_out, dloss_dout;
	uint32_t counters[4] = {0, 0, 0, 0};

	// RNG state (testbed.cu:2223-2237)
	Pcg32 rng, density_grid_rng, trainer_rng;
	uint32_t density_grid_ema_step = 0;

	// controller (Counters, testbed.h:627-640)
	uint32_t training_step = 0, cur_step = 0;
	double step_vector[7] = {0, 0, 0, 0, 0, 0, 0}; // {counters[0..3], loss sums[0..2]} of the last rnb_train_step_local
	uint32_t valid_level = 0;
	uint32_t rays_per_batch = 0;
	uint32_t measured_batch_size = 0;
	uint32_t measured_batch_size_before_compaction = 0;
	uint32_t n_rays_total = 0;
	rnb_grid_exchange_fn grid_exchange = nullptr; // rnb_set_grid_exchange
	void* grid_exchange_user = nullptr;
	uint32_t optimizer_step_count = 0;
	bool opt_begun = false;
	uint64_t param_capacity = 0; // allocated length of the parameter-shaped arrays (equal data-parallel shards)
	float lr_factor = 1.f;
	// begin/end hand-off
	uint32_t cur_n_rays = 0, local_measured_before = 0;
	bool grid_updated = false;
	float prep_ms = 0.f;
	std::chrono::steady_clock::time_point step_start;
	// Emulation of the reference's half accumulation (off by default: deviations D1 / D2 are what the HIP path implements)
	bool emul_fp16_acc = false;     // D1 off: MLP dot products and weight-gradient GEMMs accumulate in half (WMMA / CUTLASS half accumulators)
	bool emul_half_atomics = false; // D2 off: hash-grid gradients accumulate in half, one rounding per atomicAdd(__half2), in sample order
	uint32_t atomic_order_seed = 0; // ORC_ATOMIC_ORDER_SEED != 0: ... in a seeded random order of the samples instead -- any order is a legal outcome of the reference's atomics,
	                                // and the distance between two orders is the floor below which no implementation of half atomics can be compared (tests/test_gpu_fullsize.py)
	std::vector<half_t> grads16;    // accumulate = RNB_ACCUM_HALF: the half gradient vector (RNB_BUF_GRADS_FP16), filled from `grads` at the end of the backward pass
};

namespace {

// ======================================================================
// Hash grid (dependencies/neus2_tcnn/include/tiny-cuda-nn/encodings/grid.h)
// ======================================================================

void build_grid_tables(orc_ctx_s* c) {
	// grid.h:977-1012
	const rnb_config& cfg = c->cfg;
	uint32_t offset = 0;
	for (uint32_t i = 0; i < cfg.n_levels; ++i) {
		const float scale = exp2f(i * std::log2(cfg.per_level_scale)) * cfg.base_resolution - 1.0f;
		const uint32_t resolution = (uint32_t)(ceilf(scale)) + 1;
		c->scale[i] = (float)(resolution - 1); // fork: integer scale, grid.h:981-983
		c->resolution[i] = resolution;
		uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
		uint32_t params_in_level = std::pow((float)resolution, 3) > (float)max_params ? max_params : resolution * resolution * resolution;
		params_in_level = next_multiple(params_in_level, 8u);
		params_in_level = std::min(params_in_level, (1u << cfg.log2_hashmap_size));
		c->offsets[i] = offset;
		offset += params_in_level;
	}
	c->offsets[cfg.n_levels] = offset;
	c->n_grid_params = (uint64_t)offset * 2;
}

// grid.h:113-148, N_DIMS=3, N_FEATURES_PER_LEVEL=2, GridType::Hash. Returns the entry index (feature 0 at 2*idx).
static inline uint32_t grid_entry(uint32_t hashmap_size, uint32_t res, const uint32_t p[3]) {
	uint32_t stride = 1;
	uint32_t index = 0;
	for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
		index += p[dim] * stride;
		stride *= res;
	}
	if (hashmap_size < stride) {
		index = (p[0] * 1u) ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u);
	}
	return index % hashmap_size;
}

// common_device.h:403-434 (fork: +0.5 offset), Linear interpolation: derivative 1.
static inline void pos_fract(float input, float* pos, uint32_t* pos_grid, float scale) {
	*pos = input * scale + 0.5f;
	int tmp = (int)floorf(*pos);
	*pos_grid = (uint32_t)tmp;
	*pos -= (float)tmp;
}

// kernel_grid (grid.h:169-364) for one sample, all levels. `grid` = half table (training or EMA weights).
void encode_sample(const orc_ctx_s* c, const half_t* grid, const float x[3], half_t feat[28], float (*dy_dx)[3]) {
	const uint32_t L = c->cfg.n_levels;
	for (uint32_t level = 0; level < L; ++level) {
		if (level > c->valid_level) { // grid.h:192-210
			feat[level * 2 + 0] = 0; feat[level * 2 + 1] = 0;
			if (dy_dx) for (int f = 0; f < 2; ++f) for (int d = 0; d < 3; ++d) dy_dx[level * 2 + f][d] = 0.f;
			continue;
		}
		const half_t* g = grid + (uint64_t)c->offsets[level] * 2;
		const uint32_t hashmap_size = c->offsets[level + 1] - c->offsets[level];
		const float scale = c->scale[level];
		const uint32_t res = c->resolution[level];
		float pos[3]; uint32_t pg[3];
		for (int d = 0; d < 3; ++d) pos_fract(x[d], &pos[d], &pg[d], scale);

		// grid.h:287-321: N-linear interpolation, accumulated in half
		half_t result[2] = {0, 0};
		for (uint32_t idx = 0; idx < 8; ++idx) {
			float weight = 1;
			uint32_t pl[3];
			for (uint32_t d = 0; d < 3; ++d) {
				if ((idx & (1u << d)) == 0) { weight *= 1 - pos[d]; pl[d] = pg[d]; }
				else { weight *= pos[d]; pl[d] = pg[d] + 1; }
			}
			const uint32_t e = grid_entry(hashmap_size, res, pl);
			for (int f = 0; f < 2; ++f) {
				float data = h2f(g[e * 2 + f]);
				result[f] = hadd(result[f], f2h(weight * data)); // grid.h:313
			}
		}
		feat[level * 2 + 0] = result[0];
		feat[level * 2 + 1] = result[1];

		if (dy_dx) { // grid.h:324-363
			float grads[2][3] = {{0, 0, 0}, {0, 0, 0}};
			for (uint32_t gd = 0; gd < 3; ++gd) {
				for (uint32_t idx = 0; idx < 4; ++idx) {
					float weight = scale;
					uint32_t pl[3];
					for (uint32_t ngd = 0; ngd < 2; ++ngd) {
						const uint32_t d = ngd >= gd ? (ngd + 1) : ngd;
						if ((idx & (1u << ngd)) == 0) { weight *= 1 - pos[d]; pl[d] = pg[d]; }
						else { weight *= pos[d]; pl[d] = pg[d] + 1; }
					}
					pl[gd] = pg[gd];
					const uint32_t el = grid_entry(hashmap_size, res, pl);
					pl[gd] = pg[gd] + 1;
					const uint32_t er = grid_entry(hashmap_size, res, pl);
					for (int f = 0; f < 2; ++f) {
						grads[f][gd] += weight * (h2f(g[er * 2 + f]) - h2f(g[el * 2 + f])) * 1.0f;
					}
				}
			}
			for (int f = 0; f < 2; ++f) for (int d = 0; d < 3; ++d) dy_dx[level * 2 + f][d] = grads[f][d];
		}
	}
	for (uint32_t k = L * 2; k < 28; ++k) { // fewer than 14 levels: remaining features stay zero
		feat[k] = 0;
		if (dy_dx) for (int d = 0; d < 3; ++d) dy_dx[k][d] = 0.f;
	}
}

// ======================================================================
// Fully fused MLPs (dependencies/neus2_tcnn/src/fully_fused_mlp.cu). ReLU hidden, no output activation.
// Weights row-major [out][in] (fully_fused_mlp.cu:786-819). Deviation D1: fp32 accumulate.
// ======================================================================

// Dot product of n half pairs (a[i * sa], b[i * sb]). Default (deviation D1): fp32 accumulation.
// acc16: EMULATION MODEL of a tensor-core path with half accumulators (wmma 16x16x16 fragments of __half,
// fully_fused_mlp.cu:68,198; CUTLASS TypeAccumulator = half, cutlass_matmul.h:83): products exact, the 16 products of one
// k-step summed in fp32, the running accumulator rounded to half after every k-step. The order inside a k-step is not
// documented by the vendor; this is the usual model, used only to SIZE the deviation.
static inline float dot_h(const half_t* a, int sa, const half_t* b, int sb, int n, bool acc16) {
	if (!acc16) {
		float acc = 0.f;
		for (int i = 0; i < n; ++i) acc += h2f(a[(size_t)i * sa]) * h2f(b[(size_t)i * sb]);
		return acc;
	}
	half_t acc = 0;
	for (int i0 = 0; i0 < n; i0 += 16) {
		float part = 0.f;
		for (int i = i0; i < std::min(n, i0 + 16); ++i) part += h2f(a[(size_t)i * sa]) * h2f(b[(size_t)i * sb]);
		acc = f2h(h2f(acc) + part);
	}
	return h2f(acc);
}

static inline void matvec(const half_t* W, int n_out, int n_in, const half_t* in, half_t* out, bool relu, bool acc16 = false) {
	for (int o = 0; o < n_out; ++o) {
		float acc = dot_h(W + (size_t)o * n_in, 1, in, 1, n_in, acc16);
		if (relu && !(acc > 0.f)) acc = 0.f; // warp_activation ReLU, common_device.h:69-115
		out[o] = f2h(acc);
	}
}
// out[i] = sum_o W[o][i] * in[o]; optional ReLU transfer using the forward activation (common_device.h:182 ff.)
static inline void matvec_t(const half_t* W, int n_out, int n_in, const half_t* in, half_t* out, const half_t* fwd_act, bool acc16 = false) {
	for (int i = 0; i < n_in; ++i) {
		float acc = dot_h(W + i, n_in, in, 1, n_out, acc16);
		if (fwd_act && !(h2f(fwd_act[i]) > 0.f)) acc = 0.f;
		out[i] = f2h(acc);
	}
}

struct NetParams {
	const half_t* sdf_w0; // [64][32]
	const half_t* sdf_w1; // [16][64]
	const half_t* rgb_w0; // [64][48]
	const half_t* rgb_w1; // [64][64]
	const half_t* rgb_w2; // [16][64]
	const half_t* grid;
	half_t variance;
	bool acc16; // orc_ctx_s::emul_fp16_acc
};

NetParams net_params(const orc_ctx_s* c, bool inference) {
	const half_t* p = inference ? c->params_ema.data() : c->params_fp16.data();
	NetParams n;
	n.sdf_w0 = p + c->off_sdf;
	n.sdf_w1 = n.sdf_w0 + 64 * 32;
	n.rgb_w0 = p + c->off_rgb;
	n.rgb_w1 = n.rgb_w0 + 64 * 48;
	n.rgb_w2 = n.rgb_w1 + 64 * 64;
	n.grid = p + c->off_grid;
	n.variance = p[c->off_var];
	n.acc16 = c->emul_fp16_acc;
	return n;
}

// NerfNetwork::sdf (nerf_network.h:454-520): encode (no dy_dx) -> [x-0.5 | feats | 0] -> SDF MLP -> out[0] + bias.
half_t sdf_sample(const orc_ctx_s* c, const NetParams& np, const float x[3]) {
	half_t feat[28];
	encode_sample(c, np.grid, x, feat, nullptr);
	half_t in[32];
	for (int d = 0; d < 3; ++d) in[d] = hsub(f2h(x[d]), f2h(0.5f)); // common_operation.cuh:187-199
	for (int k = 0; k < 28; ++k) in[3 + k] = feat[k];
	in[31] = 0;
	half_t z1[64], out[16];
	matvec(np.sdf_w0, 64, 32, in, z1, true, np.acc16);
	matvec(np.sdf_w1, 16, 64, z1, out, false, np.acc16);
	return hadd(out[0], f2h(c->cfg.sdf_bias)); // common_operation.cuh:299-309
}

// sdf_to_density_variance_buffer (common_operation.cuh:311-328), all in half arithmetic.
half_t sdf_to_density(half_t sdf, half_t variance) {
	half_t s = f2h(expf(h2f(hmul(variance, f2h(10.0f)))));
	half_t sig = f2h(1.0f / (1.0f + expf(-h2f(hmul(sdf, s)))));
	half_t density = hmul(hmul(s, sig), hsub(f2h(1.0f), sig));
	return density;
}

// NerfNetwork::forward_impl (nerf_network.h:97-253) for one sample.
void forward_sample(const orc_ctx_s* c, const NetParams& np, const float coord[7], half_t out[16], FwdCtx* ctx_out) {
	FwdCtx local;
	FwdCtx& k = ctx_out ? *ctx_out : local;
	for (int d = 0; d < 3; ++d) { k.x[d] = coord[d]; k.dir[d] = coord[4 + d]; }
	// pos encoding with input gradients (nerf_network.h:139-146)
	encode_sample(c, np.grid, k.x, k.feat, k.dy_dx);
	// density_network_input = [xyz - 0.5 | encoding | 0] (nerf_network.h:149-155)
	for (int d = 0; d < 3; ++d) k.sdf_in[d] = hsub(f2h(k.x[d]), f2h(0.5f));
	for (int j = 0; j < 28; ++j) k.sdf_in[3 + j] = k.feat[j];
	k.sdf_in[31] = 0;
	// SDF MLP forward (nerf_network.h:159-160)
	matvec(np.sdf_w0, 64, 32, k.sdf_in, k.z1, true, np.acc16);
	matvec(np.sdf_w1, 16, 64, k.z1, k.sdf_out, false, np.acc16);
	// SDF MLP backward of dL/dout = e0, parameter gradients ignored (nerf_network.h:163-176)
	for (int j = 0; j < 64; ++j) {
		float v = h2f(np.sdf_w1[0 * 64 + j]) * 1.0f;
		if (!(h2f(k.z1[j]) > 0.f)) v = 0.f;
		k.dz1[j] = f2h(v);
	}
	matvec_t(np.sdf_w0, 64, 32, k.dz1, k.dsdf_din, nullptr, np.acc16);
	// encoding backward to the input (grid.h:527-554) + the direct xyz path (nerf_network.h:177-189)
	float g[3] = {0.f, 0.f, 0.f};
	for (int j = 0; j < 28; ++j) {
		float dl = h2f(k.dsdf_din[3 + j]);
		for (int d = 0; d < 3; ++d) g[d] += dl * k.dy_dx[j][d];
	}
	for (int d = 0; d < 3; ++d) { g[d] += h2f(k.dsdf_din[d]); k.grad[d] = g[d]; }
	// rgb_network_input = [sdf_out(16) | 0(16) | xyz | ∇sdf | 0] (nerf_network.h:206-218)
	for (int j = 0; j < 48; ++j) k.c_in[j] = 0;
	for (int j = 0; j < 16; ++j) k.c_in[j] = k.sdf_out[j];
	for (int d = 0; d < 3; ++d) { k.c_in[32 + d] = f2h(k.x[d]); k.c_in[35 + d] = f2h(k.grad[d]); }
	matvec(np.rgb_w0, 64, 48, k.c_in, k.h1, true, np.acc16);
	matvec(np.rgb_w1, 64, 64, k.h1, k.h2, true, np.acc16);
	matvec(np.rgb_w2, 16, 64, k.h2, k.r, false, np.acc16);
	// output packing (nerf_network.h:221-250)
	for (int j = 0; j < 16; ++j) out[j] = k.r[j];
	out[3] = hadd(k.sdf_out[0], f2h(c->cfg.sdf_bias));
	for (int d = 0; d < 3; ++d) out[4 + d] = f2h(k.grad[d]);
	out[7] = np.variance;
	for (int d = 0; d < 3; ++d) out[8 + d] = f2h(k.dir[d]);
}

// Per-thread gradient accumulators for the MLP weights (dense) — grid gradients go to the shared fp32 buffer.
struct MlpGrads {
	std::vector<float> g1; // first-order pass (EGradientMode::Overwrite)
	std::vector<float> g2; // second-order pass (EGradientMode::Accumulate), SDF MLP only
	double var = 0.0;
	MlpGrads() : g1(RNB_N_SDF_MLP_PARAMS + RNB_N_RGB_MLP_PARAMS, 0.f), g2(RNB_N_SDF_MLP_PARAMS, 0.f) {}
};

static inline void outer_acc(float* dW, int n_out, int n_in, const half_t* dout, const half_t* in) {
	for (int o = 0; o < n_out; ++o) {
		const float d = h2f(dout[o]);
		if (d == 0.f) continue;
		float* row = dW + (size_t)o * n_in;
		for (int i = 0; i < n_in; ++i) row[i] += d * h2f(in[i]);
	}
}

static inline void atomic_add(float* p, float v) {
#pragma omp atomic
	*p += v;
}
// rnb_config::deterministic (include/rnb_neus2.h): a half-valued addend (grid.h:415-416) as a 64-bit fixed-point integer at scale 2^24 -- exact; integer sums commute
static inline void atomic_add_fixed(int64_t* p, float half_valued) {
	const int64_t v = (int64_t)(half_valued * 16777216.0f);
#pragma omp atomic
	*p += v;
}
static inline float fixed24_to_float(int64_t v) { return (float)v * 5.9604644775390625e-08f; }

// One sample's operands of the weight-gradient GEMMs and of the grid scatter, kept when the accumulation itself is emulated in
// a second pass (orc_ctx_s::emul_*).
struct SampleOps {
	half_t dr[16], h2[64], dh2[64], h1[64], dh1[64], c_in[48], dso[16], z1[64], dz[64], sdf_in[32], dz1[64], ddin[32], front[64];
	float x[3], dn[3];
	half_t dsin[32], dsdf_din[32];
};

// Hash-grid scatter of one sample at one level: first order (kernel_grid_backward, grid.h:366-495; addend (float)grad * weight
// narrowed to half, grid.h:415-416) and second order (kernel_grid_backward_input_backward_grid, grid.h:556-683) with
// dL_dy = g2. add(entry * 2 + feature, addend) performs the accumulation.
template <class Add>
static inline void scatter_level(const orc_ctx_s* c, uint32_t level, const float x[3], const float g1[2], const float g2[2], const float dn[3], Add&& add) {
	const uint32_t hashmap_size = c->offsets[level + 1] - c->offsets[level];
	const float scale = c->scale[level];
	const uint32_t res = c->resolution[level];
	float pos[3]; uint32_t pg[3];
	for (int d = 0; d < 3; ++d) pos_fract(x[d], &pos[d], &pg[d], scale);
	for (uint32_t idx = 0; idx < 8; ++idx) {
		float weight = 1;
		uint32_t pl[3];
		for (uint32_t d = 0; d < 3; ++d) {
			if ((idx & (1u << d)) == 0) { weight *= 1 - pos[d]; pl[d] = pg[d]; }
			else { weight *= pos[d]; pl[d] = pg[d] + 1; }
		}
		const uint32_t e = grid_entry(hashmap_size, res, pl);
		for (int f = 0; f < 2; ++f) add(e * 2 + f, rh(g1[f] * weight)); // grid.h:415-416
	}
	for (uint32_t gd = 0; gd < 3; ++gd) {
		const float grad_in = scale * dn[gd] * 1.0f; // grid.h:656
		for (uint32_t idx = 0; idx < 4; ++idx) {
			float weight = grad_in;
			uint32_t pl[3];
			for (uint32_t ngd = 0; ngd < 2; ++ngd) {
				const uint32_t d = ngd >= gd ? (ngd + 1) : ngd;
				if ((idx & (1u << ngd)) == 0) { weight *= 1 - pos[d]; pl[d] = pg[d]; }
				else { weight *= pos[d]; pl[d] = pg[d] + 1; }
			}
			pl[gd] = pg[gd];
			const uint32_t el = grid_entry(hashmap_size, res, pl);
			for (int f = 0; f < 2; ++f) add(el * 2 + f, rh(g2[f] * -weight));
			pl[gd] = pg[gd] + 1;
			const uint32_t er = grid_entry(hashmap_size, res, pl);
			for (int f = 0; f < 2; ++f) add(er * 2 + f, rh(g2[f] * weight));
		}
	}
}

// NerfNetwork::backward_impl (nerf_network.h:257-452) for one compacted sample. With `ops` the accumulations (weight-gradient
// outer products, grid scatter) are left to the caller's second pass and only their operands are recorded.
void backward_sample(const orc_ctx_s* c, const NetParams& np, const FwdCtx& k, const half_t dout[16], uint32_t batch_size, float* grid_grad, MlpGrads& mg, SampleOps* ops = nullptr, int64_t* grid_fixed = nullptr) {
	float* dW_sdf0 = mg.g1.data();
	float* dW_sdf1 = dW_sdf0 + 64 * 32;
	float* dW_rgb0 = mg.g1.data() + RNB_N_SDF_MLP_PARAMS;
	float* dW_rgb1 = dW_rgb0 + 64 * 48;
	float* dW_rgb2 = dW_rgb1 + 64 * 64;
	const bool a16 = np.acc16;

	// dL_drgb = rows 0..2 of dL_doutput (extract_rgb, common_operation.cuh:1010-1025)
	half_t dr[16];
	for (int j = 0; j < 16; ++j) dr[j] = 0;
	for (int j = 0; j < 3; ++j) dr[j] = dout[j];
	// color MLP backward (fully_fused_mlp.cu:914-1031)
	half_t dh2[64], dh1[64], dcin[48];
	if (!ops) outer_acc(dW_rgb2, 16, 64, dr, k.h2);
	matvec_t(np.rgb_w2, 16, 64, dr, dh2, k.h2, a16);
	if (!ops) outer_acc(dW_rgb1, 64, 64, dh2, k.h1);
	matvec_t(np.rgb_w1, 64, 64, dh2, dh1, k.h1, a16);
	if (!ops) outer_acc(dW_rgb0, 64, 48, dh1, k.c_in);
	matvec_t(np.rgb_w0, 64, 48, dh1, dcin, nullptr, a16);
	// dL/d(sdf mlp output) = dL_drgb_network_input[0:16], [0] += dL_doutput[3] (add_density_gradient, common_operation.cuh:1027-1039)
	half_t dso[16];
	for (int j = 0; j < 16; ++j) dso[j] = dcin[j];
	dso[0] = hadd(dso[0], dout[3]);
	// SDF MLP backward (nerf_network.h:298)
	half_t dz[64], dsin[32];
	if (!ops) outer_acc(dW_sdf1, 16, 64, dso, k.z1);
	matvec_t(np.sdf_w1, 16, 64, dso, dz, k.z1, a16);
	if (!ops) outer_acc(dW_sdf0, 64, 32, dz, k.sdf_in);
	matvec_t(np.sdf_w0, 64, 32, dz, dsin, nullptr, a16);

	// variance gradient: sum of dL_doutput row 7 (nerf_network.h:327-340)
	mg.var += (double)h2f(dout[7]);

	// dL/d(∇sdf) (nerf_network.h:343-373)
	float dn[3];
	for (int d = 0; d < 3; ++d) {
		float v = 0.f;
		v = h2f(dcin[35 + d]);                                   // fill_positions_view<float,T>
		v += (float)(h2f(dout[4 + d])) / (float)batch_size;     // add_positions_view_ekloss
		v += h2f(dout[8 + d]);                                  // add_positions_view
		dn[d] = v;
	}

	// pos_encoding_dy = dL/d(dL_dy) (kernel_grid_backward_input_backward_dLdoutput, grid.h:858-883)
	half_t ddin[32];
	for (int j = 0; j < 32; ++j) ddin[j] = 0;
	for (int j = 0; j < 28; ++j) {
		float r = 0.f;
		for (int d = 0; d < 3; ++d) r += k.dy_dx[j][d] * dn[d];
		ddin[3 + j] = f2h(r);
	}
	for (int d = 0; d < 3; ++d) ddin[d] = f2h(dn[d]); // nerf_network.h:430-433

	// hash-grid scatter: first order with dsin[3:31], second order with dL_dy = dsdf_din[3:31]
	const uint32_t L = c->cfg.n_levels;
	if (!ops) {
		for (uint32_t level = 0; level < L; ++level) {
			if (level > c->valid_level) continue;
			float* gg = grid_grad + (uint64_t)c->offsets[level] * 2;
			const float g1[2] = {h2f(dsin[3 + level * 2]), h2f(dsin[3 + level * 2 + 1])};
			const float g2[2] = {h2f(k.dsdf_din[3 + level * 2]), h2f(k.dsdf_din[3 + level * 2 + 1])};
			if (grid_fixed) {
				int64_t* gf = grid_fixed + (uint64_t)c->offsets[level] * 2;
				scatter_level(c, level, k.x, g1, g2, dn, [&](uint32_t q, float v) { atomic_add_fixed(&gf[q], v); });
			} else scatter_level(c, level, k.x, g1, g2, dn, [&](uint32_t q, float v) { atomic_add(&gg[q], v); });
		}
	}

	// FullyFusedMLP::backward_backward_input_impl (fully_fused_mlp.cu:1037-1142), ReLU:
	//   front = (W0 · ddin) ⊙ relu'(z1);  back = (W1ᵀ e0) ⊙ relu'(z1) = dz1
	//   dW0 += back ⊗ ddin;  dW1 += e0 ⊗ front
	half_t front[64];
	for (int o = 0; o < 64; ++o) {
		float acc = dot_h(np.sdf_w0 + (size_t)o * 32, 1, ddin, 1, 32, a16);
		if (!(h2f(k.z1[o]) > 0.f)) acc = 0.f;
		front[o] = f2h(acc);
	}
	if (!ops) {
		float* d2W0 = mg.g2.data();
		float* d2W1 = d2W0 + 64 * 32;
		outer_acc(d2W0, 64, 32, k.dz1, ddin);
		for (int j = 0; j < 64; ++j) d2W1[j] += 1.0f * h2f(front[j]);
		return;
	}
	SampleOps& o = *ops;
	std::memcpy(o.dr, dr, sizeof(dr)); std::memcpy(o.h2, k.h2, sizeof(o.h2)); std::memcpy(o.dh2, dh2, sizeof(dh2)); std::memcpy(o.h1, k.h1, sizeof(o.h1));
	std::memcpy(o.dh1, dh1, sizeof(dh1)); std::memcpy(o.c_in, k.c_in, sizeof(o.c_in)); std::memcpy(o.dso, dso, sizeof(dso)); std::memcpy(o.z1, k.z1, sizeof(o.z1));
	std::memcpy(o.dz, dz, sizeof(dz)); std::memcpy(o.sdf_in, k.sdf_in, sizeof(o.sdf_in)); std::memcpy(o.dz1, k.dz1, sizeof(o.dz1)); std::memcpy(o.ddin, ddin, sizeof(ddin));
	std::memcpy(o.front, front, sizeof(front)); std::memcpy(o.dsin, dsin, sizeof(dsin)); std::memcpy(o.dsdf_din, k.dsdf_din, sizeof(o.dsdf_din));
	for (int d = 0; d < 3; ++d) { o.x[d] = k.x[d]; o.dn[d] = dn[d]; }
}

// ======================================================================
// Occupancy grid (src/testbed_nerf.cu:439-475, 569-740, 3424-3517)
// ======================================================================

static inline int mip_from_pos(const Vec3& pos, uint32_t max_cascade = N_CASCADES - 1) { // testbed_nerf.cu:569-574
	int exponent;
	float maxval = std::max(std::max(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	frexpf(maxval, &exponent);
	return std::min((int)max_cascade, std::max(0, exponent + 1));
}
static inline int mip_from_dt(float dt, const Vec3& pos, uint32_t max_cascade = N_CASCADES - 1) { // testbed_nerf.cu:576-583
	int mip = mip_from_pos(pos, max_cascade);
	dt *= 2 * GRIDSIZE;
	if (dt < 1.f) return mip;
	int exponent;
	frexpf(dt, &exponent);
	return std::min((int)max_cascade, std::max(exponent, mip));
}
static inline uint32_t cascaded_grid_idx_at(Vec3 pos, uint32_t mip) { // testbed_nerf.cu:439-459
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos = pos - v3(0.5f, 0.5f, 0.5f);
	pos = mip_scale * pos;
	pos = pos + v3(0.5f, 0.5f, 0.5f);
	int ix = (int)(pos.x * GRIDSIZE), iy = (int)(pos.y * GRIDSIZE), iz = (int)(pos.z * GRIDSIZE);
	auto cl = [](int v) { return (uint32_t)std::min(std::max(v, 0), (int)GRIDSIZE - 1); };
	return morton3D(cl(ix), cl(iy), cl(iz));
}
static inline bool density_grid_occupied_at(const Vec3& pos, const uint8_t* bitfield, uint32_t mip) { // testbed_nerf.cu:461-465
	uint32_t idx = cascaded_grid_idx_at(pos, mip);
	return bitfield[idx / 8 + (GRID_CELLS * mip) / 8] & (1 << (idx % 8));
}
static inline float calc_dt(float t, float cone_angle) { // testbed_nerf.cu:153-155
	return fminf(fmaxf(t * cone_angle, MIN_CONE_STEPSIZE), MAX_CONE_STEPSIZE);
}
static inline float warp_dt(float dt) { // testbed_nerf.cu:429-432
	float max_stepsize = MIN_CONE_STEPSIZE * (1 << (N_CASCADES - 1));
	return (dt - MIN_CONE_STEPSIZE) / (max_stepsize - MIN_CONE_STEPSIZE);
}
static inline float unwarp_dt(float dt) { // testbed_nerf.cu:434-437
	float max_stepsize = MIN_CONE_STEPSIZE * (1 << (N_CASCADES - 1));
	return dt * (max_stepsize - MIN_CONE_STEPSIZE) + MIN_CONE_STEPSIZE;
}
static inline float sign1(float x) { return copysignf(1.0f, x); } // common.h:194-196
static inline float distance_to_next_voxel(const Vec3& pos, const Vec3& dir, const Vec3& idir, uint32_t res) { // testbed_nerf.cu:301-309
	Vec3 p = (float)res * pos;
	float tx = (floorf(p.x + 0.5f + 0.5f * sign1(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * sign1(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * sign1(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t / res, 0.0f);
}
static inline float advance_to_next_voxel(float t, float cone_angle, const Vec3& pos, const Vec3& dir, const Vec3& idir, uint32_t res) { // testbed_nerf.cu:311-323
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	do { t += calc_dt(t, cone_angle); } while (t < t_target);
	return t;
}
static inline Vec3 warp_direction(const Vec3& d) { return {(d.x + 1.0f) * 0.5f, (d.y + 1.0f) * 0.5f, (d.z + 1.0f) * 0.5f}; }   // testbed_nerf.cu:413-415
static inline Vec3 unwarp_direction(const Vec3& d) { return {d.x * 2.0f - 1.0f, d.y * 2.0f - 1.0f, d.z * 2.0f - 1.0f}; }       // testbed_nerf.cu:417-419
static inline Vec3 warp_position(const orc_ctx_s* c, const Vec3& p) { // bounding_box.cuh:86-88
	float diag = c->aabb_max - c->aabb_min;
	return {(p.x - c->aabb_min) / diag, (p.y - c->aabb_min) / diag, (p.z - c->aabb_min) / diag};
}
static inline Vec3 unwarp_position(const orc_ctx_s* c, const Vec3& p) { // testbed_nerf.cu:395-400
	float diag = c->aabb_max - c->aabb_min;
	return {c->aabb_min + p.x * diag, c->aabb_min + p.y * diag, c->aabb_min + p.z * diag};
}
static inline bool aabb_contains(const orc_ctx_s* c, const Vec3& p) { // bounding_box.cuh:208-213
	return p.x >= c->aabb_min && p.x <= c->aabb_max && p.y >= c->aabb_min && p.y <= c->aabb_max && p.z >= c->aabb_min && p.z <= c->aabb_max;
}
static inline void ray_intersect(const orc_ctx_s* c, const Vec3& pos, const Vec3& dir, float* tmin_o, float* tmax_o) { // bounding_box.cuh:163-206
	const float mn = c->aabb_min, mx = c->aabb_max;
	const float FMAX = std::numeric_limits<float>::max();
	float tmin = (mn - pos.x) / dir.x, tmax = (mx - pos.x) / dir.x;
	if (tmin > tmax) std::swap(tmin, tmax);
	float tymin = (mn - pos.y) / dir.y, tymax = (mx - pos.y) / dir.y;
	if (tymin > tymax) std::swap(tymin, tymax);
	if (tmin > tymax || tymin > tmax) { *tmin_o = FMAX; *tmax_o = FMAX; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (mn - pos.z) / dir.z, tzmax = (mx - pos.z) / dir.z;
	if (tzmin > tzmax) std::swap(tzmin, tzmax);
	if (tmin > tzmax || tzmin > tmax) { *tmin_o = FMAX; *tmax_o = FMAX; return; }
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	*tmin_o = tmin; *tmax_o = tmax;
}

// generate_grid_samples_nerf_nonuniform (testbed_nerf.cu:585-614)
void generate_grid_samples(orc_ctx_s* c, uint32_t n_elements, Pcg32 rng0, uint32_t step, float thresh, float* pos_out, uint32_t* idx_out) {
	const uint32_t n_cascades = c->max_cascade + 1;
	const float* grid_in = c->density_grid.data();
#pragma omp parallel for schedule(static)
	for (int64_t ii = 0; ii < (int64_t)n_elements; ++ii) {
		const uint32_t i = (uint32_t)ii;
		Pcg32 rng = rng0;
		rng.advance((int64_t)i * 4);
		uint32_t level = (uint32_t)(rng.next_float() * n_cascades) % n_cascades;
		uint32_t idx = 0;
		for (uint32_t j = 0; j < 10; ++j) {
			idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % GRID_CELLS;
			idx += level * GRID_CELLS;
			if (grid_in[idx] > thresh) break;
		}
		uint32_t pos_idx = idx % GRID_CELLS;
		uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
		float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
		float sc = scalbnf(1.0f, (int)level);
		Vec3 pos = {(((float)x + rx) / GRIDSIZE - 0.5f) * sc + 0.5f, (((float)y + ry) / GRIDSIZE - 0.5f) * sc + 0.5f, (((float)z + rz) / GRIDSIZE - 0.5f) * sc + 0.5f};
		Vec3 w = warp_position(c, pos);
		pos_out[(size_t)i * 3 + 0] = w.x; pos_out[(size_t)i * 3 + 1] = w.y; pos_out[(size_t)i * 3 + 2] = w.z;
		idx_out[i] = idx;
	}
}

// update_density_grid_mean_and_bitfield (testbed_nerf.cu:3497-3517, 693-740)
void update_bitfield(orc_ctx_s* c) {
	double sum = 0.0; // deviation D5
	const float* grid = c->density_grid.data();
#pragma omp parallel for reduction(+ : sum) schedule(static)
	for (int64_t i = 0; i < (int64_t)GRID_CELLS; ++i) sum += (double)(fmaxf(grid[i], 0.f) / (float)GRID_CELLS);
	c->density_mean = (float)sum;
	const uint32_t n_bytes_per_mip = GRID_CELLS / 8;
	const uint32_t n_nonzero = n_bytes_per_mip * (c->max_cascade + 1);
	const float thresh = std::min(MIN_OPTICAL_THICKNESS, c->density_mean);
	uint8_t* bf = c->bitfield.data();
	for (uint32_t i = 0; i < n_bytes_per_mip * N_CASCADES; ++i) { // grid_to_bitfield
		if (i >= n_nonzero) { bf[i] = 0; continue; }
		uint8_t bits = 0;
		for (uint8_t j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? ((uint8_t)1 << j) : 0;
		bf[i] = bits;
	}
	for (uint32_t level = 1; level < N_CASCADES; ++level) { // bitfield_max_pool
		const uint8_t* prev = bf + (size_t)n_bytes_per_mip * (level - 1);
		uint8_t* next = bf + (size_t)n_bytes_per_mip * level;
		for (uint32_t i = 0; i < GRID_CELLS / 64; ++i) {
			uint8_t bits = 0;
			for (uint8_t j = 0; j < 8; ++j) bits |= prev[(size_t)i * 8 + j] > 0 ? ((uint8_t)1 << j) : 0;
			uint32_t x = morton3D_invert(i >> 0) + GRIDSIZE / 8;
			uint32_t y = morton3D_invert(i >> 1) + GRIDSIZE / 8;
			uint32_t z = morton3D_invert(i >> 2) + GRIDSIZE / 8;
			next[morton3D(x, y, z)] |= bits;
		}
	}
}

// update_density_grid_nerf (testbed_nerf.cu:3424-3495), first half: samples, network, splat. shard: this rank's 1 / world_size of the samples only
// (rnb_update_density_grid_begin: the caller takes the element-wise max of DENSITY_GRID_TMP over the ranks before the second half).
void update_density_grid_front(orc_ctx_s* c, uint32_t n_uniform, uint32_t n_nonuniform, bool shard) {
	const uint32_t n_elements = GRID_CELLS * (c->max_cascade + 1);
	const uint32_t n_samples = n_uniform + n_nonuniform;
	if (c->training_step == 0) { // testbed_nerf.cu:3446-3452
		c->density_grid_ema_step = 0;
		std::fill(c->density_grid.begin(), c->density_grid.end(), 0.f);
	}
	std::fill(c->density_grid_tmp.begin(), c->density_grid_tmp.end(), 0.f);
	c->grid_sample_pos.resize((size_t)n_samples * 3);
	c->grid_sample_idx.resize(n_samples);
	generate_grid_samples(c, n_uniform, c->density_grid_rng, c->density_grid_ema_step, -0.01f, c->grid_sample_pos.data(), c->grid_sample_idx.data());
	c->density_grid_rng.advance();
	generate_grid_samples(c, n_nonuniform, c->density_grid_rng, c->density_grid_ema_step, MIN_OPTICAL_THICKNESS, c->grid_sample_pos.data() + (size_t)n_uniform * 3, c->grid_sample_idx.data() + n_uniform);
	c->density_grid_rng.advance();

	// density with training weights (testbed_nerf.cu:3486) + splat max (testbed_nerf.cu:616-635)
	NetParams np = net_params(c, false);
	uint32_t* tmp_bits = reinterpret_cast<uint32_t*>(c->density_grid_tmp.data());
	std::vector<float> dens(n_samples);
	const uint64_t W = shard ? c->cfg.world_size : 1u, r = shard ? c->cfg.rank : 0u;
	const int64_t lo = (int64_t)((uint64_t)n_samples * r / W), hi = (int64_t)((uint64_t)n_samples * (r + 1) / W);
#pragma omp parallel for schedule(static)
	for (int64_t i = lo; i < hi; ++i) {
		half_t sdf = sdf_sample(c, np, &c->grid_sample_pos[(size_t)i * 3]);
		dens[i] = h2f(sdf_to_density(sdf, np.variance));
	}
	for (int64_t i = lo; i < hi; ++i) {
		uint32_t b; std::memcpy(&b, &dens[i], 4);
		uint32_t& dst = tmp_bits[c->grid_sample_idx[i]];
		if (b > dst) dst = b; // atomicMax on float bits
	}
}

// second half: EMA, mean, bitfield
void update_density_grid_back(orc_ctx_s* c) {
	const uint32_t n_elements = GRID_CELLS * (c->max_cascade + 1);
	// ema_grid_samples_nerf (testbed_nerf.cu:655-685)
	const float decay = c->cfg.density_grid_decay;
	for (uint32_t i = 0; i < n_elements; ++i) {
		float importance = c->density_grid_tmp[i];
		float prev = c->density_grid[i];
		c->density_grid[i] = (prev < 0.f) ? prev : fmaxf(prev * decay, importance);
	}
	++c->density_grid_ema_step;
	update_bitfield(c);
}

// training_prep_nerf (testbed_nerf.cu:4125-4138)
void training_prep_front(orc_ctx_s* c, bool shard) {
	const uint32_t n_cascades = c->max_cascade + 1;
	if (c->training_step < 256) update_density_grid_front(c, GRID_CELLS * n_cascades, 0, shard);
	else update_density_grid_front(c, GRID_CELLS / 4 * n_cascades, GRID_CELLS / 4 * n_cascades, shard);
}
int training_prep(orc_ctx_s* c) {
	const bool shard = c->cfg.world_size > 1 && c->grid_exchange != nullptr;
	training_prep_front(c, shard);
	if (shard && c->grid_exchange(c->grid_exchange_user, c->density_grid_tmp.data(), (uint64_t)GRID_CELLS * (c->max_cascade + 1), nullptr) != 0)
		return fail(RNB_ERR_INVALID, "the occupancy grid exchange (rnb_set_grid_exchange) failed");
	update_density_grid_back(c);
	return RNB_OK;
}

// ======================================================================
// Dataset access (include/neural-graphics-primitives/common_device.cuh:31-61, 621-700)
// ======================================================================

static inline float srgb_to_linear(float srgb) {
	if (srgb <= 0.04045f) return srgb / 12.92f;
	return std::pow((srgb + 0.055f) / 1.055f, 2.4f);
}
static inline float linear_to_srgb(float linear) {
	if (linear < 0.0031308f) return 12.92f * linear;
	return 1.055f * std::pow(linear, 0.41666f) - 0.055f;
}
static inline void image_pos(const float xy[2], uint32_t w, uint32_t h, int* px, int* py) { // common_device.cuh:621-623
	int x = (int)(xy[0] * (float)w), y = (int)(xy[1] * (float)h);
	*px = std::max(std::min(x, (int)w - 1), 0);
	*py = std::max(std::min(y, (int)h - 1), 0);
}
static inline void read_rgba(const float xy[2], const rnb_view& m, const uint16_t* pixels, float rgba[4]) { // common_device.cuh:665-700 (Byte = RGBA16 here)
	int px, py; image_pos(xy, m.width, m.height, &px, &py);
	const uint16_t* v = pixels + ((size_t)px + (size_t)py * m.width) * 4;
	uint64_t raw; std::memcpy(&raw, v, 8);
	if (raw == 0x00FF00FFull) { rgba[0] = rgba[1] = rgba[2] = rgba[3] = -1.f; return; }
	float alpha = (float)v[3] * (1.0f / 65535.0f);
	rgba[0] = srgb_to_linear((float)v[0] * (1.0f / 65535.0f)) * alpha;
	rgba[1] = srgb_to_linear((float)v[1] * (1.0f / 65535.0f)) * alpha;
	rgba[2] = srgb_to_linear((float)v[2] * (1.0f / 65535.0f)) * alpha;
	rgba[3] = alpha;
}
// The ray of an image position (testbed_nerf.cu:1279-1305; no lens distortion, no rolling shutter): Eigen's fixed-size 3x3 * 3 product (esum3 per row),
// normalized() divides by the norm.
static inline void camera_ray(const rnb_view& m, const float xy[2], Vec3& o, Vec3& d_unnorm, Vec3& dir) {
	const float* X = m.xform;
	o = v3(X[3], X[7], X[11]);
	Vec3 dcam = {
		(xy[0] - m.principal_point[0]) * (float)m.width / m.focal_length[0],
		(xy[1] - m.principal_point[1]) * (float)m.height / m.focal_length[1],
		1.0f,
	};
	d_unnorm = v3(esum3(X[0] * dcam.x, X[1] * dcam.y, X[2] * dcam.z),
	              esum3(X[4] * dcam.x, X[5] * dcam.y, X[6] * dcam.z),
	              esum3(X[8] * dcam.x, X[9] * dcam.y, X[10] * dcam.z));
	dir = normalized(d_unnorm);
}
// nerf_random_image_pos_training (testbed_nerf.cu:1171-1192), no error-map CDF (default off, testbed.h:663-664)
static inline void random_image_pos(Pcg32& rng, uint32_t w, uint32_t h, bool snap, float xy[2]) {
	xy[0] = rng.next_float(); xy[1] = rng.next_float();
	if (snap) {
		float res[2] = {(float)w, (float)h};
		for (int a = 0; a < 2; ++a) {
			float p = xy[a] * res[a];
			p = std::max(p, 0.0f);
			p = std::min(p, (float)((int)(a == 0 ? w : h) - 1));
			xy[a] = (p + 0.5f) / res[a];
		}
	}
}
// image_idx (testbed_nerf.cu:1194-1214): uint32 arithmetic, wraps.
static inline uint32_t image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_rays_total, uint32_t n_images) {
	return (((base_idx + n_rays_total) * n_images) / n_rays) % n_images;
}

// The data-parallel view of the ray index space: rank r owns global rays
// [r*n_rays, (r+1)*n_rays) of a step of world_size*n_rays rays.
static inline uint32_t global_rays(const orc_ctx_s* c, uint32_t n_rays) { return n_rays * c->cfg.world_size; }
static inline uint32_t global_ray_index(const orc_ctx_s* c, uint32_t i, uint32_t n_rays) { return c->cfg.rank * n_rays + i; }

// ======================================================================
// K6: generate_training_samples_nerf_with_global_movement (testbed_nerf.cu:1216-1387)
// ======================================================================

struct RaySetup {
	bool alive;
	Vec3 o, d_unnorm, dir;
	float startt;
};

RaySetup setup_ray(const orc_ctx_s* c, uint32_t i, uint32_t n_rays, uint32_t n_rays_total) {
	RaySetup r; r.alive = false;
	const uint32_t gi = global_ray_index(c, i, n_rays);
	const uint32_t gn = global_rays(c, n_rays);
	const uint32_t img = image_idx(gi, gn, n_rays_total, (uint32_t)c->views.size());
	const View& view = c->views[img];
	const rnb_view& m = view.meta;
	Pcg32 rng = c->rng;
	rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
	float xy[2];
	random_image_pos(rng, m.width, m.height, c->cfg.snap_to_pixel_centers != 0, xy);
	float rgba[4];
	read_rgba(xy, m, view.normal.data(), rgba);
	if (rgba[0] <= 0.0f && rng.next_float() >= 0.9) return r; // testbed_nerf.cu:1264 (short-circuit draw)
	// max_level_rand_training = false (testbed.h:460): no draw
	float motionblur_time = rng.next_float(); (void)motionblur_time; // testbed_nerf.cu:1270
	camera_ray(m, xy, r.o, r.d_unnorm, r.dir);
	// first_frame_offset = 0; predict_global_movement: identity rotation, zero translation (testbed_nerf.cu:1312-1320)
	float tmin, tmax;
	ray_intersect(c, r.o, r.dir, &tmin, &tmax);
	tmin = fmaxf(tmin, 0.0f);
	float startt = tmin;
	startt += calc_dt(startt, c->cone_angle) * rng.next_float();
	r.startt = startt;
	r.alive = true;
	return r;
}

template <typename F>
uint32_t march(const orc_ctx_s* c, const RaySetup& r, uint32_t max_steps, F&& emit) {
	const Vec3 idir = {1.0f / r.dir.x, 1.0f / r.dir.y, 1.0f / r.dir.z};
	uint32_t j = 0;
	float t = r.startt;
	Vec3 pos;
	const uint8_t* bf = c->bitfield.data();
	while (aabb_contains(c, pos = r.o + t * r.dir) && j < max_steps) {
		float dt = calc_dt(t, c->cone_angle);
		uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
		if (density_grid_occupied_at(pos, bf, mip)) {
			emit(j, pos, dt);
			++j;
			t += dt;
		} else {
			uint32_t res = GRIDSIZE >> mip;
			t = advance_to_next_voxel(t, c->cone_angle, pos, r.dir, idir, res);
		}
	}
	return j;
}

void generate_training_samples(orc_ctx_s* c, uint32_t n_rays, uint32_t n_rays_total, uint32_t max_samples) {
	std::vector<RaySetup> setups(n_rays);
	std::vector<uint32_t> steps(n_rays, 0);
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t i = 0; i < (int64_t)n_rays; ++i) {
		setups[i] = setup_ray(c, (uint32_t)i, n_rays, n_rays_total);
		if (setups[i].alive) steps[i] = march(c, setups[i], RNB_MAX_STEPS, [](uint32_t, const Vec3&, float) {});
	}
	// deviation D3: slots in ray order. base counts every ray with numsteps > 0 (atomicAdd precedes the overflow test).
	uint32_t base = 0, ray_idx = 0, written = 0;
	std::vector<uint32_t> bases(n_rays, 0), slots(n_rays, 0xffffffffu);
	for (uint32_t i = 0; i < n_rays; ++i) {
		if (!setups[i].alive || steps[i] == 0) continue;
		uint32_t b = base;
		base += steps[i];
		if (b + steps[i] > max_samples) continue; // testbed_nerf.cu:1353-1355
		bases[i] = b; slots[i] = ray_idx++;
		written = b + steps[i];
	}
	c->counters[0] = base;
	c->counters[2] = ray_idx;
	c->counters[3] = written;
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t ii = 0; ii < (int64_t)n_rays; ++ii) {
		const uint32_t i = (uint32_t)ii;
		if (slots[i] == 0xffffffffu) continue;
		const RaySetup& r = setups[i];
		const uint32_t s = slots[i];
		c->ray_indices[s] = i;
		float* ro = &c->rays[(size_t)s * 6];
		ro[0] = r.o.x; ro[1] = r.o.y; ro[2] = r.o.z; ro[3] = r.d_unnorm.x; ro[4] = r.d_unnorm.y; ro[5] = r.d_unnorm.z;
		c->numsteps[(size_t)s * 2 + 0] = steps[i];
		c->numsteps[(size_t)s * 2 + 1] = bases[i];
		const Vec3 wd = warp_direction(r.dir);
		float* co = &c->coords[(size_t)bases[i] * 7];
		march(c, r, steps[i], [&](uint32_t j, const Vec3& pos, float dt) {
			Vec3 wp = warp_position(c, pos);
			float* o = co + (size_t)j * 7;
			o[0] = wp.x; o[1] = wp.y; o[2] = wp.z; o[3] = warp_dt(dt); o[4] = wd.x; o[5] = wd.y; o[6] = wd.z;
		});
	}
}

// ======================================================================
// K8: compute_loss_kernel_train_nerf_with_global_movement (testbed_nerf.cu:1396-2097)
// ======================================================================

static inline float logistic(float x) { return 1.0f / (1.0f + expf(-x)); } // common_device.h:52-54
static inline float relu(float v) { return v > 0.0f ? v : 0.0f; }            // activation_function(.., ReLU), testbed_nerf.cu:326-335
// The ray loss (testbed_nerf.cu:280-299, 1389-1394): L2 = sum of squares of d = prediction - target, gradient 2 d; L1 = sum of |d|, gradient copysign(1, d).
static inline float loss_and_gradient(bool l2, const float target[4], const float prediction[4], float grad[4]) {
	float diff[4];
	for (int k = 0; k < 4; ++k) diff[k] = prediction[k] - target[k];
	if (l2) {
		for (int k = 0; k < 4; ++k) grad[k] = 2 * diff[k];
		return diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2] + diff[3] * diff[3];
	}
	for (int k = 0; k < 4; ++k) grad[k] = copysignf(1.0f, diff[k]);
	return fabsf(diff[0]) + fabsf(diff[1]) + fabsf(diff[2]) + fabsf(diff[3]);
}

void build_light_dirs(orc_ctx_s* c) { // testbed_nerf.cu:1537-1554
	auto radians = [](float deg) { return deg * M_PI / 180.0f; };
	float tilt[3] = {(float)radians(0.0f), (float)radians(120.0f), (float)radians(240.0f)};
	float slant[3] = {(float)radians(54.74f), (float)radians(54.74f), (float)radians(54.74f)};
	for (int k = 0; k < 3; ++k) {
		c->light_dirs[0 * 3 + k] = -(sinf(slant[k]) * cosf(tilt[k]));
		c->light_dirs[1 * 3 + k] = -(sinf(slant[k]) * sinf(tilt[k]));
		c->light_dirs[2 * 3 + k] = -cosf(slant[k]);
	}
	if (c->cfg.apply_supernormal) {
		for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) c->light_dirs[r * 3 + k] = (r == k) ? 1.f : 0.f;
	}
}

struct RayLoss {
	// pass-1 results needed by pass 2
	uint32_t n_comp;
	float rgb_ray[4];
	float weight_sum_raw;
	float rgbtarget[4];
	float light[3];
	Vec3 dir;
	float mask_certainty, mask_gt;
	uint32_t img;
};

static inline void albedo_from_output(const orc_ctx_s* c, const half_t* o, float albedo[4]) { // testbed_nerf.cu:1614-1639
	if (c->cfg.apply_no_albedo) { albedo[0] = albedo[1] = albedo[2] = 1.f; albedo[3] = 0.f; return; }
	float a[3];
	for (int k = 0; k < 3; ++k) a[k] = logistic(h2f(o[k])); // rgb_activation = Logistic (testbed_nerf.cu:3121)
	albedo[0] = a[0]; albedo[1] = a[1]; albedo[2] = a[2];
	if (c->cfg.apply_rgbplus) {
		if (c->cfg.apply_L2) albedo[3] = sqrtf(std::max(0.0f, 3 - a[0] * a[0] - a[1] * a[1] - a[2] * a[2]));
		else albedo[3] = 3 - fabsf(a[0]) - fabsf(a[1]) - fabsf(a[2]);
	} else albedo[3] = 0.f;
}

// The loss kernel's per-ray targets from the two texels (testbed_nerf.cu:1500-1592): target normal, target albedo, the step's light in the camera and the world frame, the shading
// target and rgbtarget; every Eigen reduction as Eigen evaluates it (esum3). tests/golden/float_fixtures.json runs the kernel's own statements.
static inline void ray_targets(const rnb_config& F, const float* X, const float tex_normal[4], const float tex_albedo[4], const float* light_dirs, const int random_light,
                               float rgbtarget[4], float light[3]) {
	// exposure = 0 -> exposure_scale = exp(0) = 1 (testbed_nerf.cu:1503)
	const float exposure_scale = expf(0.6931471805599453f * 0.f);
	float nv[3];
	for (int k = 0; k < 3; ++k) nv[k] = linear_to_srgb(exposure_scale * tex_normal[k]) * 2.0f - 1.0f; // testbed_nerf.cu:1507
	nv[1] *= -1; nv[2] *= -1;
	{ float n = sqrtf(esum3(nv[0] * nv[0], nv[1] * nv[1], nv[2] * nv[2])); for (int k = 0; k < 3; ++k) nv[k] /= n; } // .matrix().norm()
	float albedo_value[4];
	if (F.apply_no_albedo) { albedo_value[0] = albedo_value[1] = albedo_value[2] = 1.f; albedo_value[3] = 0.f; }
	else {
		float a[3];
		for (int k = 0; k < 3; ++k) a[k] = linear_to_srgb(exposure_scale * tex_albedo[k]);
		albedo_value[0] = a[0]; albedo_value[1] = a[1]; albedo_value[2] = a[2];
		if (F.apply_rgbplus) {
			if (F.apply_L2) albedo_value[3] = sqrtf(std::max(0.0f, 3 - a[0] * a[0] - a[1] * a[1] - a[2] * a[2]));
			else albedo_value[3] = 3 - fabsf(a[0]) - fabsf(a[1]) - fabsf(a[2]);
		} else albedo_value[3] = 0.f;
	}
	// light triplet (testbed_nerf.cu:1537-1583)
	float Ld[9];
	for (int k = 0; k < 9; ++k) Ld[k] = light_dirs[k];
	if (F.apply_light_opti) { // testbed_nerf.cu:1563-1581
		float k3[3] = {-nv[1], nv[0], 0.f};
		float kn = sqrtf(esum3(k3[0] * k3[0], k3[1] * k3[1], k3[2] * k3[2]));
		for (int a = 0; a < 3; ++a) k3[a] /= kn;
		float cos_theta = nv[2];
		float sin_theta = std::sqrt(1 - cos_theta * cos_theta);
		float K[9] = {0, -k3[2], k3[1], k3[2], 0, -k3[0], -k3[1], k3[0], 0};
		float Rm[9];
		for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q)
			Rm[r * 3 + q] = cos_theta * (r == q ? 1.f : 0.f) + sin_theta * K[r * 3 + q] + (1 - cos_theta) * (k3[r] * k3[q]);
		float out[9];
		for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q)
			out[r * 3 + q] = esum3((-Rm[r * 3 + 0]) * Ld[0 * 3 + q], (-Rm[r * 3 + 1]) * Ld[1 * 3 + q], (-Rm[r * 3 + 2]) * Ld[2 * 3 + q]); // -R * light_directions
		for (int k = 0; k < 9; ++k) Ld[k] = out[k];
	}
	float light_cam[3] = {Ld[0 * 3 + random_light], Ld[1 * 3 + random_light], Ld[2 * 3 + random_light]};
	for (int r = 0; r < 3; ++r) light[r] = esum3(X[r * 4 + 0] * light_cam[0], X[r * 4 + 1] * light_cam[1], X[r * 4 + 2] * light_cam[2]); // Rt * light_cam
	float shading_target = esum3(nv[0] * light_cam[0], nv[1] * light_cam[1], nv[2] * light_cam[2]);                                         // .dot(light_cam)
	if (F.apply_relu) shading_target = shading_target > 0.f ? shading_target : 0.f;
	for (int k = 0; k < 4; ++k) rgbtarget[k] = albedo_value[k] * shading_target;
}
struct AlphaTerms {
	float inv_s, sdf_value, true_cos, iter_cos, est_next, p_div_c, alpha, dt;
	float g[3];
};
static inline AlphaTerms alpha_terms(const half_t* o, float dt, const Vec3& dir, float cos_anneal_ratio) { // testbed_nerf.cu:1652-1677
	AlphaTerms a;
	a.dt = dt;
	a.inv_s = expf(h2f(hmul(f2h(10.f), o[7])));
	a.sdf_value = h2f(o[3]);
	a.g[0] = h2f(o[4]); a.g[1] = h2f(o[5]); a.g[2] = h2f(o[6]);
	a.true_cos = (dir.x * a.g[0] + dir.y * a.g[1] + dir.z * a.g[2]);
	a.iter_cos = (float)-(relu((float)(-a.true_cos * 0.5 + 0.5)) * (1.0 - cos_anneal_ratio) + relu(-a.true_cos) * cos_anneal_ratio);
	a.est_next = (float)(a.sdf_value + a.iter_cos * dt * 0.5);
	float est_prev = (float)(a.sdf_value - a.iter_cos * dt * 0.5);
	float next_cdf = logistic(a.est_next * a.inv_s);
	float prev_cdf = logistic(est_prev * a.inv_s);
	float p = prev_cdf - next_cdf;
	float cc = prev_cdf;
	a.p_div_c = (p + 1e-5f) / (cc + 1e-5f);
	a.alpha = std::min(std::max(a.p_div_c, 0.0f), 1.0f);
	return a;
}

void loss_pass1(const orc_ctx_s* c, uint32_t i, uint32_t n_rays, uint32_t n_rays_total, RayLoss& R) {
	const uint32_t numsteps = c->numsteps[(size_t)i * 2 + 0];
	const uint32_t base = c->numsteps[(size_t)i * 2 + 1];
	const float* coords_in = &c->coords[(size_t)base * 7];
	const half_t* net = &c->mlp_out[(size_t)base * 16];
	const uint32_t ray_idx = c->ray_indices[i];
	const uint32_t gi = global_ray_index(c, ray_idx, n_rays);
	const uint32_t gn = global_rays(c, n_rays);
	Pcg32 rng = c->rng;
	rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
	const uint32_t img = image_idx(gi, gn, n_rays_total, (uint32_t)c->views.size());
	R.img = img;
	const View& view = c->views[img];
	const rnb_view& m = view.meta;
	Vec3 ray_d = v3(c->rays[(size_t)i * 6 + 3], c->rays[(size_t)i * 6 + 4], c->rays[(size_t)i * 6 + 5]);
	Vec3 dir = normalized(ray_d);
	float xy[2];
	random_image_pos(rng, m.width, m.height, c->cfg.snap_to_pixel_centers != 0, xy);
	const float* X = m.xform;
	float tex_albedo[4], tex_normal[4];
	read_rgba(xy, m, view.albedo.data(), tex_albedo);
	read_rgba(xy, m, view.normal.data(), tex_normal);
	// deviation D4: deterministic light pick = draw #7 of the ray's PCG32 stream
	Pcg32 lrng = c->rng;
	lrng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY + 7);
	const int random_light = (int)(lrng.next_uint() % 3u);
	ray_targets(c->cfg, X, tex_normal, tex_albedo, c->light_dirs, random_light, R.rgbtarget, R.light);

	// pass 1 (testbed_nerf.cu:1608-1697)
	float T = 1.f;
	const float EPSILON = 1e-4f;
	float rgb_ray[4] = {0, 0, 0, 0};
	float weight_sum = 0.f;
	uint32_t n = 0;
	for (; n < numsteps; ++n) {
		if (T < EPSILON) break;
		const half_t* o = net + (size_t)n * 16;
		float albedo[4];
		albedo_from_output(c, o, albedo);
		float dt = unwarp_dt(coords_in[(size_t)n * 7 + 3]);
		if (n == 0) { // BENT_DIR (testbed_nerf.cu:1645-1650)
			Vec3 dv = v3(h2f(o[8]), h2f(o[9]), h2f(o[10]));
			dir = normalized(unwarp_direction(dv));
		}
		AlphaTerms a = alpha_terms(o, dt, dir, 1.0f);
		const float weight = a.alpha * T;
		float shading = esum3(a.g[0] * R.light[0], a.g[1] * R.light[1], a.g[2] * R.light[2]); // normal.dot(light)
		if (c->cfg.apply_relu) shading = shading > 0.f ? shading : 0.f;
		for (int k = 0; k < 4; ++k) rgb_ray[k] += weight * albedo[k] * shading;
		weight_sum += weight;
		T *= (1.f - a.alpha);
	}
	R.n_comp = n;
	for (int k = 0; k < 4; ++k) R.rgb_ray[k] = rgb_ray[k];
	R.weight_sum_raw = weight_sum;
	R.dir = dir;
	R.mask_certainty = (float)(tex_albedo[3] > 0.99);
	R.mask_gt = (float)(tex_normal[3] > 0.99);
}

// The ray's loss terms between the loss kernel's two loops (testbed_nerf.cu:1735-1800): loss and gradient (halved for rgb+, gated by the albedo's alpha), the clamped weight sum, the
// gradient of the mask term, the two loss rows. Returns the ray's loss. tests/golden/float_fixtures.json runs the kernel's own lines.
static inline float pass2_ray_terms(const rnb_config& F, const float rgbtarget[4], const float rgb_ray[4], const float mask_certainty, const float mask_gt, const float weight_sum_raw,
                                    const float gn, float grad[4], float* weight_sum_out, float* gradient_weight_sum_out, float* loss_row, float* mask_row) {
	float loss = loss_and_gradient(F.apply_L2 != 0, rgbtarget, rgb_ray, grad);
	if (F.apply_rgbplus) { loss /= 2; for (int k = 0; k < 4; ++k) grad[k] /= 2; }
	loss *= mask_certainty;
	for (int k = 0; k < 4; ++k) grad[k] *= mask_certainty;
	float weight_sum = weight_sum_raw;
	float gradient_weight_sum;
	if (weight_sum >= 1.0 - 1e-4) { weight_sum = (float)(1.0 - 1e-4); gradient_weight_sum = 0.0f; }
	else if (weight_sum <= 1e-4) { weight_sum = 1e-4; gradient_weight_sum = 0.0f; }
	else {
		float sig = 1.0f / (1.0f + expf(-weight_sum));
		if (F.apply_bce) gradient_weight_sum = ((1 - mask_gt) / (1 - weight_sum) - mask_gt / weight_sum) * F.mask_loss_weight;
		else gradient_weight_sum = (sig - mask_gt) * F.mask_loss_weight;
	}
	*loss_row = loss / gn;
	{
		float sig = 1.0f / (1.0f + expf(-weight_sum));
		if (F.apply_bce) *mask_row = -(mask_gt * logf(weight_sum) + (1 - mask_gt) * logf(1 - weight_sum));
		else *mask_row = -(mask_gt * logf(sig) + (1 - mask_gt) * logf(1 - sig));
	}
	*weight_sum_out = weight_sum; *gradient_weight_sum_out = gradient_weight_sum;
	return loss;
}
// dL/d(network output) of one compacted sample (testbed_nerf.cu:1920-2085) from the ray's terms, the sample's network output `o`, its step `dt`, and the running values of the
// compositing recurrence right AFTER the sample: its weight, the transmittance T, the weight sum weight_sum2, the colour sums rgb_ray2. Returns the sample's gradient norm
// (the Eikonal term is formed from it). tests/golden/float_fixtures.json runs the kernel's own lines for one sample.
static inline float pass2_sample(const rnb_config& F, const float grad[4], const float rgb_ray[4], const float weight_sum, const float gradient_weight_sum, const float light[3], const Vec3& dir,
                                 const float loss_scale, const half_t* o, const float dt, const float albedo[4], const AlphaTerms& a, const float shading, const float weight, const float T,
                                 const float weight_sum2, const float rgb_ray2[4], half_t dl[16], float* inter = nullptr) {
	const float alpha = a.alpha;
	float suffix[4];
	for (int k = 0; k < 4; ++k) suffix[k] = rgb_ray[k] - rgb_ray2[k];
	// dloss_dn = weight * (light * albedo^T) * lg.gradient (testbed_nerf.cu:1924-1926): Eigen forms every coefficient of the scaled 3x4 matrix, then sums a row's four terms as
	// (x0 + x1) + (x2 + x3)
	float dloss_dn[3];
	for (int d = 0; d < 3; ++d)
		dloss_dn[d] = esum4((weight * (light[d] * albedo[0])) * grad[0], (weight * (light[d] * albedo[1])) * grad[1], (weight * (light[d] * albedo[2])) * grad[2], (weight * (light[d] * albedo[3])) * grad[3]);
	// jac_rgb (testbed_nerf.cu:1928-1949)
	float J3[3] = {0, 0, 0};
	if (F.apply_rgbplus) {
		if (F.apply_L2) for (int d = 0; d < 3; ++d) J3[d] = (float)(-2 * albedo[d] / (albedo[3] + 1e-5));
		else for (int d = 0; d < 3; ++d) J3[d] = -sign1(albedo[d]);
	}
	float drgb[3]; // (weight * shading) * jac_rgb * lg.gradient with jac_rgb = [I | J3] (testbed_nerf.cu:1928-1952): the scaled matrix's zero entries are signed zeros and stay in the sums
	const float ws = weight * shading;
	for (int d = 0; d < 3; ++d)
		drgb[d] = esum4((ws * (d == 0 ? 1.0f : 0.0f)) * grad[0], (ws * (d == 1 ? 1.0f : 0.0f)) * grad[1], (ws * (d == 2 ? 1.0f : 0.0f)) * grad[2], (ws * J3[d]) * grad[3]);
	for (int q = 0; q < 16; ++q) dl[q] = 0;
	const float opti_rgb = F.apply_no_albedo ? 0.0f : 1.0f;
	for (int d = 0; d < 3; ++d) {
		float sg = logistic(h2f(o[d]));
		dl[d] = f2h(opti_rgb * loss_scale * (drgb[d] * (sg * (1 - sg))));
	}
	const float sum_weight_suffix = weight_sum - weight_sum2;
	const float dot_term = esum4(grad[0] * (T * albedo[0] * shading - suffix[0]), grad[1] * (T * albedo[1] * shading - suffix[1]), grad[2] * (T * albedo[2] * shading - suffix[2]),
	                             grad[3] * (T * albedo[3] * shading - suffix[3])); // lg.gradient.matrix().dot(...)
	float dloss_dalpha = (float)((dot_term + (gradient_weight_sum * (T - sum_weight_suffix))) / (1.0f - alpha + 1e-5));
	float dalpha_dE = 0.f, dE_dsdf = 0.f, dE_dinvs = 0.f, dalpha_dEp = 0.f, dEp_dinvs = 0.f, dEp_ditc = 0.f, dE_ditc = 0.f;
	if (!(a.p_div_c <= 0.0f || a.p_div_c >= 1.0f)) { // testbed_nerf.cu:1982-2014
		float plus_sigmoid_x = a.inv_s * a.iter_cos * dt;
		float plus_e = expf(plus_sigmoid_x);
		float e_minus = expf(-a.est_next * a.inv_s);
		dE_dsdf = -a.inv_s * e_minus;
		dE_dinvs = -a.est_next * e_minus;
		float aa = 1 + e_minus;
		float bb = 1 + plus_e * e_minus;
		float cc = (float)(1e-5 + 1 / (1 + plus_e * e_minus));
		float delta = aa * (bb * bb) * (cc * cc);
		dalpha_dE = -(plus_e / (delta)-1 / (aa * aa * cc));
		dalpha_dEp = -e_minus / (delta);
		dEp_dinvs = plus_e * a.iter_cos * dt;
		dEp_ditc = plus_e * a.inv_s * dt;
		dE_ditc = (float)(-a.inv_s * e_minus * dt * 0.5);
	}
	float dloss_dinvs = dloss_dalpha * (dalpha_dE * dE_dinvs + dalpha_dEp * dEp_dinvs);
	float dloss_dvariance = dloss_dinvs * a.inv_s * 10;
	float d_iter_cos_true_cos = (a.true_cos >= 0) ? 0.0f : 1.0f;
	float gradient_norm = (float)std::sqrt(a.g[0] * a.g[0] + a.g[1] * a.g[1] + a.g[2] * a.g[2] + 1e-6);
	float pos_gradient_norm_inv = 1 - 1 / gradient_norm;
	float dloss_dnormal_norm = dloss_dalpha * (dalpha_dE * dE_ditc + dEp_ditc * dalpha_dEp) * d_iter_cos_true_cos;
	float dloss_dsdf = dloss_dalpha * dalpha_dE * dE_dsdf;
	dl[3] = f2h(loss_scale * dloss_dsdf);
	for (int d = 0; d < 3; ++d) dl[4 + d] = f2h(F.ek_loss_weight * 2 * LOSS_SCALE * pos_gradient_norm_inv * a.g[d]);
	dl[7] = f2h(loss_scale * dloss_dvariance);
	const float dirv[3] = {dir.x, dir.y, dir.z};
	for (int d = 0; d < 3; ++d) dl[8 + d] = f2h(loss_scale * (dloss_dn[d] + dloss_dnormal_norm * dirv[d]));
	if (inter) {
		for (int d = 0; d < 3; ++d) { inter[d] = drgb[d]; inter[3 + d] = dloss_dn[d]; }
		inter[6] = dloss_dalpha; inter[7] = dloss_dsdf; inter[8] = dloss_dvariance; inter[9] = dloss_dnormal_norm;
	}
	return gradient_norm;
}
void loss_pass2(orc_ctx_s* c, uint32_t i, uint32_t n_rays, const RayLoss& R, uint32_t compacted_base, uint32_t compacted_numsteps) {
	const uint32_t base = c->numsteps[(size_t)i * 2 + 1];
	const float* coords_in = &c->coords[(size_t)base * 7];
	const half_t* net = &c->mlp_out[(size_t)base * 16];
	const uint32_t gn = global_rays(c, n_rays);
	float* coords_out = &c->coords_compacted[(size_t)compacted_base * 7];
	half_t* dloss = &c->dloss_dout[(size_t)compacted_base * 16];

	if (compacted_numsteps == 0) return; // testbed_nerf.cu:1726-1728: returns before any loss output is written

	// loss (testbed_nerf.cu:1737-1802)
	float grad[4], weight_sum, gradient_weight_sum;
	(void)pass2_ray_terms(c->cfg, R.rgbtarget, R.rgb_ray, R.mask_certainty, R.mask_gt, R.weight_sum_raw, (float)gn, grad, &weight_sum, &gradient_weight_sum, &c->loss[i], &c->mask_loss[i]);
	c->ek_loss[i] = 0.f;

	const float loss_scale = LOSS_SCALE / (float)gn; // testbed_nerf.cu:1832
	float rgb_ray2[4] = {0, 0, 0, 0};
	float weight_sum2 = 0.f;
	float T = 1.f;
	const Vec3 dir = R.dir;
	float ek = 0.f;
	for (uint32_t j = 0; j < compacted_numsteps; ++j) {
		for (int q = 0; q < 7; ++q) coords_out[(size_t)j * 7 + q] = coords_in[(size_t)j * 7 + q];
		const half_t* o = net + (size_t)j * 16;
		float dt = unwarp_dt(coords_in[(size_t)j * 7 + 3]);
		float albedo[4];
		albedo_from_output(c, o, albedo);
		AlphaTerms a = alpha_terms(o, dt, dir, 1.0f);
		const float alpha = a.alpha;
		const float weight = alpha * T;
		float shading = esum3(a.g[0] * R.light[0], a.g[1] * R.light[1], a.g[2] * R.light[2]); // normal.dot(light)
		if (c->cfg.apply_relu) shading = shading > 0.f ? shading : 0.f;
		for (int k = 0; k < 4; ++k) rgb_ray2[k] += weight * albedo[k] * shading;
		weight_sum2 += weight;
		T *= (1.f - alpha);
		half_t dl[16];
		const float gradient_norm = pass2_sample(c->cfg, grad, R.rgb_ray, weight_sum, gradient_weight_sum, R.light, dir, loss_scale, o, dt, albedo, a, shading, weight, T, weight_sum2, rgb_ray2, dl);
		ek += (gradient_norm - 1.0f) * (gradient_norm - 1.0f);
		for (int q = 0; q < 16; ++q) dloss[(size_t)j * 16 + q] = dl[q];
	}
	c->ek_loss[i] = ek / ((float)compacted_numsteps * (float)gn);
}

void compute_loss(orc_ctx_s* c, uint32_t n_rays, uint32_t n_rays_total) {
	const uint32_t n_kept = c->counters[2];
	const uint32_t B = c->cfg.target_batch_size;
	std::fill(c->loss.begin(), c->loss.begin() + n_rays, 0.f);
	std::fill(c->ek_loss.begin(), c->ek_loss.begin() + n_rays, 0.f);
	std::fill(c->mask_loss.begin(), c->mask_loss.begin() + n_rays, 0.f);
	std::vector<RayLoss> R(n_kept);
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t i = 0; i < (int64_t)n_kept; ++i) loss_pass1(c, (uint32_t)i, n_rays, n_rays_total, R[i]);
	// compaction in ray-slot order (deviation D3; testbed_nerf.cu:1722-1728)
	std::vector<uint32_t> cbase(n_kept), cnum(n_kept);
	uint32_t counter = 0;
	for (uint32_t i = 0; i < n_kept; ++i) {
		uint32_t b = counter;
		counter += R[i].n_comp;
		cbase[i] = b;
		cnum[i] = std::min(B - std::min(B, b), R[i].n_comp);
	}
	c->counters[1] = counter;
#pragma omp parallel for schedule(dynamic, 64)
	for (int64_t i = 0; i < (int64_t)n_kept; ++i) {
		loss_pass2(c, (uint32_t)i, n_rays, R[i], cbase[i], cnum[i]);
		c->numsteps[(size_t)i * 2 + 0] = cnum[i];
		c->numsteps[(size_t)i * 2 + 1] = cbase[i];
	}
	// fill_rollover_and_rescale / fill_rollover (common_device.h:514-535; testbed_nerf.cu:4044-4049)
	const uint32_t n_in = counter;
	if (n_in > 0 && n_in < B) {
		for (uint64_t q = (uint64_t)n_in * 16; q < (uint64_t)B * 16; ++q) {
			float v = h2f(c->dloss_dout[q % ((uint64_t)n_in * 16)]);
			c->dloss_dout[q] = f2h(v * n_in / B);
		}
		for (uint64_t q = (uint64_t)n_in * 7; q < (uint64_t)B * 7; ++q) c->coords_compacted[q] = c->coords_compacted[q % ((uint64_t)n_in * 7)];
	}
}

// ======================================================================
// K10/K11 over the compacted batch, K12 optimizer
// ======================================================================

void forward_infer(orc_ctx_s* c, const float* coords, uint32_t n, half_t* out, bool inference) {
	NetParams np = net_params(c, inference);
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) forward_sample(c, np, coords + (size_t)i * 7, out + (size_t)i * 16, nullptr);
}

// dW[o][i] = sum_s Y_s[o] X_s[i] over the batch as the reference's split-K CUTLASS GEMM with half accumulators would round it
// (EMULATION MODEL, see dot_h): 4096-sample slices (split_k_factor = batch / 2^12, fully_fused_mlp.cu:953), inside a slice the
// accumulator is rounded to half after every 16-sample k-step, the slices' results are summed in half (GemmSplitKParallel's
// reduction, cutlass_matmul.h:315-322).
template <class GetY, class GetX>
static float emulated_dw(uint32_t B, bool acc16, GetY&& y, GetX&& x) {
	if (!acc16) {
		float acc = 0.f;
		for (uint32_t s = 0; s < B; ++s) acc += y(s) * x(s);
		return acc;
	}
	half_t total = 0;
	for (uint32_t s0 = 0; s0 < B; s0 += 4096) {
		half_t acc = 0;
		for (uint32_t k0 = s0; k0 < std::min(B, s0 + 4096); k0 += 16) {
			float part = 0.f;
			for (uint32_t s = k0; s < std::min(B, k0 + 16); ++s) part += y(s) * x(s);
			acc = f2h(h2f(acc) + part);
		}
		total = hadd(total, acc);
	}
	return h2f(total);
}

// Second pass of forward_backward when an accumulation is emulated: the GEMMs and the scatter from the recorded operands.
static void emulated_accumulate(orc_ctx_s* c, const std::vector<SampleOps>& ops) {
	const uint32_t B = (uint32_t)ops.size();
	const bool a16 = c->emul_fp16_acc;
	float* g = c->grads.data();
	struct Gemm { uint64_t off; int n_out, n_in; int which; };
	const uint64_t s0 = c->off_sdf, r0 = c->off_rgb;
	const Gemm gemms[5] = {{s0, 64, 32, 0}, {s0 + 2048, 16, 64, 1}, {r0, 64, 48, 2}, {r0 + 3072, 64, 64, 3}, {r0 + 7168, 16, 64, 4}};
	for (const Gemm& G : gemms) {
#pragma omp parallel for schedule(static)
		for (int q = 0; q < G.n_out * G.n_in; ++q) {
			const int o = q / G.n_in, i = q % G.n_in;
			float first = 0.f, second = 0.f;
			bool has_second = false;
			switch (G.which) {
				case 0: first = emulated_dw(B, a16, [&](uint32_t s) { return h2f(ops[s].dz[o]); }, [&](uint32_t s) { return h2f(ops[s].sdf_in[i]); });
				        second = emulated_dw(B, a16, [&](uint32_t s) { return h2f(ops[s].dz1[o]); }, [&](uint32_t s) { return h2f(ops[s].ddin[i]); }); has_second = true; break;
				case 1: first = emulated_dw(B, a16, [&](uint32_t s) { return h2f(ops[s].dso[o]); }, [&](uint32_t s) { return h2f(ops[s].z1[i]); });
				        second = o == 0 ? emulated_dw(B, a16, [&](uint32_t) { return 1.0f; }, [&](uint32_t s) { return h2f(ops[s].front[i]); }) : 0.f; has_second = true; break;
				case 2: first = emulated_dw(B, a16, [&](uint32_t s) { return h2f(ops[s].dh1[o]); }, [&](uint32_t s) { return h2f(ops[s].c_in[i]); }); break;
				case 3: first = emulated_dw(B, a16, [&](uint32_t s) { return h2f(ops[s].dh2[o]); }, [&](uint32_t s) { return h2f(ops[s].h1[i]); }); break;
				default: first = emulated_dw(B, a16, [&](uint32_t s) { return h2f(ops[s].dr[o]); }, [&](uint32_t s) { return h2f(ops[s].h2[i]); }); break;
			}
			float v = rh(first);                  // beta = 0: the GEMM result is stored as half
			if (has_second) v = rh(second + v);   // beta = 1 (EGradientMode::Accumulate, fully_fused_mlp.cu:1127)
			g[G.off + q] = v;
		}
	}
	const uint32_t L = c->cfg.n_levels;
#pragma omp parallel for schedule(dynamic, 1)
	for (int level = 0; level < (int)L; ++level) {
		if ((uint32_t)level > c->valid_level) continue;
		float* gg = g + c->off_grid + (uint64_t)c->offsets[level] * 2;
		const size_t n = (size_t)(c->offsets[level + 1] - c->offsets[level]) * 2;
		if (c->cfg.deterministic) { // exact integer sums, narrowed once (whatever the accumulate mode: the order of the half atomics no longer exists)
			std::vector<int64_t> gf(n, 0);
			for (uint32_t s = 0; s < B; ++s) {
				const SampleOps& o = ops[s];
				const float g1[2] = {h2f(o.dsin[3 + level * 2]), h2f(o.dsin[3 + level * 2 + 1])};
				const float g2[2] = {h2f(o.dsdf_din[3 + level * 2]), h2f(o.dsdf_din[3 + level * 2 + 1])};
				scatter_level(c, level, o.x, g1, g2, o.dn, [&](uint32_t q, float v) { gf[q] += (int64_t)(v * 16777216.0f); });
			}
			for (size_t q = 0; q < n; ++q) gg[q] = fixed24_to_float(gf[q]);
			continue;
		}
		std::vector<half_t> gh(c->emul_half_atomics ? n : 0, 0);
		std::vector<uint32_t> order;
		if (c->emul_half_atomics && c->atomic_order_seed) {
			order.resize(B);
			for (uint32_t s = 0; s < B; ++s) order[s] = s;
			std::mt19937 gen(c->atomic_order_seed * 7919u + (uint32_t)level);
			std::shuffle(order.begin(), order.end(), gen);
		}
		for (uint32_t si = 0; si < B; ++si) {
			const uint32_t s = order.empty() ? si : order[si];
			const SampleOps& o = ops[s];
			const float g1[2] = {h2f(o.dsin[3 + level * 2]), h2f(o.dsin[3 + level * 2 + 1])};
			const float g2[2] = {h2f(o.dsdf_din[3 + level * 2]), h2f(o.dsdf_din[3 + level * 2 + 1])};
			if (c->emul_half_atomics) scatter_level(c, level, o.x, g1, g2, o.dn, [&](uint32_t q, float v) { gh[q] = hadd(gh[q], f2h(v)); }); // atomicAdd(__half2), grid.h:416
			else scatter_level(c, level, o.x, g1, g2, o.dn, [&](uint32_t q, float v) { gg[q] += v; });
		}
		if (c->emul_half_atomics) for (size_t q = 0; q < n; ++q) gg[q] = h2f(gh[q]);
	}
}

void forward_backward(orc_ctx_s* c) {
	const uint32_t B = c->cfg.target_batch_size;
	NetParams np = net_params(c, false);
	std::fill(c->grads.begin(), c->grads.end(), 0.f);
	float* grid_grad = c->grads.data() + c->off_grid;
	int n_threads = 1;
#ifdef _OPENMP
	n_threads = omp_get_max_threads();
#endif
	std::vector<MlpGrads> partial(n_threads);
	const bool emulate = c->emul_fp16_acc || c->emul_half_atomics;
	std::vector<int64_t> fixed((c->cfg.deterministic && !emulate) ? c->n_params - c->off_grid : 0, 0); // rnb_config::deterministic: the hash-grid sums as integers
	std::vector<SampleOps> ops(emulate ? B : 0);
#pragma omp parallel
	{
		int tid = 0;
#ifdef _OPENMP
		tid = omp_get_thread_num();
#endif
		MlpGrads& mg = partial[tid];
#pragma omp for schedule(static)
		for (int64_t i = 0; i < (int64_t)B; ++i) {
			FwdCtx k;
			half_t out[16];
			forward_sample(c, np, &c->coords_compacted[(size_t)i * 7], out, &k);
			// batch_size of the Eikonal term = the samples of the whole step (nerf_network.h:359-365): all ranks' batches
			backward_sample(c, np, k, &c->dloss_dout[(size_t)i * 16], B * c->cfg.world_size, grid_grad, mg, emulate ? &ops[i] : nullptr, fixed.empty() ? nullptr : fixed.data());
		}
	}
	double var = 0.0;
	for (int t = 0; t < n_threads; ++t) var += partial[t].var;
	if (!fixed.empty()) for (uint64_t q = 0; q < c->off_var - c->off_grid; ++q) grid_grad[q] = fixed24_to_float(fixed[q]); // the one rounding (k_fixed_narrow)
	if (emulate) {
		emulated_accumulate(c, ops);
	} else {
		// reduce: weight gradients are stored half after each GEMM (first order: beta = 0; second order: beta = 1)
		const uint32_t n_mlp = RNB_N_SDF_MLP_PARAMS + RNB_N_RGB_MLP_PARAMS;
		for (uint32_t q = 0; q < n_mlp; ++q) {
			float s1 = 0.f;
			for (int t = 0; t < n_threads; ++t) s1 += partial[t].g1[q];
			float g = rh(s1);
			if (q < RNB_N_SDF_MLP_PARAMS) {
				float s2 = 0.f;
				for (int t = 0; t < n_threads; ++t) s2 += partial[t].g2[q];
				g = rh(s2 + g);
			}
			c->grads[c->off_sdf + q] = g;
		}
	}
	// variance (nerf_network.h:338-339: fp32 sum narrowed to half) and hash-grid gradients stay fp32 sums here;
	// the optimizer narrows every gradient to half once (deviation D2), after the data-parallel all-reduce.
	c->grads[c->off_var] = (float)var;
	for (int q = 1; q < RNB_N_VARIANCE_PARAMS; ++q) c->grads[c->off_var + q] = 0.f;
	// accumulate = RNB_ACCUM_HALF: the gradient vector IS half (trainer.h:78-84) -- what RNB_BUF_GRADS_FP16 hands out, what data-parallel callers exchange and what the
	// optimizer reads (the weight and grid sums above are half values already; the variance's fp32 sum is narrowed here, nerf_network.h:338-339)
	if (c->cfg.accumulate == RNB_ACCUM_HALF) for (uint64_t i = 0; i < c->n_params; ++i) c->grads16[i] = f2h(c->grads[i]);
}

// ExponentialDecayOptimizer::step (exponential_decay.h:61-72): once per optimizer step.
void optimizer_begin(orc_ctx_s* c) {
	if (c->opt_begun) return;
	const rnb_config& cfg = c->cfg;
	const uint32_t step0 = c->optimizer_step_count;
	if (step0 == 0) c->lr_factor = 1.0f;
	if (step0 >= cfg.lr_decay_start && (step0 - cfg.lr_decay_start) % cfg.lr_decay_interval == 0 && step0 <= 10000000u) c->lr_factor *= cfg.lr_decay_base;
	++c->optimizer_step_count;
	c->opt_begun = true;
}

// Parameters [lo, hi) of one optimizer step (the update is independent per parameter).
void optimizer_range(orc_ctx_s* c, const uint64_t lo, const uint64_t hi) {
	const rnb_config& cfg = c->cfg;
	const float base_lr = cfg.learning_rate * c->lr_factor;
	// AdamOptimizer::step (adam.h:309-368) + adam_step (adam.h:52-202)
	const uint32_t current_step = c->optimizer_step_count;
	const uint64_t n_matrix = RNB_N_SDF_MLP_PARAMS + RNB_N_RGB_MLP_PARAMS; // layer_sizes(), nerf_network.h:769-774
	float* w32 = c->params_fp32.data();
	half_t* w16 = c->params_fp16.data();
#pragma omp parallel for schedule(static)
	for (int64_t ii = (int64_t)lo; ii < (int64_t)hi; ++ii) {
		const uint64_t i = (uint64_t)ii;
		float gradient = (c->cfg.accumulate == RNB_ACCUM_HALF ? h2f(c->grads16[i]) : h2f(f2h(c->grads[i]))) / LOSS_SCALE;
		const bool is_matrix = i < n_matrix;
		if (!is_matrix && gradient == 0) continue;
		if (cfg.only_sdf_training && i >= c->off_rgb && i < c->off_grid) continue; // found_reflectance && only_sdf_training (adam.h:121-165)
		const float weight_fp = w32[i];
		if (is_matrix) gradient += cfg.l2_reg * weight_fp;
		const float gradient_sq = gradient * gradient;
		float first_moment = c->adam_m[i] = cfg.beta1 * c->adam_m[i] + (1 - cfg.beta1) * gradient;
		const float second_moment = c->adam_v[i] = cfg.beta2 * c->adam_v[i] + (1 - cfg.beta2) * gradient_sq;
		float learning_rate = base_lr;
		const uint32_t cs = ++c->adam_steps[i];
		learning_rate *= sqrtf(1 - powf(cfg.beta2, (float)cs)) / (1 - powf(cfg.beta1, (float)cs));
		const float lower = 0.f, upper = std::numeric_limits<float>::max();
		const float effective_learning_rate = fminf(fmaxf(learning_rate / (sqrtf(second_moment) + cfg.epsilon), lower), upper);
		const float decayed_weight = (1 - 0.f * learning_rate) * weight_fp - copysignf(0.f * learning_rate, weight_fp);
		const float new_weight = decayed_weight - effective_learning_rate * first_moment;
		w32[i] = new_weight;
		w16[i] = f2h(new_weight);
	}
	// EmaOptimizer::step (ema.h:111-147), half-precision variant (ema.h:63-78)
	const float ema_decay = cfg.ema_decay;
	const float ema_debias_old = 1 - (float)std::pow(ema_decay, current_step - 1);
	const float ema_debias_new = 1.0f / (1 - (float)std::pow(ema_decay, current_step));
	half_t* ema = c->params_ema.data();
#pragma omp parallel for schedule(static)
	for (int64_t ii = (int64_t)lo; ii < (int64_t)hi; ++ii) {
		float filtered = (h2f(ema[ii]) * ema_decay * ema_debias_old + h2f(w16[ii]) * (1 - ema_decay)) * ema_debias_new;
		ema[ii] = f2h(filtered);
	}
}

void optimizer_step(orc_ctx_s* c) {
	optimizer_begin(c);
	optimizer_range(c, 0, c->n_params);
	c->opt_begun = false;
}

// Blocks of the sharded data-parallel optimizer (rnb_shard_layout): the splits sit in front of the four and of the two finest levels, as
// in the HIP library's overlapped schedule for the default network, so that the three-block protocol is exercised.
void shard_layout(const orc_ctx_s* c, rnb_shard_part parts[RNB_MAX_SHARD_PARTS], uint32_t* n_parts) {
	const uint64_t W = std::max(1u, c->cfg.world_size), r = c->cfg.rank, q = 4 * W;
	const uint32_t L = c->cfg.n_levels;
	const uint64_t split = L > 4 ? c->off_grid + (uint64_t)c->offsets[L - 4] * 2 : 0, mid = L > 4 ? c->off_grid + (uint64_t)c->offsets[L - 2] * 2 : 0;
	const uint64_t m0 = split / q * q, m1 = std::max(m0, mid / q * q);
	const bool three = m1 > m0 && m1 < c->param_capacity;
	uint32_t n = 0;
	if (m0) { parts[n].lo = 0; parts[n].hi = m0; ++n; }
	if (three) { parts[n].lo = m0; parts[n].hi = m1; ++n; }
	parts[n].lo = three ? m1 : m0; parts[n].hi = c->param_capacity; ++n;
	for (uint32_t k = 0; k < n; ++k) {
		const uint64_t chunk = (parts[k].hi - parts[k].lo) / W;
		parts[k].own_lo = parts[k].lo + r * chunk;
		parts[k].own_hi = parts[k].own_lo + chunk;
	}
	*n_parts = n;
}

uint32_t compute_valid_level(const rnb_config& cfg, int training_step) { // grid.h:1430-1437
	if (training_step <= 0) return cfg.n_levels;
	float v = cfg.base_valid_level_scale * cfg.n_levels + cfg.valid_level_scale * std::max(0, (int)(training_step - (int)cfg.base_training_step));
	return std::min(cfg.n_levels, (uint32_t)ceilf(v));
}

} // namespace

// ======================================================================
// C-ABI (orc_ prefix)
// ======================================================================
extern "C" {

const char* rnb_last_error(void) { return g_err.c_str(); }
uint32_t rnb_abi_version(void) { return RNB_ABI_VERSION; }

int rnb_default_config(rnb_config* cfg) {
	if (!cfg) return fail(RNB_ERR_INVALID, "cfg is null");
	std::memset(cfg, 0, sizeof(*cfg));
	cfg->abi_version = RNB_ABI_VERSION;
	cfg->n_levels = 14; cfg->log2_hashmap_size = 19; cfg->base_resolution = 16;
	cfg->per_level_scale = std::exp(std::log(2048.0f * 1.0f / 16.0f) / (14 - 1)); // testbed.cu:2320-2323
	cfg->valid_level_scale = 0.02f; cfg->base_valid_level_scale = 0.2f; cfg->base_training_step = 100;
	cfg->sdf_bias = -0.1f;
	cfg->target_batch_size = 1u << 18; cfg->initial_rays_per_batch = 1u << 12; cfg->max_rays_per_batch = 1u << 18;
	cfg->aabb_scale = 1; cfg->seed = 1337;
	cfg->mask_loss_weight = 1.0f; cfg->ek_loss_weight = 0.01f;
	cfg->apply_L2 = 1; cfg->apply_rgbplus = 1; cfg->apply_no_albedo = 0; cfg->apply_light_opti = 0;
	cfg->apply_supernormal = 0; cfg->apply_relu = 0; cfg->apply_bce = 0; cfg->snap_to_pixel_centers = 1;
	cfg->learning_rate = 1e-3f; cfg->beta1 = 0.9f; cfg->beta2 = 0.99f; cfg->epsilon = 1e-15f; cfg->l2_reg = 1e-6f;
	cfg->ema_decay = 0.95f; cfg->lr_decay_start = 20000; cfg->lr_decay_interval = 10000; cfg->lr_decay_base = 0.33f;
	cfg->density_grid_decay = 0.95f;
	cfg->world_size = 1; cfg->rank = 0;
	cfg->overlap = 1; // scheduling hint of the HIP library; the oracle is serial
	return RNB_OK;
}

int rnb_create(const rnb_config* cfg, orc_ctx_s** out) {
	if (!cfg || !out) return fail(RNB_ERR_INVALID, "null argument");
	if (cfg->abi_version != RNB_ABI_VERSION) return fail(RNB_ERR_INVALID, "abi_version mismatch");
	if (cfg->n_levels == 0 || cfg->n_levels > 14) return fail(RNB_ERR_INVALID, "n_levels must be in [1,14]");
	if (cfg->aabb_scale == 0 || (cfg->aabb_scale & (cfg->aabb_scale - 1)) != 0 || cfg->aabb_scale > 128) return fail(RNB_ERR_INVALID, "aabb_scale must be a power of two <= 128");
	if (cfg->target_batch_size == 0 || cfg->target_batch_size % 128 != 0) return fail(RNB_ERR_INVALID, "target_batch_size must be a positive multiple of 128");
	if (cfg->world_size == 0 || cfg->rank >= cfg->world_size) return fail(RNB_ERR_INVALID, "bad rank/world_size");
	orc_ctx_s* c = new orc_ctx_s();
	c->cfg = *cfg;
	build_grid_tables(c);
	c->off_sdf = 0;
	c->off_rgb = RNB_N_SDF_MLP_PARAMS;
	c->off_grid = c->off_rgb + RNB_N_RGB_MLP_PARAMS;
	c->off_var = c->off_grid + c->n_grid_params;
	c->n_params = c->off_var + RNB_N_VARIANCE_PARAMS;
	// load_nerf (testbed_nerf.cu:3198-3214)
	c->aabb_min = 0.5f - 0.5f * std::min(1u << (N_CASCADES - 1), cfg->aabb_scale);
	c->aabb_max = 0.5f + 0.5f * std::min(1u << (N_CASCADES - 1), cfg->aabb_scale);
	c->max_cascade = 0;
	while ((1u << c->max_cascade) < cfg->aabb_scale) ++c->max_cascade;
	c->cone_angle = cfg->aabb_scale <= 1 ? 0.0f : (1.0f / 256.0f);
	{
		const uint64_t q = 4ull * std::max(1u, cfg->world_size);
		c->param_capacity = (c->n_params + q - 1) / q * q;
	}
	c->params_fp32.assign(c->param_capacity, 0.f);
	c->params_fp16.assign(c->param_capacity, 0);
	c->params_ema.assign(c->param_capacity, 0);
	c->grads.assign(c->param_capacity, 0.f);
	c->grads16.assign(c->cfg.accumulate == RNB_ACCUM_HALF ? c->param_capacity : 0, (half_t)0);
	c->adam_m.assign(c->param_capacity, 0.f);
	c->adam_v.assign(c->param_capacity, 0.f);
	c->adam_steps.assign(c->param_capacity, 0);
	const uint32_t n_grid = GRID_CELLS * (c->max_cascade + 1);
	c->density_grid.assign(n_grid, 0.f);
	c->density_grid_tmp.assign(n_grid, 0.f);
	c->bitfield.assign((size_t)GRID_CELLS / 8 * N_CASCADES, 0);
	const uint32_t B = cfg->target_batch_size;
	const uint32_t maxr = cfg->max_rays_per_batch;
	c->ray_indices.assign(maxr, 0);
	c->rays.assign((size_t)maxr * 6, 0.f);
	c->numsteps.assign((size_t)maxr * 2, 0);
	c->coords.assign((size_t)B * 16 * 7, 0.f);
	c->mlp_out.assign((size_t)B * 16 * 16, 0);
	c->dloss_dout.assign((size_t)B * 16, 0);
	c->coords_compacted.assign((size_t)B * 7, 0.f);
	c->loss.assign(maxr, 0.f); c->ek_loss.assign(maxr, 0.f); c->mask_loss.assign(maxr, 0.f);
	// Testbed::reset_network (testbed.cu:2223-2237)
	c->rng = Pcg32{cfg->seed};
	c->density_grid_rng = Pcg32{c->rng.next_uint()};
	(void)c->rng.next_uint(); // tv_loss_rng
	c->rays_per_batch = cfg->initial_rays_per_batch;
	c->measured_batch_size_before_compaction = 0;
	c->training_step = 0;
	c->valid_level = compute_valid_level(c->cfg, 0);
	build_light_dirs(c);
	if (cfg->accumulate > RNB_ACCUM_HALF) { delete c; return fail(RNB_ERR_INVALID, "accumulate must be RNB_ACCUM_FP32 or RNB_ACCUM_HALF"); }
	if (cfg->deterministic > 1) { delete c; return fail(RNB_ERR_INVALID, "deterministic must be 0 or 1"); }
	// rnb_config::accumulate = RNB_ACCUM_HALF: the model of the reference as coded, both parts; the two environment variables switch the parts on one by one (tools/oracle_deviation_report.py)
	c->emul_fp16_acc = cfg->accumulate == RNB_ACCUM_HALF || (getenv("ORC_EMULATE_FP16_ACCUM") != nullptr && atoi(getenv("ORC_EMULATE_FP16_ACCUM")) != 0);
	c->emul_half_atomics = cfg->accumulate == RNB_ACCUM_HALF || (getenv("ORC_EMULATE_HALF_ATOMICS") != nullptr && atoi(getenv("ORC_EMULATE_HALF_ATOMICS")) != 0);
	c->atomic_order_seed = getenv("ORC_ATOMIC_ORDER_SEED") ? (uint32_t)atoi(getenv("ORC_ATOMIC_ORDER_SEED")) : 0u;
	*out = c;
	return RNB_OK;
}

int rnb_destroy(orc_ctx_s* c) { delete c; return RNB_OK; }
int rnb_update_config(orc_ctx_s* c, const rnb_config* cfg) {
	if (!c || !cfg) return fail(RNB_ERR_INVALID, "null argument");
	if (cfg->abi_version != RNB_ABI_VERSION) return fail(RNB_ERR_INVALID, "abi_version mismatch");
	rnb_config& dst = c->cfg;
	if (cfg->n_levels != dst.n_levels || cfg->log2_hashmap_size != dst.log2_hashmap_size || cfg->base_resolution != dst.base_resolution ||
	    cfg->per_level_scale != dst.per_level_scale || cfg->target_batch_size != dst.target_batch_size || cfg->max_rays_per_batch != dst.max_rays_per_batch ||
	    cfg->aabb_scale != dst.aabb_scale || cfg->seed != dst.seed || cfg->world_size != dst.world_size || cfg->rank != dst.rank || cfg->accumulate != dst.accumulate || cfg->deterministic != dst.deterministic)
		return fail(RNB_ERR_INVALID, "rnb_update_config: geometry fields differ from the context's");
	dst = *cfg;
	build_light_dirs(c);
	return RNB_OK;
}

uint64_t rnb_n_params(const orc_ctx_s* c) { return c ? c->n_params : 0; }

int rnb_param_layout(const orc_ctx_s* c, uint64_t offsets[5]) {
	if (!c || !offsets) return fail(RNB_ERR_INVALID, "null argument");
	offsets[0] = c->off_sdf; offsets[1] = c->off_rgb; offsets[2] = c->off_grid; offsets[3] = c->off_var; offsets[4] = c->n_params;
	return RNB_OK;
}

int rnb_grid_tables(const orc_ctx_s* c, uint32_t* offsets, uint32_t* resolution, float* scale) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	for (uint32_t i = 0; i <= c->cfg.n_levels; ++i) if (offsets) offsets[i] = c->offsets[i];
	for (uint32_t i = 0; i < c->cfg.n_levels; ++i) { if (resolution) resolution[i] = c->resolution[i]; if (scale) scale[i] = c->scale[i]; }
	return RNB_OK;
}

static void derive_half_params(orc_ctx_s* c) {
	for (uint64_t i = 0; i < c->n_params; ++i) c->params_fp16[i] = f2h(c->params_fp32[i]); // trainer.h:103-107
}

int rnb_init_params(orc_ctx_s* c, const float* sdf_w) {
	if (!c || !sdf_w) return fail(RNB_ERR_INVALID, "null argument");
	// Trainer ctor (trainer.h:54-61)
	std::seed_seq seq{c->cfg.seed};
	std::vector<uint32_t> seeds(2);
	seq.generate(std::begin(seeds), std::end(seeds));
	Pcg32 rnd{seeds.front()};
	std::fill(c->params_fp32.begin(), c->params_fp32.end(), 0.f);
	float* p = c->params_fp32.data();
	auto xavier = [&](float* dst, int rows, int cols) { // gpu_matrix.h:292-306; T=float
		float scale = 1.f;
		scale *= std::sqrt(6.0f / (float)(cols + rows));
		for (int i = 0; i < rows * cols; ++i) dst[i] = (float)(rnd.next_float() * 2.0f * scale - scale);
	};
	// density network: Xavier (consumes draws), then overwritten by the geometric init (nerf_network.h:629-643)
	xavier(p + c->off_sdf, 64, 32);
	xavier(p + c->off_sdf + 64 * 32, 16, 64);
	std::memcpy(p + c->off_sdf, sdf_w, sizeof(float) * RNB_N_SDF_MLP_PARAMS);
	// rgb network (nerf_network.h:647-656)
	xavier(p + c->off_rgb, 64, 48);
	xavier(p + c->off_rgb + 64 * 48, 64, 64);
	xavier(p + c->off_rgb + 64 * 48 + 64 * 64, 16, 64);
	// hash grid U(-1e-4, 1e-4) (grid.h:1379-1384; random.h:67-93: thread i draws elements i + n_threads*j from stream offset 4*i)
	{
		const uint64_t n = c->n_grid_params;
		const uint64_t n_thr = (n + 3) / 4;
		const uint64_t n_threads_total = ((n_thr + 127) / 128) * 128;
		float* g = p + c->off_grid;
#pragma omp parallel for schedule(static)
		for (int64_t i = 0; i < (int64_t)n_threads_total; ++i) {
			Pcg32 r = rnd;
			r.advance(i * 4);
			for (uint64_t j = 0; j < 4; ++j) {
				const uint64_t idx = (uint64_t)i + n_threads_total * j;
				if (idx >= n) break;
				float val = r.next_float();
				g[idx] = val * (1e-4f - (-1e-4f)) + (-1e-4f);
			}
		}
		rnd.advance((int64_t)n);
	}
	// dir encoding: no params. variance: own pcg32{1337}, U(0.3,0.3) (nerf_network.h:691-692)
	for (int q = 0; q < RNB_N_VARIANCE_PARAMS; ++q) {
		Pcg32 vr{1337};
		vr.advance(q); // 4 elements -> one thread, consecutive draws
		float val = vr.next_float();
		p[c->off_var + q] = val * (0.300f - 0.300f) + 0.300f;
	}
	c->trainer_rng = rnd;
	derive_half_params(c);
	std::fill(c->params_ema.begin(), c->params_ema.end(), 0);
	std::fill(c->adam_m.begin(), c->adam_m.end(), 0.f);
	std::fill(c->adam_v.begin(), c->adam_v.end(), 0.f);
	std::fill(c->adam_steps.begin(), c->adam_steps.end(), 0);
	c->optimizer_step_count = 0;
	return RNB_OK;
}

int rnb_set_params(orc_ctx_s* c, const float* params) {
	if (!c || !params) return fail(RNB_ERR_INVALID, "null argument");
	std::memcpy(c->params_fp32.data(), params, sizeof(float) * c->n_params);
	derive_half_params(c);
	c->params_ema = c->params_fp16;
	std::fill(c->adam_m.begin(), c->adam_m.end(), 0.f);
	std::fill(c->adam_v.begin(), c->adam_v.end(), 0.f);
	std::fill(c->adam_steps.begin(), c->adam_steps.end(), 0);
	c->optimizer_step_count = 0;
	return RNB_OK;
}

int rnb_buffer(orc_ctx_s* c, int id, void** ptr, uint64_t* n_bytes) {
	if (!c || !ptr || !n_bytes) return fail(RNB_ERR_INVALID, "null argument");
	id &= ~RNB_BUF_READONLY; // the checker keeps no cached forms: reads and writes through the pointers need no announcement
#define BUF(vec) do { *ptr = (void*)(vec).data(); *n_bytes = (vec).size() * sizeof((vec)[0]); return RNB_OK; } while (0)
#define BUF_P(vec) do { *ptr = (void*)(vec).data(); *n_bytes = c->n_params * sizeof((vec)[0]); return RNB_OK; } while (0)
	switch (id) {
		case RNB_BUF_PARAMS_FP32: BUF_P(c->params_fp32);
		case RNB_BUF_PARAMS_FP16: BUF_P(c->params_fp16);
		case RNB_BUF_PARAMS_EMA: BUF_P(c->params_ema);
		case RNB_BUF_GRADS_FP32: if (c->cfg.accumulate == RNB_ACCUM_HALF) return fail(RNB_ERR_INVALID, "accumulate = RNB_ACCUM_HALF: the gradient vector is RNB_BUF_GRADS_FP16"); BUF_P(c->grads);
		case RNB_BUF_GRADS_FP16: if (c->cfg.accumulate != RNB_ACCUM_HALF) return fail(RNB_ERR_INVALID, "accumulate = RNB_ACCUM_FP32: the gradient accumulators are RNB_BUF_GRADS_FP32"); BUF_P(c->grads16);
		case RNB_BUF_ADAM_M: BUF_P(c->adam_m);
		case RNB_BUF_ADAM_V: BUF_P(c->adam_v);
		case RNB_BUF_ADAM_STEPS: BUF_P(c->adam_steps);
		case RNB_BUF_DENSITY_GRID: BUF(c->density_grid);
		case RNB_BUF_DENSITY_BITFIELD: BUF(c->bitfield);
		case RNB_BUF_DENSITY_MEAN: *ptr = &c->density_mean; *n_bytes = 4; return RNB_OK;
		case RNB_BUF_RAY_INDICES: BUF(c->ray_indices);
		case RNB_BUF_RAYS: BUF(c->rays);
		case RNB_BUF_NUMSTEPS: BUF(c->numsteps);
		case RNB_BUF_COORDS: BUF(c->coords);
		case RNB_BUF_MLP_OUT: BUF(c->mlp_out);
		case RNB_BUF_DLOSS_DOUT: BUF(c->dloss_dout);
		case RNB_BUF_COORDS_COMPACTED: BUF(c->coords_compacted);
		case RNB_BUF_LOSS: BUF(c->loss);
		case RNB_BUF_EK_LOSS: BUF(c->ek_loss);
		case RNB_BUF_MASK_LOSS: BUF(c->mask_loss);
		case RNB_BUF_COUNTERS: *ptr = c->counters; *n_bytes = sizeof(c->counters); return RNB_OK;
		case RNB_BUF_DENSITY_GRID_TMP: BUF(c->density_grid_tmp);
		case RNB_BUF_GRID_SAMPLE_POS: BUF(c->grid_sample_pos);
		case RNB_BUF_GRID_SAMPLE_IDX: BUF(c->grid_sample_idx);
		case RNB_BUF_STEP_VECTOR: *ptr = c->step_vector; *n_bytes = sizeof(c->step_vector); return RNB_OK;
		default: return fail(RNB_ERR_INVALID, "unknown buffer id");
	}
#undef BUF
}

int rnb_params_changed(orc_ctx_s* c) { return c ? RNB_OK : fail(RNB_ERR_INVALID, "null ctx"); } // nothing is cached here
int rnb_bitfield_changed(orc_ctx_s* c) { return c ? RNB_OK : fail(RNB_ERR_INVALID, "null ctx"); } // the march reads the bitfield itself
int rnb_device_malloc(orc_ctx_s* c, uint64_t n_bytes, void** ptr) {
	if (!c || !ptr) return fail(RNB_ERR_INVALID, "null argument");
	*ptr = n_bytes ? std::malloc(n_bytes) : nullptr;
	if (n_bytes && !*ptr) return fail(RNB_ERR_NOMEM, "malloc failed");
	return RNB_OK;
}
int rnb_device_free(orc_ctx_s* c, void* ptr) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	std::free(ptr);
	return RNB_OK;
}

int rnb_memcpy(orc_ctx_s*, void* dst, const void* src, uint64_t n_bytes, int) {
	if (!dst || !src) return fail(RNB_ERR_INVALID, "null argument");
	std::memmove(dst, src, n_bytes);
	return RNB_OK;
}

int rnb_set_dataset(orc_ctx_s* c, uint32_t n_views, const rnb_view* views, const uint16_t* const* normals, const uint16_t* const* albedos) {
	if (!c || !views || !normals || !albedos || n_views == 0) return fail(RNB_ERR_INVALID, "bad dataset");
	c->views.clear();
	c->views.resize(n_views);
	for (uint32_t v = 0; v < n_views; ++v) {
		c->views[v].meta = views[v];
		size_t n = (size_t)views[v].width * views[v].height * 4;
		if (n == 0) return fail(RNB_ERR_INVALID, "empty view");
		c->views[v].normal.assign(normals[v], normals[v] + n);
		c->views[v].albedo.assign(albedos[v], albedos[v] + n);
	}
	return RNB_OK;
}

int rnb_set_training_step(orc_ctx_s* c, uint32_t step) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	c->training_step = step;
	c->valid_level = compute_valid_level(c->cfg, (int)step);
	return RNB_OK;
}
uint32_t rnb_valid_level(const orc_ctx_s* c) { return c ? c->valid_level : 0; }

int rnb_update_density_grid(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	return training_prep(c);
}
int rnb_set_grid_exchange(orc_ctx_s* c, rnb_grid_exchange_fn fn, void* user) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	c->grid_exchange = fn; c->grid_exchange_user = user;
	return RNB_OK;
}
int rnb_update_density_grid_begin(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	training_prep_front(c, c->cfg.world_size > 1);
	return RNB_OK;
}
int rnb_update_density_grid_end(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	update_density_grid_back(c);
	return RNB_OK;
}
int rnb_update_density_bitfield(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	update_bitfield(c);
	return RNB_OK;
}

int rnb_sdf(orc_ctx_s* c, void*, const float* xyz, uint32_t n, uint16_t* out, int inference) {
	if (!c || (!xyz && n) || (!out && n)) return fail(RNB_ERR_INVALID, "null argument");
	NetParams np = net_params(c, inference != 0);
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = sdf_sample(c, np, xyz + (size_t)i * 3);
	return RNB_OK;
}
// get_density_on_grid for the SDF (src/testbed_nerf.cu:4218-4269, 541-553): host twin of rnb_sdf_lattice.
int rnb_sdf_lattice(orc_ctx_s* c, void*, const uint32_t res[3], float lattice_min, float lattice_max, float* out, int inference) {
	if (!c || !res || !out) return fail(RNB_ERR_INVALID, "null argument");
	if (!res[0] || !res[1] || !res[2]) return fail(RNB_ERR_INVALID, "empty lattice");
	NetParams np = net_params(c, inference != 0);
	const int64_t n = (int64_t)res[0] * res[1] * res[2];
	const float size = lattice_max - lattice_min, diag = c->aabb_max - c->aabb_min;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < n; ++i) {
		const uint32_t p[3] = {(uint32_t)(i % res[0]), (uint32_t)((i / res[0]) % res[1]), (uint32_t)(i / ((int64_t)res[0] * res[1]))};
		float x[3];
		for (int k = 0; k < 3; ++k) {
			const float inv = 1.f / (float)res[k];
			const float w = (float)p[k] * inv * size + lattice_min;
			x[k] = (w - c->aabb_min) / diag; // warp_position
		}
		out[i] = h2f(sdf_sample(c, np, x));
	}
	return RNB_OK;
}

// marching_cubes_gpu (src/marching_cubes.cu:794-822): the checker's twin of rnb_marching_cubes (orc_mesh.h).
int rnb_marching_cubes(orc_ctx_s* c, void*, const float* density, const uint32_t res[3], const float aabb_min[3], const float aabb_max[3], float thresh,
                       float** verts, uint32_t** indices, uint32_t* n_verts, uint32_t* n_indices) {
	if (!c || !density || !res || !aabb_min || !aabb_max || !verts || !indices || !n_verts || !n_indices) return fail(RNB_ERR_INVALID, "null argument");
	std::vector<float> v;
	std::vector<uint32_t> idx;
	if (!orc_mesh::marching_cubes(density, res, aabb_min, aabb_max, thresh, v, idx)) return fail(RNB_ERR_INVALID, "marching cubes: missing edge vertex");
	*n_verts = (uint32_t)(v.size() / 3); *n_indices = (uint32_t)idx.size();
	*verts = v.empty() ? nullptr : (float*)std::malloc(v.size() * 4);
	*indices = idx.empty() ? nullptr : (uint32_t*)std::malloc(idx.size() * 4);
	if (*verts) std::memcpy(*verts, v.data(), v.size() * 4);
	if (*indices) std::memcpy(*indices, idx.data(), idx.size() * 4);
	return RNB_OK;
}

int rnb_density(orc_ctx_s* c, void*, const float* xyz, uint32_t n, uint16_t* out, int inference) {
	if (!c || (!xyz && n) || (!out && n)) return fail(RNB_ERR_INVALID, "null argument");
	NetParams np = net_params(c, inference != 0);
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = sdf_to_density(sdf_sample(c, np, xyz + (size_t)i * 3), np.variance);
	return RNB_OK;
}
int rnb_forward_infer(orc_ctx_s* c, void*, const float* coords, uint32_t n, uint16_t* out, int inference) {
	if (!c || (!coords && n) || (!out && n)) return fail(RNB_ERR_INVALID, "null argument");
	forward_infer(c, coords, n, out, inference != 0);
	return RNB_OK;
}

int rnb_generate_training_samples(orc_ctx_s* c, void*, uint32_t n_rays, uint32_t n_rays_total, uint32_t max_samples) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (c->views.empty()) return fail(RNB_ERR_INVALID, "no dataset");
	if (n_rays == 0 || n_rays > c->cfg.max_rays_per_batch) return fail(RNB_ERR_INVALID, "n_rays out of range");
	if (max_samples > c->cfg.target_batch_size * 16) return fail(RNB_ERR_INVALID, "max_samples exceeds 16*target_batch_size");
	generate_training_samples(c, n_rays, n_rays_total, max_samples);
	return RNB_OK;
}

int rnb_compute_loss(orc_ctx_s* c, void*, uint32_t n_rays, uint32_t n_rays_total) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (c->views.empty()) return fail(RNB_ERR_INVALID, "no dataset");
	compute_loss(c, n_rays, n_rays_total);
	return RNB_OK;
}

int rnb_forward_backward(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	forward_backward(c);
	return RNB_OK;
}

int rnb_optimizer_step(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	optimizer_step(c);
	return RNB_OK;
}

// Testbed::train (testbed.cu:2776-2872) up to the backward pass of train_nerf_step (testbed_nerf.cu:3844-4123).
int rnb_train_step_begin(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (c->views.empty()) return fail(RNB_ERR_INVALID, "no dataset");
	c->cur_step = c->training_step;
	c->valid_level = compute_valid_level(c->cfg, (int)c->training_step); // testbed.cu:2792
	c->grid_updated = false;
	c->prep_ms = 0.f;
	const uint32_t n_prep_to_skip = std::min(std::max(c->training_step / 16u, 1u), 16u); // testbed.cu:2805
	if (c->training_step % n_prep_to_skip == 0) {
		auto t0 = std::chrono::steady_clock::now();
		{ const int rc = training_prep(c); if (rc != RNB_OK) return rc; }
		c->grid_updated = true;
		c->prep_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / n_prep_to_skip;
	}
	c->step_start = std::chrono::steady_clock::now();
	const uint32_t B = c->cfg.target_batch_size;
	const uint32_t max_samples = B * 16;
	uint32_t max_inference;
	if (c->measured_batch_size_before_compaction == 0) { // testbed_nerf.cu:3891-3896
		c->measured_batch_size_before_compaction = max_inference = max_samples;
	} else {
		max_inference = next_multiple(std::min(c->measured_batch_size_before_compaction, max_samples), 128u);
	}
	if (c->training_step == 0) c->n_rays_total = 0; // testbed_nerf.cu:3906-3908
	const uint32_t n_rays_total = c->n_rays_total;
	const uint32_t n_rays = c->rays_per_batch;
	c->n_rays_total += n_rays * c->cfg.world_size;
	c->cur_n_rays = n_rays;
	generate_training_samples(c, n_rays, n_rays_total, max_inference);
	forward_infer(c, c->coords.data(), c->counters[3], c->mlp_out.data(), false);
	compute_loss(c, n_rays, n_rays_total);
	forward_backward(c);
	c->rng.advance(); // testbed_nerf.cu:4118
	return RNB_OK;
}

int rnb_train_step_apply(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	optimizer_step(c);
	++c->training_step;
	return RNB_OK;
}

int rnb_train_step_local(orc_ctx_s* c, void*, uint64_t counters[4], double sums[3]) {
	if (!c || !counters || !sums) return fail(RNB_ERR_INVALID, "null argument");
	for (int k = 0; k < 4; ++k) counters[k] = c->counters[k];
	double s0 = 0, s1 = 0, s2 = 0;
	const uint32_t n = std::min(c->counters[2], c->cur_n_rays);
	for (uint32_t i = 0; i < n; ++i) { s0 += c->loss[i]; s1 += c->ek_loss[i]; s2 += c->mask_loss[i]; }
	sums[0] = s0; sums[1] = s1; sums[2] = s2;
	for (int k = 0; k < 4; ++k) c->step_vector[k] = (double)c->counters[k];
	c->step_vector[4] = s0; c->step_vector[5] = s1; c->step_vector[6] = s2;
	c->local_measured_before = c->counters[0];
	return RNB_OK;
}

// Counters::update_after_training (testbed_nerf.cu:3532-3558) on counters summed over the data-parallel ranks.
int rnb_train_step_finish(orc_ctx_s* c, const uint64_t counters[4], const double sums[3], rnb_step_stats* stats) {
	if (!c || !counters || !sums) return fail(RNB_ERR_INVALID, "null argument");
	const uint64_t Bg = (uint64_t)c->cfg.target_batch_size * c->cfg.world_size;
	const uint32_t n_rays = c->cur_n_rays;
	c->measured_batch_size = 0;
	c->measured_batch_size_before_compaction = 0;
	float loss_scalar = 0.f, ek_scalar = 0.f, mask_scalar = 0.f;
	uint32_t next_rays = c->rays_per_batch;
	int rc = RNB_OK;
	if (counters[0] == 0 || counters[1] == 0) {
		rc = RNB_ERR_NO_SAMPLES;
		g_err = "Nerf training generated 0 samples.";
	} else {
		c->measured_batch_size_before_compaction = c->local_measured_before;
		c->measured_batch_size = (uint32_t)(counters[1] / c->cfg.world_size);
		const float measured = (float)counters[1], target = (float)Bg;
		loss_scalar = (float)sums[0] * measured / target;
		ek_scalar = (float)sums[1] * measured / target;
		mask_scalar = (float)sums[2] * measured / target;
		next_rays = (uint32_t)((float)c->rays_per_batch * target / measured);
		next_rays = std::min(next_multiple(next_rays, 128u), c->cfg.max_rays_per_batch);
	}
	if (stats) {
		stats->training_step = c->cur_step + 1; // _finish may be called before _apply
		stats->rays_per_batch = n_rays;
		stats->next_rays_per_batch = next_rays;
		stats->measured_batch_size = (uint32_t)(counters[1] / c->cfg.world_size);
		stats->measured_batch_size_before_compaction = (uint32_t)(counters[0] / c->cfg.world_size);
		stats->n_rays_kept = (uint32_t)(counters[2] / c->cfg.world_size);
		stats->density_grid_updated = c->grid_updated ? 1 : 0;
		stats->loss = loss_scalar; stats->ek_loss = ek_scalar; stats->mask_loss = mask_scalar;
		stats->prep_ms = c->prep_ms;
		stats->step_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c->step_start).count();
	}
	c->rays_per_batch = next_rays;
	return rc;
}

int rnb_train_step_end(orc_ctx_s* c, void* stream, rnb_step_stats* stats) {
	int rc = rnb_train_step_apply(c, stream);
	if (rc != RNB_OK) return rc;
	uint64_t counters[4];
	double sums[3];
	rc = rnb_train_step_local(c, stream, counters, sums);
	if (rc != RNB_OK) return rc;
	return rnb_train_step_finish(c, counters, sums, stats);
}

int rnb_train_step(orc_ctx_s* c, void* stream, rnb_step_stats* stats) {
	int rc = rnb_train_step_begin(c, stream);
	if (rc != RNB_OK) return rc;
	return rnb_train_step_end(c, stream, stats);
}

// The checker keeps no per-kernel timers: the entry points exist so both libraries export the same ABI.
int rnb_profile_enable(orc_ctx_s* c, int) { return c ? RNB_OK : fail(RNB_ERR_INVALID, "null ctx"); }
int rnb_profile_count(const orc_ctx_s*) { return 0; }
int rnb_profile_get(const orc_ctx_s*, int, const char**, double*, uint64_t*, double*) { return fail(RNB_ERR_INVALID, "the CPU checker has no profile entries"); }

uint32_t rnb_training_step(const orc_ctx_s* c) { return c ? c->training_step : 0; }
uint32_t rnb_rays_per_batch(const orc_ctx_s* c) { return c ? c->rays_per_batch : 0; }

// rnb_eval_primitives (include/rnb_neus2.h): the checker's own statements of the index primitives, item by item.
int rnb_eval_primitives(orc_ctx_s* c, int kind, const uint32_t* in, uint32_t n_items, uint32_t* out) {
	if (!c || (!in && n_items) || (!out && n_items)) return fail(RNB_ERR_INVALID, "null argument");
	if (kind < 0 || kind > RNB_PRIM_DW_SLICED) return fail(RNB_ERR_INVALID, "unknown primitive kind");
	constexpr uint32_t DW_S = 8256; // RNB_PRIM_DW_SLICED: samples per GEMM
	static const uint32_t IN_W[20] = {6, 3, 1, 8, 9, 1, 9, 9, 9, 7, 32, 20, 35, 37, 16, 263, 10, 2, 1, 4 + 8 * DW_S / 2}, OUT_W[20] = {4, 4, 2, 3, 7, 3, 11, 5, 3, 3, 5, 9, 7, 28, 9, 16, 23, 1, 2, 16};
	auto f = [](uint32_t u) { float v; std::memcpy(&v, &u, 4); return v; };
	auto u = [](float v) { uint32_t w; std::memcpy(&w, &v, 4); return w; };
	std::vector<uint8_t> bf;
	if (kind == RNB_PRIM_MARCH || kind == RNB_PRIM_MARCH_RAY) {
		bf.resize((size_t)GRID_CELLS / 8 * N_CASCADES);
		Pcg32 q{5};
		for (auto& b : bf) b = (uint8_t)(q.next_uint() >> 24);
	}
	orc_ctx_s box; // ray_intersect / aabb_contains read the box from a context
	for (uint32_t i = 0; i < n_items; ++i) {
		const uint32_t* a = in + (size_t)i * IN_W[kind];
		uint32_t* o = out + (size_t)i * OUT_W[kind];
		if (kind == RNB_PRIM_PCG32) {
			Pcg32 r{(uint64_t)a[0] | (uint64_t)a[1] << 32, (uint64_t)a[2] | (uint64_t)a[3] << 32};
			r.advance((int64_t)((uint64_t)a[4] | (uint64_t)a[5] << 32));
			o[0] = (uint32_t)(r.state >> 32); o[1] = (uint32_t)r.state;
			Pcg32 r2 = r;
			o[2] = r.next_uint();
			o[3] = u(r2.next_float());
		} else if (kind == RNB_PRIM_MORTON) {
			const uint32_t m = morton3D(a[0], a[1], a[2]);
			o[0] = m; o[1] = morton3D_invert(m >> 0); o[2] = morton3D_invert(m >> 1); o[3] = morton3D_invert(m >> 2);
		} else if (kind == RNB_PRIM_SRGB) {
			o[0] = u(srgb_to_linear(f(a[0]))); o[1] = u(linear_to_srgb(f(a[0])));
		} else if (kind == RNB_PRIM_RAY_BOX) {
			box.aabb_min = f(a[0]); box.aabb_max = f(a[1]);
			const Vec3 p = {f(a[2]), f(a[3]), f(a[4])}, d = {f(a[5]), f(a[6]), f(a[7])};
			float t0, t1;
			ray_intersect(&box, p, d, &t0, &t1);
			o[0] = u(t0); o[1] = u(t1); o[2] = aabb_contains(&box, p) ? 1u : 0u;
		} else if (kind == RNB_PRIM_ACTIVATION) {
			const float l = logistic(f(a[0]));
			o[0] = u(relu(f(a[0]))); o[1] = u(l); o[2] = u(l * (1 - l));
		} else if (kind == RNB_PRIM_WARP) {
			box.aabb_min = f(a[0]); box.aabb_max = f(a[1]);
			const Vec3 wp = warp_position(&box, {f(a[2]), f(a[3]), f(a[4])}), wd = warp_direction({f(a[5]), f(a[6]), f(a[7])}), ud = unwarp_direction(wd);
			const float wt = warp_dt(f(a[8]));
			o[0] = u(wp.x); o[1] = u(wp.y); o[2] = u(wp.z); o[3] = u(wd.x); o[4] = u(wd.y); o[5] = u(wd.z); o[6] = u(ud.x); o[7] = u(ud.y); o[8] = u(ud.z); o[9] = u(wt); o[10] = u(unwarp_dt(wt));
		} else if (kind == RNB_PRIM_LOSS) {
			const float t[4] = {f(a[1]), f(a[2]), f(a[3]), f(a[4])}, p[4] = {f(a[5]), f(a[6]), f(a[7]), f(a[8])};
			float g[4];
			o[0] = u(loss_and_gradient(a[0] != 0, t, p, g));
			o[1] = u(g[0]); o[2] = u(g[1]); o[3] = u(g[2]); o[4] = u(g[3]);
		} else if (kind == RNB_PRIM_PIXEL) {
			Pcg32 r{1337};
			r.advance((int64_t)((uint64_t)a[7] | (uint64_t)a[8] << 32));
			float xy[2];
			random_image_pos(r, a[4], a[5], a[6] != 0, xy);
			o[0] = image_idx(a[0], a[1], a[2], a[3]); o[1] = u(xy[0]); o[2] = u(xy[1]);
		} else if (kind == RNB_PRIM_READ_RGBA) {
			rnb_view m{};
			m.width = a[0]; m.height = a[1];
			const float xy[2] = {f(a[2]), f(a[3])};
			float cl[4];
			read_rgba(xy, m, reinterpret_cast<const uint16_t*>(a + 4), cl); // the item's own words are the image: RGBA16, two words per pixel
			o[0] = u(cl[0]); o[1] = u(cl[1]); o[2] = u(cl[2]); o[3] = u(cl[3]);
			o[4] = cl[0] <= 0.0f ? 1u : 0u; // the test of testbed_nerf.cu:1264
		} else if (kind == RNB_PRIM_CAMERA_RAY) {
			rnb_view m{};
			m.width = a[0]; m.height = a[1]; m.focal_length[0] = f(a[2]); m.focal_length[1] = f(a[3]); m.principal_point[0] = f(a[4]); m.principal_point[1] = f(a[5]);
			for (int k = 0; k < 12; ++k) m.xform[k] = f(a[8 + k]);
			const float xy[2] = {f(a[6]), f(a[7])};
			Vec3 ro, du, dir;
			camera_ray(m, xy, ro, du, dir);
			o[0] = u(ro.x); o[1] = u(ro.y); o[2] = u(ro.z); o[3] = u(du.x); o[4] = u(du.y); o[5] = u(du.z); o[6] = u(dir.x); o[7] = u(dir.y); o[8] = u(dir.z);
		} else if (kind == RNB_PRIM_RAY_TARGETS) {
			rnb_config F{};
			F.apply_no_albedo = a[0]; F.apply_rgbplus = a[1]; F.apply_L2 = a[2]; F.apply_light_opti = a[3]; F.apply_relu = a[4];
			float X[12], tn[4], ta[4], ld[9], tgt[4], lw[3];
			for (int k = 0; k < 12; ++k) X[k] = f(a[6 + k]);
			for (int k = 0; k < 4; ++k) { tn[k] = f(a[18 + k]); ta[k] = f(a[22 + k]); }
			bool own = true; // nine words 0xffffffff = the context's own light directions (build_light_dirs)
			for (int k = 0; k < 9; ++k) { ld[k] = f(a[26 + k]); own = own && a[26 + k] == 0xffffffffu; }
			if (own) std::memcpy(ld, c->light_dirs, 36);
			ray_targets(F, X, tn, ta, ld, (int)a[5], tgt, lw);
			o[0] = u(tgt[0]); o[1] = u(tgt[1]); o[2] = u(tgt[2]); o[3] = u(tgt[3]); o[4] = u(lw[0]); o[5] = u(lw[1]); o[6] = u(lw[2]);
		} else if (kind == RNB_PRIM_LOSS_SAMPLE) {
			rnb_config F{};
			F.apply_no_albedo = a[0]; F.apply_rgbplus = a[1]; F.apply_L2 = a[2]; F.apply_relu = a[3];
			half_t oh[16];
			std::memcpy(oh, a + 4, 32);
			const float dt = f(a[12]);
			const Vec3 dir = {f(a[13]), f(a[14]), f(a[15])};
			const float light[3] = {f(a[16]), f(a[17]), f(a[18])}, grad[4] = {f(a[19]), f(a[20]), f(a[21]), f(a[22])}, rgb_ray[4] = {f(a[23]), f(a[24]), f(a[25]), f(a[26])};
			float rgb_ray2[4] = {f(a[27]), f(a[28]), f(a[29]), f(a[30])};
			const float weight_sum = f(a[31]);
			float weight_sum2 = f(a[32]), T = f(a[33]);
			const float gws = f(a[34]), loss_scale = f(a[35]);
			F.ek_loss_weight = f(a[36]);
			// the forward part of the loop body (testbed_nerf.cu:1866-1917), as loss_pass2 runs it
			orc_ctx_s tmp; tmp.cfg = F;
			float albedo[4];
			albedo_from_output(&tmp, oh, albedo);
			const AlphaTerms at = alpha_terms(oh, dt, dir, 1.0f);
			const float weight = at.alpha * T;
			float shading = esum3(at.g[0] * light[0], at.g[1] * light[1], at.g[2] * light[2]);
			if (F.apply_relu) shading = shading > 0.f ? shading : 0.f;
			for (int k = 0; k < 4; ++k) rgb_ray2[k] += weight * albedo[k] * shading;
			weight_sum2 += weight;
			T *= (1.f - at.alpha);
			half_t dl[16];
			float inter[10];
			(void)pass2_sample(F, grad, rgb_ray, weight_sum, gws, light, dir, loss_scale, oh, dt, albedo, at, shading, weight, T, weight_sum2, rgb_ray2, dl, inter);
			for (int k = 0; k < 10; ++k) o[18 + k] = u(inter[k]);
			o[0] = u(at.alpha); o[1] = u(T); o[2] = u(weight_sum2);
			for (int k = 0; k < 4; ++k) o[3 + k] = u(rgb_ray2[k]);
			for (int k = 0; k < 11; ++k) { uint16_t hb; std::memcpy(&hb, &dl[k], 2); o[7 + k] = hb; }
		} else if (kind == RNB_PRIM_RAY_LOSS) {
			rnb_config F{};
			F.apply_L2 = a[0]; F.apply_rgbplus = a[1]; F.apply_bce = a[2]; F.mask_loss_weight = f(a[3]);
			const float tgt[4] = {f(a[5]), f(a[6]), f(a[7]), f(a[8])}, ray[4] = {f(a[9]), f(a[10]), f(a[11]), f(a[12])};
			float grad[4], ws, gws, lrow, mrow;
			const float loss = pass2_ray_terms(F, tgt, ray, (float)(f(a[13]) > 0.99), (float)(f(a[14]) > 0.99), f(a[15]), (float)a[4], grad, &ws, &gws, &lrow, &mrow);
			o[0] = u(loss); o[1] = u(grad[0]); o[2] = u(grad[1]); o[3] = u(grad[2]); o[4] = u(grad[3]); o[5] = u(ws); o[6] = u(gws); o[7] = u(lrow); o[8] = u(mrow);
		} else if (kind == RNB_PRIM_ENCODE) {
			orc_ctx_s lv; // a one-level encoding over the item's own table
			lv.cfg.n_levels = 1; lv.valid_level = 0;
			lv.offsets[0] = 0; lv.offsets[1] = a[0]; lv.resolution[0] = a[1]; lv.scale[0] = f(a[2]);
			const float x3[3] = {f(a[3]), f(a[4]), f(a[5])};
			half_t feat[28];
			float dy[28][3];
			encode_sample(&lv, reinterpret_cast<const half_t*>(a + 6), x3, feat, dy);
			uint16_t h0, h1; std::memcpy(&h0, &feat[0], 2); std::memcpy(&h1, &feat[1], 2);
			o[0] = o[8] = h0; o[1] = o[9] = h1; // (the checker has one form of the encoding; the library's second form fills words 8-15)
			for (int d = 0; d < 3; ++d) { o[2 + d] = o[10 + d] = u(dy[0][d]); o[5 + d] = o[13 + d] = u(dy[1][d]); }
		} else if (kind == RNB_PRIM_MARCH_RAY) {
			orc_ctx_s mc; // the box, the cone angle and the bitfield the march reads from a context
			mc.aabb_min = f(a[0]); mc.aabb_max = f(a[1]); mc.cone_angle = f(a[2]);
			mc.bitfield = bf;
			RaySetup r; r.alive = true; r.o = {f(a[3]), f(a[4]), f(a[5])}; r.dir = {f(a[6]), f(a[7]), f(a[8])}; r.d_unnorm = r.dir; r.startt = f(a[9]);
			const Vec3 wd = warp_direction(r.dir);
			uint32_t chk = 0, last[7] = {0, 0, 0, 0, 0, 0, 0};
			for (int q = 0; q < 23; ++q) o[q] = 0u;
			const uint32_t n = march(&mc, r, RNB_MAX_STEPS, [&](uint32_t j, const Vec3& pos, float dt) {
				const Vec3 wp = warp_position(&mc, pos);
				const uint32_t c7[7] = {u(wp.x), u(wp.y), u(wp.z), u(warp_dt(dt)), u(wd.x), u(wd.y), u(wd.z)}; // a NerfCoordinate (nerf.h:76-104): position, dt, direction
				for (int q = 0; q < 7; ++q) { chk += c7[q]; last[q] = c7[q]; if (j < 2) o[2 + j * 7 + q] = c7[q]; }
			});
			o[0] = n; o[1] = chk;
			for (int q = 0; q < 7; ++q) o[16 + q] = last[q];
		} else if (kind == RNB_PRIM_SDF_DENSITY) {
			const uint16_t s16 = (uint16_t)a[0], v16 = (uint16_t)a[1];
			half_t sdf, var;
			std::memcpy(&sdf, &s16, 2); std::memcpy(&var, &v16, 2);
			const half_t d = sdf_to_density(sdf, var);
			uint16_t d16; std::memcpy(&d16, &d, 2);
			o[0] = d16;
		} else if (kind == RNB_PRIM_DW_SLICED) { // the weight-gradient GEMM of the half mode: emulated_dw itself, on the caller's operands
			const half_t* rows = reinterpret_cast<const half_t*>(a + 4);
			const bool ones = (a[0] & 1u) != 0u;
			for (uint32_t oo = 0; oo < 4; ++oo)
				for (uint32_t ii = 0; ii < 4; ++ii) {
					float v = 0.f;
					if (!ones) v = emulated_dw(DW_S, true, [&](uint32_t s) { return h2f(rows[(size_t)oo * DW_S + s]); }, [&](uint32_t s) { return h2f(rows[(size_t)(4 + ii) * DW_S + s]); });
					else if (oo == 0) v = emulated_dw(DW_S, true, [&](uint32_t) { return 1.0f; }, [&](uint32_t s) { return h2f(rows[(size_t)(4 + ii) * DW_S + s]); });
					o[oo * 4 + ii] = u(v);
				}
		} else if (kind == RNB_PRIM_PREP_DUE) {
			const uint32_t n_prep_to_skip = std::min(std::max(a[0] / 16u, 1u), 16u); // testbed.cu:2805, as the step driver below states it
			o[0] = a[0] % n_prep_to_skip == 0 ? 1u : 0u; o[1] = n_prep_to_skip;
		} else if (kind == RNB_PRIM_GRID) {
			float pos; uint32_t cell;
			pos_fract(f(a[5]), &pos, &cell, f(a[6]));
			const uint32_t pg[3] = {a[2], a[3], a[4]};
			o[0] = grid_entry(a[0], a[1], pg); o[1] = u(pos); o[2] = cell;
		} else {
			const float cone = f(a[0]);
			const uint32_t max_cascade = a[1];
			const Vec3 p = {f(a[2]), f(a[3]), f(a[4])}, d = {f(a[5]), f(a[6]), f(a[7])};
			const Vec3 idir = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
			const float t = f(a[8]);
			const float dt = calc_dt(t, cone);
			const int mip = mip_from_dt(dt, p, max_cascade);
			const uint32_t res = GRIDSIZE >> mip;
			o[0] = u(dt); o[1] = (uint32_t)mip_from_pos(p, max_cascade); o[2] = (uint32_t)mip; o[3] = cascaded_grid_idx_at(p, (uint32_t)mip);
			o[4] = density_grid_occupied_at(p, bf.data(), (uint32_t)mip) ? 1u : 0u;
			o[5] = u(distance_to_next_voxel(p, d, idir, res)); o[6] = u(advance_to_next_voxel(t, cone, p, d, idir, res));
		}
	}
	return RNB_OK;
}

int rnb_set_optimizer_step(orc_ctx_s* c, uint32_t step) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	c->opt_begun = false;
	c->optimizer_step_count = step;
	c->lr_factor = 1.0f;
	for (uint64_t s0 = c->cfg.lr_decay_start; s0 < step && s0 <= 10000000u; s0 += std::max(1u, c->cfg.lr_decay_interval)) c->lr_factor *= c->cfg.lr_decay_base; // exponential_decay.h:61-72, one factor per event
	return RNB_OK;
}

int rnb_set_controller(orc_ctx_s* c, uint32_t training_step, uint32_t rays_per_batch, uint32_t measured_before, uint32_t n_rays_total) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (rays_per_batch == 0 || rays_per_batch > c->cfg.max_rays_per_batch) return fail(RNB_ERR_INVALID, "rays_per_batch out of range");
	c->training_step = training_step;
	c->valid_level = compute_valid_level(c->cfg, (int)training_step);
	c->rays_per_batch = rays_per_batch;
	c->measured_batch_size_before_compaction = measured_before;
	c->n_rays_total = n_rays_total;
	return RNB_OK;
}

// The checker computes gradients in one piece.
int rnb_gradient_parts(orc_ctx_s* c, uint64_t ranges[3][2], uint32_t* n_parts) {
	if (!c || !ranges || !n_parts) return fail(RNB_ERR_INVALID, "null argument");
	ranges[0][0] = 0; ranges[0][1] = c->n_params; *n_parts = 1;
	return RNB_OK;
}
int rnb_shard_layout(orc_ctx_s* c, rnb_shard_part parts[RNB_MAX_SHARD_PARTS], uint32_t* n_parts, uint64_t* capacity) {
	if (!c || !parts || !n_parts || !capacity) return fail(RNB_ERR_INVALID, "null argument");
	shard_layout(c, parts, n_parts);
	*capacity = c->param_capacity;
	return RNB_OK;
}
int rnb_train_step_apply_shard(orc_ctx_s* c, uint32_t part, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	rnb_shard_part parts[RNB_MAX_SHARD_PARTS];
	uint32_t n = 0;
	shard_layout(c, parts, &n);
	if (part >= n) return fail(RNB_ERR_INVALID, "rnb_train_step_apply_shard: no such block");
	optimizer_begin(c);
	const rnb_shard_part& p = parts[part];
	std::fill(c->grads.begin() + p.lo, c->grads.begin() + p.own_lo, 0.f);
	std::fill(c->grads.begin() + p.own_hi, c->grads.begin() + p.hi, 0.f);
	if (c->cfg.accumulate == RNB_ACCUM_HALF) { std::fill(c->grads16.begin() + p.lo, c->grads16.begin() + p.own_lo, (half_t)0); std::fill(c->grads16.begin() + p.own_hi, c->grads16.begin() + p.hi, (half_t)0); }
	optimizer_range(c, std::min<uint64_t>(p.own_lo, c->n_params), std::min<uint64_t>(p.own_hi, c->n_params));
	return RNB_OK;
}
int rnb_train_step_apply_done(orc_ctx_s* c, void*) {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (!c->opt_begun) return fail(RNB_ERR_INVALID, "rnb_train_step_apply_done without rnb_train_step_apply_shard");
	c->opt_begun = false;
	++c->training_step;
	return RNB_OK;
}
int rnb_train_step_apply_early(orc_ctx_s* c, void*) { return c ? RNB_OK : fail(RNB_ERR_INVALID, "null ctx"); } // one block: nothing to do early
// (every gradient is final when rnb_train_step_begin returns: nothing to wait for, whichever block of rnb_gradient_parts / rnb_shard_layout is meant)
int rnb_gradient_part_wait(orc_ctx_s* c, uint32_t part, void*) { return (c && part < RNB_MAX_SHARD_PARTS) ? RNB_OK : fail(RNB_ERR_INVALID, "bad part"); }

} // extern "C"
