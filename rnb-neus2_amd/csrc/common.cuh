// common.cuh — device-side building blocks shared by the gfx950 kernels of the RNb-NeuS2 hot path.
// Written for CDNA4 only: 64-lane wavefronts, v_mfma_f32_16x16x32_f16, 160 KB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rnb_neus2.h"

namespace rnb {

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr uint32_t GRIDSIZE = RNB_GRIDSIZE;
constexpr uint32_t GRID_CELLS = GRIDSIZE * GRIDSIZE * GRIDSIZE;
constexpr uint32_t N_CASCADES = RNB_CASCADES;
constexpr float LOSS_SCALE = 128.f;                   // testbed.h:237
constexpr uint32_t N_MAX_RANDOM_SAMPLES_PER_RAY = 8;  // testbed_nerf.cu:60
constexpr float SQRT3 = 1.73205080757f;               // testbed_nerf.cu:52
constexpr float STEPSIZE = SQRT3 / 1024;              // testbed_nerf.cu:53
constexpr float MIN_CONE_STEPSIZE = STEPSIZE;
constexpr float MAX_CONE_STEPSIZE = STEPSIZE * (1 << (N_CASCADES - 1)) * 1024 / GRIDSIZE;
constexpr float MIN_OPTICAL_THICKNESS = 0.1f;         // testbed_nerf.cu:66

// ---- hash-grid meta (grid.h:977-1012), passed by value as a kernel argument (lives in SGPRs) ----
struct GridMeta {
	uint32_t n_levels;
	uint32_t valid_level;                  // levels > valid_level emit zeros (grid.h:192-210)
	uint32_t offsets[RNB_MAX_LEVELS + 1];  // in entries (one entry = 2 halfs)
	uint32_t resolution[RNB_MAX_LEVELS];
	float scale[RNB_MAX_LEVELS];
};

// Parameter block pointers (nerf_network.h:539-583): [sdf mlp | rgb mlp | hash grid | variance]
struct NetW {
	const half_t* sdf_w0; // [64][32]
	const half_t* sdf_w1; // [16][64]
	const half_t* rgb_w0; // [64][48]
	const half_t* rgb_w1; // [64][64]
	const half_t* rgb_w2; // [16][64]
	const uint32_t* grid; // half2 per entry
	const half_t* variance;
};

// ---- PCG32 (dependencies/neus2_tcnn/dependencies/pcg32/pcg32.h:44-170) ----
struct Pcg32 {
	uint64_t state, inc;
	static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;
	__host__ __device__ Pcg32() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
	__host__ __device__ explicit Pcg32(uint64_t initstate, uint64_t initseq = 1u) { seed(initstate, initseq); }
	__host__ __device__ void seed(uint64_t initstate, uint64_t initseq = 1u) {
		state = 0U;
		inc = (initseq << 1u) | 1u;
		next_uint();
		state += initstate;
		next_uint();
	}
	__host__ __device__ uint32_t next_uint() {
		uint64_t oldstate = state;
		state = oldstate * MULT + inc;
		uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
		uint32_t rot = (uint32_t)(oldstate >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	__host__ __device__ float next_float() {
		union { uint32_t u; float f; } x;
		x.u = (next_uint() >> 9) | 0x3f800000u;
		return x.f - 1.0f;
	}
	__host__ __device__ void advance(int64_t delta_ = (1ll << 32)) {
		uint64_t cur_mult = MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		uint64_t delta = (uint64_t)delta_;
		while (delta > 0) {
			if (delta & 1) {
				acc_mult *= cur_mult;
				acc_plus = acc_plus * cur_mult + cur_plus;
			}
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta /= 2;
		}
		state = acc_mult * state + acc_plus;
	}
};

// ---- Morton (tiny-cuda-nn/common_device.h:337-363) ----
__host__ __device__ inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
__host__ __device__ inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
	return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__host__ __device__ inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249;
	x = (x | (x >> 2)) & 0xc30c30c3;
	x = (x | (x >> 4)) & 0x0f00f00f;
	x = (x | (x >> 8)) & 0xff0000ff;
	x = (x | (x >> 16)) & 0x0000ffff;
	return x;
}

// ---- half helpers ----
__device__ __forceinline__ float h2f(half_t h) { return (float)h; }
__device__ __forceinline__ half_t f2h(float f) { return (half_t)f; }
__device__ __forceinline__ float rh(float f) { return (float)(half_t)f; }
__device__ __forceinline__ uint32_t pack_h2(half_t a, half_t b) {
	h2 v = {a, b};
	return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ h2 unpack_h2(uint32_t u) { return __builtin_bit_cast(h2, u); }

// Orders LDS traffic between the lanes of ONE wavefront (per-wave private tiles: no s_barrier needed;
// the LDS executes a wave's DS operations in issue order).
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct __attribute__((packed, aligned(4))) U2 { uint32_t x, y; }; // 8-byte gather at 4-byte alignment (global_load_dwordx2)

// ---- hash-grid indexing (grid.h:113-148; N_DIMS = 3, F = 2, GridType::Hash) ----
__device__ __forceinline__ uint32_t grid_entry(uint32_t hashmap_size, uint32_t res, uint32_t px, uint32_t py, uint32_t pz) {
	// The stride loop of grid.h:137-141, unrolled; all conditions are wave-uniform (per level).
	uint32_t index = px;
	uint32_t stride = res;
	if (stride <= hashmap_size) {
		index += py * stride;
		stride *= res;
		if (stride <= hashmap_size) {
			index += pz * stride;
			stride *= res;
		}
	}
	if (hashmap_size < stride) {
		index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
	}
	// index % hashmap_size without an integer division: hashed levels have a power-of-two table; dense
	// levels hold >= res^3 entries while index < res + res^2 + res^3 < 2*res^3 (res >= 2).
	if ((hashmap_size & (hashmap_size - 1u)) == 0u) return index & (hashmap_size - 1u);
	return index >= hashmap_size ? index - hashmap_size : index;
}

// common_device.h:403-434 (fork: pos = x*scale + 0.5), linear interpolation.
__device__ __forceinline__ void pos_fract(float input, float scale, float* pos, uint32_t* pos_grid) {
	float p = input * scale + 0.5f;
	float fl = floorf(p);
	*pos_grid = (uint32_t)(int)fl;
	*pos = p - fl;
}

// One level of kernel_grid (grid.h:237-363): the 8 corners are gathered once and reused for the
// feature (accumulated in half, grid.h:310-313) and for dy/dx (fp32, grid.h:324-363).
template <bool GRAD>
__device__ __forceinline__ void encode_level_core(const uint32_t* __restrict__ g, const uint32_t hashmap_size, const uint32_t res, const float scale,
                                                  const float x, const float y, const float z,
                                                  half_t& f0, half_t& f1, float dy0[3], float dy1[3]) {
	float pos[3];
	uint32_t pg[3];
	pos_fract(x, scale, &pos[0], &pg[0]);
	pos_fract(y, scale, &pos[1], &pg[1]);
	pos_fract(z, scale, &pos[2], &pg[2]);
	// The x-neighbour of a cell is the next table entry on dense levels and, on hashed levels, whenever x is even
	// (hash prime 1): one 8-byte gather then serves both corners. Random gathers are bound by the number of lane
	// accesses the L1 processes AND by the bytes returned: measured alternatives that lost — 16-byte aligned block gathers
	// (fewer accesses, +18 % time), unconditional second gathers (+10 %), a single wave-level branch around the four
	// second gathers (+5 % on the one-wave-per-SIMD kernels). (Entry `size` is readable: the parameter block continues
	// past every level.)
	// The eight entries (grid_entry's values), formed per LEVEL rather than per corner (round 4): whether the level is dense or hashed and whether its table is a power
	// of two is wave-uniform, and (y + 1) * c = y * c + c in uint32 arithmetic, so a level costs two integer multiplies instead of up to 32 (v_mul_lo_u32 issues at a
	// quarter of the rate: they were a fifth of k_forward_chained's issue cycles, 426 of 7 678 instructions at four times the cost).
	uint32_t stride = res;
	bool dense = false;
	if (stride <= hashmap_size) { stride *= res; if (stride <= hashmap_size) { const uint32_t s2 = stride; stride *= res; dense = !(hashmap_size < stride); stride = s2; } }
	const bool pow2 = (hashmap_size & (hashmap_size - 1u)) == 0u;
	uint32_t ty[2], tz[2];
	if (dense) { ty[0] = pg[1] * res; ty[1] = ty[0] + res; tz[0] = pg[2] * stride; tz[1] = tz[0] + stride; }       // index = x + y res + z res^2
	else { ty[0] = pg[1] * 2654435761u; ty[1] = ty[0] + 2654435761u; tz[0] = pg[2] * 805459861u; tz[1] = tz[0] + 805459861u; } // index = x ^ y P1 ^ z P2
	uint32_t v[8];
#pragma unroll
	for (uint32_t yz = 0; yz < 4; ++yz) {
		uint32_t e0, e1;
		if (dense) { e0 = pg[0] + ty[yz & 1u] + tz[yz >> 1]; e1 = e0 + 1u; }
		else { const uint32_t h = ty[yz & 1u] ^ tz[yz >> 1]; e0 = pg[0] ^ h; e1 = (pg[0] + 1u) ^ h; }
		if (pow2) { e0 &= hashmap_size - 1u; e1 &= hashmap_size - 1u; }
		else { e0 = e0 >= hashmap_size ? e0 - hashmap_size : e0; e1 = e1 >= hashmap_size ? e1 - hashmap_size : e1; }
		const U2 p = *reinterpret_cast<const U2*>(g + e0);
		uint32_t v1 = p.y;
		if (e1 != e0 + 1u) v1 = g[e1];
		v[yz * 2 + 0] = p.x;
		v[yz * 2 + 1] = v1;
	}
	half_t r0 = (half_t)0.f, r1 = (half_t)0.f;
#pragma unroll
	for (uint32_t idx = 0; idx < 8; ++idx) {
		float weight = 1;
#pragma unroll
		for (uint32_t d = 0; d < 3; ++d) {
			if ((idx & (1u << d)) == 0) weight *= 1 - pos[d];
			else weight *= pos[d];
		}
		const h2 val = unpack_h2(v[idx]);
		r0 = r0 + f2h(weight * h2f(val[0]));
		r1 = r1 + f2h(weight * h2f(val[1]));
	}
	f0 = r0;
	f1 = r1;
	if (GRAD) {
#pragma unroll
		for (uint32_t gd = 0; gd < 3; ++gd) {
			float a0 = 0.f, a1 = 0.f;
#pragma unroll
			for (uint32_t idx = 0; idx < 4; ++idx) {
				float weight = scale;
				uint32_t corner = 0;
#pragma unroll
				for (uint32_t ngd = 0; ngd < 2; ++ngd) {
					const uint32_t d = ngd >= gd ? (ngd + 1) : ngd;
					if ((idx & (1u << ngd)) == 0) weight *= 1 - pos[d];
					else { weight *= pos[d]; corner |= (1u << d); }
				}
				const h2 vl = unpack_h2(v[corner]);
				const h2 vr = unpack_h2(v[corner | (1u << gd)]);
				a0 += weight * (h2f(vr[0]) - h2f(vl[0])) * 1.0f;
				a1 += weight * (h2f(vr[1]) - h2f(vl[1])) * 1.0f;
			}
			dy0[gd] = a0;
			dy1[gd] = a1;
		}
	}
}

template <bool GRAD>
__device__ __forceinline__ void encode_level(const GridMeta& G, const uint32_t* __restrict__ grid, const uint32_t level,
                                             const float x, const float y, const float z,
                                             half_t& f0, half_t& f1, float dy0[3], float dy1[3]) {
	encode_level_core<GRAD>(grid + G.offsets[level], G.offsets[level + 1] - G.offsets[level], G.resolution[level], G.scale[level], x, y, z, f0, f1, dy0, dy1);
}

// Per-level constants staged in LDS: 42 kernel-argument SGPRs that every unrolled level keeps alive would otherwise be
// spilled to VGPR lanes (v_readlane / v_writelane traffic in the hot loop). One broadcast ds_read_b128 + 4 readfirstlane.
// (round 5) + what level_issue needs to form the table indices without a branch: the y / z multipliers of the index (dense: res, res^2; hashed: the two primes) and
// whether the level is dense / its table a power of two -- grid_entry's wave-uniform decisions, taken once per workgroup instead of once per level and sample.
struct __attribute__((aligned(16))) LevelMeta { uint32_t offset, size, res; float scale; uint32_t my, mz, dense, pow2; };

__device__ __forceinline__ void fill_level_meta(LevelMeta* __restrict__ lm, const GridMeta& G, const int tid) {
	if (tid < RNB_MAX_LEVELS) {
		LevelMeta m;
		m.offset = G.offsets[tid]; m.size = G.offsets[tid + 1] - G.offsets[tid]; m.res = G.resolution[tid]; m.scale = G.scale[tid];
		uint32_t stride = m.res, s2 = m.res;
		bool dense = false;
		if (stride <= m.size) { stride *= m.res; if (stride <= m.size) { s2 = stride; stride *= m.res; dense = !(m.size < stride); } } // (the stride loop of grid.h:137-141, as encode_level_core walks it)
		m.dense = dense ? 1u : 0u;
		m.pow2 = (m.size & (m.size - 1u)) == 0u ? 1u : 0u;
		m.my = dense ? m.res : 2654435761u;
		m.mz = dense ? s2 : 805459861u;
		lm[tid] = m;
	}
}

template <bool GRAD>
__device__ __forceinline__ void encode_level_lm(const LevelMeta* __restrict__ lm, const uint32_t* __restrict__ grid, const uint32_t level,
                                                const float x, const float y, const float z,
                                                half_t& f0, half_t& f1, float dy0[3], float dy1[3]) {
	const uint4 raw = *reinterpret_cast<const uint4*>(lm + level);
	const uint32_t off = __builtin_amdgcn_readfirstlane(raw.x), size = __builtin_amdgcn_readfirstlane(raw.y), res = __builtin_amdgcn_readfirstlane(raw.z);
	const float scale = __uint_as_float(__builtin_amdgcn_readfirstlane(raw.w));
	encode_level_core<GRAD>(grid + off, size, res, scale, x, y, z, f0, f1, dy0, dy1);
}

// ---- the same level in two halves, for encodes that keep the gathers of several levels in flight (round 5) ----
// encode_level_core issues a level's eight 4-byte gathers and consumes them at once, behind wave-uniform BRANCHES (dense / hashed, power-of-two table, the second
// gather of an x-pair): every level is a scheduling region of its own and ends in s_waitcnt vmcnt(0), so a wavefront walks its 14 levels as 14 dependent round trips
// with eight loads in flight -- at two wavefronts per SIMD the network evaluations were bound by exactly that chain (profiles/r05_*: 41 % of the cycles waiting).
// level_issue computes the eight table indices WITHOUT branches (the uniform conditions become selects) and issues the eight loads; level_consume recomputes the
// interpolation weights (a few VALU instructions) and reduces the eight values: the same expressions on the same operands as encode_level_core, hence the same bits.
// A caller issues DEPTH levels ahead of the one it consumes (vmcnt counts in order: consuming level l waits until at most 8 (DEPTH - 1) loads are outstanding).
// PAIR (round 6): the caller promises that `level` is a DENSE level (checked on the host: the kernels are instantiated for the number of leading dense levels of the configuration): the x-neighbour of a
// corner is then the next table entry, and one 8-byte gather serves both x-corners -- four gathers per level instead of eight (the wrapped index of the neighbour differs from e0 + 1 only when e0 is the
// table's last entry: that lane takes the table's first entry, read once through a wave-uniform address). The same eight values, hence the same bits.
template <bool PAIR = false>
__device__ __forceinline__ void level_issue(const LevelMeta* __restrict__ lm, const uint32_t* __restrict__ grid, const uint32_t level, const float x, const float y, const float z, uint32_t (&v)[8]) {
	const uint4 raw = reinterpret_cast<const uint4*>(lm + level)[0], raw2 = reinterpret_cast<const uint4*>(lm + level)[1];
	const uint32_t off = __builtin_amdgcn_readfirstlane(raw.x), hashmap_size = __builtin_amdgcn_readfirstlane(raw.y);
	const float scale = __uint_as_float(__builtin_amdgcn_readfirstlane(raw.w));
	const uint32_t my = __builtin_amdgcn_readfirstlane(raw2.x), mz = __builtin_amdgcn_readfirstlane(raw2.y);
	const bool dense = __builtin_amdgcn_readfirstlane(raw2.z) != 0u, pow2 = __builtin_amdgcn_readfirstlane(raw2.w) != 0u;
	const uint32_t* __restrict__ g = grid + off;
	float pos[3];
	uint32_t pg[3];
	pos_fract(x, scale, &pos[0], &pg[0]);
	pos_fract(y, scale, &pos[1], &pg[1]);
	pos_fract(z, scale, &pos[2], &pg[2]);
	// grid_entry's value per corner (see encode_level_core): dense: index = x + y res + z res^2, else the hash; wrapped into the table. Selects, no branches.
	const uint32_t ty0 = pg[1] * my, ty1 = ty0 + my, tz0 = pg[2] * mz, tz1 = tz0 + mz;
#pragma unroll
	for (uint32_t yz = 0; yz < 4; ++yz) {
		const uint32_t ty = (yz & 1u) ? ty1 : ty0, tz = (yz >> 1) ? tz1 : tz0;
		const uint32_t d0 = pg[0] + ty + tz, h = ty ^ tz;
		uint32_t e0 = (PAIR || dense) ? d0 : (pg[0] ^ h), e1 = (PAIR || dense) ? d0 + 1u : ((pg[0] + 1u) ^ h);
		const uint32_t w0 = e0 >= hashmap_size ? e0 - hashmap_size : e0, w1 = e1 >= hashmap_size ? e1 - hashmap_size : e1;
		e0 = pow2 ? (e0 & (hashmap_size - 1u)) : w0;
		e1 = pow2 ? (e1 & (hashmap_size - 1u)) : w1;
		if (PAIR) {
			const U2 p = *reinterpret_cast<const U2*>(g + e0);
			v[yz * 2 + 0] = p.x;
			v[yz * 2 + 1] = e1 == e0 + 1u ? p.y : g[0];
		} else {
			v[yz * 2 + 0] = g[e0];
			v[yz * 2 + 1] = g[e1];
		}
	}
}
template <bool GRAD>
__device__ __forceinline__ void level_consume(const LevelMeta* __restrict__ lm, const uint32_t level, const float x, const float y, const float z, const uint32_t (&v)[8],
                                              half_t& f0, half_t& f1, float dy0[3], float dy1[3]) {
	const float scale = __uint_as_float(__builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t*>(lm + level)[3]));
	float pos[3];
	uint32_t pg[3];
	pos_fract(x, scale, &pos[0], &pg[0]);
	pos_fract(y, scale, &pos[1], &pg[1]);
	pos_fract(z, scale, &pos[2], &pg[2]);
	half_t r0 = (half_t)0.f, r1 = (half_t)0.f;
#pragma unroll
	for (uint32_t idx = 0; idx < 8; ++idx) {
		float weight = 1;
#pragma unroll
		for (uint32_t d = 0; d < 3; ++d) {
			if ((idx & (1u << d)) == 0) weight *= 1 - pos[d];
			else weight *= pos[d];
		}
		const h2 val = unpack_h2(v[idx]);
		r0 = r0 + f2h(weight * h2f(val[0]));
		r1 = r1 + f2h(weight * h2f(val[1]));
	}
	f0 = r0;
	f1 = r1;
	if (GRAD) {
#pragma unroll
		for (uint32_t gd = 0; gd < 3; ++gd) {
			float a0 = 0.f, a1 = 0.f;
#pragma unroll
			for (uint32_t idx = 0; idx < 4; ++idx) {
				float weight = scale;
				uint32_t corner = 0;
#pragma unroll
				for (uint32_t ngd = 0; ngd < 2; ++ngd) {
					const uint32_t d = ngd >= gd ? (ngd + 1) : ngd;
					if ((idx & (1u << ngd)) == 0) weight *= 1 - pos[d];
					else { weight *= pos[d]; corner |= (1u << d); }
				}
				const h2 vl = unpack_h2(v[corner]);
				const h2 vr = unpack_h2(v[corner | (1u << gd)]);
				a0 += weight * (h2f(vr[0]) - h2f(vl[0])) * 1.0f;
				a1 += weight * (h2f(vr[1]) - h2f(vl[1])) * 1.0f;
			}
			dy0[gd] = a0;
			dy1[gd] = a1;
		}
	}
}

// ---- occupancy helpers (src/testbed_nerf.cu:439-475, 569-583, 153-155, 301-323, 429-437) ----
struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 v3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
// Eigen's reduction of a fixed-size vector splits the range in halves (Redux.h, redux_novec_unroller; no SIMD under a GPU compiler): three terms are x0 + (x1 + x2), four
// (x0 + x1) + (x2 + x3). That is what .dot(), .norm(), .normalized() and the fixed-size matrix products of the reference's kernels evaluate (tests/golden/float_fixtures.json
// runs the kernel's own statements through its vendored Eigen); sums the reference spells out term by term stay left to right.
__device__ __forceinline__ float esum3(float x0, float x1, float x2) { return x0 + (x1 + x2); }
__device__ __forceinline__ float esum4(float x0, float x1, float x2, float x3) { return (x0 + x1) + (x2 + x3); }
__device__ __forceinline__ float dot(Vec3 a, Vec3 b) { return esum3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ Vec3 normalized(Vec3 a) { float n = sqrtf(dot(a, a)); return {a.x / n, a.y / n, a.z / n}; }

__device__ __forceinline__ int mip_from_pos(const Vec3& pos, uint32_t max_cascade = N_CASCADES - 1) {
	int exponent;
	float maxval = fmaxf(fmaxf(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	frexpf(maxval, &exponent);
	return min((int)max_cascade, max(0, exponent + 1));
}
__device__ __forceinline__ int mip_from_dt(float dt, const Vec3& pos, uint32_t max_cascade = N_CASCADES - 1) {
	int mip = mip_from_pos(pos, max_cascade);
	dt *= 2 * GRIDSIZE;
	if (dt < 1.f) return mip;
	int exponent;
	frexpf(dt, &exponent);
	return min((int)max_cascade, max(exponent, mip));
}
__device__ __forceinline__ uint32_t cascaded_grid_idx_at(Vec3 pos, uint32_t mip) {
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos = pos - v3(0.5f, 0.5f, 0.5f);
	pos = mip_scale * pos;
	pos = pos + v3(0.5f, 0.5f, 0.5f);
	int ix = (int)(pos.x * GRIDSIZE), iy = (int)(pos.y * GRIDSIZE), iz = (int)(pos.z * GRIDSIZE);
	return morton3D((uint32_t)min(max(ix, 0), (int)GRIDSIZE - 1), (uint32_t)min(max(iy, 0), (int)GRIDSIZE - 1), (uint32_t)min(max(iz, 0), (int)GRIDSIZE - 1));
}
__device__ __forceinline__ bool density_grid_occupied_at(const Vec3& pos, const uint8_t* __restrict__ bitfield, uint32_t mip) {
	uint32_t idx = cascaded_grid_idx_at(pos, mip);
	return bitfield[idx / 8 + (GRID_CELLS * mip) / 8] & (1 << (idx % 8));
}
__device__ __forceinline__ float calc_dt(float t, float cone_angle) {
	return fminf(fmaxf(t * cone_angle, MIN_CONE_STEPSIZE), MAX_CONE_STEPSIZE);
}
__device__ __forceinline__ float warp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE * (1 << (N_CASCADES - 1));
	return (dt - MIN_CONE_STEPSIZE) / (max_stepsize - MIN_CONE_STEPSIZE);
}
__device__ __forceinline__ float unwarp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE * (1 << (N_CASCADES - 1));
	return dt * (max_stepsize - MIN_CONE_STEPSIZE) + MIN_CONE_STEPSIZE;
}
__device__ __forceinline__ float sign1(float x) { return copysignf(1.0f, x); }
__device__ __forceinline__ float distance_to_next_voxel(const Vec3& pos, const Vec3& dir, const Vec3& idir, uint32_t res) {
	Vec3 p = (float)res * pos;
	float tx = (floorf(p.x + 0.5f + 0.5f * sign1(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * sign1(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * sign1(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t / res, 0.0f);
}
__device__ __forceinline__ float advance_to_next_voxel(float t, float cone_angle, const Vec3& pos, const Vec3& dir, const Vec3& idir, uint32_t res) {
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	do { t += calc_dt(t, cone_angle); } while (t < t_target);
	return t;
}

__device__ __forceinline__ float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float relu(float v) { return v > 0.0f ? v : 0.0f; } // activation_function(.., ReLU), testbed_nerf.cu:326-335
__device__ __forceinline__ Vec3 warp_direction(const Vec3& d) { return {(d.x + 1.0f) * 0.5f, (d.y + 1.0f) * 0.5f, (d.z + 1.0f) * 0.5f}; }     // testbed_nerf.cu:413-415
__device__ __forceinline__ Vec3 unwarp_direction(const Vec3& d) { return {d.x * 2.0f - 1.0f, d.y * 2.0f - 1.0f, d.z * 2.0f - 1.0f}; }         // testbed_nerf.cu:417-419
// The ray loss of testbed_nerf.cu:280-299, 1389-1394 on the composited rgb+ 4-vector: L2 = sum of squares, gradient 2 d; L1 = sum of |d|, gradient copysign(1, d). d = prediction - target.
__device__ __forceinline__ float loss_and_gradient(const bool l2, const float (&target)[4], const float (&prediction)[4], float (&grad)[4]) {
	float diff[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) diff[k] = prediction[k] - target[k];
	if (l2) {
#pragma unroll
		for (int k = 0; k < 4; ++k) grad[k] = 2 * diff[k];
		return diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2] + diff[3] * diff[3];
	}
#pragma unroll
	for (int k = 0; k < 4; ++k) grad[k] = copysignf(1.0f, diff[k]);
	return fabsf(diff[0]) + fabsf(diff[1]) + fabsf(diff[2]) + fabsf(diff[3]);
}

} // namespace rnb
