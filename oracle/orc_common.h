/*
 * oracle/orc_common.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar building blocks of the CPU restatement of the RNb-NeuS2 training hot
 * path: IEEE half emulation, PCG32, Morton codes. Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load the library built
 * from this directory; the product (rnb-neus2_amd/) never links or calls it.
 *
 * Parity status: PCG32 and Morton are pinned by known-answer vectors
 * (tests/golden/pcg32_kat.json: the published pcg32-demo vector and the draws
 * the reference's own pcg32.h produced for seed 1337). The
 * floating-point restatement of the network/loss path is UNPINNED: the
 * reference ships no tests, fixtures or CPU path for it and cannot be built
 * here (CUDA/WMMA/CUTLASS only).
 *
 * All citations are relative to /root/reference.
 */
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

#if defined(__F16C__)
#include <immintrin.h>
#endif

namespace orc {

// ---------------------------------------------------------------- half ----
// __half storage emulation with round-to-nearest-even, as CUDA's __float2half_rn.
typedef uint16_t half_t;

static inline float h2f_soft(half_t h) {
	uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1fu;
	uint32_t man = h & 0x3ffu;
	uint32_t bits;
	if (exp == 0) {
		if (man == 0) {
			bits = sign;
		} else {
			// subnormal: normalise
			int e = -1;
			do { ++e; man <<= 1; } while ((man & 0x400u) == 0);
			bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
		}
	} else if (exp == 31) {
		bits = sign | 0x7f800000u | man << 13;
	} else {
		bits = sign | (exp + 127 - 15) << 23 | man << 13;
	}
	float f; std::memcpy(&f, &bits, 4); return f;
}

static inline half_t f2h_soft(float f) {
	uint32_t x; std::memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t ax = x & 0x7fffffffu;
	if (ax >= 0x7f800000u) { // inf / nan
		return (half_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0));
	}
	if (ax >= 0x477ff000u) { // rounds to >= 65520 -> inf
		return (half_t)(sign | 0x7c00u);
	}
	if (ax < 0x33000001u) { // < 2^-25 (or == 2^-25 tie -> even = 0)
		return (half_t)sign;
	}
	int e = (int)(ax >> 23) - 127;
	uint32_t m = (ax & 0x7fffffu) | 0x800000u;
	int shift;
	uint32_t hexp;
	if (e < -14) { // subnormal result
		shift = 13 + (-14 - e);
		hexp = 0;
	} else {
		shift = 13;
		hexp = (uint32_t)(e + 15);
	}
	uint32_t hm = m >> shift;
	uint32_t rem = m & ((1u << shift) - 1u);
	uint32_t halfway = 1u << (shift - 1);
	if (rem > halfway || (rem == halfway && (hm & 1u))) ++hm;
	// hm includes the implicit bit for normals (0x400) -> add exponent accordingly
	uint32_t out;
	if (hexp == 0) out = hm;              // may carry into exponent 1: correct
	else out = ((hexp - 1) << 10) + hm;   // hm in [0x400, 0x800]
	return (half_t)(sign | out);
}

static inline float h2f(half_t h) {
#if defined(__F16C__)
	return _cvtsh_ss(h);
#else
	return h2f_soft(h);
#endif
}
static inline half_t f2h(float f) {
#if defined(__F16C__)
	return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
#else
	return f2h_soft(f);
#endif
}
// native half arithmetic (__hadd/__hmul/__hsub): one rounding of the exact result.
static inline half_t hadd(half_t a, half_t b) { return f2h(h2f(a) + h2f(b)); }
static inline half_t hsub(half_t a, half_t b) { return f2h(h2f(a) - h2f(b)); }
static inline half_t hmul(half_t a, half_t b) { return f2h(h2f(a) * h2f(b)); }
// round a float through half
static inline float rh(float f) { return h2f(f2h(f)); }

// --------------------------------------------------------------- pcg32 ----
// dependencies/neus2_tcnn/dependencies/pcg32/pcg32.h:44-170 (Wenzel Jakob's pcg32, O'Neill's PCG-XSH-RR)
struct Pcg32 {
	static constexpr uint64_t DEFAULT_STATE = 0x853c49e6748fea9bULL;
	static constexpr uint64_t DEFAULT_STREAM = 0xda3e39cb94b95bdbULL;
	static constexpr uint64_t MULT = 0x5851f42d4c957f2dULL;
	uint64_t state, inc;
	Pcg32() : state(DEFAULT_STATE), inc(DEFAULT_STREAM) {}
	explicit Pcg32(uint64_t initstate, uint64_t initseq = 1u) { seed(initstate, initseq); }
	void seed(uint64_t initstate, uint64_t initseq = 1u) { // pcg32.h:56-62
		state = 0U;
		inc = (initseq << 1u) | 1u;
		next_uint();
		state += initstate;
		next_uint();
	}
	uint32_t next_uint() { // pcg32.h:65-71
		uint64_t oldstate = state;
		state = oldstate * MULT + inc;
		uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
		uint32_t rot = (uint32_t)(oldstate >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	float next_float() { // pcg32.h:104-113
		union { uint32_t u; float f; } x;
		x.u = (next_uint() >> 9) | 0x3f800000u;
		return x.f - 1.0f;
	}
	void advance(int64_t delta_ = (1ll << 32)) { // pcg32.h:144-162
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

// -------------------------------------------------------------- morton ----
// dependencies/neus2_tcnn/include/tiny-cuda-nn/common_device.h:337-363
static inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
	return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249;
	x = (x | (x >> 2)) & 0xc30c30c3;
	x = (x | (x >> 4)) & 0x0f00f00f;
	x = (x | (x >> 8)) & 0xff0000ff;
	x = (x | (x >> 16)) & 0x0000ffff;
	return x;
}

static inline uint32_t next_multiple(uint32_t v, uint32_t m) { return ((v + m - 1) / m) * m; }

} // namespace orc
