"""Writes tests/golden/int_fixtures.json: outputs of the host-compilable fragments of the REFERENCE for the integer / index arithmetic of the
hot path, produced by compiling those fragments -- read from /root/reference at run time, never copied into this
repository -- with g++ in the build container and running them on seeded inputs. The tests read only the JSON.

  python tests/golden/make_int_fixtures.py

What is taken, verbatim, from the reference's files (function or struct body located by its signature, braces matched):
  dependencies/neus2_tcnn/dependencies/pcg32/pcg32.h                 struct pcg32                         (RNG streams: pixel, jitter, light, init)
  dependencies/neus2_tcnn/include/tiny-cuda-nn/common_device.h       expand_bits, morton3D, morton3D_invert
  dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h              clamp, host_device_swap
  include/neural-graphics-primitives/common.h                       sign
  include/neural-graphics-primitives/common_device.cuh               srgb_to_linear, linear_to_srgb (scalar)
  include/neural-graphics-primitives/bounding_box.cuh                BoundingBox::ray_intersect, ::contains
  include/neural-graphics-primitives/nerf.h                          NERF_GRIDSIZE
  src/testbed_nerf.cu                                                NERF_STEPS .. MAX_CONE_STEPSIZE, grid_mip_offset, calc_dt, distance_to_next_voxel,
                                                                     advance_to_next_voxel, cascaded_grid_idx_at, density_grid_occupied_at, mip_from_pos, mip_from_dt
The CUDA decorations are defined away (__host__, __device__, TCNN_HOST_DEVICE, NGP_HOST_DEVICE) and Eigen comes from the reference's vendored
dependencies/eigen. Floating-point results are stored as bit patterns; host libm stands in for the device's (floorf, frexpf, scalbnf, copysignf are
exact; pow is not -- the sRGB vectors are compared with an ulp tolerance on the GPU, bit for bit against the CPU checker, which calls the same libm).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _block(src, start):
    """Text from `start` through the brace that closes the first '{' at or after it."""
    i = src.index("{", start)
    depth, j = 0, i
    while True:
        c = src[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return src[start:j + 1]
        j += 1


def fragment(path, signature, occurrence=0):
    src = open(os.path.join(REF, path)).read()
    pos = -1
    for _ in range(occurrence + 1):
        pos = src.index(signature, pos + 1)
    return _block(src, pos)


def build_program():
    f = fragment
    tn = "src/testbed_nerf.cu"
    cd = "dependencies/neus2_tcnn/include/tiny-cuda-nn/common_device.h"
    parts = ["""
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <limits>
#include <algorithm>
#include <Eigen/Dense>
#define __host__
#define __device__
#define __restrict__
#define TCNN_HOST_DEVICE
#define NGP_HOST_DEVICE
// CUDA's overload set of min / max for the argument types these fragments use (math_functions.hpp: mixed signedness compares as unsigned)
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
static inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
#define PCG32_DEFAULT_STATE  0x853c49e6748fea9bULL
#define PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define PCG32_MULT           0x5851f42d4c957f2dULL
namespace tcnn {
template <typename T> """ + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "TCNN_HOST_DEVICE T clamp(T val, T lower, T upper)"),
             "template <typename T> " + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "TCNN_HOST_DEVICE void host_device_swap(T& a, T& b)"),
             f("dependencies/neus2_tcnn/dependencies/pcg32/pcg32.h", "struct pcg32 {") + ";",
             f(cd, "__host__ __device__ inline uint32_t expand_bits(uint32_t v)"),
             f(cd, "__host__ __device__ inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z)"),
             f(cd, "__host__ __device__ inline uint32_t morton3D_invert(uint32_t x)"),
             "}\nusing namespace Eigen;",
             f("include/neural-graphics-primitives/common.h", "inline NGP_HOST_DEVICE float sign(float x)"),
             f("include/neural-graphics-primitives/common_device.cuh", "inline __host__ __device__ float srgb_to_linear(float srgb)"),
             f("include/neural-graphics-primitives/common_device.cuh", "inline __host__ __device__ float linear_to_srgb(float linear)"),
             f("include/neural-graphics-primitives/nerf.h", "inline constexpr __device__ uint32_t NERF_GRIDSIZE()")]
    for sig in ("inline constexpr __device__ uint32_t NERF_STEPS()", "inline constexpr __device__ uint32_t NERF_CASCADES()", "inline constexpr __device__ float SQRT3()",
                "inline constexpr __device__ float STEPSIZE()", "inline constexpr __device__ float MIN_CONE_STEPSIZE()", "inline constexpr __device__ float MAX_CONE_STEPSIZE()",
                "inline __host__ __device__ uint32_t grid_mip_offset(uint32_t mip)", "inline __host__ __device__ float calc_dt(float t, float cone_angle)",
                "inline __device__ float distance_to_next_voxel(", "inline __device__ float advance_to_next_voxel(", "__device__ uint32_t cascaded_grid_idx_at(Vector3f pos, uint32_t mip)",
                "__device__ bool density_grid_occupied_at(", "inline __device__ int mip_from_pos(", "inline __device__ int mip_from_dt("):
        parts.append(f(tn, sig))
    bb = "include/neural-graphics-primitives/bounding_box.cuh"
    parts.append("struct BoundingBox { Eigen::Vector3f min, max;\n" + f(bb, "NGP_HOST_DEVICE Eigen::Vector2f ray_intersect(") + "\n" + f(bb, "NGP_HOST_DEVICE bool contains(const Eigen::Vector3f& p) const") + "};")
    parts.append(r"""
static uint32_t fb(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static void arr_u(const char* name, const uint32_t* v, size_t n, bool last = false) { printf("\"%s\": [", name); for (size_t i = 0; i < n; ++i) printf("%u%s", v[i], i + 1 < n ? "," : ""); printf("]%s\n", last ? "" : ","); }
int main() {
	printf("{\n");
	{ // ---- pcg32: draws, floats, advance (the march jumps (ray * 8), the trainer (n params), the grid samples (i * 4))
		std::vector<uint32_t> out;
		const uint64_t seeds[4] = {1337, 42, 0, 0xdeadbeefcafeULL};
		for (uint64_t s : seeds) { tcnn::pcg32 r{s}; for (int k = 0; k < 6; ++k) out.push_back(r.next_uint()); }
		arr_u("pcg32_next_uint_seeds_1337_42_0_deadbeefcafe_x6", out.data(), out.size());
		out.clear();
		{ tcnn::pcg32 r{1337}; for (int k = 0; k < 8; ++k) out.push_back(fb(r.next_float())); }
		arr_u("pcg32_next_float_bits_seed_1337_x8", out.data(), out.size());
		out.clear();
		const int64_t deltas[8] = {1, 8, 800, 100000 * 8, 262143ll * 8, 1ll << 32, 10548128, -5};
		for (int64_t d : deltas) { tcnn::pcg32 r{1337}; r.advance(d); out.push_back((uint32_t)(d >> 32)); out.push_back((uint32_t)d); out.push_back((uint32_t)(r.state >> 32)); out.push_back((uint32_t)r.state); out.push_back(r.next_uint()); }
		arr_u("pcg32_advance_seed_1337_deltahi_deltalo_statehi_statelo_next", out.data(), out.size());
		out.clear();
		{ tcnn::pcg32 r{1337, 54}; for (int k = 0; k < 4; ++k) out.push_back(r.next_uint()); }
		arr_u("pcg32_seed_1337_seq_54_x4", out.data(), out.size());
	}
	{ // ---- Morton codes
		std::vector<uint32_t> out;
		tcnn::pcg32 r{7};
		for (int k = 0; k < 64; ++k) {
			const uint32_t x = r.next_uint() & 1023u, y = r.next_uint() & 1023u, z = r.next_uint() & 1023u;
			const uint32_t m = tcnn::morton3D(x, y, z);
			out.push_back(x); out.push_back(y); out.push_back(z); out.push_back(m);
			out.push_back(tcnn::morton3D_invert(m)); out.push_back(tcnn::morton3D_invert(m >> 1)); out.push_back(tcnn::morton3D_invert(m >> 2));
		}
		arr_u("morton_x_y_z_code_ix_iy_iz", out.data(), out.size());
	}
	{ // ---- sRGB transfer on all the 16-bit codes the pixel decode can meet at a stride, and the inverse
		std::vector<uint32_t> out;
		for (uint32_t v = 0; v < 65536; v += 257) { const float s = (float)v * (1.0f / 65535.0f); out.push_back(v); out.push_back(fb(srgb_to_linear(s))); out.push_back(fb(linear_to_srgb(s))); }
		arr_u("srgb_code_tolinear_bits_tosrgb_bits", out.data(), out.size());
	}
	{ // ---- ray / box: the unit box of every RNb scene and a box of aabb_scale 4, rays from outside, inside, parallel to a face, missing
		std::vector<uint32_t> out;
		tcnn::pcg32 r{11};
		for (int k = 0; k < 96; ++k) {
			BoundingBox b;
			const float lo = (k & 1) ? -1.5f : 0.0f, hi = (k & 1) ? 2.5f : 1.0f;
			b.min = Eigen::Vector3f::Constant(lo); b.max = Eigen::Vector3f::Constant(hi);
			Eigen::Vector3f o(r.next_float() * 6 - 2.5f, r.next_float() * 6 - 2.5f, r.next_float() * 6 - 2.5f);
			Eigen::Vector3f d(r.next_float() * 2 - 1, r.next_float() * 2 - 1, r.next_float() * 2 - 1);
			if (k % 8 == 7) d.x() = 0.0f;            // parallel to a face: division by zero, inf arithmetic
			if (k % 16 == 3) o = Eigen::Vector3f(0.5f, 0.25f, 0.75f); // origin inside
			d.normalize();
			const Eigen::Vector2f t = b.ray_intersect(o, d);
			out.push_back(fb(lo)); out.push_back(fb(hi));
			for (int q = 0; q < 3; ++q) out.push_back(fb(o[q]));
			for (int q = 0; q < 3; ++q) out.push_back(fb(d[q]));
			out.push_back(fb(t.x())); out.push_back(fb(t.y())); out.push_back(b.contains(o) ? 1u : 0u);
		}
		arr_u("ray_box_lo_hi_o3_d3_tmin_tmax_contains", out.data(), out.size());
	}
	{ // ---- march helpers: dt, mip, cell index, voxel stepping (single cascade cone 0 and aabb_scale 4 cone 1/256)
		std::vector<uint32_t> out;
		tcnn::pcg32 r{23};
		std::vector<uint8_t> bitfield(128 * 128 * 128 / 8 * 8);
		{ tcnn::pcg32 q{5}; for (auto& b : bitfield) b = (uint8_t)(q.next_uint() >> 24); }
		for (int k = 0; k < 128; ++k) {
			const bool wide = (k & 1) != 0;
			const float cone = wide ? 1.0f / 256.0f : 0.0f;
			const uint32_t max_cascade = wide ? 2u : 0u;
			Eigen::Vector3f p(r.next_float(), r.next_float(), r.next_float());
			if (wide) p = (p - Eigen::Vector3f::Constant(0.5f)) * 3.9f + Eigen::Vector3f::Constant(0.5f);
			Eigen::Vector3f d(r.next_float() * 2 - 1, r.next_float() * 2 - 1, r.next_float() * 2 - 1);
			d.normalize();
			const Eigen::Vector3f idir = d.cwiseInverse();
			const float t = r.next_float() * 4.0f;
			const float dt = calc_dt(t, cone);
			const int mip = mip_from_dt(dt, p, max_cascade);
			const uint32_t res = NERF_GRIDSIZE() >> mip;
			const uint32_t idx = cascaded_grid_idx_at(p, (uint32_t)mip);
			out.push_back(fb(cone)); out.push_back(max_cascade);
			for (int q = 0; q < 3; ++q) out.push_back(fb(p[q]));
			for (int q = 0; q < 3; ++q) out.push_back(fb(d[q]));
			out.push_back(fb(t)); out.push_back(fb(dt)); out.push_back((uint32_t)mip_from_pos(p, max_cascade)); out.push_back((uint32_t)mip); out.push_back(idx);
			out.push_back(density_grid_occupied_at(p, bitfield.data(), (uint32_t)mip) ? 1u : 0u);
			out.push_back(fb(distance_to_next_voxel(p, d, idir, res))); out.push_back(fb(advance_to_next_voxel(t, cone, p, d, idir, res)));
		}
		arr_u("march_cone_maxcascade_p3_d3_t_dt_mipfrompos_mip_idx_occupied_dist_advance", out.data(), out.size());
		const uint32_t consts[6] = {NERF_STEPS(), NERF_CASCADES(), NERF_GRIDSIZE(), fb(STEPSIZE()), fb(MIN_CONE_STEPSIZE()), fb(MAX_CONE_STEPSIZE())};
		arr_u("constants_steps_cascades_gridsize_stepsize_min_max_cone_stepsize", consts, 6, true);
	}
	printf("}\n");
	return 0;
}
""")
    return "\n".join(parts)


def main():
    prog = build_program()
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "int_fixtures.cpp")
        open(src, "w").write("#include <cstring>\n#include <vector>\n" + prog)
        exe = os.path.join(d, "int_fixtures")
        # -ffp-contract=off: nvcc contracts to FMA on the device, g++ on x86-64 does not by default either way; the library and the checker are built without contraction
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-DEIGEN_DONT_VECTORIZE", "-w", "-I", os.path.join(REF, "dependencies", "eigen"), src, "-o", exe])
        text = subprocess.check_output([exe]).decode()
    data = json.loads(text)
    data = {"_source": "tests/golden/make_int_fixtures.py: fragments of /root/reference compiled with g++ in the build container (see the script's header); floats as IEEE-754 bit patterns",
            **data}
    out = os.path.join(HERE, "int_fixtures.json")
    with open(out, "w") as f:
        json.dump(data, f, separators=(",", ":"))
    print("wrote", out, {k: len(v) for k, v in data.items() if isinstance(v, list)})
    # the survey's three draws must come out again
    assert data["pcg32_next_uint_seeds_1337_42_0_deadbeefcafe_x6"][:3] == [634364130, 2023056239, 747258445]


if __name__ == "__main__":
    sys.exit(main())
