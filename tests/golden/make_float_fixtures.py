"""Writes tests/golden/float_fixtures.json: outputs of the host-compilable FLOATING-POINT fragments of the REFERENCE on the hot path --
the scalar device functions its kernels call for the hash-grid index / fraction, the coordinate warps of `NerfCoordinate`, the activations of the NeuS alpha
and of the albedo, the L1 / L2 ray loss, and the pixel / image choice of a training ray -- produced, like int_fixtures.json, by compiling those fragments (read
from /root/reference at run time, never copied into this repository) for the host in the build container and running them on seeded inputs. Tests read only the JSON.

  python tests/golden/make_float_fixtures.py

What is taken, verbatim (function or struct body located by its signature, braces matched):
  dependencies/neus2_tcnn/include/tiny-cuda-nn/common_device.h   logistic, identity_fun, pos_fract (the 3-argument-functor-free form: pos, pos_grid)
  dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h          clamp
  dependencies/neus2_tcnn/include/tiny-cuda-nn/encodings/grid.h  fast_hash, grid_index, to_string(GridType), the level loop of GridEncodingTemplated's constructor (:977-1010;
                                                                 its members bound as arguments) with common.h powi / next_multiple, set_training_step (:1430-1437; `override` dropped)  (enum GridType restated: its three names)
  dependencies/neus2_tcnn/dependencies/pcg32/pcg32.h             struct pcg32
  include/neural-graphics-primitives/random_val.cuh              random_val_2d; sobol .. ld_random_val and common.h binary_search, testbed_nerf.cu sample_cdf_2d (the error-map branches of
                                                                 image_idx / nerf_random_image_pos_training: compiled so that those two are taken unchanged, not reached -- null CDFs)
  include/neural-graphics-primitives/common_device.cuh           srgb_to_linear, image_pos, pixel_idx, the body of read_rgba's `case EImageDataType::Byte` (this fork's RGBA16 pixels)
  include/neural-graphics-primitives/bounding_box.cuh            BoundingBox::diag, ::relative_pos
  include/neural-graphics-primitives/nerf.h                      NERF_GRIDSIZE
  include/neural-graphics-primitives/nerf_loader.h               NerfDataset::nerf_matrix_to_ngp (in a struct with the four members it reads)
  include/neural-graphics-primitives/common.h                    struct Ray
  dependencies/neus2_tcnn/include/tiny-cuda-nn/encodings/grid.h  kernel_grid's body behind its index lines (one sample, one level), gpu_matrix.h MatrixView, common_device.h pos_fract / smoothstep
  dependencies/neus2_tcnn/include/tiny-cuda-nn/optimizers/*.h    adam_step's body behind its index lines, ema_step_half_precision's arithmetic line + EmaOptimizer::step's debias statements
  src/testbed_nerf.cu                                            the per-ray targets of the loss kernel (:1500-1592, three runs of its lines: all but the texel fetches and curand),
                                                                 the pinhole-ray statements of generate_training_samples_nerf (:1279-1305, located by their text),
                                                                 NERF_STEPS .. MIN_CONE_STEPSIZE, struct LossAndGradient, copysign(Array4f), mse_loss, l1_loss, loss_and_gradient,
                                                                 activation_function, network_to_rgb, network_to_rgb_derivative, warp_position, unwarp_position, warp_direction,
                                                                 unwarp_direction, warp_dt, unwarp_dt, nerf_random_image_pos_training, image_idx
  include/neural-graphics-primitives/common.h                    enum ELossType, enum ENerfActivation (restated: the enumerators in the file's order, checked against its text)
The CUDA decorations are defined away, `__expf` is `expf`, Eigen comes from the reference's vendored dependencies/eigen. What the host build cannot reproduce of the
device: nvcc contracts a * b + c to an FMA by default and its expf is not libm's -- the library and the checker are built WITHOUT contraction (-ffp-contract=off), so the
products / sums here are the ones they compute; the expf-based values are compared with an ulp tolerance on the GPU and bit for bit against the CPU checker (same libm).
"""
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_int_fixtures import REF, HERE, fragment, _block  # noqa: E402

LOSS_KERNEL = "__global__ void compute_loss_kernel_train_nerf_with_global_movement("
CXX = "/opt/rocm/lib/llvm/bin/clang++"  # host compilation only; clang has the native half type (_Float16) on x86-64 that g++ 11 lacks


def enum_names(path, head):
    """The enumerators of `enum class X : int {...}` in the reference's header, in order (the restated enum below must list the same names)."""
    src = open(os.path.join(REF, path)).read()
    i = src.index(head)
    body = src[src.index("{", i) + 1:src.index("}", i)]
    return [t.strip().split("=")[0].strip() for t in body.split(",") if t.strip()]


def block_ignoring_comments(path, signature):
    """fragment() for code whose // comments hold braces: the text from `signature` through the brace that closes its first '{', braces behind `//` not counted."""
    src = open(os.path.join(REF, path)).read()
    start = src.index(signature)
    i = src.index("{", start)
    depth, j, in_comment = 0, i, False
    while True:
        c = src[j]
        if c == "\n":
            in_comment = False
        elif src.startswith("//", j):
            in_comment = True
        elif not in_comment:
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
                if depth == 0:
                    return src[start:j + 1]
        j += 1


def statement(path, text):
    """A statement of the reference located by its exact text (asserted to be there, once or more -- the training and the inference sampler hold the same lines)."""
    src = open(os.path.join(REF, path)).read()
    assert text in src, text
    return text


def span(path, first, last, after=None):
    """The reference's text from the statement `first` through the statement `last` (both asserted present), searched from the text `after` on: a run of a kernel's own lines."""
    src = open(os.path.join(REF, path)).read()
    base = src.index(after) if after else 0
    i = src.index(first, base)
    j = src.index(last, i)
    return src[i:j + len(last)]


def span_until(path, first, before, after=None):
    """Like span(), but ends right in front of the text `before`."""
    src = open(os.path.join(REF, path)).read()
    base = src.index(after) if after else 0
    i = src.index(first, base)
    return src[i:src.index(before, i)]


def build_program():
    f = fragment
    tn = "src/testbed_nerf.cu"
    cd = "dependencies/neus2_tcnn/include/tiny-cuda-nn/common_device.h"
    gh = "dependencies/neus2_tcnn/include/tiny-cuda-nn/encodings/grid.h"
    bb = "include/neural-graphics-primitives/bounding_box.cuh"
    rv = "include/neural-graphics-primitives/random_val.cuh"
    cdc = "include/neural-graphics-primitives/common_device.cuh"
    src_tn = open(os.path.join(REF, tn)).read()
    uniform_fraction = src_tn.split("static constexpr float UNIFORM_SAMPLING_FRACTION =")[1].split(";")[0].strip()
    normals_normalized = src_tn.split("#define NORMAL_VECTORS_NORMALIZED")[1].split("\n")[0].strip()
    assert normals_normalized in ("0", "1")
    loss_names = enum_names("include/neural-graphics-primitives/common.h", "enum class ELossType")
    act_names = enum_names("include/neural-graphics-primitives/common.h", "enum class ENerfActivation")
    grid_names = enum_names(gh, "enum class GridType")
    assert loss_names[:2] == ["L2", "L1"] and act_names == ["None", "ReLU", "Logistic", "Exponential"] and grid_names == ["Hash", "Dense", "Tiled"], (loss_names, act_names, grid_names)
    parts = ["""
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cassert>
#include <vector>
#include <string>
#include <stdexcept>
#include <limits>
#include <algorithm>
#include <Eigen/Dense>
#define __host__
#define __device__
#define __restrict__
#define TCNN_HOST_DEVICE
#define NGP_HOST_DEVICE
#define __expf expf
#define NGP_PRAGMA_UNROLL
typedef _Float16 __half; // CUDA's half type is the host compiler's native one (clang): conversions round to nearest even, a product of two halves is rounded once
#define NORMAL_VECTORS_NORMALIZED """ + normals_normalized + """
#define PCG32_DEFAULT_STATE  0x853c49e6748fea9bULL
#define PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define PCG32_MULT           0x5851f42d4c957f2dULL
enum class ELossType : int { """ + ", ".join(loss_names) + """ };
enum class ENerfActivation : int { """ + ", ".join(act_names) + """ };
namespace tcnn {
enum class GridType { """ + ", ".join(grid_names) + """ };
template <typename T> """ + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "TCNN_HOST_DEVICE T clamp(T val, T lower, T upper)"),
             f("dependencies/neus2_tcnn/dependencies/pcg32/pcg32.h", "struct pcg32 {") + ";",
             "using network_precision_t = __half; // common.h: TCNN_HALF_PRECISION\ntemplate <typename T, uint32_t N_ELEMS>\n" + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "struct vector_t {") + ";",
             "using default_rng_t = pcg32;",
             f(cd, "__host__ __device__ inline float logistic(const float x)"),
             f(cd, "__device__ inline float identity_fun(float val)"),
             "template <typename F>\n" + f(cd, "__device__ inline void pos_fract(const float input, float* pos, uint32_t* pos_grid, float scale, F interpolation_fun)"),
             "template <uint32_t N_DIMS>\n" + f(gh, "__device__ uint32_t fast_hash(const uint32_t pos_grid[N_DIMS])"),
             "template <uint32_t N_DIMS, uint32_t N_FEATURES_PER_LEVEL>\n" + f(gh, "__device__ uint32_t grid_index(const GridType grid_type, const uint32_t feature, const uint32_t hashmap_size, const uint32_t grid_resolution, const uint32_t pos_grid[N_DIMS])"),
             f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "inline uint32_t powi(uint32_t base, uint32_t exponent)"),
             "template <typename T> " + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "TCNN_HOST_DEVICE T div_round_up(T val, T divisor)"),
             "template <typename T> " + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "TCNN_HOST_DEVICE T next_multiple(T val, T divisor)"),
             f(gh, "inline std::string to_string(GridType grid_type)"),
             # the level loop of GridEncodingTemplated's constructor (grid.h:977-1010), verbatim, with the members it reads / writes bound as arguments
             """template <uint32_t N_POS_DIMS>
static void level_tables(const uint32_t m_n_levels, const float per_level_scale, const uint32_t base_resolution, const uint32_t log2_hashmap_size, const GridType grid_type,
                         std::vector<uint32_t>& m_hashmap_offsets_table_cpu, std::vector<uint32_t>& m_resolution_table_cpu, std::vector<float>& m_scale_table_cpu) {
	uint32_t offset = 0;
	""" + f(gh, "for (uint32_t i = 0; i < m_n_levels; ++i) {\n\t\t\t// Compute dense params required for the given level") + """
	m_hashmap_offsets_table_cpu[m_n_levels] = offset;
}""",
             # GridEncodingTemplated::set_training_step (grid.h:1430-1437), verbatim, inside a struct that declares the members it touches with the reference's types (:1477-1481)
             """static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
struct ValidLevel {
	uint32_t m_valid_level; int m_training_step; uint32_t m_base_training_step; float m_base_valid_level_scale; float m_valid_level_scale; uint32_t m_n_levels;
	""" + f(gh, "void set_training_step(int training_step) override {").replace("void set_training_step(int training_step) override {", "void set_training_step(int training_step) {", 1) + """
};""",
             "}\nusing namespace Eigen;\nusing default_rng_t = tcnn::default_rng_t;",
             f("include/neural-graphics-primitives/common.h", "struct Ray {") + ";",
             # the pinhole ray of generate_training_samples_nerf (testbed_nerf.cu:1279-1305): the kernel's own statements, located by their text, in its order, between the
             # declarations of the names they use (distortion modes other than None, rolling shutter and global movement are off on this path)
             """static void camera_ray_statements(const Vector2f& xy, const Vector2f& principal_point, const Vector2i& resolution, const Vector2f& focal_length, const Matrix<float, 3, 4>& xform,
                                  Vector3f& o_out, Vector3f& d_out, Vector3f& dir_out) {
	Ray ray_unnormalized;
	""" + statement(tn, "ray_unnormalized.o = xform.col(3);") + "\n\t" + _block(open(os.path.join(REF, tn)).read(), open(os.path.join(REF, tn)).read().index("ray_unnormalized.d = {\n\t\t\t(xy.x()-principal_point.x())*resolution.x() / focal_length.x(),")) + ";\n\t"
             + statement(tn, "ray_unnormalized.d = (xform.block<3, 3>(0, 0) * ray_unnormalized.d);") + "\n\t" + statement(tn, "Eigen::Vector3f dir = ray_unnormalized.d.normalized();") + """
	o_out = ray_unnormalized.o; d_out = ray_unnormalized.d; dir_out = dir;
}""",
             f(cdc, "inline __host__ __device__ float linear_to_srgb(float linear)"),
             f(cdc, "inline __host__ __device__ Eigen::Array3f linear_to_srgb(const Eigen::Array3f& x)"),
             # the per-ray targets of compute_loss_kernel_train_nerf_with_global_movement (testbed_nerf.cu:1500-1592): the kernel's own lines in three runs -- everything between
             # `xform` and `rgbtarget` except the two read_rgba calls (the texels are arguments here) and the three curand lines (the light index is an argument: deviation D4)
             """static inline float max(float a, float b) { return fmaxf(a, b); }
using std::abs; using std::exp; using std::log; // CUDA's global overload set: abs / exp / log of a float are the float functions (the C library's global names are the double ones)
float activation_function(float val, ENerfActivation activation); // (the reference's, defined below from its file)
struct RayTargets { Array3f normal_value; Array4f albedo_value; Matrix3f light_directions_before; Vector3f light; float shading_target; Array4f rgbtarget; };
static RayTargets ray_target_statements(const Matrix<float, 3, 4>& xform_in, const Array4f& texsamp_albedo, const Array4f& texsamp_normal, const bool apply_no_albedo, const bool apply_rgbplus,
                                        const bool apply_L2, const bool apply_supernormal, const bool apply_light_opti, const bool apply_relu, const int random_light) {
	const uint32_t img = 0;
	const Array3f exposure_values[1] = {Array3f::Zero()};
	const Array3f* exposure = exposure_values;
	struct { Matrix<float, 3, 4> start; } training_xforms[1] = {{xform_in}};
	""" + span(tn, "const Matrix<float, 3, 4>& xform = training_xforms[img].start;", "Array3f exposure_scale = (0.6931471805599453f * exposure[img]).exp();", LOSS_KERNEL) + "\n\t"
             + span(tn, "Array3f normal_value = linear_to_srgb(exposure_scale * texsamp_normal.head<3>())*2.0f - 1.0f;", "light_directions = Eigen::Matrix3f::Identity(); \n\t}", LOSS_KERNEL) + """
	const Matrix3f light_directions_before = light_directions;
	""" + span(tn, "if (apply_light_opti){", "Array4f rgbtarget = albedo_value * shading_target;", LOSS_KERNEL) + """
	return {normal_value, albedo_value, light_directions_before, light, shading_target, rgbtarget};
}""",
             # NerfDataset::nerf_matrix_to_ngp (nerf_loader.h:180-201), verbatim, inside a struct with the four members it reads
             "struct DatasetAxes { float scale; Eigen::Vector3f offset; bool from_mitsuba; bool from_na;\n" + f("include/neural-graphics-primitives/nerf_loader.h", "Eigen::Matrix<float, 3, 4> nerf_matrix_to_ngp(const Eigen::Matrix<float, 3, 4>& nerf_matrix)") + "\n};",
             "template <typename RNG>\n" + f("include/neural-graphics-primitives/random_val.cuh", "inline __host__ __device__ Eigen::Vector2f random_val_2d(RNG& rng)"),
             f("include/neural-graphics-primitives/nerf.h", "inline constexpr __device__ uint32_t NERF_GRIDSIZE()"),
             f(cdc, "inline __host__ __device__ float srgb_to_linear(float srgb)"),
             f(cdc, "inline NGP_HOST_DEVICE Eigen::Vector2i image_pos(const Eigen::Vector2f& pos, const Eigen::Vector2i& resolution)"),
             f(cdc, "inline NGP_HOST_DEVICE uint64_t pixel_idx(const Eigen::Vector2i& pos, const Eigen::Vector2i& resolution, uint32_t img)"),
             # read_rgba (common_device.cuh:665-696): the body of its `case EImageDataType::Byte` -- this fork's RGBA16 pixels -- verbatim, as a function of the same arguments
             # (the Half case next to it is written in CUDA's __half and is not what the path's images take)
             "static Eigen::Array4f read_rgba_byte_case(Eigen::Vector2i px, const Eigen::Vector2i& resolution, const void* pixels, uint32_t img) "
             + f(cdc, "case EImageDataType::Byte: {")[len("case EImageDataType::Byte: "):],
             "struct BoundingBox { Eigen::Vector3f min, max;\n" + f(bb, "NGP_HOST_DEVICE Eigen::Vector3f diag() const") + "\n" + f(bb, "NGP_HOST_DEVICE Eigen::Vector3f relative_pos(const Eigen::Vector3f& pos) const") + "\n" + f(bb, "NGP_HOST_DEVICE void inflate(float amount)") + "\n" + f(bb, "NGP_HOST_DEVICE bool contains(const Eigen::Vector3f& p) const") + "};",
             # what image_idx / nerf_random_image_pos_training call when error-map CDFs are given (never here: null pointers) -- the reference's own functions, so that the two compile unchanged
             f(rv, "inline __host__ __device__ uint32_t sobol(uint32_t index, uint32_t dim)"), f(rv, "inline __host__ __device__ uint32_t hash_combine(uint32_t seed, uint32_t v)"),
             f(rv, "inline __host__ __device__ uint32_t reverse_bits(uint32_t x)"), f(rv, "inline __host__ __device__ uint32_t laine_karras_permutation(uint32_t x, uint32_t seed)"),
             f(rv, "inline __host__ __device__ uint32_t nested_uniform_scramble_base2(uint32_t x, uint32_t seed)"),
             f(rv, "inline __host__ __device__ float ld_random_val(uint32_t index, uint32_t seed, uint32_t dim = 0)"),
             f("include/neural-graphics-primitives/common.h", "inline NGP_HOST_DEVICE uint32_t binary_search(float val, const float* data, uint32_t length)"),
             "static constexpr float UNIFORM_SAMPLING_FRACTION = %s;" % uniform_fraction,
             f(tn, "inline __device__ Vector2f sample_cdf_2d(")]
    for sig in ("inline constexpr __device__ uint32_t NERF_STEPS()", "inline constexpr __device__ uint32_t NERF_CASCADES()", "inline constexpr __device__ float SQRT3()",
                "inline constexpr __device__ float STEPSIZE()", "inline constexpr __device__ float MIN_CONE_STEPSIZE()",
                "struct LossAndGradient {"):
        parts.append(f(tn, sig) + (";" if sig.startswith("struct") else ""))
    for sig in ("inline __device__ Array4f copysign(const Array4f& a, const Array4f& b)", "inline __device__ LossAndGradient mse_loss(const Array4f& target, const Array4f& prediction)",
                "inline __device__ LossAndGradient l1_loss(const Array4f& target, const Array4f& prediction)",
                "__device__ LossAndGradient loss_and_gradient(const Vector4f& target, const Vector4f& prediction, ELossType loss_type)",
                "__device__ float activation_function(float val, ENerfActivation activation)", "__device__ float network_to_rgb(float val, ENerfActivation activation)",
                "__device__ float network_to_rgb_derivative(float val, ENerfActivation activation)",
                "__device__ Vector3f warp_position(const Vector3f& pos, const BoundingBox& aabb)", "__device__ Vector3f unwarp_position(const Vector3f& pos, const BoundingBox& aabb)",
                "__host__ __device__ Vector3f warp_direction(const Vector3f& dir)", "__device__ Vector3f unwarp_direction(const Vector3f& dir)",
                "__device__ float warp_dt(float dt)", "__device__ float unwarp_dt(float dt)",
                "inline __device__ Vector2f nerf_random_image_pos_training(", "inline __device__ uint32_t image_idx("):
        parts.append(f(tn, sig).replace("float* __restrict__ pdf = nullptr", "float* pdf = nullptr").replace("const float* __restrict__ cdf = nullptr", "const float* cdf = nullptr"))
    parts.append(f("include/neural-graphics-primitives/common.h", "inline NGP_HOST_DEVICE float sign(float x)"))
    parts.append(f(tn, "__device__ Array3f network_to_pos_gradient(const tcnn::vector_t<tcnn::network_precision_t, 16>& local_network_output, ENerfActivation activation)"))
    parts.append(f(tn, "__device__ Array3f network_to_rgb(const tcnn::vector_t<tcnn::network_precision_t, 16>& local_network_output, ENerfActivation activation)"))
    # one iteration of the loss kernel's SECOND loop (testbed_nerf.cu:1855-2085, "now do it again computing gradients"): its own lines from the load of the network output
    # to the last element of dL/doutput it sets, between declarations of the names the loop body reads and updates
    parts.append("""struct SampleBackward { float alpha, shading, T_after, weight_sum2_after; Array4f rgb_ray2_after; float ek_term; __half dl[11]; float inter[10]; };
static SampleBackward backward_sample_statements(const __half* network_output, const float dt, const Vector3f dir, const Vector3f light, const Vector3f pos, const Vector3f ray_o,
                                                 const bool apply_no_albedo, const bool apply_rgbplus, const bool apply_L2, const bool apply_relu, const float cos_anneal_ratio,
                                                 LossAndGradient lg, const Array4f rgb_ray, Array4f rgb_ray2, const float weight_sum, float weight_sum2, float T, const float gradient_weight_sum,
                                                 const float loss_scale, const float original_loss_scale, const float ek_loss_weight) {
	const ENerfActivation rgb_activation = ENerfActivation::Logistic; // testbed_nerf.cu:3121
	float depth_ray2 = 0.f;
	const float depth = (pos - ray_o).norm();
	float ek_value = 0.f;
	float* ek_loss_output = &ek_value;
	const uint32_t i = 0;
	""" + span_until(tn, "const tcnn::vector_t<tcnn::network_precision_t, 16> local_network_output = *(tcnn::vector_t<tcnn::network_precision_t, 16>*)network_output;",
                     "*(tcnn::vector_t<tcnn::network_precision_t, 16>*)dloss_doutput = local_dL_doutput;", "// now do it again computing gradients") + """
	SampleBackward r{alpha, shading, T, weight_sum2, rgb_ray2, ek_value, {}, {dloss_by_drgb.x(), dloss_by_drgb.y(), dloss_by_drgb.z(), dloss_dn.x(), dloss_dn.y(), dloss_dn.z(), dloss_dalpha, dloss_dsdf,
	                                                                              dloss_dvariance, dloss_dnormal_norm}};
	for (int q = 0; q < 11; ++q) r.dl[q] = local_dL_doutput[q];
	return r;
}""")
    # the loss kernel between its two loops (testbed_nerf.cu:1735-1800): the ray's loss and gradient, the clamped weight sum and the mask term's gradient, the two loss rows
    parts.append("""struct RayLossTerms { float loss; Array4f gradient; float weight_sum, gradient_weight_sum, loss_row, mask_row; };
static RayLossTerms ray_loss_statements(const Array4f rgbtarget, const Array4f rgb_ray, const Array4f texsamp_albedo, const Array4f texsamp_normal, float weight_sum, const bool apply_L2,
                                        const bool apply_rgbplus, const bool apply_bce, const float mask_loss_weight, const uint32_t n_rays) {
	ELossType loss_type;
	const float img_pdf = 1.0f, xy_pdf = 1.0f; // no error-map sampling
	float loss_row = 0.f, mask_row = 0.f;
	float* loss_output = &loss_row;
	float* mask_loss_output = &mask_row;
	const uint32_t i = 0;
	""" + span_until(tn, "float mask_certainty = (float) (texsamp_albedo.w() > 0.99);", "if (ek_loss_output) {\n\t\tek_loss_output[i] = 0.f;", LOSS_KERNEL) + """
	return {lg.loss, lg.gradient, weight_sum, gradient_weight_sum, loss_row, mask_row};
}""")
    # one (sample, level) of tcnn's kernel_grid (encodings/grid.h:168-364): the kernel's body behind its three index lines (`i` and `level` bound as arguments): interpolated
    # features accumulated in half, dy/dx in float
    itp = enum_names("dependencies/neus2_tcnn/include/tiny-cuda-nn/encoding.h", "enum class InterpolationType")
    assert itp == ["Nearest", "Linear", "Smoothstep"]
    grid_kernel = block_ignoring_comments(gh, "__global__ void kernel_grid(")
    grid_body = grid_kernel[grid_kernel.index("if (level > valid_level) {"):grid_kernel.rindex("}")]
    parts.append("namespace tcnn {\nenum class InterpolationType { " + ", ".join(itp) + " };\n" + statement("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "template <uint32_t N_FLOATS>\nusing vector_fullp_t = vector_t<float, N_FLOATS>;")
                 + "\ntemplate <typename T>\n" + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/gpu_matrix.h", "struct MatrixView {") + ";\n"
                 + f(cd, "__device__ inline float smoothstep(float val)") + "\n" + f(cd, "__device__ inline float smoothstep_derivative(float val)") + "\n" + f(cd, "__device__ inline float identity_derivative(float val)") + "\n"
                 + "template <typename F, typename FPRIME>\n" + f(cd, "__device__ inline void pos_fract(const float input, float* pos, float* pos_derivative, uint32_t* pos_grid, float scale, F interpolation_fun, FPRIME interpolation_fun_derivative)") + """
template <typename T, uint32_t N_POS_DIMS, uint32_t N_FEATURES_PER_LEVEL>
static void kernel_grid_element(const uint32_t i, const uint32_t level, const uint32_t num_elements, const uint32_t num_grid_features, const uint32_t* hashmap_offset_table,
	const uint32_t* resolution_table, const float* scale_table, const uint32_t valid_level, const float quantize_threshold, float max_level, const float* max_level_gpu,
	const InterpolationType interpolation_type, const GridType grid_type, const T* grid, MatrixView<const float> positions_in, T* encoded_positions, float* dy_dx) {
	""" + grid_body + """
}
}""")
    # one sample of an occupancy update: generate_grid_samples_nerf_nonuniform's body behind its two index lines (testbed_nerf.cu:585-614), with NERF_MIN_OPTICAL_THICKNESS as the
    # file's `#if SDF_GRID` selects it
    sdf_grid = src_tn.split("#define SDF_GRID")[1].split("\n")[0].strip()
    gs_kernel = block_ignoring_comments(tn, "__global__ void generate_grid_samples_nerf_nonuniform(")
    gs_body = gs_kernel[gs_kernel.index("// 1 random number to select the level, 3 to select the position."):gs_kernel.rindex("}")]
    parts.append("#define SDF_GRID " + sdf_grid + "\n" + span(tn, "#if SDF_GRID\ninline constexpr __device__ float NERF_MIN_OPTICAL_THICKNESS()", "#endif") + "\n"
                 + f("include/neural-graphics-primitives/nerf.h", "struct NerfPosition {") + ";\n"
                 + "template <typename RNG>\n" + f(rv, "inline __host__ __device__ float random_val(RNG& rng)") + "\ntemplate <typename RNG>\n" + f(rv, "inline __host__ __device__ Eigen::Vector3f random_val_3d(RNG& rng)") + """
namespace tcnn { """ + f(cd, "__host__ __device__ inline uint32_t expand_bits(uint32_t v)") + f(cd, "__host__ __device__ inline uint32_t morton3D_invert(uint32_t x)") + """ }
static void grid_sample_element(const uint32_t i, const uint32_t n_elements, default_rng_t rng, const uint32_t step, BoundingBox aabb, const float* grid_in, NerfPosition* out, uint32_t* indices,
                                uint32_t n_cascades, float thresh) {
	""" + gs_body + """
}""")
    # occupancy grid -> bitfield and its seven max-pooled mips: the bodies of grid_to_bitfield and bitfield_max_pool behind their index lines (testbed_nerf.cu:693-740), driven as
    # update_density_grid_mean_and_bitfield drives them (:3496-3517)
    g2b = block_ignoring_comments(tn, "__global__ void grid_to_bitfield(")
    bmp = block_ignoring_comments(tn, "__global__ void bitfield_max_pool(")
    parts.append("namespace tcnn { " + f(cd, "__host__ __device__ inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z)") + " }\n" + f(tn, "inline __host__ __device__ uint32_t grid_mip_offset(uint32_t mip)") + """
static void grid_to_bitfield_element(const uint32_t i, const uint32_t n_nonzero_elements, const float* grid, uint8_t* grid_bitfield, const float* mean_density_ptr) {
	""" + g2b[g2b.index("if (i >= n_nonzero_elements) {"):g2b.rindex("}")] + """
}
static void bitfield_max_pool_element(const uint32_t i, const uint8_t* prev_level, uint8_t* next_level) {
	""" + bmp[bmp.index("uint8_t bits = 0;"):bmp.rindex("}")] + """
}""")
    # the sampler's two march loops (testbed_nerf.cu:1330-1380): the counting loop and the loop that writes the NerfCoordinates, the kernel's own lines, with the functions they call
    # (calc_dt .. mip_from_dt, the int fixtures' fragments) and the reference's NerfCoordinate / PitchedPtr
    parts.append("""static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
static inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
namespace tcnn { template <typename T>\n""" + f("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "struct PitchedPtr {") + "; }\n"
                 + f("include/neural-graphics-primitives/nerf.h", "struct NerfDirection {") + ";\n" + f("include/neural-graphics-primitives/nerf.h", "struct NerfCoordinate {") + ";\n"
                 + "\n".join(f(tn, sig) for sig in ("inline constexpr __device__ float MAX_CONE_STEPSIZE()", "inline __host__ __device__ float calc_dt(float t, float cone_angle)",
                                                     "inline __device__ float distance_to_next_voxel(", "inline __device__ float advance_to_next_voxel(",
                                                     "__device__ uint32_t cascaded_grid_idx_at(Vector3f pos, uint32_t mip)", "__device__ bool density_grid_occupied_at(",
                                                     "inline __device__ int mip_from_pos(", "inline __device__ int mip_from_dt(")) + """
struct MarchResult { uint32_t numsteps; std::vector<float> coords; };
static MarchResult march_statements(const Vector3f ray_o, const Vector3f dir, const float startt, const float cone_angle, const BoundingBox aabb, const uint8_t* density_grid) {
	std::vector<float> storage(7 * NERF_STEPS());
	tcnn::PitchedPtr<NerfCoordinate> coords_out((NerfCoordinate*)storage.data(), 1, 0, 0);
	const float* extra_dims = nullptr;
	""" + span_until(tn, "Vector3f idir = dir.cwiseInverse();", "if (j == 0 && !train_envmap) {", "__global__ void generate_training_samples_nerf_with_global_movement(") + """
	uint32_t numsteps = j;
	""" + span_until(tn, "Vector3f warped_dir = warp_direction(dir);", "if (max_level_rand_training) {\n\t\tmax_level_ptr += base;", "__global__ void generate_training_samples_nerf_with_global_movement(") + """
	storage.resize(7 * (size_t)numsteps);
	return {numsteps, storage};
}""")
    # the ray-batch controller: the two statements of Counters::update_after_training that set the next step's rays_per_batch (testbed_nerf.cu:3554-3555), host code in the reference too
    parts.append("namespace tcnn { " + statement("dependencies/neus2_tcnn/include/tiny-cuda-nn/common.h", "constexpr uint32_t batch_size_granularity = 128;") + """ }
static uint32_t controller_statements(uint32_t rays_per_batch, const uint32_t target_batch_size, const uint32_t measured_batch_size) {
	using tcnn::next_multiple;
	""" + statement(tn, "rays_per_batch = (uint32_t)((float)rays_per_batch * (float)target_batch_size / (float)measured_batch_size);") + "\n\t"
                 + statement(tn, "rays_per_batch = std::min(next_multiple(rays_per_batch, tcnn::batch_size_granularity), 1u << 18);") + """
	return rays_per_batch;
}""")
    # SDF -> the occupancy grid's density: sdf_to_density_variance_buffer's body behind its index lines (common_operation.cuh:311-328), half arithmetic throughout
    # (-ffloat16-excess-precision=none: every half operation rounds, as CUDA's __half operators do)
    s2d = block_ignoring_comments("include/neural-graphics-primitives/common_operation.cuh", "__global__ void sdf_to_density_variance_buffer(")
    parts.append("""template <typename T>
static void sdf_to_density_element(const uint32_t i, const T* variance_output, tcnn::MatrixView<T> sdf_network_output) {
	""" + s2d[s2d.index("T sdf = sdf_network_output(0, i);"):s2d.rindex("}")] + """
}""")
    # which steps begin with an occupancy update: the two lines of Testbed::train (src/testbed.cu:2805-2806) that decide it
    tb = open(os.path.join(REF, "src", "testbed.cu")).read()
    skip_stmt = "uint32_t n_prep_to_skip = (m_testbed_mode == ETestbedMode::Nerf) ? tcnn::clamp(m_canonical_training_step / 16u, 1u, 16u) : 1u;"
    due_cond = "m_canonical_training_step % n_prep_to_skip == 0"
    assert skip_stmt in tb and ("if (" + due_cond + ") {") in tb
    modes = enum_names("include/neural-graphics-primitives/common.h", "enum class ETestbedMode")
    assert modes[0] == "Nerf"
    parts.append("enum class ETestbedMode : int { " + ", ".join(modes) + """ };
static void prep_statements(const uint32_t m_canonical_training_step, uint32_t* skip_out, uint32_t* due_out) {
	const ETestbedMode m_testbed_mode = ETestbedMode::Nerf;
	""" + skip_stmt + """
	*skip_out = n_prep_to_skip;
	*due_out = (""" + due_cond + """) ? 1u : 0u;
}""")
    # the learning-rate schedule: the two if-blocks at the head of ExponentialDecayOptimizer::step (optimizers/exponential_decay.h:61-72), in a struct with the members they use;
    # step() is the nested optimizer's count of steps taken so far
    dec_h = "dependencies/neus2_tcnn/include/tiny-cuda-nn/optimizers/exponential_decay.h"
    dec_step = fragment(dec_h, "void step(cudaStream_t stream, float loss_scale, float* weights_full_precision, T* weights, const T* gradients) override {")
    parts.append("""struct DecaySchedule {
	float m_learning_rate_factor; uint32_t m_decay_start, m_decay_interval, m_decay_end; float m_decay_base; uint32_t steps_taken;
	uint32_t step() const { return steps_taken; }
	void before_nested_step() {
		""" + _block(dec_step, dec_step.index("if (step() == 0) {")) + "\n\t\t" + _block(dec_step, dec_step.index("if (step() >= m_decay_start")) + """
	}
};""")
    # the optimizer: one element of tcnn's adam_step (optimizers/adam.h:52-202: the kernel's body behind its two index lines, `i` bound as an argument), the half-precision EMA
    # step (ema.h:63-78, its one arithmetic line) with the two debias statements of EmaOptimizer::step (ema.h:115-116)
    adam_h = "dependencies/neus2_tcnn/include/tiny-cuda-nn/optimizers/adam.h"
    ema_h = "dependencies/neus2_tcnn/include/tiny-cuda-nn/optimizers/ema.h"
    adam_kernel = block_ignoring_comments(adam_h, "__global__ void adam_step(")
    adam_body = adam_kernel[adam_kernel.index("float gradient = (float)gradients[i] / loss_scale;"):adam_kernel.rindex("}")]
    parts.append("#include <utility>\nnamespace tcnn {\n" + f(cd, "__device__ inline float weight_decay(float relative_weight_decay, float absolute_weight_decay, float weight)") + """
template <typename T>
static void adam_step_element(const uint32_t i, const uint32_t n_components, const uint32_t n_weights_delta, const uint32_t n_weights_canonical_covered_by_matrices,
	const uint32_t n_weights_delta_covered_by_matrices, const float relative_weight_decay, const float absolute_weight_decay, const float loss_scale, float learning_rate,
	const float non_matrix_learning_rate_factor, const bool optimize_matrix_params, const bool optimize_non_matrix_params, const bool optimize_canonical_params, const bool optimize_delta_params,
	const float beta1, const float beta2, const float epsilon, const float lower_lr_bound, const float upper_lr_bound, const float l2_reg, float* weights_full_precision, T* weights,
	const T* gradients, float* first_moments, float* second_moments, uint32_t* param_steps, std::pair<uint32_t, bool>* n_weights_optimize, const bool only_sdf_training,
	const bool only_reflectance_training) {
	""" + adam_body + """
}
template <typename T>
static void ema_step_element(const uint32_t i, const float m_ema_decay, const uint32_t current_step, const T* weights, T* weights_ema) {
	const float ema_decay = m_ema_decay;
	""" + statement(ema_h, "float ema_debias_old = 1 - (float)std::pow(m_ema_decay, current_step-1);") + "\n\t" + statement(ema_h, "float ema_debias_new = 1.0f / (1 - (float)std::pow(m_ema_decay, current_step));") + "\n\t"
                 + statement(ema_h, "float filtered_val = ((float)weights_ema[i] * ema_decay * ema_debias_old + (float)weights[i] * (1 - ema_decay)) * ema_debias_new;") + "\n\t"
                 + statement(ema_h, "weights_ema[i] = (T)filtered_val;") + """
}
}""")
    base = json.load(open(os.path.join(REF, "configs", "nerf", "base.json")))["optimizer"]
    adam_cfg = base["nested"]["nested"]
    assert base["otype"] == "Ema" and base["nested"]["otype"] == "ExponentialDecay" and adam_cfg["otype"] == "Adam"
    opt = {k: repr(float(adam_cfg[k])) + "f" for k in ("learning_rate", "beta1", "beta2", "epsilon", "l2_reg")}
    opt["ema_decay"] = repr(float(base["decay"])) + "f"
    dec = {"decay_start": str(int(base["nested"]["decay_start"])), "decay_interval": str(int(base["nested"]["decay_interval"])), "decay_base": repr(float(base["nested"]["decay_base"])) + "f"}
    parts.append(r"""
static uint32_t fb(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static void arr_u(const char* name, const std::vector<uint32_t>& v, bool last = false) { printf("\"%s\": [", name); for (size_t i = 0; i < v.size(); ++i) printf("%u%s", v[i], i + 1 < v.size() ? "," : ""); printf("]%s\n", last ? "" : ","); }
int main() {
	printf("{\n");
	tcnn::pcg32 gen{20240930};
	auto uni = [&](float lo, float hi) { return lo + (hi - lo) * gen.next_float(); };
	{ // ---- activations of the loss kernel (testbed_nerf.cu:968-975 ReLU / Logistic through activation_function; 1620 network_to_rgb; its derivative in the backward part)
		std::vector<uint32_t> out;
		std::vector<float> vals = {0.0f, -0.0f, 1.0f, -1.0f, 1e-8f, -1e-8f, 0.5f, -0.5f, 10.0f, -10.0f, 16.5f, -16.5f, 30.0f, -30.0f, 87.0f, -87.0f, 88.8f, -88.8f, 100.0f, -100.0f, 1e4f, -1e4f};
		for (int k = 0; k < 234; ++k) vals.push_back(uni(-20.0f, 20.0f));
		for (float v : vals) {
			out.push_back(fb(v));
			out.push_back(fb(activation_function(v, ENerfActivation::ReLU)));
			out.push_back(fb(activation_function(v, ENerfActivation::Logistic)));
			out.push_back(fb(network_to_rgb(v, ENerfActivation::Logistic)));
			out.push_back(fb(network_to_rgb_derivative(v, ENerfActivation::Logistic)));
		}
		arr_u("activation_val_relu_logistic_rgb_rgbderivative", out);
	}
	{ // ---- NerfCoordinate warps (testbed_nerf.cu:1366-1373 writes them, 1641-1648 reads them back)
		std::vector<uint32_t> out;
		const float boxes[3][2] = {{0.0f, 1.0f}, {-0.5f, 1.5f}, {-1.5f, 2.5f}};
		const float max_stepsize = MIN_CONE_STEPSIZE() * (1 << (NERF_CASCADES() - 1));
		for (int k = 0; k < 192; ++k) {
			const float* b = boxes[k % 3];
			BoundingBox box{Vector3f::Constant(b[0]), Vector3f::Constant(b[1])};
			Vector3f p{uni(b[0], b[1]), uni(b[0], b[1]), uni(b[0], b[1])};
			Vector3f d = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}.normalized();
			float dt = k < 8 ? (k % 2 ? max_stepsize : MIN_CONE_STEPSIZE()) : uni(MIN_CONE_STEPSIZE(), max_stepsize);
			if (k == 8) p = Vector3f::Constant(b[0]);
			if (k == 9) p = Vector3f::Constant(b[1]);
			Vector3f wp = warp_position(p, box), up = unwarp_position(wp, box), wd = warp_direction(d), ud = unwarp_direction(wd);
			float wt = warp_dt(dt), ut = unwarp_dt(wt);
			for (float v : {b[0], b[1], p.x(), p.y(), p.z(), d.x(), d.y(), d.z(), dt, wp.x(), wp.y(), wp.z(), up.x(), up.y(), up.z(), wd.x(), wd.y(), wd.z(), ud.x(), ud.y(), ud.z(), wt, ut}) out.push_back(fb(v));
		}
		arr_u("warp_lo_hi_p3_d3_dt_warpedp3_unwarpedp3_warpedd3_unwarpedd3_warpeddt_unwarpeddt", out);
	}
	{ // ---- the ray loss (testbed_nerf.cu:1389-1394, called at 1800 with the composited rgb+ vector and its target)
		std::vector<uint32_t> out;
		for (int k = 0; k < 128; ++k) {
			Vector4f t{uni(0, 1), uni(0, 1), uni(0, 1), uni(0, 3)}, p{uni(0, 1), uni(0, 1), uni(0, 1), uni(0, 3)};
			if (k < 8) p[k % 4] = t[k % 4];            // a zero difference: copysign(1, +0)
			if (k >= 8 && k < 12) { t[k % 4] = 0.25f; p[k % 4] = 0.25f - 0.0f; t[(k + 1) % 4] = p[(k + 1) % 4] + 0.0f; }
			const ELossType type = (k & 1) ? ELossType::L1 : ELossType::L2;
			LossAndGradient lg = loss_and_gradient(t, p, type);
			out.push_back(type == ELossType::L2 ? 1u : 0u);
			for (int c = 0; c < 4; ++c) out.push_back(fb(t[c]));
			for (int c = 0; c < 4; ++c) out.push_back(fb(p[c]));
			out.push_back(fb(lg.loss));
			for (int c = 0; c < 4; ++c) out.push_back(fb(lg.gradient[c]));
		}
		arr_u("loss_isL2_target4_prediction4_loss_gradient4", out);
	}
	{ // ---- which image and which pixel a training ray takes (testbed_nerf.cu:1255-1262: image_idx, then the ray's generator advanced by i * 8, then the position)
		std::vector<uint32_t> out;
		const uint32_t shapes[4][2] = {{800, 800}, {256, 256}, {612, 512}, {1, 7}};
		for (int k = 0; k < 256; ++k) {
			const uint32_t n_rays = k < 4 ? 1u : 1u + gen.next_uint() % 300000u;
			const uint32_t base = gen.next_uint() % n_rays;
			const uint32_t total = k % 5 == 0 ? 0u : (k % 5 == 1 ? 4294967295u - gen.next_uint() % 1000u : gen.next_uint());
			const uint32_t n_img = k % 7 == 0 ? 1u : 1u + gen.next_uint() % 100u;
			const uint32_t w = shapes[k % 4][0], h = shapes[k % 4][1], snap = (k / 4) % 2;
			const uint64_t seed = 1337, adv = (uint64_t)(base + total) * 8;
			tcnn::pcg32 r{seed};
			r.advance((int64_t)adv);
			const uint32_t img = image_idx(base, n_rays, total, n_img);
			Vector2f xy = nerf_random_image_pos_training(r, Vector2i{(int)w, (int)h}, snap != 0, nullptr, nullptr, Vector2i{0, 0}, img);
			for (uint32_t v : {base, n_rays, total, n_img, w, h, snap, (uint32_t)adv, (uint32_t)(adv >> 32), img, fb(xy.x()), fb(xy.y())}) out.push_back(v);
		}
		arr_u("pixel_base_nrays_total_nimg_w_h_snap_advlo_advhi_img_x_y", out);
	}
	{ // ---- hash-grid index and fraction (grid.h:113-148, common_device.h:427-434 as called by kernel_grid with identity_fun: pos = x * scale + 0.5)
		std::vector<uint32_t> out;
		const uint32_t tables[8][2] = {{4096, 16}, {13824, 24}, {39304, 34}, {125000, 50}, {373248, 72}, {524288, 104}, {524288, 971}, {524288, 2049}};
		for (int k = 0; k < 512; ++k) {
			const uint32_t size = tables[k % 8][0], res = tables[k % 8][1];
			uint32_t pg[3] = {gen.next_uint() % (res + 1), gen.next_uint() % (res + 1), gen.next_uint() % (res + 1)};
			if (k < 8) { pg[0] = pg[1] = pg[2] = res; }      // the far corner of the last cell
			if (k >= 8 && k < 16) { pg[0] = pg[1] = pg[2] = 0; }
			const uint32_t i0 = tcnn::grid_index<3, 2>(tcnn::GridType::Hash, 0, size, res, pg), i1 = tcnn::grid_index<3, 2>(tcnn::GridType::Hash, 1, size, res, pg);
			const float x = k % 16 == 15 ? 1.0f : (k % 16 == 14 ? 0.0f : gen.next_float()), scale = (float)(res - 1) - (k % 3 == 0 ? 0.0f : 0.37f * gen.next_float());
			float pos; uint32_t cell;
			tcnn::pos_fract(x, &pos, &cell, scale, tcnn::identity_fun);
			for (uint32_t v : {size, res, pg[0], pg[1], pg[2], i0, i1, fb(x), fb(scale), fb(pos), cell}) out.push_back(v);
		}
		arr_u("grid_size_res_pg3_index0_index1_x_scale_pos_cell", out);
	}
	{ // ---- the level tables (grid.h:977-1012): resolution, scale (this fork: resolution - 1) and table offsets of every level -- the parameter layout of the hash grid
		std::vector<uint32_t> out;
		const float auto_scale = std::exp(std::log(2048.0f * 1.0f / 16.0f) / (14 - 1)); // testbed.cu:2322 with top_resolution 2048, aabb_scale 1, base 16, 14 levels
		struct C { uint32_t n, base, log2; float pls; };
		const C cfgs[10] = {{14, 16, 19, auto_scale}, {14, 16, 15, auto_scale}, {14, 16, 22, auto_scale}, {8, 16, 19, 1.5f}, {6, 8, 19, 2.0f}, {2, 16, 19, auto_scale},
		                    {14, 32, 19, 1.26f}, {10, 16, 17, 1.3819129f}, {1, 16, 19, auto_scale}, {12, 4, 14, 1.7f}};
		for (const C& c : cfgs) {
			std::vector<uint32_t> off(c.n + 1), res(c.n);
			std::vector<float> sc(c.n);
			tcnn::level_tables<3>(c.n, c.pls, c.base, c.log2, tcnn::GridType::Hash, off, res, sc);
			out.push_back(c.n); out.push_back(c.base); out.push_back(c.log2); out.push_back(fb(c.pls));
			for (uint32_t v : off) out.push_back(v);
			for (uint32_t v : res) out.push_back(v);
			for (float v : sc) out.push_back(fb(v));
		}
		arr_u("levels_n_base_log2hash_scalebits_offsets_resolutions_scales", out);
	}
	{ // ---- how many levels are live at a training step (grid.h:1430-1437; the values of configs/nerf/base.json and three others)
		std::vector<uint32_t> out;
		struct C { uint32_t n; float base_scale, scale; uint32_t base_step; };
		const C cfgs[4] = {{14, 0.2f, 0.02f, 100}, {14, 0.5f, 0.01f, 0}, {8, 0.1f, 0.05f, 250}, {6, 0.3f, 0.004f, 40}};
		for (const C& c : cfgs) {
			for (int step = -2; step <= 1500; step += (step < 130 ? 1 : 7)) {
				tcnn::ValidLevel v{0, 0, c.base_step, c.base_scale, c.scale, c.n};
				v.set_training_step(step);
				out.push_back(c.n); out.push_back(fb(c.base_scale)); out.push_back(fb(c.scale)); out.push_back(c.base_step); out.push_back((uint32_t)step); out.push_back(v.m_valid_level);
			}
		}
		arr_u("validlevel_n_basescale_scale_basestep_step_level", out);
	}
	{ // ---- a pixel of a training image (common_device.cuh:621-627, 665-700): position -> pixel, the 0x00FF00FF sentinel, sRGB decode, alpha premultiplication
		std::vector<uint32_t> out;
		const int shapes[4][2] = {{5, 2}, {4, 3}, {7, 2}, {2, 7}};
		for (int k = 0; k < 256; ++k) {
			const int w = shapes[k % 4][0], h = shapes[k % 4][1];
			uint16_t px[14 * 4];
			for (int q = 0; q < 14 * 4; ++q) px[q] = (uint16_t)(gen.next_uint() >> 16);
			for (int q = 0; q < 14; ++q) {
				const uint32_t r = gen.next_uint() % 16;
				if (r == 0) { px[q * 4 + 0] = 0x00FF; px[q * 4 + 1] = 0x00FF; px[q * 4 + 2] = 0; px[q * 4 + 3] = 0; }   // the sentinel word
				if (r == 1) px[q * 4 + 3] = 0;        // transparent
				if (r == 2) px[q * 4 + 0] = 0;        // red 0: the ray is kept with probability 0.1 (testbed_nerf.cu:1264)
				if (r == 3) px[q * 4 + 3] = 65535;
				if (r == 4) { px[q * 4 + 0] = 0x00FF; px[q * 4 + 1] = 0x00FF; px[q * 4 + 2] = 0; px[q * 4 + 3] = 1; }   // almost the sentinel
				if (r == 5) { px[q * 4 + 0] = 700; px[q * 4 + 1] = 2651; px[q * 4 + 2] = 2652; }                          // around the sRGB knee (0.04045 * 65535 = 2650.9)
			}
			float x = gen.next_float(), y = gen.next_float();
			if (k % 16 == 1) x = 1.0f;
			if (k % 16 == 2) y = 1.0f;
			if (k % 16 == 3) { x = 0.0f; y = 0.0f; }
			if (k % 16 == 4) x = -0.25f;
			if (k % 16 == 5) y = 1.75f;
			const Vector2i res{w, h};
			Array4f c = read_rgba_byte_case(image_pos(Vector2f{x, y}, res), res, px, 0);
			out.push_back((uint32_t)w); out.push_back((uint32_t)h); out.push_back(fb(x)); out.push_back(fb(y));
			for (int q = 0; q < 14; ++q) { out.push_back((uint32_t)px[q * 4] | (uint32_t)px[q * 4 + 1] << 16); out.push_back((uint32_t)px[q * 4 + 2] | (uint32_t)px[q * 4 + 3] << 16); }
			for (int q = 0; q < 4; ++q) out.push_back(fb(c[q]));
			out.push_back(c.x() <= 0.0f ? 1u : 0u);
		}
		arr_u("readrgba_w_h_x_y_pixels28_rgba4_rednonpositive", out);
	}
	{ // ---- camera matrices of transform.json -> the training frame (nerf_loader.h:180-201): the three axis conventions, scale and offset of the position
		std::vector<uint32_t> out;
		for (int k = 0; k < 36; ++k) {
			const int mode = k % 3; // 0 default (rows cycled), 1 from_na, 2 from_mitsuba
			DatasetAxes d{mode == 2 ? 0.66f : (k % 2 ? 0.33f : uni(0.1f, 2.0f)), Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}, mode == 2, mode == 1};
			if (mode == 2) d.offset = Vector3f::Constant(0.25f * d.scale); // what the loader sets for Mitsuba scenes (nerf_loader.cu:399-402)
			Eigen::Matrix<float, 3, 4> m;
			for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) m(r, c) = uni(-3, 3);
			const Eigen::Matrix<float, 3, 4> g = d.nerf_matrix_to_ngp(m);
			out.push_back((uint32_t)mode); out.push_back(fb(d.scale));
			for (int c = 0; c < 3; ++c) out.push_back(fb(d.offset[c]));
			for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out.push_back(fb(m(r, c)));
			for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out.push_back(fb(g(r, c)));
		}
		arr_u("axes_mode_scale_offset3_matrix12_ngp12", out);
	}
	{ // ---- the ray of an image position (testbed_nerf.cu:1279-1305)
		std::vector<uint32_t> out;
		const int shapes[3][2] = {{800, 800}, {256, 256}, {612, 512}};
		for (int k = 0; k < 192; ++k) {
			const int w = shapes[k % 3][0], h = shapes[k % 3][1];
			const Vector2f focal{uni(200, 1600), uni(200, 1600)}, pp{uni(0.4f, 0.6f), uni(0.4f, 0.6f)};
			Vector2f xy{gen.next_float(), gen.next_float()};
			if (k % 16 == 0) xy = pp;                       // the principal ray
			if (k % 16 == 1) xy = Vector2f{0.0f, 0.0f};
			Matrix<float, 3, 4> X;
			Vector3f a = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}.normalized(), b = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)};
			b = (b - a * a.dot(b)).normalized();
			X.col(0) = a; X.col(1) = b; X.col(2) = a.cross(b); X.col(3) = Vector3f{uni(-2, 3), uni(-2, 3), uni(-2, 3)};
			if (k % 4 == 3) X.block<3, 3>(0, 0) *= uni(0.5f, 2.0f);   // a scaled camera frame (n2w): the direction is normalised after the product
			Vector3f o, d, dir;
			camera_ray_statements(xy, pp, Vector2i{w, h}, focal, X, o, d, dir);
			out.push_back((uint32_t)w); out.push_back((uint32_t)h);
			for (float v : {focal.x(), focal.y(), pp.x(), pp.y(), xy.x(), xy.y()}) out.push_back(fb(v));
			for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out.push_back(fb(X(r, c)));
			for (float v : {o.x(), o.y(), o.z(), d.x(), d.y(), d.z(), dir.x(), dir.y(), dir.z()}) out.push_back(fb(v));
		}
		arr_u("cameraray_w_h_focal2_pp2_xy2_xform12_o3_d3_dir3", out);
	}
	{ // ---- the loss kernel's per-ray targets (testbed_nerf.cu:1500-1592)
		std::vector<uint32_t> out;
		for (int k = 0; k < 192; ++k) {
			const bool no_albedo = k % 4 == 1, rgbplus = k % 8 < 6, L2 = k % 3 != 2, supernormal = k % 16 == 5, light_opti = k % 5 == 2, relu_ = k % 7 == 3;
			const int random_light = k % 3;
			Matrix<float, 3, 4> X;
			Vector3f a = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}.normalized(), b = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)};
			b = (b - a * a.dot(b)).normalized();
			X.col(0) = a; X.col(1) = b; X.col(2) = a.cross(b); X.col(3) = Vector3f{uni(-2, 3), uni(-2, 3), uni(-2, 3)};
			// texels as read_rgba returns them: linear colour x alpha, alpha (a normal map pointing roughly at the camera; an albedo)
			const float an = k % 9 == 0 ? 0.5f : 1.0f, aa = k % 11 == 0 ? 0.25f : 1.0f;
			Vector3f nrm = Vector3f{uni(-0.6f, 0.6f), uni(-0.6f, 0.6f), 1.0f}.normalized();
			auto enc = [](float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 0.41666f) - 0.055f; }; (void)enc;
			Array4f tn_{srgb_to_linear(nrm.x() * 0.5f + 0.5f) * an, srgb_to_linear(-nrm.y() * 0.5f + 0.5f) * an, srgb_to_linear(-nrm.z() * 0.5f + 0.5f) * an, an};
			Array4f ta_{uni(0.02f, 0.9f) * aa, uni(0.02f, 0.9f) * aa, uni(0.02f, 0.9f) * aa, aa};
			const RayTargets r = ray_target_statements(X, ta_, tn_, no_albedo, rgbplus, L2, supernormal, light_opti, relu_, random_light);
			for (uint32_t v : {(uint32_t)no_albedo, (uint32_t)rgbplus, (uint32_t)L2, (uint32_t)light_opti, (uint32_t)relu_, (uint32_t)random_light}) out.push_back(v);
			for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 4; ++c) out.push_back(fb(X(rr, c)));
			for (int c = 0; c < 4; ++c) out.push_back(fb(tn_[c]));
			for (int c = 0; c < 4; ++c) out.push_back(fb(ta_[c]));
			for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) out.push_back(fb(r.light_directions_before(rr, c)));
			for (int c = 0; c < 4; ++c) out.push_back(fb(r.rgbtarget[c]));
			for (int c = 0; c < 3; ++c) out.push_back(fb(r.light[c]));
			for (int c = 0; c < 3; ++c) out.push_back(fb(r.normal_value[c]));
			out.push_back(fb(r.shading_target)); out.push_back((uint32_t)supernormal);
		}
		arr_u("raytargets_flags5_light_xform12_texnormal4_texalbedo4_lightdirs9_rgbtarget4_light3_normal3_shading_supernormal", out);
	}
	{ // ---- one sample of the loss kernel's backward loop (testbed_nerf.cu:1855-2085): alpha, the compositing sums, dL/d(network output)[0..10]
		std::vector<uint32_t> out;
		auto hb = [](__half h) { uint16_t u; memcpy(&u, &h, 2); return (uint32_t)u; };
		for (int k = 0; k < 384; ++k) {
			const bool no_albedo = k % 4 == 1, rgbplus = k % 8 < 6, L2 = k % 3 != 2, relu_ = k % 7 == 3;
			__half o[16];
			for (int q = 0; q < 16; ++q) o[q] = (__half)0.0f;
			for (int q = 0; q < 3; ++q) o[q] = (__half)uni(-2.5f, 2.5f);
			const float sdf_range = k % 5 == 0 ? 0.2f : (k % 5 == 1 ? 0.0005f : 0.02f);   // far from / at / near the surface
			o[3] = (__half)uni(-sdf_range, sdf_range);
			Vector3f g = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}.normalized() * uni(0.6f, 1.4f);
			for (int q = 0; q < 3; ++q) o[4 + q] = (__half)g[q];
			o[7] = (__half)uni(0.15f, 0.72f);
			for (int q = 0; q < 3; ++q) o[8 + q] = (__half)uni(0.0f, 1.0f);
			const float dt = uni(MIN_CONE_STEPSIZE(), 2.0f * MIN_CONE_STEPSIZE());
			Vector3f dir = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}.normalized();
			if (k % 2) dir = -g.normalized() * 0.9f + dir * 0.1f, dir.normalize();        // mostly facing the surface (true_cos < 0), as along a training ray
			const Vector3f light = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}.normalized();
			const Vector3f pos{uni(0, 1), uni(0, 1), uni(0, 1)}, ray_o{uni(-1, 2), uni(-1, 2), uni(-1, 2)};
			LossAndGradient lg{uni(0, 1), Array4f{uni(-1, 1), uni(-1, 1), uni(-1, 1), uni(-1, 1)}};
			if (!L2) lg.gradient = Array4f{k % 2 ? 1.f : -1.f, k % 4 < 2 ? 1.f : -1.f, 1.f, -1.f} * (rgbplus ? 0.5f : 1.0f);
			const Array4f rgb_ray{uni(0, 1), uni(0, 1), uni(0, 1), uni(0, 2)};
			const float prefix = uni(0, 1);
			const Array4f rgb_ray2 = rgb_ray * prefix;
			const float weight_sum = k % 16 == 7 ? (float)(1.0 - 1e-4) : uni(0.2f, 0.999f), weight_sum2 = weight_sum * prefix, T = k % 16 == 9 ? 1.0f : uni(0.01f, 1.0f);
			const float gws = k % 16 == 7 ? 0.0f : uni(-1, 1), original_loss_scale = 128.0f, loss_scale = original_loss_scale / (float)(2000 + k * 37), ekw = 0.1f;
			const SampleBackward r = backward_sample_statements(o, dt, dir, light, pos, ray_o, no_albedo, rgbplus, L2, relu_, 1.0f, lg, rgb_ray, rgb_ray2, weight_sum, weight_sum2, T, gws,
			                                                    loss_scale, original_loss_scale, ekw);
			for (uint32_t v : {(uint32_t)no_albedo, (uint32_t)rgbplus, (uint32_t)L2, (uint32_t)relu_}) out.push_back(v);
			for (int q = 0; q < 8; ++q) out.push_back(hb(o[2 * q]) | hb(o[2 * q + 1]) << 16);
			for (float v : {dt, dir.x(), dir.y(), dir.z(), light.x(), light.y(), light.z(), lg.gradient[0], lg.gradient[1], lg.gradient[2], lg.gradient[3], rgb_ray[0], rgb_ray[1], rgb_ray[2], rgb_ray[3],
			                rgb_ray2[0], rgb_ray2[1], rgb_ray2[2], rgb_ray2[3], weight_sum, weight_sum2, T, gws, loss_scale, ekw}) out.push_back(fb(v));
			for (float v : {r.alpha, r.T_after, r.weight_sum2_after, r.rgb_ray2_after[0], r.rgb_ray2_after[1], r.rgb_ray2_after[2], r.rgb_ray2_after[3]}) out.push_back(fb(v));
			for (int q = 0; q < 11; ++q) out.push_back(hb(r.dl[q]));
			out.push_back(fb(r.shading)); out.push_back(fb(r.ek_term));
			for (int q = 0; q < 10; ++q) out.push_back(fb(r.inter[q]));
		}
		arr_u("losssample_flags4_out8_in25_alpha_T_w2_rgb4_dl11_shading_ek_inter10", out);
	}
	{ // ---- the ray's loss terms between the two loops (testbed_nerf.cu:1735-1800)
		std::vector<uint32_t> out;
		for (int k = 0; k < 192; ++k) {
			const bool L2 = k % 3 != 2, rgbplus = k % 8 < 6, bce = k % 5 == 4;
			const Array4f target{uni(0, 1), uni(0, 1), uni(0, 1), uni(0, 2)};
			Array4f ray{uni(0, 1), uni(0, 1), uni(0, 1), uni(0, 2)};
			if (k % 16 == 3) ray[k % 4] = target[k % 4];
			const float aa = k % 6 == 0 ? 0.5f : (k % 6 == 1 ? 0.99f : 1.0f), an = k % 7 == 0 ? 0.0f : 1.0f;
			float ws = uni(0.0f, 1.05f);
			if (k % 16 == 5) ws = 0.0f;
			if (k % 16 == 6) ws = (float)(1.0 - 1e-4);
			if (k % 16 == 7) ws = 1e-4f;
			if (k % 16 == 8) ws = 0.99995f;
			const float mw = k % 2 ? 1.0f : 0.1f;
			const uint32_t n_rays = 4096 + 128 * (uint32_t)k;
			const RayLossTerms r = ray_loss_statements(target, ray, Array4f{0, 0, 0, aa}, Array4f{0, 0, 0, an}, ws, L2, rgbplus, bce, mw, n_rays);
			for (uint32_t v : {(uint32_t)L2, (uint32_t)rgbplus, (uint32_t)bce, fb(mw), n_rays}) out.push_back(v);
			for (int c = 0; c < 4; ++c) out.push_back(fb(target[c]));
			for (int c = 0; c < 4; ++c) out.push_back(fb(ray[c]));
			out.push_back(fb(aa)); out.push_back(fb(an)); out.push_back(fb(ws));
			out.push_back(fb(r.loss));
			for (int c = 0; c < 4; ++c) out.push_back(fb(r.gradient[c]));
			for (float v : {r.weight_sum, r.gradient_weight_sum, r.loss_row, r.mask_row}) out.push_back(fb(v));
		}
		arr_u("rayloss_L2_rgbplus_bce_maskweight_nrays_target4_ray4_albedoalpha_normalalpha_weightsum_loss_grad4_ws_gws_lossrow_maskrow", out);
	}
	{ // ---- the optimizer on single parameters: Adam (adam.h:52-202) then the EMA of the half weights (ema.h:63-78, 115-116), with the values of configs/nerf/base.json
		std::vector<uint32_t> out;
		auto hb = [](__half h) { uint16_t u; memcpy(&u, &h, 2); return (uint32_t)u; };
		const uint32_t n_matrix = 3072 + 8192;
		const float lr = """ + opt["learning_rate"] + r""", beta1 = """ + opt["beta1"] + r""", beta2 = """ + opt["beta2"] + r""", eps = """ + opt["epsilon"] + r""", l2 = """ + opt["l2_reg"] + r""", loss_scale = 128.0f, ema_decay = """ + opt["ema_decay"] + r"""; // configs/nerf/base.json
		for (uint32_t v : {fb(lr), fb(beta1), fb(beta2), fb(eps), fb(l2), fb(loss_scale), fb(ema_decay), n_matrix}) out.push_back(v);
		for (int k = 0; k < 256; ++k) {
			const bool is_matrix = k % 2 == 0;
			const uint32_t i = is_matrix ? (uint32_t)k : n_matrix + (uint32_t)k;
			const uint32_t steps_before[8] = {0, 1, 2, 9, 99, 2999, 65534, 70000};
			uint32_t step = steps_before[(k / 2) % 8];
			const uint32_t current_step = k % 16 == 0 ? 1u : step + 1 + (uint32_t)(gen.next_uint() % 50);   // the optimizer's own step count (>= any parameter's)
			float w = uni(-0.5f, 0.5f) * (is_matrix ? 1.0f : 1e-2f);
			__half g = (__half)(uni(-1, 1) * (k % 3 == 0 ? 60000.0f : (k % 3 == 1 ? 1.0f : 1e-4f)));
			if (k % 8 == 5 || k % 8 == 4) g = (__half)0.0f;                         // a hash-grid entry without gradient is left alone; an MLP weight still decays
			float m = step ? uni(-1e-3f, 1e-3f) : 0.0f, v = step ? uni(0, 1e-5f) : 0.0f;
			__half w16 = (__half)w, ema = (__half)(current_step > 1 ? w * uni(0.8f, 1.2f) : 0.0f);
			const uint32_t step0 = step; const float w0 = w, m0 = m, v0 = v; const __half w160 = w16, ema0 = ema;
			// arrays of one element addressed as element i
			tcnn::adam_step_element<__half>(i, 0, 0, n_matrix, 0, 0.0f, 0.0f, loss_scale, lr, 1.0f, true, true, true, true, beta1, beta2, eps, 0.0f, std::numeric_limits<float>::max(), l2,
			                                &w - i, &w16 - i, &g - i, &m - i, &v - i, &step - i, nullptr, false, false);
			tcnn::ema_step_element<__half>(i, ema_decay, current_step, &w16 - i, &ema - i);
			for (uint32_t x : {(uint32_t)is_matrix, step0, current_step, fb(w0), hb(w160), hb(g), fb(m0), fb(v0), hb(ema0), fb(w), hb(w16), fb(m), fb(v), step, hb(ema)}) out.push_back(x);
		}
		arr_u("adam_globals8_then_ismatrix_step_optstep_w_w16_g16_m_v_ema16_neww_neww16_newm_newv_newstep_newema16", out);
	}
	{ // ---- one (sample, level) of the hash-grid encoding (grid.h:168-364): features (half sums) and dy/dx (float) from a small table that travels with the item
		std::vector<uint32_t> out;
		auto hb = [](__half h) { uint16_t u; memcpy(&u, &h, 2); return (uint32_t)u; };
		const uint32_t shapes[8][2] = {{32, 3}, {216, 6}, {128, 5}, {64, 151}, {128, 971}, {256, 2049}, {256, 50}, {64, 4}}; // table entries, resolution: dense (27 -> 32, 216, 125 -> 128, 64) and hashed
		for (int k = 0; k < 160; ++k) {
			const uint32_t size = shapes[k % 8][0], res = shapes[k % 8][1];
			const float scale = (float)(res - 1);
			__half table[257 * 2];
			for (int q = 0; q < 257 * 2; ++q) table[q] = (__half)(q < (int)size * 2 ? uni(-1, 1) * (k % 3 == 0 ? 1e-4f : (k % 3 == 1 ? 0.1f : 4.0f)) : 0.0f);
			float xyz[3] = {gen.next_float(), gen.next_float(), gen.next_float()};
			if (k % 32 == 31) xyz[0] = 1.0f;
			if (k % 32 == 30) { xyz[1] = 0.0f; xyz[2] = 1.0f; }
			const uint32_t offsets[2] = {0, size};
			__half feat[2] = {(__half)7.0f, (__half)7.0f};
			float dydx[6] = {7, 7, 7, 7, 7, 7};
			tcnn::kernel_grid_element<__half, 3, 2>(0, 0, 1, 2, offsets, &res, &scale, 0, 0.0f, 1.0f, nullptr, tcnn::InterpolationType::Linear, tcnn::GridType::Hash, table,
			                                         tcnn::MatrixView<const float>(xyz, 1, 3), feat, dydx);
			for (uint32_t v : {size, res, fb(scale), fb(xyz[0]), fb(xyz[1]), fb(xyz[2])}) out.push_back(v);
			for (int q = 0; q < 257; ++q) out.push_back(hb(table[2 * q]) | hb(table[2 * q + 1]) << 16);
			out.push_back(hb(feat[0])); out.push_back(hb(feat[1]));
			for (int q = 0; q < 6; ++q) out.push_back(fb(dydx[q]));
		}
		arr_u("encode_size_res_scale_xyz_table257_f0_f1_dydx6", out);
	}
	{ // ---- the samples of two occupancy updates (testbed_nerf.cu:3424-3494 drives generate_grid_samples_nerf_nonuniform): the first update of a run (every cell of the zeroed grid)
	  //      and a later one (n/4 samples anywhere, then n/4 in cells above NERF_MIN_OPTICAL_THICKNESS) over the grid pattern cell c -> 0.5 if (c * 2654435761 >> 29) == 0 else 0
		std::vector<uint32_t> out;
		const uint32_t cells = NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_GRIDSIZE();
		std::vector<float> grid(cells, 0.0f);
		tcnn::pcg32 m_rng{1337};                                        // Testbed::reset_network (testbed.cu:2223-2237): seed 1337, the grid's generator seeded with its first draw
		tcnn::pcg32 density_grid_rng{m_rng.next_uint()};
		BoundingBox box{Vector3f::Constant(0.0f), Vector3f::Constant(1.0f)};
		// the kernel advances its generator by i * 4 itself and indexes out[i] / indices[i]: call it with the true i and arrays shifted by -i
		auto run_true = [&](const uint32_t n_elements, const uint32_t step, const float thresh, const uint32_t offset, const uint32_t call) {
			for (int k = 0; k < 512; ++k) {
				const uint32_t i = k < 4 ? (uint32_t)k : (k < 8 ? n_elements - 1 - (uint32_t)(k - 4) : gen.next_uint() % n_elements);
				NerfPosition p{Vector3f::Zero(), 0.0f};
				uint32_t idx = 0;
				grid_sample_element(i, n_elements, density_grid_rng, step, box, grid.data(), &p - i, &idx - i, 1, thresh);
				out.push_back(call); out.push_back(offset + i); out.push_back(idx); out.push_back(fb(p.p.x())); out.push_back(fb(p.p.y())); out.push_back(fb(p.p.z()));
			}
		};
		run_true(cells, 0, -0.01f, 0, 0);                                 // update 1 (training step 0): n_uniform = all cells, n_nonuniform = 0; density_grid_ema_step 0
		density_grid_rng.advance(); density_grid_rng.advance();           // (both launches are followed by an advance, the empty one too)
		for (uint32_t c = 0; c < cells; ++c) grid[c] = ((c * 2654435761u) >> 29) == 0 ? 0.5f : 0.0f;
		run_true(cells / 4, 1, -0.01f, 0, 1);                             // update 2 (training step >= 256): density_grid_ema_step 1
		density_grid_rng.advance();
		run_true(cells / 4, 1, NERF_MIN_OPTICAL_THICKNESS(), cells / 4, 2);
		arr_u("gridsamples_call_slot_idx_pos3", out);
	}
	{ // ---- occupancy grid -> bitfield + 7 mips (testbed_nerf.cu:3496-3517), two grid patterns whose mean is exact in any summation order (values are multiples of 2^-10):
	  //      cell c -> table[(c * 2654435761) >> 29], sparse (mean < 0.1: the threshold is the mean) and dense (mean > 0.1: the threshold is NERF_MIN_OPTICAL_THICKNESS)
		std::vector<uint32_t> out;
		const uint32_t n_elements = NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_GRIDSIZE();
		const float tables[2][8] = {{0.5f, 0.0f, 0.0f, -1.0f, 0.0625f, 0.0f, 0.0f, 0.0f}, {0.5f, 0.5f, 0.09765625f, -1.0f, 0.1015625f, 0.5f, 0.5f, 0.0f}};
		for (int pat = 0; pat < 2; ++pat) {
			std::vector<float> grid(n_elements);
			double sum = 0.0;
			for (uint32_t c = 0; c < n_elements; ++c) { grid[c] = tables[pat][(c * 2654435761u) >> 29]; sum += fmaxf(grid[c], 0.f) / (n_elements); }
			const float mean = (float)sum; // = what reduce_sum leaves in any order: every term and every partial sum is exact
			std::vector<uint8_t> bits(grid_mip_offset(NERF_CASCADES()) / 8, 0);
			for (uint32_t i = 0; i < n_elements / 8 * NERF_CASCADES(); ++i) grid_to_bitfield_element(i, n_elements / 8 * 1, grid.data(), bits.data(), &mean);
			for (uint32_t level = 1; level < NERF_CASCADES(); ++level)
				for (uint32_t i = 0; i < n_elements / 64; ++i) bitfield_max_pool_element(i, bits.data() + grid_mip_offset(level - 1) / 8, bits.data() + grid_mip_offset(level) / 8);
			out.push_back((uint32_t)pat); out.push_back(fb(mean));
			for (int q = 0; q < 8; ++q) out.push_back(fb(tables[pat][q]));
			for (uint32_t level = 0; level < NERF_CASCADES(); ++level) { // per mip: set bits, and a position-weighted checksum of its bytes
				uint32_t set = 0, chk = 0;
				const uint8_t* b = bits.data() + grid_mip_offset(level) / 8;
				for (uint32_t i = 0; i < n_elements / 8; ++i) { set += (uint32_t)__builtin_popcount(b[i]); chk += (uint32_t)b[i] * (i * 2654435761u + 1u); }
				out.push_back(set); out.push_back(chk);
			}
		}
		arr_u("bitfield_pattern_mean_table8_then_setbits_checksum_per_mip", out);
	}
	{ // ---- a ray through the occupancy bitfield (testbed_nerf.cu:1330-1380): how many samples, and the NerfCoordinates written (7 floats each: warped position, warped dt, warped direction);
	  //      bitfield byte i = pcg32{5} draw i >> 24, as for the int fixtures' march items
		std::vector<uint32_t> out;
		static_assert(sizeof(NerfCoordinate) == 28, "NerfCoordinate is seven floats");
		std::vector<uint8_t> bf(grid_mip_offset(NERF_CASCADES()) / 8);
		{ tcnn::pcg32 q{5}; for (auto& b : bf) b = (uint8_t)(q.next_uint() >> 24); }
		for (int k = 0; k < 192; ++k) {
			const float lo = k % 3 == 2 ? -1.5f : 0.0f, hi = k % 3 == 2 ? 2.5f : 1.0f, cone = k % 3 == 2 ? 1.0f / 256.0f : 0.0f;   // one cascade with the constant step; four with cone stepping
			BoundingBox box{Vector3f::Constant(lo), Vector3f::Constant(hi)};
			Vector3f o{uni(lo, hi), uni(lo, hi), uni(lo, hi)};
			Vector3f d = Vector3f{uni(-1, 1), uni(-1, 1), uni(-1, 1)}.normalized();
			if (k % 16 == 7) d = Vector3f{1.0f, 0.0f, 0.0f};
			if (k % 16 == 8) { d = Vector3f{0.0f, -1.0f, 0.0f}; o.y() = hi; }      // starts ON the box's face
			const float startt = k % 16 == 8 ? 0.0f : uni(0.0f, 0.02f);
			const MarchResult r = march_statements(o, d, startt, cone, box, bf.data());
			for (float v : {lo, hi, cone, o.x(), o.y(), o.z(), d.x(), d.y(), d.z(), startt}) out.push_back(fb(v));
			out.push_back(r.numsteps);
			uint32_t chk = 0;
			for (float v : r.coords) chk += fb(v);
			out.push_back(chk);
			for (int q = 0; q < 14; ++q) out.push_back(q < (int)r.coords.size() ? fb(r.coords[q]) : 0u);                                 // the first two samples
			for (int q = 0; q < 7; ++q) out.push_back(r.numsteps ? fb(r.coords[(size_t)(r.numsteps - 1) * 7 + q]) : 0u);               // the last one
		}
		arr_u("marchray_lo_hi_cone_o3_d3_startt_numsteps_checksum_first14_last7", out);
	}
	{ // ---- the next step's rays_per_batch from this step's compacted sample count (testbed_nerf.cu:3554-3555)
		std::vector<uint32_t> out;
		for (int k = 0; k < 256; ++k) {
			const uint32_t target = k % 5 == 4 ? (1u << 16) : (1u << 18);
			const uint32_t rays = 128u * (1u + gen.next_uint() % (k % 2 ? 2048u : 128u));
			uint32_t measured = 1u + gen.next_uint() % (k % 3 == 0 ? (1u << 14) : (1u << 19));
			if (k < 4) measured = target;                       // on target: unchanged up to the granularity
			if (k >= 4 && k < 8) measured = target / 2 + (uint32_t)k;
			out.push_back(rays); out.push_back(target); out.push_back(measured); out.push_back(controller_statements(rays, target, measured));
		}
		arr_u("controller_rays_target_measured_nextrays", out);
	}
	{ // ---- SDF -> density of the occupancy grid (common_operation.cuh:311-328): density = s sigmoid(sdf s) (1 - sigmoid(sdf s)), s = exp(10 variance), every operation in half
		std::vector<uint32_t> out;
		auto hb = [](__half h) { uint16_t u; memcpy(&u, &h, 2); return (uint32_t)u; };
		for (int k = 0; k < 512; ++k) {
			__half sdf = (__half)(uni(-1, 1) * (k % 4 == 0 ? 0.5f : (k % 4 == 1 ? 0.02f : 0.002f)));
			if (k % 32 == 5) sdf = (__half)0.0f;
			__half var = (__half)(k % 8 == 7 ? 1.2f : uni(0.05f, 0.8f));   // 1.2: s = exp(12) overflows half
			__half io = sdf;
			sdf_to_density_element<__half>(0, &var, tcnn::MatrixView<__half>(&io, 1, 1));
			out.push_back(hb(sdf)); out.push_back(hb(var)); out.push_back(hb(io));
		}
		arr_u("sdfdensity_sdf16_variance16_density16", out);
	}
	{ // ---- which training steps begin with an occupancy update (src/testbed.cu:2805-2806): every step up to 31, then every 2nd, 3rd, ... 16th
		std::vector<uint32_t> out;
		for (uint32_t step = 0; step < 700; ++step) { uint32_t skip, due; prep_statements(step, &skip, &due); out.push_back(step); out.push_back(due); out.push_back(skip); }
		for (uint32_t step : {1000u, 1008u, 4095u, 4096u, 65535u, 65536u, 1000000u, 4294967280u, 4294967295u}) { uint32_t skip, due; prep_statements(step, &skip, &due); out.push_back(step); out.push_back(due); out.push_back(skip); }
		arr_u("prep_step_due_skip", out);
	}
	{ // ---- the learning-rate factor the k-th optimizer step runs with (exponential_decay.h:61-72), configs/nerf/base.json's schedule and a dense one
		std::vector<uint32_t> out;
		const uint32_t cfgs[2][2] = {{""" + dec["decay_start"] + r"""u, """ + dec["decay_interval"] + r"""u}, {3u, 2u}};
		const float bases[2] = {""" + dec["decay_base"] + r""", 0.5f};
		for (int c = 0; c < 2; ++c) {
			DecaySchedule d{1.0f, cfgs[c][0], cfgs[c][1], 10000000u, bases[c], 0};
			const uint32_t last = c == 0 ? 60001u : 40u;
			for (uint32_t k = 0; k <= last; ++k) {
				d.steps_taken = k;
				d.before_nested_step();
				const bool record = c == 1 || k < 3 || (k % 10000u) <= 1u || (k % 10000u) == 9999u;
				if (record) { out.push_back(cfgs[c][0]); out.push_back(cfgs[c][1]); out.push_back(fb(bases[c])); out.push_back(k); out.push_back(fb(d.m_learning_rate_factor)); }
			}
		}
		arr_u("lrdecay_start_interval_base_step_factor", out, true);
	}
	printf("}\n");
	return 0;
}
""")
    return "\n".join(parts)


def main():
    prog = build_program()
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "float_fixtures.cpp")
        open(src, "w").write(prog)
        exe = os.path.join(d, "float_fixtures")
        subprocess.check_call([CXX, "-O1", "-std=c++17", "-ffp-contract=off", "-Xclang", "-ffloat16-excess-precision=none", "-DEIGEN_DONT_VECTORIZE", "-w", "-I", os.path.join(REF, "dependencies", "eigen"), src, "-o", exe])
        text = subprocess.check_output([exe]).decode()
    data = json.loads(text)
    data = {"_source": "tests/golden/make_float_fixtures.py: floating-point fragments of /root/reference compiled for the host with clang++ (-ffp-contract=off, -DEIGEN_DONT_VECTORIZE as Eigen configures itself under a GPU compiler) in the build container "
                       "(see the script's header); floats as IEEE-754 bit patterns", **data}
    out = os.path.join(HERE, "float_fixtures.json")
    with open(out, "w") as f:
        json.dump(data, f, separators=(",", ":"))
    print("wrote", out, {k: len(v) for k, v in data.items() if isinstance(v, list)}, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
