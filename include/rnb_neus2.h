/*
 * rnb_neus2.h — C-ABI of the MI355X-native RNb-NeuS2 training hot path.
 *
 * The reference (RobinBruneau/RNb-NeuS2) has no FFI for this path: the hot
 * path sits behind the in-process C++ interface tcnn::Network<float, half>
 * (dependencies/neus2_tcnn/include/tiny-cuda-nn/object.h:96-295) and the
 * Testbed member functions that drive it (src/testbed_nerf.cu). This header
 * is the plain-C boundary a maintainer would bind in their place. Every entry
 * point cites the reference interface it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, a negative rnb_status otherwise;
 *    no exception crosses the boundary; rnb_last_error() gives the message.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *    device work is enqueued on it; functions documented "syncs" block.
 *  - pointers named *_dev are device pointers, *_host host pointers.
 *  - a context is not thread-safe; use one context per GPU / per process.
 *
 * The same signatures, with the prefix orc_ instead of rnb_ and host
 * pointers everywhere, are exported by oracle/liborc.so (the CPU checker).
 */
#ifndef RNB_NEUS2_H
#define RNB_NEUS2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): RNB_BUF_PARAMS_FP32 / ADAM_M / ADAM_V / ADAM_STEPS became staging views (see rnb_buffer); buffer ids 25, 26 and
 * rnb_bitfield_changed were added after 1 without a bump -- a binary built against 1 must not link silently.
 * 3 (round 4): rnb_shard_layout fills up to RNB_MAX_SHARD_PARTS = 3 blocks (2: two) -- a caller's array must have room for them.
 * 4 (round 5): rnb_config::accumulate (taken from `reserved`; 0 keeps the behaviour of ABI 3) and RNB_BUF_GRADS_FP16.
 * 5 (round 6): rnb_config::deterministic (taken from `reserved`; 0 keeps the behaviour of ABI 4). */
#define RNB_ABI_VERSION 5

typedef enum rnb_status {
	RNB_OK = 0,
	RNB_ERR_INVALID = -1,   /* bad argument / bad state */
	RNB_ERR_DEVICE = -2,    /* HIP runtime error */
	RNB_ERR_NOMEM = -3,
	RNB_ERR_NO_SAMPLES = -4 /* "Nerf training generated 0 samples" (src/testbed_nerf.cu:3662-3668) */
} rnb_status;

/* Fixed architecture of the reference (configs/nerf/base.json, nerf_network.h:40-83). */
#define RNB_N_POS_DIMS 3
#define RNB_COORD_FLOATS 7          /* NerfCoordinate: pos3, dt, dir3 (nerf.h:76-102) */
#define RNB_OUT_WIDTH 16            /* padded network output width */
#define RNB_SDF_IN 32
#define RNB_SDF_HID 64
#define RNB_SDF_OUT 16
#define RNB_RGB_IN 48
#define RNB_RGB_HID 64
#define RNB_RGB_OUT 16
#define RNB_N_SDF_MLP_PARAMS (RNB_SDF_HID * RNB_SDF_IN + RNB_SDF_OUT * RNB_SDF_HID)                       /* 3072 */
#define RNB_N_RGB_MLP_PARAMS (RNB_RGB_HID * RNB_RGB_IN + RNB_RGB_HID * RNB_RGB_HID + RNB_RGB_OUT * RNB_RGB_HID) /* 8192 */
#define RNB_N_VARIANCE_PARAMS 4
#define RNB_GRIDSIZE 128            /* NERF_GRIDSIZE (nerf.h:24-26) */
#define RNB_CASCADES 8              /* NERF_CASCADES (testbed_nerf.cu:50) */
#define RNB_MAX_STEPS 1024          /* NERF_STEPS (testbed_nerf.cu:49) */
#define RNB_MAX_LEVELS 16

/* Mirrors the values the reference reads from configs/nerf/base.json, the CLI
 * flags of src/main.cu:83-258 that reach the loss kernel, and testbed.h defaults. */
typedef struct rnb_config {
	uint32_t abi_version;             /* = RNB_ABI_VERSION */
	/* hash grid — base.json:30-40, grid.h:946-1027 */
	uint32_t n_levels;                /* 14 */
	uint32_t log2_hashmap_size;       /* 19 */
	uint32_t base_resolution;         /* 16 */
	float    per_level_scale;         /* exp(log(top_resolution*aabb_scale/base)/(L-1)), testbed.cu:2320-2323 */
	float    valid_level_scale;       /* 0.02 */
	float    base_valid_level_scale;  /* 0.2 */
	uint32_t base_training_step;      /* 100 */
	float    sdf_bias;                /* -0.1, nerf_network.h:74 */
	/* batch geometry — testbed.h:908, testbed.cu:2229, testbed_nerf.cu:3554-3555 */
	uint32_t target_batch_size;       /* 1<<18 */
	uint32_t initial_rays_per_batch;  /* 1<<12 */
	uint32_t max_rays_per_batch;      /* 1<<18 */
	uint32_t aabb_scale;              /* 1 (power of two) */
	uint32_t seed;                    /* 1337, testbed.h:550 */
	/* loss — testbed.h:490-521, main.cu:349-410 */
	float    mask_loss_weight;        /* --mask-weight */
	float    ek_loss_weight;          /* 0.01 */
	uint32_t apply_L2;                /* !--lone */
	uint32_t apply_rgbplus;           /* !--no-rgbplus */
	uint32_t apply_no_albedo;         /* --no-albedo */
	uint32_t apply_light_opti;        /* --opti-lights */
	uint32_t apply_supernormal;       /* --supernormal */
	uint32_t apply_relu;              /* --relu */
	uint32_t apply_bce;               /* --bce */
	uint32_t snap_to_pixel_centers;   /* 1 unless --disable-snap-to-center */
	/* optimizer — base.json:5-28 */
	float    learning_rate;           /* 1e-3 */
	float    beta1, beta2, epsilon;   /* 0.9, 0.99, 1e-15 */
	float    l2_reg;                  /* 1e-6 */
	float    ema_decay;               /* 0.95 */
	uint32_t lr_decay_start;          /* 20000 */
	uint32_t lr_decay_interval;       /* 10000 */
	float    lr_decay_base;           /* 0.33 */
	float    density_grid_decay;      /* 0.95, testbed.h:671 */
	/* data parallel: this process is `rank` of `world_size`; see DESIGN.md §multi-GPU */
	uint32_t world_size;              /* 1 */
	uint32_t rank;                    /* 0 */
	uint32_t only_sdf_training;       /* 0; Adam skips the colour MLP (adam.h:121-165), set by --fractional-training (testbed.cu:1886-1895) */
	uint32_t overlap;                 /* 1; run independent stages of consecutive steps on side streams (same results); 0 = strictly serial */
	uint32_t accumulate;              /* rnb_accumulate: the width of the accumulators. RNB_ACCUM_FP32 (0, default): every dot product and gradient sum in fp32, narrowed to
	                                     half where the reference stores half. RNB_ACCUM_HALF (1): the reference's arithmetic as coded -- the MLPs' dot products round their
	                                     accumulator to half after every 16-wide k-step (WMMA half fragments, fully_fused_mlp.cu:59-68, 198), the hash-grid gradients are summed
	                                     by atomicAdd(__half2) into a half gradient vector (grid.h:410-430, trainer.h:78-84: RNB_BUF_GRADS_FP16 replaces RNB_BUF_GRADS_FP32).
	                                     The weight-gradient GEMMs follow tcnn's CUTLASS split-K order (slices of 4096 samples, half accumulators rounded after every 16-sample
	                                     k-step, the slices reduced in half: cutlass_matmul.h:83, 315-322; RNB_PRIM_DW_SLICED). Fixed at rnb_create. One stated departure from the
	                                     reference's code in this mode: the default scatter sums a cell run / a workgroup's slice in fp32 before its one
	                                     packed half atomic (RNB_SCATTER_PLAIN=1 issues the reference's own sequence: every addend its own atomicAdd(__half2)). */
	uint32_t deterministic;           /* 0 (default): the hash-grid gradients are summed by floating-point atomics, as in the reference (grid.h:410-430) -- the sum depends on the order
	                                     the hardware retires them in, so two runs from one state differ in the last bits and a training run is not reproducible (neither is the
	                                     reference's: src/testbed_nerf.cu:1352, 1557-1561). 1: every addend -- rounded to half first exactly as the reference rounds it (grid.h:415-416),
	                                     hence a multiple of 2^-24 -- is added as a 64-bit FIXED-POINT INTEGER (scale 2^24; integer atomics commute, the sum is exact and independent
	                                     of any order), and the sum is narrowed ONCE into the gradient vector of the accumulate mode (fp32 accumulators, or half). Every other stage
	                                     is order-independent already (prefix-sum slots, fixed-order weight-gradient and loss sums, atomicMax splat), so with it a training run is
	                                     bit-reproducible: same state in, same bits out, on any schedule (overlap on or off) and -- the sums being exact -- for any split of the
	                                     batch over workgroups. Works with both accumulate modes and with the data-parallel entry points (the gradient vector they describe is
	                                     complete, narrowed, at the same points). Fixed at rnb_create. */
	uint32_t reserved[4];
} rnb_config;
typedef enum rnb_accumulate { RNB_ACCUM_FP32 = 0, RNB_ACCUM_HALF = 1 } rnb_accumulate;

/* One training view — TrainingImageMetadata + TrainingXForm (nerf_loader.h:33-49). */
typedef struct rnb_view {
	uint32_t width, height;
	float focal_length[2];     /* pixels */
	float principal_point[2];  /* normalised to [0,1] */
	float xform[12];           /* camera-to-world 3x4, row-major: xform[r*4+c] */
} rnb_view;

typedef struct rnb_step_stats {
	uint32_t training_step;            /* value after the step */
	uint32_t rays_per_batch;           /* rays generated by this step (per rank) */
	uint32_t next_rays_per_batch;      /* controller output, testbed_nerf.cu:3554-3555 */
	uint32_t measured_batch_size;      /* compacted samples */
	uint32_t measured_batch_size_before_compaction;
	uint32_t n_rays_kept;              /* rays that survived marching (ray_counter) */
	uint32_t density_grid_updated;     /* 1 when training_prep_nerf ran this step */
	float loss, ek_loss, mask_loss;    /* Counters::update_after_training scalars */
	float prep_ms, step_ms;            /* the two ScopeGuard timers, testbed.cu:2807-2810, 2853-2856 */
} rnb_step_stats;

typedef struct rnb_ctx rnb_ctx;

/* Buffers reachable through rnb_buffer(). Device pointers for librnb_neus2_hip, host for liborc. */
typedef enum rnb_buffer_id {
	RNB_BUF_PARAMS_FP32 = 0,   /* float[n_params]  master weights (trainer.h:78-84) */
	RNB_BUF_PARAMS_FP16 = 1,   /* half[n_params]   training weights */
	RNB_BUF_PARAMS_EMA = 2,    /* half[n_params]   EMA = inference weights (ema.h:63-78) */
	RNB_BUF_GRADS_FP32 = 3,    /* float[n_params]  gradient accumulators (x loss_scale 128); accumulate = RNB_ACCUM_HALF: not there (RNB_ERR_INVALID), see RNB_BUF_GRADS_FP16 */
	RNB_BUF_ADAM_M = 4,        /* float[n_params] */
	RNB_BUF_ADAM_V = 5,        /* float[n_params] */
	RNB_BUF_ADAM_STEPS = 6,    /* uint32[n_params] */
	RNB_BUF_DENSITY_GRID = 7,  /* float[128^3 * (max_cascade+1)]; a caller that writes it other than through rnb_memcpy follows with rnb_update_density_bitfield */
	RNB_BUF_DENSITY_BITFIELD = 8, /* uint8[128^3/8 * 8 mips] */
	RNB_BUF_DENSITY_MEAN = 9,  /* float[1] */
	RNB_BUF_RAY_INDICES = 10,  /* uint32[rays] */
	RNB_BUF_RAYS = 11,         /* float[rays*6] origin, unnormalised dir */
	RNB_BUF_NUMSTEPS = 12,     /* uint32[rays*2] (numsteps, base) */
	RNB_BUF_COORDS = 13,       /* float[16*B*7] */
	RNB_BUF_MLP_OUT = 14,      /* half[16*B*16] */
	RNB_BUF_DLOSS_DOUT = 15,   /* half[B*16] */
	RNB_BUF_COORDS_COMPACTED = 16, /* float[B*7] */
	RNB_BUF_LOSS = 17,         /* float[rays] */
	RNB_BUF_EK_LOSS = 18,
	RNB_BUF_MASK_LOSS = 19,
	RNB_BUF_COUNTERS = 20,     /* uint32[4]: numsteps_counter, numsteps_counter_compacted, ray_counter, samples_written */
	RNB_BUF_DENSITY_GRID_TMP = 21, /* float[128^3 * (max_cascade+1)] */
	RNB_BUF_GRID_SAMPLE_POS = 22,  /* float[n*3] positions of the last density-grid update */
	RNB_BUF_GRID_SAMPLE_IDX = 23,  /* uint32[n] */
	RNB_BUF_STEP_VECTOR = 24,  /* double[7]: this rank's {counters[0..3], loss sums[0..2]} of the running step, final once the loss pass is
	                              (rnb_train_step_local has returned); data-parallel callers all-reduce it in place */
	RNB_BUF_GRID_SAMPLE_POS_EVAL = 25, /* float[n*3] / uint32[n]: the same samples in the order the last update evaluated them (cell order, prepared an */
	RNB_BUF_GRID_SAMPLE_IDX_EVAL = 26, /* update interval ahead); 0 bytes when it evaluated them in the reference's order (GRID_SAMPLE_POS / _IDX). HIP library only */
	RNB_BUF_GRADS_FP16 = 27,   /* half[n_params]: the gradient vector of accumulate = RNB_ACCUM_HALF (x loss_scale 128); RNB_ERR_INVALID in the fp32 mode. The data-parallel
	                              entry points (rnb_gradient_parts, rnb_shard_layout ...) then describe ranges of THIS buffer and the caller sums halfs */
	RNB_BUF_COUNT
} rnb_buffer_id;
/* OR-ed into the buffer id: the caller only reads through the pointer (see rnb_buffer). */
#define RNB_BUF_READONLY 0x100

typedef enum rnb_memcpy_kind { RNB_H2D = 0, RNB_D2H = 1, RNB_D2D = 2 } rnb_memcpy_kind;

/* ---- lifecycle -------------------------------------------------------- */
const char* rnb_last_error(void);
uint32_t rnb_abi_version(void);
/* Fills *cfg with the defaults of configs/nerf/base.json + testbed.h + main.cu (no CLI flags given). */
int rnb_default_config(rnb_config* cfg);
/* Testbed::reset_network (src/testbed.cu:2220-2485): allocates parameters, optimizer state,
 * occupancy grid and step scratch on the current HIP device. */
int rnb_create(const rnb_config* cfg, rnb_ctx** out);
int rnb_destroy(rnb_ctx* ctx);
/* Re-reads the run-time switches from cfg — the Testbed setters of src/testbed.cu:255-323 (apply_L2, apply_no_albedo,
 * set_mask_weight, ...), the optimizer hyper-parameters re-applied every step (testbed.cu:2823-2835) and
 * only_sdf_training. Geometry fields (levels, batch size, seed, ranks) must be unchanged. */
int rnb_update_config(rnb_ctx* ctx, const rnb_config* cfg);

/* ---- parameters ------------------------------------------------------- */
/* NerfNetwork::n_params (nerf_network.h:722-724): 3072 + 8192 + grid + 4. */
uint64_t rnb_n_params(const rnb_ctx* ctx);
/* Offsets (in elements) of [sdf mlp | rgb mlp | hash grid | variance] (nerf_network.h:539-583). */
int rnb_param_layout(const rnb_ctx* ctx, uint64_t offsets[5]);
/* Per-level tables (grid.h:977-1012): hashmap offsets (n_levels+1, in entries), resolution, scale. */
int rnb_grid_tables(const rnb_ctx* ctx, uint32_t* offsets, uint32_t* resolution, float* scale);
/* Trainer::initialize_params + NerfNetwork::initialize_params (trainer.h:72-109, nerf_network.h:625-696):
 * PCG32 streams identical to the reference; sdf_mlp_weights_host = the 3072 floats of
 * utils/mlp_weights_hidden_layer_num_1_hidden_size_32.txt. Syncs. */
int rnb_init_params(rnb_ctx* ctx, const float* sdf_mlp_weights_host);
/* Trainer::deserialize-like: overwrite fp32 master weights from host, re-derive fp16/EMA copies,
 * reset Adam state (trainer.h:263-275). Syncs. */
int rnb_set_params(rnb_ctx* ctx, const float* params_host);
/* Pointer + size of a context buffer. Reads need nothing further, with one exception; writes through a kept pointer must be announced,
 * because the kernels work from cached forms of three buffers:
 *   RNB_BUF_PARAMS_FP16        written  -> rnb_params_changed            (LDS weight images)
 *   RNB_BUF_DENSITY_BITFIELD   written  -> rnb_bitfield_changed          (LDS occupancy of the march, batches generated ahead)
 *   RNB_BUF_DENSITY_GRID       written  -> through rnb_memcpy, or followed by rnb_update_density_bitfield (occupancy samples prepared ahead)
 * rnb_buffer(id) without RNB_BUF_READONLY is itself treated as the announcement of a write that happens BEFORE the caller's next call into
 * the library (the cached forms are dropped on the spot: conservative, and what a binary written against ABI 1 expects); a pointer KEPT
 * across library calls and written later needs the explicit call. rnb_buffer(id | RNB_BUF_READONLY) has no side effect on these three.
 * The exception: RNB_BUF_PARAMS_FP32 / ADAM_M / ADAM_V / ADAM_STEPS are STAGING VIEWS. The optimizer keeps these four as one 64-byte
 * record per 4-parameter group; rnb_buffer() on any of them synchronises the device, unpacks the records into the four
 * plain arrays and returns the array asked for. The contents are a snapshot: call rnb_buffer() again after any call that trains. Writes
 * through the pointer (rnb_memcpy or the caller's own copies / collectives, complete before the next rnb_train_step* / rnb_optimizer_step
 * call returns control to the device, i.e. queued on any stream by then) are honoured: unless RNB_BUF_READONLY was given, the next
 * optimizer launch synchronises the device and packs all four arrays back into the records. */
int rnb_buffer(rnb_ctx* ctx, int buffer_id, void** ptr, uint64_t* n_bytes);
/* A caller that keeps a pointer from rnb_buffer(RNB_BUF_PARAMS_FP16) and writes training weights through it later (e.g. an
 * all-gather of a sharded optimizer) says so here: the kernels' cached LDS weight images are dropped and rebuilt from the
 * weights at the next launch (rnb_train_step_apply_done rebuilds them). No reference counterpart: tcnn reads the weights from global memory on every launch (fully_fused_mlp.cu:624-758). */
int rnb_params_changed(rnb_ctx* ctx);
/* The occupancy bitfield was written through a pointer kept from rnb_buffer(RNB_BUF_DENSITY_BITFIELD): the march kernels' LDS form
 * of it is rebuilt in front of the next march and a batch already generated ahead of time with the old bits is dropped.
 * Reading the buffer needs no call. (none in the reference: its march reads the bitfield from global memory,
 * src/testbed_nerf.cu:1320-1345; the LDS form is this build's.) */
int rnb_bitfield_changed(rnb_ctx* ctx);
/* Caller-owned device scratch (GPUMemory<T> in the reference, e.g. the lattice of get_density_on_grid,
 * src/testbed_nerf.cu:4218-4269). Host memory in the CPU checker. */
int rnb_device_malloc(rnb_ctx* ctx, uint64_t n_bytes, void** ptr);
int rnb_device_free(rnb_ctx* ctx, void* ptr);
int rnb_memcpy(rnb_ctx* ctx, void* dst, const void* src, uint64_t n_bytes, int kind); /* syncs */

/* ---- dataset ---------------------------------------------------------- */
/* NerfDataset as uploaded by Testbed::load_nerf (src/testbed_nerf.cu:3078-3218): per view one RGBA16
 * normal map and one RGBA16 albedo map (8 bytes/pixel, row-major, alpha = mask). Host pointers; copies. Syncs. */
int rnb_set_dataset(rnb_ctx* ctx, uint32_t n_views, const rnb_view* views,
                    const uint16_t* const* normals_host, const uint16_t* const* albedos_host);

/* ---- hot-path stages (each replaces the cited reference function) ------ */
/* GridEncoding::set_training_step + NerfNetwork::m_training_step (grid.h:1430-1437, testbed.cu:2787-2793). */
int rnb_set_training_step(rnb_ctx* ctx, uint32_t step);
uint32_t rnb_valid_level(const rnb_ctx* ctx);
/* Testbed::training_prep_nerf -> update_density_grid_nerf (testbed_nerf.cu:4125-4138, 3424-3495), K1-K5. */
int rnb_update_density_grid(rnb_ctx* ctx, void* stream);
/* Data parallel: the occupancy update sharded over the ranks. The update evaluates the network on a SET of 2^20 sample points and splats the densities
 * with atomicMax into DENSITY_GRID_TMP (testbed_nerf.cu:616-635); with world_size > 1
 *     rnb_update_density_grid_begin   K1 (all samples: every rank draws the same ones) + K2-K3 on samples [rank n / W, (rank + 1) n / W) of the evaluation order
 *     element-wise MAX of DENSITY_GRID_TMP over the ranks (float[128^3 (max_cascade + 1)]) taken on the words as UINT32, the order of the single-rank atomicMax
 *                                     (testbed_nerf.cu:634; values are >= 0 or NaN, and a NaN with its sign bit set must win as it does on one rank). The pointer
 *                                     behind RNB_BUF_DENSITY_GRID_TMP changes from update to update (two targets swap roles): fetch it after every _begin
 *     rnb_update_density_grid_end     K4-K5: EMA, mean, bitfield, pools
 * leaves every rank with the grid a single rank computes, bit for bit, for 1 / W of the network evaluations. rnb_update_density_grid = begin; [the exchange]; end,
 * where the exchange is the callback of rnb_set_grid_exchange (stream-ordered work on `stream`, e.g. one ncclAllReduce(ncclMax); returns 0 on success) -- which is
 * also how the training step (rnb_train_step_begin) shards its updates. Without a callback the update stays replicated (every rank evaluates every sample: no
 * communication). With world_size 1 begin / end are the two halves of rnb_update_density_grid. No reference counterpart (the reference is single-GPU). */
typedef int (*rnb_grid_exchange_fn)(void* user, void* grid_tmp, uint64_t n_elements, void* stream);
int rnb_set_grid_exchange(rnb_ctx* ctx, rnb_grid_exchange_fn fn, void* user);
int rnb_update_density_grid_begin(rnb_ctx* ctx, void* stream);
int rnb_update_density_grid_end(rnb_ctx* ctx, void* stream);
/* Testbed::update_density_grid_mean_and_bitfield (testbed_nerf.cu:3497-3517), K5 from the current grid. */
int rnb_update_density_bitfield(rnb_ctx* ctx, void* stream);
/* NerfNetwork::density (nerf_network.h:522-537): xyz[n,3] f32 -> density half[n], training weights. */
int rnb_density(rnb_ctx* ctx, void* stream, const float* xyz_dev, uint32_t n, uint16_t* out_dev, int use_inference_params);
/* NerfNetwork::sdf (nerf_network.h:454-520): xyz[n,3] -> sdf+bias half[n] (EMA weights for meshing). */
int rnb_sdf(rnb_ctx* ctx, void* stream, const float* xyz_dev, uint32_t n, uint16_t* out_dev, int use_inference_params);
/* Network::inference_mixed_precision -> NerfNetwork::forward_impl (nerf_network.h:87-253):
 * coords[n,7] f32 -> out[n,16] half. */
int rnb_forward_infer(rnb_ctx* ctx, void* stream, const float* coords_dev, uint32_t n, uint16_t* out_dev, int use_inference_params);
/* get_density_on_grid (src/testbed_nerf.cu:4218-4269) for the SDF: the lattice res[0] x res[1] x res[2] (x fastest) of the cube
 * [lattice_min, lattice_max)^3 -- point (x,y,z) at lattice_min + (x,y,z) / res * (lattice_max - lattice_min), as
 * generate_grid_samples_nerf_uniform places it (src/testbed_nerf.cu:541-553) -- evaluated into `out` (float[res^3], device) as
 * rnb_sdf does (sdf + bias). Syncs. */
int rnb_sdf_lattice(rnb_ctx* ctx, void* stream, const uint32_t res[3], float lattice_min, float lattice_max, float* out, int inference);
/* marching_cubes_gpu (src/marching_cubes.cu:794-822; gen_vertices :276-327, gen_faces :377-430/676-717) on a device lattice:
 * one vertex per lattice edge that crosses `thresh` (linear interpolation), triangles from the 256-case table, numbered in
 * lattice order (deterministic; the reference numbers them by atomicAdd). *verts (float[n_verts][3]) and *indices
 * (uint32[n_indices], 3 per triangle, counter-clockwise around the value > thresh side's outward normal) are allocated here
 * and belong to the caller (rnb_device_free). Scratch: 12 bytes per lattice point (the reference's vertidx_grid). Syncs. */
int rnb_marching_cubes(rnb_ctx* ctx, void* stream, const float* density, const uint32_t res[3], const float aabb_min[3], const float aabb_max[3], float thresh,
                       float** verts, uint32_t** indices, uint32_t* n_verts, uint32_t* n_indices);
/* generate_training_samples_nerf_with_global_movement (testbed_nerf.cu:1216-1387), K6.
 * Fills RAY_INDICES/RAYS/NUMSTEPS/COORDS/COUNTERS. Deterministic slot order. */
int rnb_generate_training_samples(rnb_ctx* ctx, void* stream, uint32_t n_rays, uint32_t n_rays_total, uint32_t max_samples);
/* compute_loss_kernel_train_nerf_with_global_movement + fill_rollover* (testbed_nerf.cu:1396-2097, 4044-4052), K8+K9.
 * Reads MLP_OUT for COORDS; fills DLOSS_DOUT, COORDS_COMPACTED, the three LOSS buffers and COUNTERS[1]. */
int rnb_compute_loss(rnb_ctx* ctx, void* stream, uint32_t n_rays, uint32_t n_rays_total);
/* Network::forward + Network::backward on the compacted batch (testbed_nerf.cu:4067-4068;
 * nerf_network.h:97-452), K10+K11. Accumulates into GRADS_FP32 (zeroed first). */
int rnb_forward_backward(rnb_ctx* ctx, void* stream);
/* Trainer::optimizer_step: ExponentialDecay -> Adam -> EMA (trainer.h:170-172; adam.h:52-202; ema.h:63-147), K12. */
int rnb_optimizer_step(rnb_ctx* ctx, void* stream);

/* ---- whole step ------------------------------------------------------- */
/* Testbed::train (src/testbed.cu:2776-2872) = prep schedule + train_nerf (testbed_nerf.cu:3560-3668).
 * rnb_train_step == begin; [all-reduce of GRADS_FP32 and rnb_step_exchange across ranks]; end. Syncs. */
int rnb_train_step(rnb_ctx* ctx, void* stream, rnb_step_stats* stats);
/* Everything up to and including the backward pass; leaves gradients in GRADS_FP32. */
int rnb_train_step_begin(rnb_ctx* ctx, void* stream);
/* Optimizer step, ++training_step, Counters::update_after_training (testbed_nerf.cu:3532-3558). Syncs.
 * == apply; local; finish(local values) -- in the overlapped schedule (cfg.overlap) queued as local; finish; apply: the next step's march goes out the moment
 * the loss readback arrives, the optimizer's launches behind it (same results; the optimizer runs even when the step produced no samples, whose error code is
 * then returned). Data-parallel hosts call the three pieces and sum the local counters /
 * loss sums over the ranks before finish, so every rank draws the same rays_per_batch for the next step. */
int rnb_train_step_end(rnb_ctx* ctx, void* stream, rnb_step_stats* stats);
int rnb_train_step_apply(rnb_ctx* ctx, void* stream);
/* counters = {numsteps_counter, numsteps_counter_compacted, ray_counter, samples_written}; loss_sums = sums of the LOSS,
 * EK_LOSS, MASK_LOSS arrays. Syncs the stream. */
int rnb_train_step_local(rnb_ctx* ctx, void* stream, uint64_t counters[4], double loss_sums[3]);
int rnb_train_step_finish(rnb_ctx* ctx, const uint64_t counters[4], const double loss_sums[3], rnb_step_stats* stats);
/* Per-kernel timing with HIP events on the step's stream (the reference only keeps the two wall-clock EMAs
 * m_training_prep_ms / m_training_ms, testbed.h:863-867). enable(1) also clears the accumulators. Entry idx of
 * [0, rnb_profile_count): kernel-group name, accumulated milliseconds, timed launches, units processed
 * (samples / rays / parameters). */
int rnb_profile_enable(rnb_ctx* ctx, int on);
int rnb_profile_count(const rnb_ctx* ctx);
int rnb_profile_get(const rnb_ctx* ctx, int idx, const char** name, double* total_ms, uint64_t* launches, double* units);
uint32_t rnb_training_step(const rnb_ctx* ctx);
uint32_t rnb_rays_per_batch(const rnb_ctx* ctx);
/* Re-seat the optimizer's step counter: what AdamOptimizer / ExponentialDecayOptimizer::deserialize restore from a snapshot that carries the
 * optimizer state (adam.h:486-495 "current_step", exponential_decay.h:143-147 "learning_rate_factor"). `step` = optimizer steps taken so far; the
 * learning-rate factor becomes lr_decay_base ^ (number of decay events the steps [0, step) have seen, exponential_decay.h:61-72), multiplied up
 * in the same order as the running optimizer does. The per-parameter state (moments, step counts) is written through rnb_buffer. */
int rnb_set_optimizer_step(rnb_ctx* ctx, uint32_t step);
/* Re-seat the controller state (snapshot resume, testbed.cu:3333-3390). */
int rnb_set_controller(rnb_ctx* ctx, uint32_t training_step, uint32_t rays_per_batch,
                       uint32_t measured_batch_size_before_compaction, uint32_t n_rays_total);

/* Self-test of the integer / index primitives the path is built on, evaluated by the library's own device functions (one thread per item) --
 * what tests/golden/int_fixtures.json (outputs of the reference's host-compilable fragments) is compared with. Words are uint32; floats
 * travel as bit patterns. Items in / out per kind:
 *   RNB_PRIM_PCG32   in  initstate lo hi, initseq lo hi, delta lo hi           pcg32{initstate, initseq}; advance(delta)  (pcg32.h:44-170)
 *                    out state hi lo (after the advance), next_uint(), bits(next_float()) (each the draw at that position)
 *   RNB_PRIM_MORTON  in  x y z          out morton3D, morton3D_invert(code >> 0, 1, 2)                  (tiny-cuda-nn/common_device.h:337-363)
 *   RNB_PRIM_SRGB    in  bits(v)        out bits(srgb_to_linear(v)), bits(linear_to_srgb(v))              (common_device.cuh:31-61)
 *   RNB_PRIM_RAY_BOX in  box lo hi, origin 3, direction 3     out tmin, tmax, contains(origin)            (bounding_box.cuh:163-213)
 *   RNB_PRIM_MARCH   in  cone_angle, max_cascade, pos 3, dir 3, t   out calc_dt(t), mip_from_pos, mip_from_dt(dt), cascaded_grid_idx_at(pos, mip),
 *                    density_grid_occupied_at (bitfield byte i = pcg32{5} draw i >> 24), distance_to_next_voxel, advance_to_next_voxel at res = 128 >> mip
 *                                                                                                        (src/testbed_nerf.cu:153-155, 301-323, 439-465, 569-583)
 * Floating-point primitives (tests/golden/float_fixtures.json, the reference's fragments compiled without FMA contraction, as this library is):
 *   RNB_PRIM_ACTIVATION in bits(v)     out relu(v), logistic(v), logistic'(v) = l (1 - l)   (activation_function / network_to_rgb[_derivative], testbed_nerf.cu:326-357;
 *                                                                                             tcnn common_device.h:52-54) -- the NeuS alpha's two CDFs, the albedo and its gradient
 *   RNB_PRIM_WARP    in  box lo hi, pos 3, dir 3, dt   out warp_position 3, warp_direction 3, unwarp_direction(warped) 3, warp_dt, unwarp_dt(warped)
 *                                                                                                        (NerfCoordinate's fields, testbed_nerf.cu:390-437)
 *   RNB_PRIM_LOSS    in  is_L2, target 4, prediction 4   out loss, gradient 4                              (loss_and_gradient, testbed_nerf.cu:280-299, 1389-1394)
 *   RNB_PRIM_PIXEL   in  base_idx, n_rays, n_rays_total, n_images, w, h, snap, advance lo hi     out image_idx, x, y of nerf_random_image_pos_training with
 *                                                                       pcg32{1337} advanced by `advance` (testbed_nerf.cu:1171-1214; no error-map CDFs)
 *   RNB_PRIM_GRID    in  hashmap_size, resolution, pos_grid 3, bits(x), bits(scale)   out grid_index<3,2>(Hash, feature 0, ..) / 2, and pos_fract(x, scale): pos, pos_grid
 *                                                                                                        (tcnn encodings/grid.h:113-148, common_device.h:427-434)
 *   RNB_PRIM_READ_RGBA in w, h (w h <= 14), bits(x), bits(y), 28 words = the image, RGBA16, two words per pixel    out read_rgba(pos) r g b a, and the test
 *                                                                       `red <= 0` of testbed_nerf.cu:1264 (common_device.cuh:621-627, 665-700)
 *   RNB_PRIM_CAMERA_RAY in w, h, focal 2, principal point 2, position x y, camera matrix 12 (row-major 3x4)   out origin 3, direction before and after normalisation 3 + 3
 *                                                                       (the pinhole ray of testbed_nerf.cu:1279-1305 as Eigen evaluates it)
 *   RNB_PRIM_RAY_TARGETS in apply_no_albedo, apply_rgbplus, apply_L2, apply_light_opti, apply_relu, light index, camera matrix 12, normal texel 4, albedo texel 4,
 *                    light_directions 9 (row-major, camera frame; nine words 0xffffffff = the context's own, testbed_nerf.cu:1537-1554)   out rgbtarget 4, the light in the world frame 3   (the loss kernel's per-ray targets, testbed_nerf.cu:1500-1592)
 *   RNB_PRIM_LOSS_SAMPLE in apply_no_albedo, apply_rgbplus, apply_L2, apply_relu, the sample's network output (16 halves = 8 words), dt, ray direction 3, light 3, the ray's loss
 *                    gradient 4, rgb_ray 4, then the running values BEFORE the sample: rgb_ray2 4, (clamped) weight_sum, weight_sum2, T, then gradient_weight_sum, loss_scale,
 *                    ek_loss_weight   out alpha, and AFTER the sample T, weight_sum2, rgb_ray2 4, then dL/d(network output)[0..10] as half bit patterns, then the float values behind
 *                    them: dloss_by_drgb 3, dloss_dn 3, dloss_dalpha, dloss_dsdf, dloss_dvariance, dloss_dnormal_norm
 *                    (one iteration of the loss kernel's second loop, testbed_nerf.cu:1855-2085; cos_anneal_ratio 1, rgb_activation Logistic)
 *   RNB_PRIM_RAY_LOSS in apply_L2, apply_rgbplus, apply_bce, bits(mask_loss_weight), n_rays, rgbtarget 4, rgb_ray 4, the albedo texel's alpha, the normal texel's alpha, weight_sum
 *                    out the ray's loss, its gradient 4, the clamped weight_sum, gradient_weight_sum, the loss row (loss / n_rays), the mask-loss row
 *                    (the loss kernel between its two loops, testbed_nerf.cu:1735-1800)
 *   RNB_PRIM_ENCODE  in  table entries (<= 256), resolution, bits(scale), x y z, then 257 words = the level's table (half2 per entry, + one readable word)
 *                    out features f0 f1 (half bits), d f0 / d xyz 3, d f1 / d xyz 3 -- twice: the training kernels' form, then the evaluation kernels' pipelined form
 *                    (one sample, one level of kernel_grid, tcnn encodings/grid.h:168-364)
 *   RNB_PRIM_MARCH_RAY in box lo hi, cone_angle, origin 3, direction 3, start t    out the number of samples the march takes through the RNB_PRIM_MARCH bitfield, the sum of the bit
 *                    patterns of every NerfCoordinate word it writes (7 per sample: warped position, warped dt, warped direction), the first two samples' 14 words, the last one's 7
 *                    (the sampler's two march loops, testbed_nerf.cu:1330-1380)
 *   RNB_PRIM_SDF_DENSITY in sdf, variance (half bit patterns)    out the occupancy grid's density s sigmoid(sdf s) (1 - sigmoid(sdf s)), s = exp(10 variance), half arithmetic throughout
 *                    (sdf_to_density_variance_buffer, common_operation.cuh:311-328)
 *   RNB_PRIM_PREP_DUE in training step    out 1 if that step begins with an occupancy update, and the interval n_prep_to_skip it is due at (Testbed::train, src/testbed.cu:2805-2806)
 *   RNB_PRIM_DW_SLICED in 4 words (word 0: 1 = the Y operand is the constant 1 on row 0, rows 1..3 zero), then 8 rows of 8256 halves (two per word): Y rows 0..3, X rows 0..3
 *                    out the 4 x 4 weight gradient dW[o][i] = sum_s Y[o][s] X[i][s] as accumulate = RNB_ACCUM_HALF forms it (float bit patterns of half values): slices of 4096 samples
 *                    (here 4096 + 4096 + 64), a half accumulator per slice rounded after every 16-sample k-step, the slices' results added in half
 *                    (tcnn cutlass_matmul.h:83, 315-322 as the oracle's emulated_dw models it; the library's own k_dw_sliced / k_dw_finish code)
 * Host pointers; syncs. */
typedef enum rnb_primitive { RNB_PRIM_PCG32 = 0, RNB_PRIM_MORTON = 1, RNB_PRIM_SRGB = 2, RNB_PRIM_RAY_BOX = 3, RNB_PRIM_MARCH = 4,
                             RNB_PRIM_ACTIVATION = 5, RNB_PRIM_WARP = 6, RNB_PRIM_LOSS = 7, RNB_PRIM_PIXEL = 8, RNB_PRIM_GRID = 9, RNB_PRIM_READ_RGBA = 10, RNB_PRIM_CAMERA_RAY = 11, RNB_PRIM_RAY_TARGETS = 12, RNB_PRIM_LOSS_SAMPLE = 13, RNB_PRIM_RAY_LOSS = 14, RNB_PRIM_ENCODE = 15, RNB_PRIM_MARCH_RAY = 16, RNB_PRIM_SDF_DENSITY = 17, RNB_PRIM_PREP_DUE = 18, RNB_PRIM_DW_SLICED = 19 } rnb_primitive;
int rnb_eval_primitives(rnb_ctx* ctx, int kind, const uint32_t* in_host, uint32_t n_items, uint32_t* out_host);

/* Data parallel only: gradient blocks in the order they become final during the backward pass queued by
 * rnb_train_step_begin, so that a caller can exchange the early block while the rest is still being accumulated.
 * ranges[k] = {first, last+1} parameter indices into RNB_BUF_GRADS_FP32, k = 0 .. *n_parts-1 (at most 3); the blocks
 * partition [0, n_params). rnb_gradient_part_wait makes `stream` wait (device side) until block k is final.
 * Without side-stream overlap there is one block, final when the stream given to _begin is. No reference counterpart
 * (the reference is single-GPU). */
int rnb_gradient_parts(rnb_ctx* ctx, uint64_t ranges[3][2], uint32_t* n_parts);
int rnb_gradient_part_wait(rnb_ctx* ctx, uint32_t part, void* stream);
/* Optimizer on block 0 only, queued on `stream` (after the caller's exchange of that block on the same stream);
 * rnb_train_step_apply then covers the remaining blocks and joins. Optional. */
int rnb_train_step_apply_early(rnb_ctx* ctx, void* stream);

/* Data parallel, sharded optimizer: instead of all-reducing the gradients and stepping every parameter on every rank, the
 * caller reduce-scatters each block, rank r steps chunk r (Adam moments, fp32 masters and EMA of the other chunks are not
 * touched on this rank), and the fp16 training weights (RNB_BUF_PARAMS_FP16) are all-gathered. Per block, in order:
 *     rnb_gradient_part_wait(ctx, k, stream)                         block k's gradients are final
 *     reduce-scatter  GRADS_FP32[lo, hi)   -> [own_lo, own_hi)       (in place, sum)
 *     rnb_train_step_apply_shard(ctx, k, stream)                     Adam + EMA on the own chunk, the rest of the block's accumulators cleared
 *     all-gather      PARAMS_FP16[own_lo, own_hi) -> [lo, hi)        (in place)
 * then rnb_train_step_apply_done on a stream that has joined the blocks' streams (it replaces rnb_train_step_apply).
 * Blocks, in the order they become final (MLPs + coarse and middle levels | first half of the fine levels | second half + variance: the exchange of each runs
 * beside the scatter of the next), are world_size equal chunks of a multiple of 4 parameters; the last block ends at *capacity >= n_params, the
 * allocated (zero-padded) length of every parameter-shaped buffer of rnb_buffer. The fp32 masters / EMA / Adam state of a
 * rank are complete only on its own chunks: all-gather them over the same layout before reading them as a whole
 * (snapshots, inference with EMA weights). No reference counterpart (the reference is single-GPU). */
#define RNB_MAX_SHARD_PARTS 3 /* ABI 3 (ABI 2: two blocks) */
typedef struct rnb_shard_part { uint64_t lo, hi, own_lo, own_hi; } rnb_shard_part;
int rnb_shard_layout(rnb_ctx* ctx, rnb_shard_part parts[RNB_MAX_SHARD_PARTS], uint32_t* n_parts, uint64_t* capacity);
int rnb_train_step_apply_shard(rnb_ctx* ctx, uint32_t part, void* stream);
int rnb_train_step_apply_done(rnb_ctx* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RNB_NEUS2_H */
