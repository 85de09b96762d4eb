// rnb_neus2_hip.hip — C-ABI of include/rnb_neus2.h over the gfx950 kernels (host driver).
// Host-side sequencing follows Testbed::train / train_nerf / train_nerf_step (src/testbed.cu:2776-2872,
// src/testbed_nerf.cu:3560-4123). No CPU fallback: every entry point runs HIP kernels or fails.
#include "kernels_net.cuh"
#include "kernels_ray.cuh"
#include "kernels_mesh.cuh"
#include "../host/mesh.hpp" // the marching-cubes case table generator (header only)

#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <new>
#include <stdexcept>
#include <random>
#include <string>
#include <vector>

using namespace rnb;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                         \
	do {                                                                                                      \
		hipError_t e_ = (expr);                                                                               \
		if (e_ != hipSuccess) return fail(RNB_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

template <typename T>
struct DevBuf {
	T* p = nullptr;
	size_t n = 0;
	hipError_t alloc(size_t count) {
		n = count;
		if (count == 0) { p = nullptr; return hipSuccess; }
		return hipMalloc((void**)&p, count * sizeof(T));
	}
	// `count` elements in use, `capacity` >= count allocated and zeroed (the tail is padding for equal data-parallel shards)
	hipError_t alloc_padded(size_t count, size_t capacity) {
		n = count;
		hipError_t e = hipMalloc((void**)&p, capacity * sizeof(T));
		if (e != hipSuccess) return e;
		return hipMemset(p, 0, capacity * sizeof(T));
	}
	void free() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
	size_t bytes() const { return n * sizeof(T); }
};

uint32_t next_multiple_u32(uint32_t v, uint32_t m) { return ((v + m - 1) / m) * m; }

// Kernel launch whose completion IS the event `ev` (null: plain launch). hipEventRecord puts a marker packet of its own
// into the queue, which costs the stream ~6 us between two kernels (measured); binding the event to the kernel's own
// completion signal costs nothing, and the step's critical stream records 4-5 events.
#define LAUNCH_EV(kernel, grid, block, lds, stream, ev, ...) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ev, 0, __VA_ARGS__)
// the same with the launch flags spelled out (hipExtAnyOrderLaunch: the packet carries no barrier bit -- it may start while the kernel in front of it on the stream still runs)
#define LAUNCH_EVF(kernel, grid, block, lds, stream, ev, flags, ...) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ev, flags, __VA_ARGS__)

uint32_t compute_valid_level(const rnb_config& cfg, int training_step) { // grid.h:1430-1437
	if (training_step <= 0) return cfg.n_levels;
	float v = cfg.base_valid_level_scale * cfg.n_levels + cfg.valid_level_scale * std::max(0, (int)(training_step - (int)cfg.base_training_step));
	return std::min(cfg.n_levels, (uint32_t)ceilf(v));
}

} // namespace

// Per-kernel-group timing with HIP events on the caller's stream (bench.py's roofline leg).
enum ProfId { P_NONE = -1, P_GRID_SAMPLES = 0, P_POINT_QUERY, P_EMA_BITFIELD, P_MARCH_COUNT, P_SCAN_RAYS, P_MARCH_WRITE, P_FORWARD, P_LOSS_PASS1,
              P_SCAN_COMPACT, P_LOSS_PASS2, P_FWD_BWD, P_DW, P_SCATTER, P_ADAM, P_REDUCE, P_SCATTER_LDS, P_SCATTER_RL, P_SCATTER_QUAD, P_COUNT };
static const char* const PROF_NAMES[P_COUNT] = {"k_grid_samples", "k_point_query", "k_ema_grid+bitfield", "k_march_count", "k_scan_rays", "k_march_write", "k_forward",
                                                "k_loss_pass1", "k_scan_compact", "k_loss_pass2+k_rollover", "k_fwd_bwd", "k_dw*7+k_dw_finish", "k_grid_scatter", "k_adam_ema", "k_reduce_losses",
                                                // the three kernels of the group "k_grid_scatter" one by one (the group's entry is their sum, kept for records of earlier rounds)
                                                "k_grid_scatter_lds", "k_grid_scatter_quad_rl", "k_grid_scatter_quad"};
struct Profiler {
	bool on = false;
	std::vector<hipEvent_t> ev;
	std::vector<int> ids;
	size_t n = 0;
	double total_ms[P_COUNT] = {0};
	uint64_t launches[P_COUNT] = {0};
	double units[P_COUNT] = {0};
	void mark(hipStream_t s, int id) {
		if (!on) return;
		if (n == ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev.push_back(e); ids.push_back(0); }
		ids[n] = id;
		(void)hipEventRecord(ev[n], s);
		++n;
	}
	void collect() { // after a stream sync
		for (size_t i = 1; i < n; ++i) {
			if (ids[i] < 0) continue;
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, ev[i - 1], ev[i]) == hipSuccess) {
				total_ms[ids[i]] += ms; ++launches[ids[i]];
				if (ids[i] == P_SCATTER_LDS || ids[i] == P_SCATTER_RL || ids[i] == P_SCATTER_QUAD) { total_ms[P_SCATTER] += ms; if (ids[i] == P_SCATTER_QUAD) ++launches[P_SCATTER]; } // the group's entry
			}
		}
		n = 0;
	}
	void reset() { for (int i = 0; i < P_COUNT; ++i) { total_ms[i] = 0; launches[i] = 0; units[i] = 0; } n = 0; }
	void destroy() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); ids.clear(); n = 0; }
};

static inline uint32_t ilog2(uint32_t v) { uint32_t r = 0; while (v > 1) { v >>= 1; ++r; } return r; }

struct rnb_ctx {
	rnb_config cfg;
	Profiler prof;
	GridMeta grid;
	uint64_t n_grid_params = 0, n_params = 0;
	uint32_t n_dense_lead = 0; // build_grid_tables
	uint64_t off_sdf = 0, off_rgb = 0, off_grid = 0, off_var = 0;
	SceneAabb aabb;
	int n_cus = 256;

	DevBuf<float> params_fp32, grads, adam_m, adam_v;
	DevBuf<half_t> params_fp16, params_ema;
	DevBuf<half_t> grads16; // cfg.accumulate = RNB_ACCUM_HALF: the half gradient vector (RNB_BUF_GRADS_FP16) instead of the fp32 accumulators `grads`
	bool half_acc() const { return cfg.accumulate == RNB_ACCUM_HALF; }
	DevBuf<unsigned long long> grads_fixed; // cfg.deterministic: [n_grid_params] 64-bit fixed-point sums of the hash-grid scatter (scale 2^24), narrowed into grads / grads16 by k_fixed_narrow
	bool fixed_acc() const { return cfg.deterministic != 0; }
	size_t grad_elem() const { return half_acc() ? sizeof(half_t) : sizeof(float); }
	char* grad_ptr(uint64_t i) const { return half_acc() ? reinterpret_cast<char*>(grads16.p + i) : reinterpret_cast<char*>(grads.p + i); }
	DevBuf<uint32_t> adam_steps;
	// Optimizer state as k_adam_ema keeps it: one 64-byte record {fp32 weight, m, v, step count} per 4-parameter group (kernels_net.cuh). params_fp32 /
	// adam_m / adam_v / adam_steps above are the staging views rnb_buffer hands out: unpacked from the records when a caller asks for one
	// (opt_plain_current), packed back in front of the next optimizer launch (opt_rec_current = false: the caller may have written through the pointer).
	DevBuf<float> opt_rec;
	bool opt_plain_current = true, opt_rec_current = false;
	DevBuf<float> adam_lr_table; // k_adam_lr_table
	float lr_table_beta1 = -1.f, lr_table_beta2 = -1.f; // the betas the table was filled for
	DevBuf<float> density_grid, density_grid_tmp, density_grid_tmp_alt, density_mean; // _alt: cleared on a side stream for the NEXT update (tmp_alt_clear), the two swap roles
	bool tmp_alt_clear = false;
	rnb_grid_exchange_fn grid_exchange = nullptr; // data parallel: element-wise max of density_grid_tmp over the ranks between the two halves of an occupancy update
	void* grid_exchange_user = nullptr;
	bool bitfield_foreign = false; // a caller may have written the bitfield: levels >= 1 are not known to be zero outside the pooled supports (update_bitfield takes the zero-filling kernels once)
	DevBuf<double> mean_partial, loss_sums;
	DevBuf<uint8_t> bitfield;
	DevBuf<unsigned long long> march_stats; // RNB_MARCH_STATS=1: k_march_count_skip's counters, printed by rnb_destroy
	DevBuf<uint32_t> coarse_bits, coarse_count; // k_coarse_bitfield: cascade 0's occupancy in the form the march kernels keep in LDS; its number of non-empty blocks
	bool coarse_valid = false;
	uint32_t* host_coarse = nullptr; // mapped host memory: k_coarse_bitfield's block count as last written by the device (0xffffffff: never)
	uint32_t* host_coarse_dev = nullptr;
	DevBuf<float> grid_sample_pos;
	DevBuf<uint32_t> grid_sample_idx;
	uint32_t n_grid_samples = 0;
	// the next update's samples in cell order (pregenerate_grid_samples): what they were generated from, their buffers, the placement scratch
	struct { bool valid = false; uint32_t ema_step = 0, n_uniform = 0, n_nonuniform = 0; uint64_t rng_state = 0, rng_inc = 0; } gs_pre;
	bool last_update_sorted = false;
	struct { bool pending = false; uint32_t n_uniform = 0, n_nonuniform = 0; } gs_todo; // an update has run: prepare the next one's samples
	DevBuf<float> gs_sorted_pos, gs_stage_pos, gs_eval_pos;   // sorted / stage: being prepared for the next update; eval: what the last update evaluated
	DevBuf<uint32_t> gs_sorted_idx, gs_stage_idx, gs_eval_idx, gs_hist, gs_range; // gs_range: [2] this rank's share of the cell-ordered samples (k_shard_range)
	hipEvent_t ev_grid = nullptr, ev_gs = nullptr;

	// dataset
	uint32_t n_views = 0;
	DevBuf<ViewDev> views;
	DevBuf<uint16_t> pixels;
	float light_dirs[9];

	// step scratch
	DevBuf<uint32_t> ray_indices, numsteps, counters;
	DevBuf<float> rays, coords, coords_compacted, loss; // loss: [3][max_rays] = colour, eikonal, mask terms per ray
	float *ek_loss = nullptr, *mask_loss = nullptr;       // rows 1, 2 of `loss`
	DevBuf<half_t> mlp_out, dloss_dout;
	DevBuf<double> wg_partial; // loss sums per workgroup of k_loss_pass2_rays
	bool loss_reduced = false; // the running step's loss sums were reduced and published by k_loss_pass2_samples (launch_reduce_losses has nothing left to do)
	DevBuf<float> ray_grad; DevBuf<uint32_t> ray_of, slot_of; // pass 2 of the loss in two launches (k_loss_pass2_rays -> k_loss_pass2_samples)
	DevBuf<float> chain_rec; // per marched-sample slot: the compositing recurrence's running values, left by pass 1 of the loss for pass 2 (LossArgs::chain_rec)
	DevBuf<float> ray_setup, ray_dunnorm, ray_t;
	DevBuf<double> loss_partial; // per-tile loss sums of large batches
	DevBuf<uint32_t> ray_steps, ray_base, ray_slot, ncomp, cbase, scan_tiles; // scan_tiles: [64] tile sums + [64][3] dependent tile sums of the multi-workgroup ray scan
	// two-round network evaluation (step_front): head of every ray first, tails of the rays that need them second
	DevBuf<uint32_t> ray_base1, idx1, idx2, fwd_counts;
	DevBuf<unsigned long long> scan_words; // k_scan_rays_chain / k_scan_compact_chain: [64 tiles][4] sums with the launch's ticket, one block each
	uint32_t scan_ticket = 0;
	DevBuf<half_t> wimg_fwd, wimg_fbs, wimg_train, wimg_rgb; // LDS weight images of the training weights, rebuilt after every optimizer step
	// albedo mode: k_rgb_fwd_bwd + k_fwd_bwd_sdf_full (kernels_net.cuh). cin_eval = the colour MLP's input row of every evaluated sample (written by the
	// network evaluation), src_slot = the slot of every compacted sample (loss pass 2), dcin = dL/d(input row) between the two training kernels
	DevBuf<half_t> cin_eval, dcin, rgb_out_scratch;
	DevBuf<uint32_t> src_slot;
	bool cin_flow = false; // this step's network evaluation has exported cin_eval and the loss pass src_slot (rnb_train_step_begin); consumed by its backward pass, cleared by the stage entry points and when the step's front fails
	bool rgb_split() const { return !cfg.apply_no_albedo && !knobs.fwd_bwd_generic; }
	bool wimg_valid = false;
	DevBuf<float> ray_const; // per kept ray: loss constants worked out beside the march (k_march_write)
	DevBuf<uint32_t> unfinished;
	uint32_t fwd_k1 = 48;      // head length of the two-round network evaluation (0 = one round); the longest head of the adaptive rule
	bool fwd_k1_fixed = false; // RNB_FWD_K1: that value for every step
	bool gen_split = false; // the last generated batch wrote its sample records in two launches (knobs.march_write_split)
	uint32_t gen_k1 = 0, cur_k1 = 0; // head length the last generated batch / the running step was laid out with (k1_for)
	// Tuning / A-B knobs, read from the environment once at creation (measurement aids, not part of the interface).
	struct Knobs {
		bool march_narrow = false, fwd_bwd_generic = false, loss_wave_per_ray = false;
		// RNB_SCATTER_WG_PER_CU: resident workgroups of the atomic scatter kernels per CU (0 = one workgroup per 64 samples, round 2's launch). Default
		// by mode, measured in round 3 (window, ms/step): --no-albedo 0: 0.650, 2: 0.666, 4: 0.667, 8: 0.664; albedo 0: 0.826, 2: 0.796, 4: 0.808, 8: 0.828
		// (there the weight-gradient GEMMs of the side stream are the long pole and need the wave slots)
		int scatter_wg_per_cu = -1;
		uint32_t fbs_wg_per_cu = 2; // RNB_FBS_WG_PER_CU: workgroups of k_fwd_bwd_sdf per CU (its launch bounds allow two)
		bool march_running_sums = false; // RNB_MARCH_RUNNING_SUMS: the 16-lanes-per-ray march with the 16 running sums per round everywhere (round 2), no closed form
		bool march_late = false; // RNB_MARCH_LATE: the next step's march waits for k_fwd_bwd instead of starting after the loss pass
		uint32_t march_narrow_from = 18432; // rays per step from which the per-ray kernels switch to their large-batch forms (RNB_MARCH_NARROW_FROM). ms/step small / large forms, end of round 2: 16.2 k rays 0.697 / 0.707, 19.1 k 0.713 / 0.704, 22.3 k 0.740 / 0.710
		bool dp_order = false; // scatter order of the data-parallel exchange even with one rank (RNB_DP_FORCE_COLLECTIVES)
		bool scan_chain = true; // RNB_SCAN_CHAIN=0: the ray scans as in rounds 1-3 (one 1024-thread workgroup for small batches, three tiled launches for large ones)
		uint32_t march_wave_per_ray_below = 4096; // RNB_MARCH_WAVE_PER_RAY_BELOW=n: one wavefront per ray for batches of at most n rays (single-cascade scenes). At an eighth of the batch
		                                          // (1.8 k rays per step): 0.2985 -> 0.2890 ms/step with 4096 (2560: 0.2887); bit-exact at every size (the full-size tests were run with n = 100 000)
		int scatter_order = -1; // RNB_SCATTER_ORDER: 0 = B, A1, A2, C (rounds 1-3); 1 = A1, A2, B, C; 2 = A (one launch), B, C; default: 2 below march_narrow_from rays per step, 0 from there on
		bool scatter_c_side = false; // RNB_SCATTER_C_SIDE=1 (A/B, round 6): rnb_ctx::sc.c_side -- group C and its optimizer chunk at the end of the weight-gradient stream instead of last on the critical stream. Late 0.6040 -> 0.5991,
		                             // but steps 1000-1200 0.5617 -> 0.5802, window 0.5483 -> 0.5589: an event that crosses streams costs ~16 us from the producer's end to the consumer's start however idle the
		                             // consumer's queue has been (C starts 16 us behind group B and takes 62 instead of 37 us beside the optimizer's chunk; the evaluation still starts 16 us behind the last event):
		                             // profiles/r06_ab_c_side.txt. Off.
		bool unsafe_skip_joins = false; // RNB_UNSAFE_SKIP_JOINS=1: MEASUREMENT ONLY (the results are then unordered): the critical stream waits for none of the side streams in front of the network evaluation --
		                                // what the three barrier packets cost
		bool join_fold = false; // RNB_JOIN_FOLD=1 (A/B, round 6): rnb_ctx::join_pending -- one event in front of the network evaluation instead of three. SLOWER: window 0.5470 -> 0.5553, late 0.6022 -> 0.6065
		                        // (the packets whose events are long signalled cost the critical stream little; the folded event arrives two hops later): profiles/r06_ab_join_fold.txt. Off.
		bool defer_tail = true; // RNB_DEFER_TAIL=0: the critical stream itself waits for the side stream's weight images at the end of the optimizer (rounds 1-3)
		bool poll_loss = true; // RNB_POLL_LOSS=0: the host waits for the completion event of k_reduce_losses_rollover (rounds 1-3) instead of polling the readback's sequence word
		bool fused_update = true; // RNB_FUSED_UPDATE=0: the occupancy update's grid / bitfield chain as the seven launches of rounds 1-3 (k_ema_grid, k_mean_*, k_grid_to_bitfield, pools, k_coarse_bitfield)
		bool loss_scan_fused_always = false; // RNB_LOSS_SCAN_FUSED=2 (tests): at every batch size
		bool loss_scan_fused = true; // RNB_LOSS_SCAN_FUSED=0: the compaction offsets by k_scan_compact* in front of the two pass-2 launches (default: formed inside k_loss_pass2_rays)
		bool loss_flat = true; // RNB_LOSS_FLAT=0: pass 2 of the loss as one launch with 64 / 16 lanes per ray (default: k_loss_pass2_rays, then k_loss_pass2_samples with one lane per compacted sample; needs the chain records)
		bool loss_chain_records = true; // RNB_LOSS_CHAIN_RECORDS=0: pass 2 of the loss replays the compositing recurrence itself (rounds 1-3) instead of reading the running values pass 1 left
		int ray_const_dense = 2; // RNB_RAY_CONST_DENSE=0: the loss's per-ray constants inside k_march_write (rounds 2-4); 1: k_ray_constants (one thread per kept ray) behind a k_march_write that is never split;
		                         // 2 (default): behind k_march_write split as before. ms/step at steps 1000 / 2000 / 6000: 0: 0.6019 / 0.5884 / 0.6292, 1: 0.6088 / 0.5954 / 0.6269, 2: 0.5999 / 0.5877 / (= 1)
		int march_write_split = -1; // RNB_MARCH_WRITE_SPLIT=0|1: k_march_write of a march generated ahead as one launch (rounds 1-3) / always split; default: split below 65 536 rays per step. Split: what the first network evaluation reads (idx1, the heads'
		                               // coordinates) in a first launch, whose completion the critical stream waits for; the rest (ray constants, ray records, the tails' coordinates) in a second one
		                               // that runs beside that evaluation and is joined in front of the loss pass
		bool scatter_plain = false; // RNB_SCATTER_PLAIN=1 (A/B, tests): no LDS-privatised and no run-length scatter -- every corner of every level is its own L2 atomic, as in the reference; with
		                            // accumulate = RNB_ACCUM_HALF every one of a corner's four addends is (k_grid_scatter_quad_h_per_addend), which reproduces the reference's sequential half sums
		                            // on the coarse levels too
		int dbg_scatter_lo = -1, dbg_scatter_hi = -1; // RNB_DEBUG_SCATTER_LEVELS=lo,hi (measurement aid, WRONG gradients): while the per-kernel profiler is on, the atomic scatter kernels only walk levels [lo, hi)
		bool scatter_c_early = false; // RNB_SCATTER_C_EARLY=1: group C (LDS-privatised coarse levels, no global atomics to speak of) + its optimizer chunk on the optimizer's stream beside group A
		                              // instead of last on the caller's stream. Measured again in round 5 (round 2: 42 -> 158 us): the kernel stretches 40 -> 167 us beside the atomic kernels and the march
		                              // (which keep their times) and holds the optimizer's chunks back: 0.5846 -> 0.6404 ms/step at step 1000, 0.6217 -> 0.6753 at 6000 (profiles/r05_ab_scatter_c_early.txt). Off.
		int scatter_rl_staged = -1; // RNB_SCATTER_RL_STAGED=0|1: the run-length scatter loads its operands from global memory inside the walk (rounds 2-4) / stages them in LDS (round 5). Alone the two take
		                            // the same time (112 / 101 / 112 us staged vs 112 / 97 / 111 direct at steps 1000 / 2000 / 6000: the walk is NOT a chain of load -> atomic-acknowledge round trips,
		                            // which is what the staging removes); in the step the staged form's 40 registers and 32 KB of LDS leave the march beside it more of the CU while the batch is
		                            // few long rays: 0.6078 -> 0.5971 ms/step at step 1000, 0.5900 -> 0.5929 at 2000, 0.6334 -> 0.6360 at 6000 (profiles/r05_ab_scatter_rl_staged.txt).
		                            // Default: staged below march_narrow_from rays per step (the regime of the A-B-C scatter order), direct from there on
		bool march_skip_narrow = false; // RNB_MARCH_SKIP_NARROW=1: the same skipping in the one-thread-per-ray march of the large batches. Bit-identical (tests, 6100 lockstep steps) and SLOWER: k_march_count 192 -> 221 us
		                                // at step 2000, window 0.5611 -> 0.5715 ms/step, late 0.6073 -> 0.6204: a wavefront's 64 rays finish with the slowest, and the one ray in 64 that cannot skip keeps the old
		                                // cost while every lane pays the 64-point scan and the re-entry search (profiles/r06_ab_march_skip_narrow.txt). Off.
		uint32_t march_narrow_wgs = 128; // RNB_MARCH_NARROW_WGS=64|128|256|512 (A/B, round 6): threads (= rays) per workgroup of the thread-per-ray march. Window / late, ms/step: 64: 0.5512 / 0.6069, 128: 0.5502 / 0.6043,
		                                 // 256: 0.5589 / 0.6117, 512: 0.5712 / 0.6190 (a workgroup keeps its slots until its slowest ray is through) -- profiles/r06_ab_march_wgs.txt
		int march_wgs = 1024; // RNB_MARCH_WGS=256|512|1024 (round 6): threads per workgroup of k_march_count_skip. Every workgroup first loads the occupancy's LDS form (~44 KB at step 1000) for its WGS / 16 rays:
		                      // 64 rays per load instead of 16. Steps 1000-1200 / window: 256: 0.5637 / 0.5502, 512: 0.5608 / 0.5498, 1024: 0.5592 / 0.5492 (bit-identical: the pinned states)
		int march_skip = 1; // RNB_MARCH_SKIP=0: k_march_count_wide<16> as in rounds 2-5 (every round from box entry to box exit); 1 (round 6): k_march_count_skip; 2: its start-over path forced (tests)
		bool dw_late = false; // RNB_DW_LATE=1 (A/B)
		int march_bbox = 1; // RNB_MARCH_BBOX=0: the thread-per-ray march walks from the scene box's entry to its exit (rounds 1-5); 1 (round 6, default): it ends where the ray leaves the occupied region's bounding
		                    // box: window 0.5609 -> 0.5531 ms/step, late 0.6152 -> 0.6086; 2: + one jump to that box's entry (k_march_count_bbox): bit-identical, but the jump and its re-entry search cost what they save
		                    // (0.5624 / 0.6165; profiles/r06_ab_march_bbox.txt) -- kept as a knob
		int march_prio = 0; // RNB_MARCH_PRIO=0..3 (A/B): s_setprio of k_march_count / k_march_count_skip
		int scatter_prio = 0; // RNB_SCATTER_PRIO=0..3 (A/B): s_setprio of the scatter kernels' wavefronts
		bool scatter_share = true;  // RNB_SCATTER_SHARE=1 (A/B, round 6): face sharing in the run-length scatter (kernels_net.cuh: share_face)
		int scatter_kmin = 0, scatter_rl_upto = 0; // RNB_SCATTER_KMIN, RNB_SCATTER_RL_UPTO (A/B, plan_scatter_groups)
		bool scatter_anyorder = true;  // RNB_SCATTER_ANYORDER=1 (A/B): the scatter groups behind the first one are launched with hipExtAnyOrderLaunch -- they touch other levels, so a group may start in the tail of the one in front of it
		int encode_depth = 4; // RNB_ENCODE_DEPTH=0|2|4|7: levels whose gathers k_forward_chained / k_point_query_chained keep in flight (round 5; 0: one level at a time behind branches, rounds 1-4).
		                      // Interleaved medians, ms/step at steps 1000 / 2000 / 6000: 0: 0.5964 / 0.5884 / 0.6298; 2: 0.5773 / 0.5810 / 0.6209; 4: 0.5775 / 0.5769 / 0.6199; 7: 0.5781 / 0.5776 / 0.6262
		                      // (profiles/r05_ab_encode_depth.txt). The half mode's evaluation kernels take depth 4 too (254 VGPRs, 4 spilled dwords); the training kernels (rolled level loop, two workgroups per CU: no gain) keep the old form
		int march_write_wg = 256; // RNB_MARCH_WRITE_WG=1024 (A/B, round 6): threads per workgroup of k_march_write
		bool chain_plain = true; // RNB_CHAIN_PLAIN=0 (A/B, round 6): ScanChainArgs::plain -- k_scan_rays_chain's tiles exchange their sums by agent-scope atomic stores / loads instead of read-modify-write atomics (which wait behind
		                         // the scatter's backlog at the memory side): window 0.5524 -> 0.5512, late 0.6084 -> 0.6066, medians of 4 (profiles/r06_ab_chain_plain.txt)
		bool point_xcd = true; // RNB_POINT_XCD=0 (A/B, round 6): PointArgs::xcd -- the occupancy update's cell-ordered points in eight contiguous parts, one per XCD (workgroups go round the XCDs): 286 -> 274 us per update
		bool dw_sliced = true; // RNB_DW_SLICED=0 (A/B): the half mode's weight gradients in the training kernel's own tiling (deviation D1', rounds 4-5) instead of the reference's split-K order
		bool encode_pair = false; // RNB_ENCODE_PAIR=1 (A/B, round 6): when the configuration's first five levels are dense (the default's are: 16^3 ... 71^3), the depth-4 evaluation kernels gather their x-pairs with one 8-byte load
		                          // (level_issue<true>): bit-identical, 18 % fewer gather instructions -- and SLOWER: 0.5538 vs 0.5500 ms/step over steps 1000-2000, 0.6092 vs 0.6058 at step 6000 (profiles/r06_ab_encode_pair.txt;
		                          // a 4-byte-aligned 8-byte gather that straddles a 64-byte line costs a second pass, and the coarse levels were L2 hits to begin with). Off.
		bool grid_presort = true; // RNB_GRID_PRESORT=0: occupancy updates evaluate their samples in the reference's order (no pregenerate_grid_samples)
	} knobs;
	DevBuf<RayLoss> ray_loss;
	DevBuf<McTable> mc_table; // marching-cubes case table, uploaded on first use
	// training scratch
	DevBuf<half_t> fm;       // feature-major operand arrays
	DevBuf<uint32_t> g12;
	DevBuf<float> srec, var_partial, dw_partial;
	TrainScratch ts;
	uint32_t dw_nwg = 0, dw_chunk = 0;
	uint32_t fwd_grid = 0;
	bool grads_clean = true;

	Pcg32 rng, density_grid_rng, trainer_rng;
	uint32_t density_grid_ema_step = 0;
	uint32_t training_step = 0, valid_level = 0, rays_per_batch = 0;
	uint32_t measured_batch_size = 0, measured_batch_size_before_compaction = 0, n_rays_total = 0;
	uint32_t optimizer_step_count = 0;
	float lr_factor = 1.f;
	uint32_t cur_n_rays = 0, cur_n_rays_total = 0, local_measured_before = 0, cur_step = 0;
	bool grid_updated = false;
	float prep_ms = 0.f;
	std::chrono::steady_clock::time_point step_start;

	// Overlap machinery (cfg.overlap): the weight-gradient GEMMs run beside the grid scatter (s_dw), and the NEXT step's ray
	// generation + march — which depends on the occupancy bitfield and the RNG, not on the weights — runs beside this step's
	// backward pass and optimizer (s_march). Results are identical to the serial order; see DESIGN.md §5.
	hipStream_t s_march = nullptr, s_dw = nullptr, s_adam = nullptr; // with the caller's stream: the 4 hardware queues HIP multiplexes streams onto
	// The step's side stream s_dw ends with the MLPs' Adam and the LDS weight images (ev_tail). Every barrier packet in front of the next network evaluation
	// costs the critical stream ~5 us (tools/probe_barriers.hip), so when the next step's march is about to be queued, the wait for ev_tail goes onto ITS stream,
	// in front of k_march_write (slack there), and the critical stream reaches it through ev_march. tail_pending: nobody has waited for ev_tail yet.
	bool tail_pending = false;
	// (round 6) join_pending: the next step's march was already queued when the optimizer was launched, so the weight-gradient stream -- idle behind its weight images -- has taken the
	// joins: it waits for ev_adam and ev_march and records ev_join, and the critical stream reaches all three side streams through that ONE event at the head of the next step
	// (three barrier packets: 17-18 us of idle queue in front of every network evaluation, profiles/r06_timeline_*). ev_march_rest (a split k_march_write) stays a wait of its own
	// behind the first network evaluation. Whoever else consumes what the side streams wrote (join_tail_host) waits for it on the host.
	// (On the march stream behind k_ray_constants instead, measured: the first evaluation then waits for the second k_march_write too, window 0.5517 -> 0.5748.)
	bool join_pending = false;
	hipEvent_t ev_join = nullptr;
	hipEvent_t ev_loss = nullptr, ev_march = nullptr, ev_fb = nullptr, ev_dw = nullptr, ev_adam = nullptr, ev_tail = nullptr, ev_march_rest = nullptr, ev_all = nullptr, ev_sc[4] = {nullptr, nullptr, nullptr, nullptr};
	// Scatter groups of the queued backward pass: B = middle levels [split1, split0) (final at ev_sc[0]), A = fine levels [split0, off_var) in two
	// halves (ev_sc[1], ev_sc[3]; the second starts at split_mid), C = coarse levels [off_grid, split1) last; the MLPs + variance follow the dW GEMMs (ev_dw).
	struct { bool valid = false, exchanged = false, dp = false, sharded = false, all_final_recorded = false, dw_joined = true, c_early = false; uint64_t split[2] = {0, 0}, split_mid = 0; int order = 0;
	         // (round 6, RNB_SCATTER_C_SIDE) c_side: group C has NOT been launched by the backward pass: the optimizer launches it (c_launch) with its optimizer chunk at the END of the
	         // weight-gradient stream, behind the last atomic group (c_after), so that the critical stream falls idle behind its last atomic group and works off the barrier packets of the
	         // joins while the side streams finish (each costs it ~5 us, tools/probe_barriers.hip -- 16-18 us of idle queue in front of every network evaluation)
	         bool c_side = false; hipEvent_t c_after = nullptr; std::function<void(hipStream_t, hipEvent_t)> c_launch; } sc;
	// level groups of the gradient scatter (forward_backward), fixed at creation: C = [0, e_c) LDS, B = [e_c, l_fine) run-length quads, A = [l_fine, L) plain quads
	struct ScatterGroups { uint32_t e_c = 0, l_fine = 0, Ks[RNB_MAX_LEVELS] = {}; uint64_t k_log2 = 0; } sg;
	hipStream_t backward_stream = nullptr; // the stream the last backward pass was queued on
	hipStream_t step_stream = nullptr;     // the stream of the running step (rnb_train_step_begin): what wait_loss_readback synchronises before it declares a readback lost
	uint64_t dp_split = 0, dp_mid = 0; // first parameter of the plain-quad levels (group A) / of their second half (A2): boundaries of the data-parallel gradient blocks
	uint64_t param_capacity = 0; // allocated length of the parameter-shaped arrays: padded so that the data-parallel shards are equal
	bool dp_order() const { return cfg.world_size > 1 || knobs.dp_order; }
	struct { bool begun = false, early_done = false; AdamArgs args; } opt; // optimizer state of the running step (it may be applied in two pieces) // scatter groups of the current backward pass (see forward_backward)
	struct { bool valid = false, loss_cleared = false; uint32_t n_rays = 0, n_rays_total = 0, max_inference = 0, k1 = 0; bool split = false, march_joined = false; } pre; // samples already generated for the next step
	struct Readback { double sums[3]; uint32_t counters[4]; uint32_t fwd[2]; uint32_t seq, pad; }* host_rb = nullptr; // pinned, device-mapped; same layout as the device block k_reduce_losses fills; seq: see poll_loss()
	uint32_t rb_seq = 0;     // sequence number of the last step whose readback was launched in polling mode
	bool loss_polled = true; // the host has seen that step's readback (or the step publishes through ev_loss instead)
	bool poll_loss() const { return overlap() && !dp_order() && knobs.poll_loss; }
	void* host_rb_dev = nullptr;
	bool overlap() const { return cfg.overlap != 0 && !prof.on && s_march != nullptr; }

	NetW net(bool inference) const {
		const half_t* p = inference ? params_ema.p : params_fp16.p;
		NetW n;
		n.sdf_w0 = p + off_sdf;
		n.sdf_w1 = n.sdf_w0 + 64 * 32;
		n.rgb_w0 = p + off_rgb;
		n.rgb_w1 = n.rgb_w0 + 64 * 48;
		n.rgb_w2 = n.rgb_w1 + 64 * 64;
		n.grid = reinterpret_cast<const uint32_t*>(p + off_grid);
		n.variance = p + off_var;
		return n;
	}
	GridMeta meta() const { GridMeta g = grid; g.valid_level = valid_level; return g; }
};

static void discard_premarch(rnb_ctx* c);
static bool prep_due(uint32_t step) { // testbed.cu:2805
	const uint32_t n_prep_to_skip = std::min(std::max(step / 16u, 1u), 16u);
	return step % n_prep_to_skip == 0;
}
// Group C of a training step's scatter was left to the optimizer (sc.c_side) and something else needs the gradients first: launch it on `s` now.
static void flush_c_side(rnb_ctx* c, hipStream_t s) {
	if (!c->sc.c_side) return;
	c->sc.c_side = false;
	c->sc.c_launch(s, nullptr);
}
// Safety net of the deferred join (rnb_ctx::tail_pending) for launches outside the training step's own sequence: wait on the host.
static void join_tail_host(rnb_ctx* c) {
	if (c->join_pending) { (void)hipEventSynchronize(c->ev_join); c->join_pending = false; }
	if (c->tail_pending) { (void)hipEventSynchronize(c->ev_tail); c->tail_pending = false; }
}

// The one-launch scans (kernels_ray.cuh, chain_prefix) report a wait that gave up through two mapped host words; read after a synchronisation.
// The launch itself has poisoned its result (zero counters -- the fused scan of k_loss_pass2_rays through k_loss_pass2_samples), so nothing was trained on it.
static int check_scan_errors(rnb_ctx* c) {
	const bool rays = c->host_coarse[5] != 0, compact = c->host_coarse[6] != 0;
	if (!rays && !compact) return RNB_OK;
	c->host_coarse[5] = 0; c->host_coarse[6] = 0;
	g_err = std::string(rays ? "k_scan_rays_chain" : "k_scan_compact_chain") + ": a tile's sums did not arrive (workgroup not scheduled); the launch reported zero counters";
	return RNB_ERR_DEVICE;
}

namespace {

constexpr size_t LDS_TRAIN = (size_t)(W_TRAIN_END + WAVES_PER_WG * 3 * ACT_TILE_HALFS) * sizeof(half_t);
static_assert(LDS_TRAIN <= 160 * 1024, "training kernel LDS exceeds 160 KiB");

void build_grid_tables(rnb_ctx* c) { // grid.h:977-1012
	const rnb_config& cfg = c->cfg;
	std::memset(&c->grid, 0, sizeof(c->grid));
	c->grid.n_levels = cfg.n_levels;
	uint32_t offset = 0;
	for (uint32_t i = 0; i < cfg.n_levels; ++i) {
		const float scale = exp2f(i * std::log2(cfg.per_level_scale)) * cfg.base_resolution - 1.0f;
		const uint32_t resolution = (uint32_t)(ceilf(scale)) + 1;
		c->grid.scale[i] = (float)(resolution - 1);
		c->grid.resolution[i] = resolution;
		uint32_t max_params = std::numeric_limits<uint32_t>::max() / 2;
		uint32_t params_in_level = std::pow((float)resolution, 3) > (float)max_params ? max_params : resolution * resolution * resolution;
		params_in_level = next_multiple_u32(params_in_level, 8u);
		params_in_level = std::min(params_in_level, (1u << cfg.log2_hashmap_size));
		c->grid.offsets[i] = offset;
		offset += params_in_level;
	}
	for (uint32_t i = cfg.n_levels; i <= RNB_MAX_LEVELS; ++i) c->grid.offsets[i] = offset;
	c->n_grid_params = (uint64_t)offset * 2;
	c->n_dense_lead = 0; // the leading levels whose tables hold the whole lattice (fill_level_meta's rule): the pipelined encode gathers their x-pairs with one load (level_issue<true>)
	for (uint32_t i = 0; i < cfg.n_levels; ++i) {
		const uint64_t r = c->grid.resolution[i], size = c->grid.offsets[i + 1] - c->grid.offsets[i];
		if (r * r * r > size) break;
		c->n_dense_lead = i + 1;
	}
}

void build_light_dirs(rnb_ctx* c) { // testbed_nerf.cu:1537-1554
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

int derive_half_params(rnb_ctx* c, hipStream_t s) { // trainer.h:103-107
	hipLaunchKernelGGL(k_fp32_to_half, dim3(2048), dim3(256), 0, s, c->params_fp32.p, c->params_fp16.p, c->n_params);
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

// Brings the optimizer records up to date with the staging views (after rnb_init_params / rnb_set_params / a caller's rnb_buffer) -- in front of
// an optimizer launch -- or the views with the records (rnb_buffer). Both are whole-array copies behind a device synchronisation: rare events.
static int ensure_opt_records(rnb_ctx* c) {
	if (c->opt_rec_current) return RNB_OK;
	HIP_TRY(hipDeviceSynchronize());
	hipLaunchKernelGGL(k_opt_records_pack, dim3(4096), dim3(256), 0, 0, c->param_capacity / 4, c->params_fp32.p, c->adam_m.p, c->adam_v.p, c->adam_steps.p, c->opt_rec.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	c->opt_rec_current = true;
	return RNB_OK;
}
static int ensure_opt_views(rnb_ctx* c) {
	if (c->opt_plain_current) return RNB_OK;
	HIP_TRY(hipDeviceSynchronize()); // a pipelined step may still be running its optimizer
	hipLaunchKernelGGL(k_opt_records_unpack, dim3(4096), dim3(256), 0, 0, c->param_capacity / 4, c->opt_rec.p, c->params_fp32.p, c->adam_m.p, c->adam_v.p, c->adam_steps.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	c->opt_plain_current = true;
	return RNB_OK;
}

// (callers have written, or are about to overwrite, the whole fp32 master array of the staging views)
int reset_optimizer_state(rnb_ctx* c) {
	HIP_TRY(hipMemset(c->adam_m.p, 0, c->adam_m.bytes()));
	HIP_TRY(hipMemset(c->adam_v.p, 0, c->adam_v.bytes()));
	HIP_TRY(hipMemset(c->adam_steps.p, 0, c->adam_steps.bytes()));
	HIP_TRY(hipMemset(c->grad_ptr(0), 0, c->param_capacity * c->grad_elem()));
	c->grads_clean = true;
	c->optimizer_step_count = 0;
	c->opt_plain_current = true; c->opt_rec_current = false; // the views are the truth until the next optimizer launch packs them
	return RNB_OK;
}

// ---- K5 ----
// The LDS form of the occupancy for the march kernels (kernels_ray.cuh) + its block count on the host (the launches size their LDS
// by it). Runs after an occupancy update (every 16th step, which synchronises anyway) or when a caller may have written the bitfield.
static int rebuild_coarse(rnb_ctx* c, hipStream_t s, bool wait = true) {
	// ev_grid = this kernel's completion: an occupancy update is over (pregenerate_grid_samples starts behind it, beside the march)
	LAUNCH_EV(k_coarse_bitfield, dim3(1), dim3(1024), 0, s, (c->overlap() && !wait) ? c->ev_grid : nullptr, c->bitfield.p, c->coarse_bits.p, c->coarse_count.p, c->host_coarse_dev);
	HIP_TRY(hipGetLastError());
	c->coarse_valid = true;
	// The block count only sizes the LDS image of later march launches; a launch with any other size takes the same decisions (blocks beyond its
	// budget are read from the bitfield, load_coarse / occupied_mip0). The training loop therefore does not wait for it: the kernel writes the count
	// into mapped host memory and march_args reads whatever has arrived (the previous update's count until then) plus a margin. A caller-written
	// bitfield (rnb_bitfield_changed) can be of any shape, so that path waits as before.
	if (wait || *c->host_coarse == 0xffffffffu) HIP_TRY(hipStreamSynchronize(s));
	return RNB_OK;
}

// have_partials: k_ema_mean has left the 1024 partial sums of the mean (update_density_grid); otherwise they are formed here.
int update_bitfield(rnb_ctx* c, hipStream_t s, bool wait = true, bool have_partials = false) { // testbed_nerf.cu:3497-3517
	const uint32_t n_blocks = 1024;
	if (!have_partials) hipLaunchKernelGGL(k_mean_partial, dim3(n_blocks), dim3(256), 0, s, c->density_grid.p, c->mean_partial.p);
	const uint32_t n_bytes_per_mip = GRID_CELLS / 8;
	if (c->aabb.max_cascade == 0 && !c->bitfield_foreign && c->knobs.fused_update) {
		// single-cascade scene: mean + level 0 + level 1 in one launch, the upper levels + the march's LDS form in a second one (kernels_ray.cuh)
		hipLaunchKernelGGL(k_bitfield_sc, dim3(n_bytes_per_mip / 256), dim3(256), 0, s, c->density_grid.p, c->bitfield.p, c->mean_partial.p, c->density_mean.p);
		LAUNCH_EV(k_pool_tail_coarse, dim3(2), dim3(1024), 0, s, (c->overlap() && !wait) ? c->ev_grid : nullptr, 2u, c->bitfield.p, c->coarse_bits.p, c->coarse_count.p, c->host_coarse_dev);
		HIP_TRY(hipGetLastError());
		c->coarse_valid = true;
		if (wait || *c->host_coarse == 0xffffffffu) HIP_TRY(hipStreamSynchronize(s)); // (see rebuild_coarse)
		return RNB_OK;
	}
	hipLaunchKernelGGL(k_mean_final, dim3(1), dim3(64), 0, s, c->mean_partial.p, n_blocks, c->density_mean.p);
	const uint32_t n_el = n_bytes_per_mip * N_CASCADES;
	hipLaunchKernelGGL(k_grid_to_bitfield, dim3((n_el + 127) / 128), dim3(128), 0, s, n_el, n_bytes_per_mip * (c->aabb.max_cascade + 1), c->density_grid.p, c->bitfield.p, c->density_mean.p);
	// levels that pool a level with bits of its own (the scene's cascades): the full kernel; the levels above them in one small launch
	const uint32_t first_tail = std::min<uint32_t>(c->aabb.max_cascade + 2, N_CASCADES);
	for (uint32_t level = 1; level < first_tail; ++level) {
		hipLaunchKernelGGL(k_bitfield_max_pool, dim3((GRID_CELLS / 64 + 127) / 128), dim3(128), 0, s, GRID_CELLS / 64,
		                   c->bitfield.p + (size_t)n_bytes_per_mip * (level - 1), c->bitfield.p + (size_t)n_bytes_per_mip * level);
	}
	if (first_tail < N_CASCADES) hipLaunchKernelGGL(k_bitfield_max_pool_tail, dim3(1), dim3(1024), 0, s, first_tail, c->bitfield.p);
	HIP_TRY(hipGetLastError());
	c->bitfield_foreign = false; // k_grid_to_bitfield has zero-filled every level
	return rebuild_coarse(c, s, wait);
}

int launch_point_query(rnb_ctx* c, hipStream_t s, const float* xyz, uint32_t n, half_t* out, const uint32_t* splat_idx, float* grid_tmp, int want_density, bool inference, const uint32_t* range = nullptr) {
	if (n == 0) return RNB_OK;
	join_tail_host(c); // (inference launches too: the same side-stream launch writes the MLPs' and the variance's EMA weights)
	PointArgs a;
	a.xyz = xyz; a.n = n; a.out = out; a.splat_idx = splat_idx; a.grid_tmp = grid_tmp; a.want_density = want_density; a.sdf_bias = c->cfg.sdf_bias; a.range = range; a.xcd = c->knobs.point_xcd ? 1u : 0u;
	const uint32_t n_tiles = (n + TILE - 1) / TILE;
	const uint32_t grid = std::min<uint32_t>((n_tiles + WAVES_PER_WG - 1) / WAVES_PER_WG, (uint32_t)c->n_cus * 5); // 86 VGPRs, 28 KB of LDS: five workgroups per CU
	if (c->half_acc() && c->knobs.encode_depth) hipLaunchKernelGGL(k_point_query_chained_emul_pipe<4>, dim3(grid), dim3(WG), LDS_POINT2, s, c->meta(), c->net(inference), a, (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr);
	else if (c->half_acc()) hipLaunchKernelGGL(k_point_query_chained_emul, dim3(grid), dim3(WG), LDS_POINT2, s, c->meta(), c->net(inference), a, (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr);
	else if (c->knobs.encode_depth == 4 && c->knobs.encode_pair && c->n_dense_lead >= 5) hipLaunchKernelGGL((k_point_query_chained_pipe<4, 5>), dim3(grid), dim3(WG), LDS_POINT2, s, c->meta(), c->net(inference), a, (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr);
	else if (c->knobs.encode_depth == 4) hipLaunchKernelGGL(k_point_query_chained_pipe<4>, dim3(grid), dim3(WG), LDS_POINT2, s, c->meta(), c->net(inference), a, (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr);
	else if (c->knobs.encode_depth == 7) hipLaunchKernelGGL(k_point_query_chained_pipe<7>, dim3(grid), dim3(WG), LDS_POINT2, s, c->meta(), c->net(inference), a, (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr);
	else if (c->knobs.encode_depth == 2) hipLaunchKernelGGL(k_point_query_chained_pipe<2>, dim3(grid), dim3(WG), LDS_POINT2, s, c->meta(), c->net(inference), a, (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr);
	else hipLaunchKernelGGL(k_point_query_chained, dim3(grid), dim3(WG), LDS_POINT2, s, c->meta(), c->net(inference), a, (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr);
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

// ---- K1-K5 ----
// K1 for one update: the uniform and the non-uniform pass (testbed_nerf.cu:3385-3409) from the generator state `rng` (advanced twice).
static int launch_grid_samples(rnb_ctx* c, hipStream_t s, Pcg32& rng, uint32_t ema_step, uint32_t n_uniform, uint32_t n_nonuniform, float* pos, uint32_t* idx, uint32_t* hist, uint32_t key_shift) {
	if (n_uniform) hipLaunchKernelGGL(k_grid_samples, dim3((n_uniform + 127) / 128), dim3(128), 0, s, n_uniform, rng, ema_step, c->aabb, c->density_grid.p,
	                                  pos, idx, -0.01f, hist, key_shift);
	rng.advance();
	if (n_nonuniform) hipLaunchKernelGGL(k_grid_samples, dim3((n_nonuniform + 127) / 128), dim3(128), 0, s, n_nonuniform, rng, ema_step, c->aabb, c->density_grid.p,
	                                     pos + (size_t)n_uniform * 3, idx + n_uniform, MIN_OPTICAL_THICKNESS, hist, key_shift);
	rng.advance();
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

// The samples of the NEXT occupancy update, generated and put into cell order (k_grid_samples_place) as soon as this update's grid is final:
// they depend on that grid, the generator state and the update counter only. In the overlapped schedule this runs on a side stream beside
// the step's march, off the critical path; the next update (16 steps later) finds them ready, checks that nothing they depend on has changed
// (ema step, generator state, sizes; entry points through which a caller can change the grid invalidate them) and evaluates the network in
// that order. Same sample SET as the reference's order, same atomicMax splat, same grid.
static int pregenerate_grid_samples(rnb_ctx* c, hipStream_t s_main) {
	if (!c->gs_todo.pending) return RNB_OK;
	c->gs_todo.pending = false;
	c->gs_pre.valid = false;
	const uint32_t n_uniform = c->gs_todo.n_uniform, n_nonuniform = c->gs_todo.n_nonuniform;
	if (!c->knobs.grid_presort || c->training_step < 256 || n_nonuniform == 0) return RNB_OK; // the first 256 steps sample every cell each step: nothing to gain
	const uint32_t n_elements = GRID_CELLS * (c->aabb.max_cascade + 1);
	const uint32_t n = n_uniform + n_nonuniform;
	uint32_t shift = 3;
	while ((n_elements >> shift) > (1u << 20)) ++shift; // the two-level scan below covers 2^20 keys
	const uint32_t n_keys = n_elements >> shift;
	if (!c->gs_sorted_pos.p) {
		if (c->gs_range.alloc(2) != hipSuccess || c->gs_sorted_pos.alloc((size_t)n * 3) != hipSuccess || c->gs_sorted_idx.alloc(n) != hipSuccess || c->gs_hist.alloc((size_t)(1u << 20) + 2048) != hipSuccess ||
		    c->gs_stage_pos.alloc(c->grid_sample_pos.n) != hipSuccess || c->gs_stage_idx.alloc(c->grid_sample_idx.n) != hipSuccess ||
		    c->gs_eval_pos.alloc((size_t)n * 3) != hipSuccess || c->gs_eval_idx.alloc(n) != hipSuccess) {
			// a sharded update divides the samples by the order they are evaluated in: one rank that falls back to the reference's order alone would leave
			// cells uncovered without any error, so a data-parallel job fails here instead (RNB_GRID_PRESORT must likewise be the same on every rank)
			if (c->cfg.world_size > 1 && c->grid_exchange) return fail(RNB_ERR_NOMEM, "hipMalloc failed for the cell-ordered occupancy samples (data parallel: every rank must evaluate the same order)");
			c->knobs.grid_presort = false; // not essential: the update then keeps the reference's order
			return RNB_OK;
		}
	}
	hipStream_t s = s_main;
	if (c->overlap()) { // after this update's k_ema_grid (ev_grid), beside what follows it on s_main: the march. (Behind the march chain instead -- the placement's
		// returning atomics slow k_scan_rays_chain's polls, 11 -> 54 us at an eighth of the batch -- it runs beside the network evaluation, which is bound by the same
		// memory system: k_forward_chained 72 -> 130 us twice, the update step 1134 -> 1265 us; round 4, dropped.)
		s = c->s_dw;
		HIP_TRY(hipStreamWaitEvent(s, c->ev_grid, 0));
	}
	HIP_TRY(hipMemsetAsync(c->gs_hist.p, 0, sizeof(uint32_t) * n_keys, s));
	if (c->density_grid_tmp_alt.p) { // the next update's splat target, cleared here instead of in front of its network evaluation (ordered by ev_gs like the samples)
		HIP_TRY(hipMemsetAsync(c->density_grid_tmp_alt.p, 0, sizeof(float) * n_elements, s));
		c->tmp_alt_clear = true;
	}
	Pcg32 rng = c->density_grid_rng;
	c->gs_pre.rng_state = rng.state; c->gs_pre.rng_inc = rng.inc;
	int rc = launch_grid_samples(c, s, rng, c->density_grid_ema_step, n_uniform, n_nonuniform, c->gs_stage_pos.p, c->gs_stage_idx.p, c->gs_hist.p, shift);
	if (rc != RNB_OK) return rc;
	const uint32_t nb = (n_keys + 1023) / 1024;
	uint32_t* sums = c->gs_hist.p + (1u << 20);
	hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, s, c->gs_hist.p, (uint64_t)n_keys, sums);
	hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, sums, (uint64_t)nb, sums + 1024);
	hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, s, c->gs_hist.p, (uint64_t)n_keys, sums);
	hipLaunchKernelGGL(k_grid_samples_place, dim3((n + 127) / 128), dim3(128), 0, s, n, c->gs_stage_pos.p, c->gs_stage_idx.p, c->gs_hist.p, shift, c->gs_sorted_pos.p, c->gs_sorted_idx.p);
	if (c->cfg.world_size > 1) hipLaunchKernelGGL(k_shard_range, dim3(1), dim3(64), 0, s, c->gs_hist.p, n_keys, n, c->cfg.world_size, c->cfg.rank, c->gs_range.p); // this rank's share of the update (rnb_update_density_grid_begin)
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(c->ev_gs, s));
	c->gs_pre.valid = true; c->gs_pre.ema_step = c->density_grid_ema_step; c->gs_pre.n_uniform = n_uniform; c->gs_pre.n_nonuniform = n_nonuniform;
	return RNB_OK;
}

// First half of an occupancy update (K1-K3). shard: this rank evaluates its 1 / world_size of the samples only (the caller takes the element-wise max of
// density_grid_tmp over the ranks before update_density_grid_back).
int update_density_grid_front(rnb_ctx* c, hipStream_t s, uint32_t n_uniform, uint32_t n_nonuniform, bool shard) { // testbed_nerf.cu:3424-3490
	const uint32_t n_elements = GRID_CELLS * (c->aabb.max_cascade + 1);
	const uint32_t n_samples = n_uniform + n_nonuniform;
	c->prof.mark(s, P_NONE);
	if (c->training_step == 0) {
		c->density_grid_ema_step = 0;
		HIP_TRY(hipMemsetAsync(c->density_grid.p, 0, sizeof(float) * n_elements, s));
		c->gs_pre.valid = false;
	}
	const bool sorted = c->gs_pre.valid && c->gs_pre.ema_step == c->density_grid_ema_step && c->gs_pre.n_uniform == n_uniform && c->gs_pre.n_nonuniform == n_nonuniform &&
	                    c->gs_pre.rng_state == c->density_grid_rng.state && c->gs_pre.rng_inc == c->density_grid_rng.inc;
	c->gs_pre.valid = false;
	c->last_update_sorted = sorted;
	if (sorted && c->tmp_alt_clear) std::swap(c->density_grid_tmp, c->density_grid_tmp_alt); // cleared behind the previous update, ordered by ev_gs below
	else HIP_TRY(hipMemsetAsync(c->density_grid_tmp.p, 0, sizeof(float) * n_elements, s));
	c->tmp_alt_clear = false;
	if (sorted) { // generated after the previous update (pregenerate_grid_samples): the staging pair holds the reference's order, gs_sorted_* the cell order
		HIP_TRY(hipStreamWaitEvent(s, c->ev_gs, 0));
		std::swap(c->grid_sample_pos, c->gs_stage_pos); // RNB_BUF_GRID_SAMPLE_POS / _IDX: the samples of the LAST update, as always
		std::swap(c->grid_sample_idx, c->gs_stage_idx);
		std::swap(c->gs_sorted_pos, c->gs_eval_pos);     // the next pregenerate_grid_samples writes the other pair
		std::swap(c->gs_sorted_idx, c->gs_eval_idx);
		c->density_grid_rng.advance();
		c->density_grid_rng.advance();
	} else {
		int rc = launch_grid_samples(c, s, c->density_grid_rng, c->density_grid_ema_step, n_uniform, n_nonuniform, c->grid_sample_pos.p, c->grid_sample_idx.p, nullptr, 0);
		if (rc != RNB_OK) return rc;
	}
	c->prof.mark(s, P_GRID_SAMPLES);
	c->n_grid_samples = n_samples;
	int rc;
	const uint64_t W = shard ? c->cfg.world_size : 1u, r = shard ? c->cfg.rank : 0u;
	if (sorted) { // cell order: the share's bounds sit at cell-block boundaries and live on the device (k_shard_range, behind the placement)
		rc = launch_point_query(c, s, c->gs_eval_pos.p, n_samples, nullptr, c->gs_eval_idx.p, c->density_grid_tmp.p, 1, false, W > 1 ? c->gs_range.p : nullptr);
	} else { // the reference's order: the same on every rank, shares by count
		const uint32_t lo = (uint32_t)((uint64_t)n_samples * r / W), hi = (uint32_t)((uint64_t)n_samples * (r + 1) / W);
		rc = launch_point_query(c, s, c->grid_sample_pos.p + (size_t)lo * 3, hi - lo, nullptr, c->grid_sample_idx.p + lo, c->density_grid_tmp.p, 1, false);
	}
	if (rc != RNB_OK) return rc;
	c->prof.mark(s, P_POINT_QUERY);
	c->prof.units[P_POINT_QUERY] += n_samples / W;
	c->gs_todo.n_uniform = n_uniform; c->gs_todo.n_nonuniform = n_nonuniform; // (pending is set by the second half)
	return RNB_OK;
}

// Second half (K4-K5) from density_grid_tmp.
int update_density_grid_back(rnb_ctx* c, hipStream_t s) { // testbed_nerf.cu:3491-3517
	const uint32_t n_elements = GRID_CELLS * (c->aabb.max_cascade + 1);
	int rc;
	if (c->knobs.fused_update) hipLaunchKernelGGL(k_ema_mean, dim3(n_elements / 2048), dim3(256), 0, s, n_elements, c->cfg.density_grid_decay, c->density_grid.p, c->density_grid_tmp.p, c->mean_partial.p);
	else hipLaunchKernelGGL(k_ema_grid, dim3((n_elements + 127) / 128), dim3(128), 0, s, n_elements, c->cfg.density_grid_decay, c->density_grid.p, c->density_grid_tmp.p);
	HIP_TRY(hipGetLastError());
	++c->density_grid_ema_step;
	rc = update_bitfield(c, s, !c->overlap(), c->knobs.fused_update);
	c->prof.mark(s, P_EMA_BITFIELD);
	c->gs_todo.pending = true; // queued by the caller once the kernels that wait for THIS update are in their queue
	return rc;
}

int training_prep_front(rnb_ctx* c, hipStream_t s, bool shard) { // testbed_nerf.cu:4125-4138
	const uint32_t n_cascades = c->aabb.max_cascade + 1;
	if (c->training_step < 256) return update_density_grid_front(c, s, GRID_CELLS * n_cascades, 0, shard);
	return update_density_grid_front(c, s, GRID_CELLS / 4 * n_cascades, GRID_CELLS / 4 * n_cascades, shard);
}
// The whole update; sharded over the data-parallel ranks when the host has given an exchange (rnb_set_grid_exchange).
int training_prep(rnb_ctx* c, hipStream_t s) {
	const bool shard = c->grid_exchange != nullptr && c->dp_order(); // (a world of one with RNB_DP_FORCE_COLLECTIVES: the whole update, and the exchange is called: the path is exercised)
	int rc = training_prep_front(c, s, shard);
	if (rc != RNB_OK) return rc;
	if (shard) {
		const int xrc = c->grid_exchange(c->grid_exchange_user, c->density_grid_tmp.p, (uint64_t)GRID_CELLS * (c->aabb.max_cascade + 1), (void*)s);
		if (xrc != 0) return fail(RNB_ERR_INVALID, "the occupancy grid exchange (rnb_set_grid_exchange) failed");
	}
	return update_density_grid_back(c, s);
}

// The buffers of the albedo mode's training kernels, on first use (the mode can be switched on by rnb_update_config).
static int ensure_rgb_buffers(rnb_ctx* c) {
	if (c->cin_eval.p) return RNB_OK;
	const size_t B = c->cfg.target_batch_size;
	if (c->cin_eval.alloc(B * 16 * 32) != hipSuccess || c->dcin.alloc(B * 32) != hipSuccess || c->rgb_out_scratch.alloc(B * 16) != hipSuccess || c->src_slot.alloc(B) != hipSuccess)
		return fail(RNB_ERR_NOMEM, "hipMalloc failed for the colour-MLP training buffers");
	return RNB_OK;
}

int launch_forward(rnb_ctx* c, hipStream_t s, const float* coords, const uint32_t* n_ptr, uint32_t n_max, half_t* out, bool inference, const uint32_t* idx = nullptr, half_t* cin_out = nullptr) {
	if (n_max == 0) return RNB_OK;
	join_tail_host(c); // (inference launches too: the same side-stream launch writes the MLPs' and the variance's EMA weights)
	FwdArgs a;
	a.coords = coords; a.n_ptr = n_ptr; a.n_max = n_max; a.out = out; a.sdf_bias = c->cfg.sdf_bias; a.idx = idx; a.cin_out = cin_out;
	a.wimg = (!inference && c->wimg_valid) ? c->wimg_fwd.p : nullptr;
	const uint32_t n_tiles = (n_max + TILE - 1) / TILE;
	const uint32_t grid = std::min<uint32_t>((n_tiles + WAVES_PER_WG - 1) / WAVES_PER_WG, (uint32_t)c->n_cus * 2);
	if (c->half_acc() && c->knobs.encode_depth) hipLaunchKernelGGL(k_forward_chained_emul_pipe<4>, dim3(grid), dim3(WG), LDS_FWD2, s, c->meta(), c->net(inference), a);
	else if (c->half_acc()) hipLaunchKernelGGL(k_forward_chained_emul, dim3(grid), dim3(WG), LDS_FWD2, s, c->meta(), c->net(inference), a);
	else if (c->knobs.encode_depth == 2) hipLaunchKernelGGL(k_forward_chained_pipe<2>, dim3(grid), dim3(WG), LDS_FWD2, s, c->meta(), c->net(inference), a);
	else if (c->knobs.encode_depth == 4 && c->knobs.encode_pair && c->n_dense_lead >= 5) hipLaunchKernelGGL((k_forward_chained_pipe<4, 5>), dim3(grid), dim3(WG), LDS_FWD2, s, c->meta(), c->net(inference), a);
	else if (c->knobs.encode_depth == 4) hipLaunchKernelGGL(k_forward_chained_pipe<4>, dim3(grid), dim3(WG), LDS_FWD2, s, c->meta(), c->net(inference), a);
	else if (c->knobs.encode_depth == 7) hipLaunchKernelGGL(k_forward_chained_pipe<7>, dim3(grid), dim3(WG), LDS_FWD2, s, c->meta(), c->net(inference), a);
	else hipLaunchKernelGGL(k_forward_chained, dim3(grid), dim3(WG), LDS_FWD2, s, c->meta(), c->net(inference), a);
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

static LossFlags loss_flags(const rnb_ctx* c) {
	LossFlags F;
	F.apply_L2 = c->cfg.apply_L2; F.apply_rgbplus = c->cfg.apply_rgbplus; F.apply_no_albedo = c->cfg.apply_no_albedo; F.apply_light_opti = c->cfg.apply_light_opti;
	F.apply_relu = c->cfg.apply_relu; F.apply_bce = c->cfg.apply_bce; F.snap = c->cfg.snap_to_pixel_centers;
	F.mask_loss_weight = c->cfg.mask_loss_weight; F.ek_loss_weight = c->cfg.ek_loss_weight;
	return F;
}

// Head length of the two-round network evaluation for a batch of n_rays rays. The results do not depend on it (tests/test_gpu_fullsize.py); the step time
// does, sharply (tools/sweep_k1.sh, profiles/r04_sweep_k1.txt, ms/step of 200-step slices with a fixed head of 12 / 16 / 24 / 32 / 40 / 48 / 64):
//   step 1000, 13 k rays: .755 .753 .728 .656 .621 .615 .620     step 2000, 48 k: .626 .608 .602 .617 .628 .644 .658
//   step 1400, 22 k rays: .715 .692 .627 .598 .597 .602 .628     step 3000, 85 k: .627 .619 .638 .650 .664 .669 .678
//   step 1700, 33 k rays: .666 .624 .596 .589 .603 .618 .646     step 6000, 95 k: .637 .636 .657 .671 .680 .685 .692
// The optimum follows 8 + 3 B / n_rays (the controller holds the compacted batch at B, so B / n_rays is the kept samples per ray), in multiples of 8 within
// [16, 48]. (Round 2's rule, 4.5 B / n_rays within [12, 48], was fitted to that round's kernels: 48 instead of 40 at 22 k rays, 14 instead of 16 at 85 k.)
static uint32_t k1_for(const rnb_ctx* c, uint32_t n_rays) {
	if (c->fwd_k1 == 0 || c->fwd_k1_fixed) return c->fwd_k1;
	const float k = 8.f + 3.f * (float)c->cfg.target_batch_size / (float)std::max(1u, n_rays);
	const uint32_t k8 = (uint32_t)std::min(48.f, std::max(16.f, k) + 4.f) / 8u * 8u;
	return std::min(c->fwd_k1, std::max(16u, k8));
}

// Is t -> fl(t + C) the addition of ONE constant inside each binade of [0.25, 8)? It is unless C sits exactly half way between two
// multiples of the binade's ulp (a tie, rounded to even: the increment would then alternate with the parity of t). Checked on an even and
// an odd multiple of the ulp at both ends of every binade; k_march_count_wide's closed form relies on it.
static bool lattice_is_linear() {
	const volatile float C = STEPSIZE;
	for (int e = -1; e <= 3; ++e) { // binades [2^(e-1), 2^e)
		const float base = std::ldexp(0.5f, e), ulp = std::ldexp(1.0f, e - 1 - 23);
		const volatile float d0 = (base + C) - base;
		for (const float t : {base, base + ulp, base + 2 * ulp, base + 3 * ulp, std::ldexp(1.0f, e) - 64 * d0, std::ldexp(1.0f, e) - 64 * d0 + ulp}) {
			const volatile float sum = t + C;
			if (sum - t != d0) return false;
		}
	}
	return true;
}

MarchArgs march_args(rnb_ctx* c, uint32_t n_rays, uint32_t n_rays_total, uint32_t max_samples) {
	MarchArgs a;
	static const bool linear = lattice_is_linear();
	a.lattice_ok = (linear && !c->knobs.march_running_sums) ? 1u : 0u;
	a.n_rays = n_rays;
	a.n_rays_global = n_rays * c->cfg.world_size;
	a.ray_offset = c->cfg.rank * n_rays;
	a.n_rays_total = n_rays_total;
	a.max_samples = max_samples;
	a.n_images = c->n_views;
	a.snap = c->cfg.snap_to_pixel_centers;
	a.rng = c->rng;
	a.A = c->aabb;
	a.views = c->views.p;
	a.bitfield = c->bitfield.p; a.coarse = c->coarse_bits.p;
	{ // LDS image: the count the device last reported + 1/8 + 64 blocks (the update in flight may have grown the surface; any size is exact)
		const uint32_t seen = *reinterpret_cast<volatile uint32_t*>(c->host_coarse);
		const uint32_t budget = seen == 0xffffffffu ? 0u : (seen + seen / 8 + 64 + 63) / 64 * 64;
		a.n_blocks_lds = seen <= COARSE_MAX_BLOCKS ? std::min(budget, COARSE_MAX_BLOCKS) : 0u; // a volume rather than a surface (early training): coarse bits only, the cell bits from the bitfield
	}
	a.setup = c->ray_setup.p; a.ray_t = c->ray_t.p; a.d_unnorm = c->ray_dunnorm.p; a.steps = c->ray_steps.p; a.base = c->ray_base.p; a.slot = c->ray_slot.p;
	a.ray_indices = c->ray_indices.p; a.rays = c->rays.p; a.numsteps = c->numsteps.p; a.coords = c->coords.p; a.counters = c->counters.p;
	a.base1 = c->ray_base1.p; a.idx1 = c->idx1.p; a.k1 = k1_for(c, n_rays);
	a.F = loss_flags(c);
	for (int k = 0; k < 9; ++k) a.light_dirs[k] = c->light_dirs[k];
	a.ray_const = c->ray_const.p;
	a.part = 0;
	a.stats = c->march_stats.p;
	a.prio = (uint32_t)c->knobs.march_prio;
	a.use_bbox = (uint32_t)c->knobs.march_bbox;
	return a;
}

int generate_training_samples(rnb_ctx* c, hipStream_t s, uint32_t n_rays, uint32_t n_rays_total, uint32_t max_samples, hipEvent_t done = nullptr, hipEvent_t wait_before_write = nullptr, hipEvent_t rest_done = nullptr, hipEvent_t wait_before_scans = nullptr) {
	if (!c->coarse_valid) { const int rc0 = rebuild_coarse(c, s); if (rc0 != RNB_OK) return rc0; } // a caller may have written the bitfield (rnb_buffer)
	MarchArgs a = march_args(c, n_rays, n_rays_total, max_samples);
	c->gen_k1 = a.k1;
	const uint32_t blocks = (n_rays + 127) / 128;
	c->prof.mark(s, P_NONE);
	// One thread per ray once there are enough rays to fill the chip that way (late in training the converged occupancy grid
	// lets the controller raise the batch from 15 k to 100 k rays): 0.23 ms at 94 k rays against 0.77 ms for the 16-lanes-per-ray
	// kernel, which in turn wins below ~30 k rays (0.20 vs 0.30 ms at 14 k), where a ray per thread leaves the GPU to latency.
	const bool sc = c->aabb.cone_angle == 0.f && c->aabb.max_cascade == 0; // one cascade, constant step: the specialised instances
	const size_t march_lds = sc ? (size_t)(2 * COARSE_WORDS + 2 * a.n_blocks_lds) * sizeof(uint32_t) : 0;
	if (c->knobs.march_narrow || n_rays >= c->knobs.march_narrow_from) {
		if (sc && c->knobs.march_skip && c->knobs.march_skip_narrow && a.lattice_ok) { // RNB_MARCH_SKIP_NARROW=1 (dropped, kept as an A/B knob): the thread-per-ray march minus the stretches that cannot hold a sample (kernels_ray.cuh: march_skip_narrow)
			if (c->knobs.march_skip == 2) a.lattice_ok |= 2u;
			hipLaunchKernelGGL(k_march_count_skip_narrow, dim3(blocks), dim3(128), march_lds + COARSE_WORDS * sizeof(uint32_t), s, a);
			a.lattice_ok &= 1u;
		} else if (sc && c->knobs.march_bbox >= 2 && a.lattice_ok) {
			if (c->knobs.march_skip == 2) a.lattice_ok |= 2u; // (tests: no re-entry cell is accepted -- every ray walks every voxel up to the occupied box's exit)
			hipLaunchKernelGGL(k_march_count_bbox, dim3(blocks), dim3(128), march_lds, s, a);
			a.lattice_ok &= 1u;
		} else if (sc) hipLaunchKernelGGL(k_march_count<true>, dim3((n_rays + c->knobs.march_narrow_wgs - 1) / c->knobs.march_narrow_wgs), dim3(c->knobs.march_narrow_wgs), march_lds, s, a);
		else hipLaunchKernelGGL(k_march_count<false>, dim3(blocks), dim3(128), 0, s, a);
	} else if (sc && n_rays <= c->knobs.march_wave_per_ray_below) {
		// a batch so small that 16 lanes per ray leave most SIMDs without a wavefront (a rank of a strong-scaling job: 1.8 k rays = 0.4 wavefronts per SIMD): the kernel
		// is then one wavefront's dependent chain of ~50 rounds, and a whole wavefront per ray quarters the rounds (at 12.6 k rays the same form lost, 660 vs 184 us: DESIGN.md section 6)
		hipLaunchKernelGGL((k_march_count_wide<64, true, 256>), dim3((n_rays + 3) / 4), dim3(256), march_lds, s, a);
	} else {
		const dim3 grid((n_rays + 15) / 16); // 16 lanes per ray (8: 0.19 ms alone but a slower step; 32: 0.32 ms, measured in round 1)
		if (sc && c->knobs.march_skip && a.lattice_ok) {
			// round 6: the same rounds, minus the stretches of the ray that cannot hold a sample (kernels_ray.cuh: k_march_count_skip); bit-identical sample set and t
			if (c->knobs.march_skip == 2) a.lattice_ok |= 2u;
			if (c->knobs.march_wgs == 512) hipLaunchKernelGGL((k_march_count_skip<512>), dim3((n_rays + 31) / 32), dim3(512), march_lds + COARSE_WORDS * sizeof(uint32_t), s, a);
			else if (c->knobs.march_wgs == 1024) hipLaunchKernelGGL((k_march_count_skip<1024>), dim3((n_rays + 63) / 64), dim3(1024), march_lds + COARSE_WORDS * sizeof(uint32_t), s, a);
			else hipLaunchKernelGGL((k_march_count_skip<256>), grid, dim3(256), march_lds + COARSE_WORDS * sizeof(uint32_t), s, a);
			a.lattice_ok &= 1u;
		} else if (sc) hipLaunchKernelGGL((k_march_count_wide<16, true>), grid, dim3(256), march_lds, s, a);
		else hipLaunchKernelGGL((k_march_count_wide<16, false>), grid, dim3(256), 0, s, a);
	}
	c->prof.mark(s, P_MARCH_COUNT);
	if (wait_before_scans) HIP_TRY(hipStreamWaitEvent(s, wait_before_scans, 0)); // (launch_premarch: the previous step's second loss pass may still be reading what the scans and k_march_write overwrite)
	const uint32_t n_scan_tiles = (n_rays + SCAN_TILE - 1) / SCAN_TILE;
	if (c->knobs.scan_chain && n_scan_tiles <= 64) { // one launch, one workgroup per 4096-ray tile (k_scan_rays_chain)
		ScanChainArgs q;
		q.n = n_rays; q.max_samples = max_samples; q.k1 = a.k1; q.steps = c->ray_steps.p; q.base = c->ray_base.p; q.slot = c->ray_slot.p; q.base1 = c->ray_base1.p;
		q.counters = c->counters.p; q.fwd_counts = c->fwd_counts.p; q.words = c->scan_words.p; q.words2 = c->scan_words.p + 2 * 64 * 4; q.ticket = ++c->scan_ticket; q.error = c->host_coarse_dev + 5; q.plain = c->knobs.chain_plain ? 1u : 0u;
		hipLaunchKernelGGL(k_scan_rays_chain, dim3(n_scan_tiles), dim3(SCAN_WG), 0, s, q);
	} else if (n_rays >= c->knobs.march_narrow_from) { // one workgroup per 4096-ray tile (<= 64 tiles) instead of one workgroup walking them
		const uint32_t n_tiles = (n_rays + SCAN_TILE - 1) / SCAN_TILE;
		hipLaunchKernelGGL(k_scan_rays_sums, dim3(n_tiles), dim3(SCAN_WG), 0, s, n_rays, c->ray_steps.p, c->scan_tiles.p);
		hipLaunchKernelGGL(k_scan_rays_base, dim3(n_tiles), dim3(SCAN_WG), 0, s, n_rays, max_samples, a.k1, c->ray_steps.p, c->scan_tiles.p, c->ray_base.p, c->scan_tiles.p + 64, c->counters.p);
		hipLaunchKernelGGL(k_scan_rays_slots, dim3(n_tiles), dim3(SCAN_WG), 0, s, n_rays, max_samples, a.k1, c->ray_steps.p, c->ray_base.p, c->scan_tiles.p + 64, c->ray_slot.p, c->ray_base1.p, c->counters.p, c->fwd_counts.p);
	} else
		hipLaunchKernelGGL(k_scan_rays, dim3(1), dim3(1024), 0, s, n_rays, max_samples, c->ray_steps.p, c->ray_base.p, c->ray_slot.p, c->counters.p, a.k1, c->ray_base1.p, c->fwd_counts.p);
	c->prof.mark(s, P_SCAN_RAYS);
	if (wait_before_write) HIP_TRY(hipStreamWaitEvent(s, wait_before_write, 0)); // (rnb_ctx::tail_pending)
	const bool dense_const = c->knobs.ray_const_dense != 0;
	float* const ray_const = a.ray_const;
	if (dense_const) a.ray_const = nullptr; // k_ray_constants below
	c->gen_split = rest_done != nullptr && a.k1 != 0 && (c->knobs.ray_const_dense == 1 ? false : (c->knobs.march_write_split < 0 ? n_rays < 65536u : c->knobs.march_write_split != 0)); // (measured: -2.7 / -3.5 us per step at 12.6 k / 50 k rays, +3.2 at 94 k: the join costs a barrier packet)
	for (uint32_t part = c->gen_split ? 1u : 0u; part <= (c->gen_split ? 2u : 0u); ++part) {
		a.part = part;
		hipEvent_t ev = part == 2 ? rest_done : done;
		if (dense_const && part != 1) ev = nullptr; // the constants' launch carries the event
		if (c->knobs.march_write_wg == 1024) { // RNB_MARCH_WRITE_WG=1024 (A/B): rounds 2-5
			if (n_rays >= c->knobs.march_narrow_from) LAUNCH_EV((k_march_write<16, 1024>), dim3((n_rays + 63) / 64), dim3(1024), 0, s, ev, a);
			else LAUNCH_EV((k_march_write<64, 1024>), dim3((n_rays + 15) / 16), dim3(1024), 0, s, ev, a);
		} else if (c->knobs.march_write_wg == 64) {
			if (n_rays >= c->knobs.march_narrow_from) LAUNCH_EV((k_march_write<16, 64>), dim3((n_rays + 3) / 4), dim3(64), 0, s, ev, a);
			else LAUNCH_EV((k_march_write<64, 64>), dim3(n_rays), dim3(64), 0, s, ev, a);
		} else if (c->knobs.march_write_wg == 128) {
			if (n_rays >= c->knobs.march_narrow_from) LAUNCH_EV((k_march_write<16, 128>), dim3((n_rays + 7) / 8), dim3(128), 0, s, ev, a);
			else LAUNCH_EV((k_march_write<64, 128>), dim3((n_rays + 1) / 2), dim3(128), 0, s, ev, a);
		} else if (n_rays >= c->knobs.march_narrow_from) LAUNCH_EV((k_march_write<16, 256>), dim3((n_rays + 15) / 16), dim3(256), 0, s, ev, a);
		else LAUNCH_EV((k_march_write<64, 256>), dim3((n_rays + 3) / 4), dim3(256), 0, s, ev, a);
	}
	if (dense_const) LAUNCH_EV(k_ray_constants, dim3((n_rays + 63) / 64), dim3(64), 0, s, c->gen_split ? rest_done : done, a, ray_const); // one thread per kept ray
	c->prof.mark(s, P_MARCH_WRITE);
	c->prof.units[P_MARCH_COUNT] += n_rays;
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

// defer_rollover: the training step pads the batch in the same launch as its loss reduction (launch_reduce_losses)
int compute_loss(rnb_ctx* c, hipStream_t s, uint32_t n_rays, uint32_t n_rays_total, uint32_t two_round_n_max = 0, bool defer_rollover = false, bool reduce_too = false) {
	LossArgs a;
	a.n_rays = n_rays; a.n_rays_global = n_rays * c->cfg.world_size; a.ray_offset = c->cfg.rank * n_rays; a.n_rays_total = n_rays_total;
	a.n_images = c->n_views; a.B = c->cfg.target_batch_size;
	a.rng = c->rng; a.A = c->aabb;
	a.F = loss_flags(c);
	a.ray_const = two_round_n_max ? c->ray_const.p : nullptr; // stage API: flags may have changed since the samples were generated
	for (int k = 0; k < 9; ++k) a.light_dirs[k] = c->light_dirs[k];
	a.views = c->views.p; a.counters = c->counters.p; a.ray_indices = c->ray_indices.p; a.rays = c->rays.p; a.numsteps = c->numsteps.p;
	a.coords = c->coords.p; a.mlp_out = c->mlp_out.p; a.ray_loss = c->ray_loss.p; a.ncomp = c->ncomp.p; a.cbase = c->cbase.p;
	a.coords_compacted = c->coords_compacted.p; a.dloss = c->dloss_dout.p; a.loss = c->loss.p; a.ek_loss = c->ek_loss; a.mask_loss = c->mask_loss;
	a.src_slot = c->cin_flow ? c->src_slot.p : nullptr;
	a.chain_rec = c->knobs.loss_chain_records ? c->chain_rec.p : nullptr;
	a.ray_grad = c->ray_grad.p; a.ray_of = c->ray_of.p; a.slot_of = c->slot_of.p;
	a.scan_words = nullptr; a.scan_ticket = 0; a.scan_error = nullptr; a.scan_total = nullptr;
	a.wg_partial = c->wg_partial.p; a.red_out = nullptr; a.red_host_out = nullptr; a.red_host_seq = 0;
	c->loss_reduced = false;
	// all three rows in one fill -- for the stage API only (rows beyond the kept rays read as zeros, as in the reference); the training step reads the kept rays' rows, which
	// pass 2 writes one by one (zeros for a ray without compacted samples): the fill was 7.6 us on the critical stream of every step that begins with an occupancy update
	if (!c->pre.loss_cleared && !defer_rollover) HIP_TRY(hipMemsetAsync(c->loss.p, 0, c->loss.bytes(), s));
	c->pre.loss_cleared = false;
	const uint32_t blocks = (n_rays + 3) / 4; // one wavefront per ray
	// 16 lanes per ray (four rays per wavefront) from the batch size at which the march switches kernels: many short rays
	const bool rows = n_rays >= c->knobs.march_narrow_from && !c->knobs.loss_wave_per_ray;
	const uint32_t blocks_heads = rows ? (n_rays + LOSS1_WG / 16 - 1) / (LOSS1_WG / 16) : (n_rays + LOSS1_WG / 64 - 1) / (LOSS1_WG / 64);
	auto launch_heads = [&]() {
		if (rows) hipLaunchKernelGGL(k_loss_pass1_heads<16>, dim3(blocks_heads), dim3(LOSS1_WG), 0, s, a);
		else hipLaunchKernelGGL(k_loss_pass1_heads<64>, dim3(blocks_heads), dim3(LOSS1_WG), 0, s, a);
	};
	a.cap = 0xffffffffu; a.phase = 0; a.unfinished = c->unfinished.p; a.idx2 = c->idx2.p; a.fwd_counts = c->fwd_counts.p;
	c->prof.mark(s, P_NONE);
	if (two_round_n_max) { // the caller has evaluated the head (fwd_k1 samples) of every ray; settle what that allows, evaluate the queued tails, redo those rays
		a.cap = c->cur_k1;
		launch_heads();
		c->prof.mark(s, P_LOSS_PASS1);
		int rc = launch_forward(c, s, c->coords.p, c->fwd_counts.p + 2, two_round_n_max, c->mlp_out.p, false, c->idx2.p, c->cin_flow ? c->cin_eval.p : nullptr);
		if (rc != RNB_OK) return rc;
		c->prof.mark(s, P_FORWARD);
		a.phase = 1;
	}
	if (a.phase) hipLaunchKernelGGL(k_loss_pass1, dim3(std::min(blocks, 1024u)), dim3(256), 0, s, a);
	else launch_heads();
	c->prof.mark(s, P_LOSS_PASS1);
	const bool flat = a.chain_rec && c->knobs.loss_flat;
	// the offsets formed inside k_loss_pass2_rays: step 1000 (13 k rays, 4 tiles) 0.6065 -> 0.6015 ms/step; step 6000 (95 k rays, 24 tiles: 5.9 k workgroups polling) 0.6274 -> 0.6270: up to 8 tiles
	if (flat && c->knobs.loss_scan_fused && (n_rays + SCAN_TILE - 1) / SCAN_TILE <= (c->knobs.loss_scan_fused_always ? 64u : 8u)) {
		a.scan_words = c->scan_words.p + 64 * 4; a.scan_ticket = ++c->scan_ticket; a.scan_error = c->host_coarse_dev + 6; a.scan_total = c->counters.p + 1;
	} else if (n_rays >= c->knobs.march_narrow_from && c->knobs.scan_chain && (n_rays + SCAN_TILE - 1) / SCAN_TILE <= 64) {
		hipLaunchKernelGGL(k_scan_compact_chain, dim3((n_rays + SCAN_TILE - 1) / SCAN_TILE), dim3(SCAN_WG), 0, s, n_rays, c->ncomp.p, c->cbase.p, c->counters.p,
		                   c->scan_words.p + 64 * 4, ++c->scan_ticket, c->host_coarse_dev + 6);
	} else if (n_rays >= c->knobs.march_narrow_from) {
		const uint32_t n_tiles = (n_rays + SCAN_TILE - 1) / SCAN_TILE;
		hipLaunchKernelGGL(k_scan_compact_sums, dim3(n_tiles), dim3(SCAN_WG), 0, s, n_rays, c->ncomp.p, c->scan_tiles.p + 256);
		hipLaunchKernelGGL(k_scan_compact_offsets, dim3(n_tiles), dim3(SCAN_WG), 0, s, n_rays, c->ncomp.p, c->scan_tiles.p + 256, c->cbase.p, c->counters.p);
	} else
		hipLaunchKernelGGL(k_scan_compact, dim3(1), dim3(1024), 0, s, n_rays, c->ncomp.p, c->cbase.p, c->counters.p);
	c->prof.mark(s, P_SCAN_COMPACT);
	if (flat) { // at every batch size (step 1000, 12 k long rays: 0.6247 -> 0.6189 ms/step; step 2000: 0.6216 -> 0.6100; step 6000: 0.6526 -> 0.6367)
		hipLaunchKernelGGL(k_loss_pass2_rays<16>, dim3((n_rays + 15) / 16), dim3(256), 0, s, a);
		hipEvent_t ev = nullptr;
		if (reduce_too) { // the training step: the loss sums and their readback in the same launch (launch_reduce_losses for the one-launch forms)
			const bool poll = c->poll_loss();
			if (poll) { if (++c->rb_seq == 0) ++c->rb_seq; c->loss_polled = false; }
			a.red_out = c->loss_sums.p; a.red_host_out = reinterpret_cast<double*>(c->host_rb_dev); a.red_host_seq = poll ? c->rb_seq : 0u;
			ev = poll ? nullptr : c->ev_loss;
			c->loss_reduced = true;
		}
		LAUNCH_EV(k_loss_pass2_samples, dim3(1 + (a.B + 255) / 256), dim3(256), 0, s, ev, a); // workgroup 0: the reduction; the samples also write their wrapped copies (no k_rollover)
		c->prof.mark(s, P_LOSS_PASS2);
		HIP_TRY(hipGetLastError());
		return RNB_OK;
	} else if (a.chain_rec) {
		if (rows) hipLaunchKernelGGL((k_loss_pass2<16, true>), dim3((n_rays + 15) / 16), dim3(256), 0, s, a);
		else hipLaunchKernelGGL((k_loss_pass2<64, true>), dim3(blocks), dim3(256), 0, s, a);
	} else {
		if (rows) hipLaunchKernelGGL((k_loss_pass2<16, false>), dim3((n_rays + 15) / 16), dim3(256), 0, s, a);
		else hipLaunchKernelGGL((k_loss_pass2<64, false>), dim3(blocks), dim3(256), 0, s, a);
	}
	if (!defer_rollover) hipLaunchKernelGGL(k_rollover, dim3(1024), dim3(256), 0, s, c->cfg.target_batch_size, c->counters.p, c->dloss_dout.p, c->coords_compacted.p, a.src_slot);
	c->prof.mark(s, P_LOSS_PASS2);
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

int forward_backward(rnb_ctx* c, hipStream_t s, bool join_dw = true) {
	const uint32_t B = c->cfg.target_batch_size;
	const rnb_ctx::ScatterGroups& sg = c->sg;
	const uint32_t L = c->cfg.n_levels;
	const uint32_t e_c = sg.e_c, l_fine = sg.l_fine;
	join_tail_host(c);
	const bool half = c->half_acc();
	if (!c->grads_clean) HIP_TRY(hipMemsetAsync(c->grad_ptr(0), 0, c->n_params * c->grad_elem(), s));
	c->grads_clean = false;
	c->sc.valid = false; c->sc.exchanged = false; c->sc.sharded = false; c->sc.all_final_recorded = false; c->sc.dw_joined = true; c->sc.c_early = false; c->sc.c_side = false;
	TrainArgs a;
	a.coords = c->coords_compacted.p; a.dout = c->dloss_dout.p; a.B = B; a.B_global = B * c->cfg.world_size; a.sdf_bias = c->cfg.sdf_bias; a.t = c->ts; a.skip_rgb = c->cfg.apply_no_albedo ? 1u : 0u;
	const bool split = c->rgb_split(); // albedo mode: k_rgb_fwd_bwd + k_fwd_bwd_sdf_full instead of the generic kernel and its weight-gradient GEMMs
	a.wimg = !c->wimg_valid ? nullptr : (!c->knobs.fwd_bwd_generic) ? c->wimg_fbs.p : c->wimg_train.p;
	const bool sdf_only = (a.skip_rgb && !c->knobs.fwd_bwd_generic) || split; // the training kernels leave one weight-gradient partial per workgroup themselves
	const uint32_t fb_grid = sdf_only ? std::min<uint32_t>((B / TILE + WAVES_PER_WG - 1) / WAVES_PER_WG, (uint32_t)c->n_cus * c->knobs.fbs_wg_per_cu) : c->fwd_grid;
	// half mode, round 6: the weight gradients in the reference's split-K order (k_dw_sliced): the training kernels export their GEMM operands feature-major instead of accumulating
	// in their own tiling (deviation D1' of rounds 4-5; RNB_DW_SLICED=0: that form).
	const bool sliced = half && sdf_only && c->knobs.dw_sliced; // (sdf_only: the SDF-only training kernel or the albedo mode's two -- not the generic kernel of RNB_FWD_BWD_GENERIC)
	const uint32_t n_slices = (B + DW_SLICE - 1) / DW_SLICE;
	// partial weight gradients: one slab per producing workgroup -- k_dw's workgroups (generic kernel) or k_fwd_bwd_sdf's own; sliced: one per 4096-sample slice (never more than workgroups)
	const size_t slab = sliced ? (size_t)n_slices : sdf_only ? (size_t)fb_grid : (size_t)c->dw_nwg;
	float* p_rgb2; float* p_rgb1; float* p_rgb0; float* p_sdf1; float* p_sdf0; float* p_sdf0b; float* p_sdf1b;
	{
		float* p = c->dw_partial.p;
		p_rgb2 = p;  p += slab * 16 * 64;
		p_rgb1 = p;  p += slab * 64 * 64;
		p_rgb0 = p;  p += slab * 64 * 32;
		p_sdf1 = p;  p += slab * 16 * 64;
		p_sdf0 = p;  p += slab * 64 * 32;
		p_sdf0b = p; p += slab * 64 * 32;
		p_sdf1b = p; p += slab * 16 * 64;
	}
	a.dw_w0 = p_sdf0; a.dw_w0b = p_sdf0b; a.dw_w1 = p_sdf1; a.dw_w1b = p_sdf1b;
	const bool side_streams = c->overlap();
	c->prof.mark(s, P_NONE);
	hipEvent_t ev_fb = side_streams ? c->ev_fb : nullptr; // the weight-gradient GEMMs start on the side stream when this kernel is done
	if (split) {
		int rc = ensure_rgb_buffers(c);
		if (rc != RNB_OK) return rc;
		const bool flow = c->cin_flow; // the step's network evaluation exported the input rows; a stage call evaluates the compacted batch for them
		c->cin_flow = false;
		if (!flow) {
			rc = launch_forward(c, s, c->coords_compacted.p, nullptr, B, c->rgb_out_scratch.p, false, nullptr, c->cin_eval.p);
			if (rc != RNB_OK) return rc;
		}
		RgbArgs r;
		r.cin = c->cin_eval.p; r.src_slot = flow ? c->src_slot.p : nullptr; r.dout = c->dloss_dout.p; r.dcin = c->dcin.p; r.B = B;
		r.wimg = c->wimg_valid ? c->wimg_rgb.p : nullptr;
		r.dw_c0 = p_rgb0; r.dw_c1 = p_rgb1; r.dw_c2 = p_rgb2;
		r.t = c->ts;
		if (half && sliced) hipLaunchKernelGGL(k_rgb_fwd_bwd_hs, dim3(fb_grid), dim3(WG), LDS_RGB, s, c->net(false), r);
		else if (half) hipLaunchKernelGGL(k_rgb_fwd_bwd_h, dim3(fb_grid), dim3(WG), LDS_RGB, s, c->net(false), r);
		else hipLaunchKernelGGL(k_rgb_fwd_bwd, dim3(fb_grid), dim3(WG), LDS_RGB, s, c->net(false), r);
		a.dcin = c->dcin.p;
		// (one workgroup per CU -- no register spills, the two-per-CU instance keeps ~80 values in scratch -- lost: 0.88 vs 0.80 ms/step, half the wavefronts to hide the gathers)
		if (half && sliced) LAUNCH_EV(k_fwd_bwd_sdf_full_hs, dim3(fb_grid), dim3(WG), LDS_FBS_FULL, s, ev_fb, c->meta(), c->net(false), a);
		else if (half) LAUNCH_EV(k_fwd_bwd_sdf_full_h, dim3(fb_grid), dim3(WG), LDS_FBS_FULL, s, ev_fb, c->meta(), c->net(false), a);
		else LAUNCH_EV(k_fwd_bwd_sdf_full, dim3(fb_grid), dim3(WG), LDS_FBS_FULL, s, ev_fb, c->meta(), c->net(false), a);
	} else if (sdf_only && half && sliced) LAUNCH_EV(k_fwd_bwd_sdf_hs, dim3(fb_grid), dim3(WG), LDS_FBS, s, ev_fb, c->meta(), c->net(false), a);
	else if (sdf_only && half) LAUNCH_EV(k_fwd_bwd_sdf_h, dim3(fb_grid), dim3(WG), LDS_FBS, s, ev_fb, c->meta(), c->net(false), a);
	else if (sdf_only) LAUNCH_EV(k_fwd_bwd_sdf, dim3(fb_grid), dim3(WG), LDS_FBS, s, ev_fb, c->meta(), c->net(false), a);
	else LAUNCH_EV(k_fwd_bwd, dim3(fb_grid), dim3(WG), LDS_TRAIN, s, ev_fb, c->meta(), c->net(false), a);
	c->prof.mark(s, P_FWD_BWD);
	c->prof.units[P_FWD_BWD] += B;
	const TrainScratch& T = c->ts;

	// ---- weight-gradient GEMMs (MFMA / streaming)
	auto launch_dw = [&](hipStream_t sd, hipEvent_t done) {
		const uint32_t nwg = c->dw_nwg, chunk = c->dw_chunk;
		DwAllArgs d;
		d.n = 0; d.nwg = nwg; d.B = B; d.chunk = chunk;
		auto add = [&](uint32_t kind, const half_t* yt, const half_t* xt, float* part) { d.kind[d.n] = kind; d.YT[d.n] = yt; d.XT[d.n] = xt; d.partial[d.n] = part; ++d.n; };
		if (!a.skip_rgb) {
			add(DW_4x4, T.dh2, T.h1, p_rgb1); // the largest first
			add(DW_4x2, T.dh1, T.cin, p_rgb0);
			add(DW_1x4, T.dr, T.h2, p_rgb2);
		}
		add(DW_4x2, T.dz, T.sdfin, p_sdf0);
		add(DW_4x2, T.dz1, T.ddin, p_sdf0b);
		add(DW_1x4, T.dso, T.z1, p_sdf1);
		add(DW_1x4_ONES, nullptr, T.front, p_sdf1b);
		for (uint32_t q = d.n; q < 7; ++q) { d.kind[q] = DW_1x4; d.YT[q] = nullptr; d.XT[q] = nullptr; d.partial[q] = nullptr; }
		// the generic kernel's feature-major operands: seven (four with --no-albedo) GEMMs in one launch (0.852 -> 0.831 ms/step); k_fwd_bwd_sdf has
		// accumulated its weight gradients itself and left one partial per workgroup
		if (!sdf_only) hipLaunchKernelGGL(k_dw_all, dim3(nwg * d.n), dim3(WG), 0, sd, d);
		if (sliced) {
			DwSlicedArgs q;
			q.n = 0; q.B = B; q.n_slices = n_slices;
			uint32_t wgs = 0;
			auto add_sliced = [&](const half_t* yt, const half_t* xt, float* out, uint32_t n_out, uint32_t n_live, uint32_t n_in, uint32_t ones) {
				q.YT[q.n] = yt; q.XT[q.n] = xt; q.out[q.n] = out; q.n_out[q.n] = n_out; q.n_out_live[q.n] = n_live; q.n_in[q.n] = n_in; q.ones[q.n] = ones;
				q.first_wg[q.n] = wgs; wgs += n_slices * ((n_out * n_in / 4 + 255) / 256); ++q.n;
			};
			if (!a.skip_rgb) { // the colour MLP (the largest first)
				add_sliced(T.dh2, T.h1, p_rgb1, 64, 64, 64, 0);
				add_sliced(T.dh1, T.cin, p_rgb0, 64, 64, 32, 0);
				add_sliced(T.dr, T.h2, p_rgb2, 16, 3, 64, 0);    // rows 3..15 of dL/d(colour output) are exact zeros (k_rgb_fwd_bwd)
			}
			add_sliced(T.dz, T.sdfin, p_sdf0, 64, 64, 32, 0);   // dW0 = dz in^T                    (fully_fused_mlp.cu:953-1030)
			add_sliced(T.dz1, T.ddin, p_sdf0b, 64, 64, 32, 0);  // dW0 += dz1 ddin^T                (:1097-1131, beta = 1)
			add_sliced(T.dso, T.z1, p_sdf1, 16, split ? 16 : 1, 64, 0); // dW1 = dso z1^T: without the colour MLP rows 1..15 of dso are exact zeros (TrainArgs::skip_rgb)
			add_sliced(nullptr, T.front, p_sdf1b, 16, 1, 64, 1); // dW1[0, :] += sum front
			for (uint32_t g = q.n; g < 7; ++g) { q.YT[g] = nullptr; q.XT[g] = nullptr; q.out[g] = nullptr; q.n_out[g] = 0; q.n_out_live[g] = 0; q.n_in[g] = 4; q.ones[g] = 0; }
			for (uint32_t g = q.n; g < 8; ++g) q.first_wg[g] = wgs;
			hipLaunchKernelGGL(k_dw_sliced, dim3(wgs), dim3(256), 0, sd, q);
		}
		DwFinishArgs f;
		f.partial[0] = p_rgb2; f.partial[1] = p_rgb1; f.partial[2] = p_rgb0; f.partial[3] = p_sdf1; f.partial[4] = p_sdf0; f.partial[5] = p_sdf0b; f.partial[6] = p_sdf1b;
		f.n_partials = (uint32_t)slab; f.sliced = sliced ? 1u : 0u;
		f.var_partial = c->var_partial.p; f.n_var_partials = fb_grid * WAVES_PER_WG;
		f.grads = c->grads.p; f.grads16 = half ? c->grads16.p : nullptr; f.off_sdf = (uint32_t)c->off_sdf; f.off_rgb = (uint32_t)c->off_rgb; f.off_var = (uint32_t)c->off_var; f.skip_rgb = a.skip_rgb;
		const uint32_t n_fin_blocks = (RNB_N_SDF_MLP_PARAMS + (a.skip_rgb ? 0 : RNB_N_RGB_MLP_PARAMS)) / DWF_PARAMS + 1; // + the variance workgroup
		LAUNCH_EV(k_dw_finish, dim3(n_fin_blocks), dim3(DWF_WG), 0, sd, done, f);
	};

	// ---- hash-grid gradient scatter, three groups of levels (kernels_net.cuh); the addends commute up to fp32 rounding, as with any atomic order:
	//   A  fine    [l_fine, L)   plain quad kernel: one cell per sample, bound by the L2 atomic line rate
	//   B  middle  [e_c, l_fine) run-length quad kernel: a cell spans several march steps (~590 / resolution); same bound + the latency of the walk
	//   C  coarse  [0, e_c)      LDS-privatised tables (beside the atomic groups on the optimizer's stream it stretches 42 -> 158 us and the
	//                            step loses 4 %, measured in round 2: it stays last on the caller's stream)
	// albedo mode, round 4 (k_rgb_fwd_bwd + k_fwd_bwd_sdf_full, march held back behind them): 0: 0.806, 1: 0.783, 2: 0.760, 3: 0.774, 4: 0.800 -- there the march is the
	// long pole beside the scatter and gets the wave slots a capped scatter leaves
	const uint32_t scatter_cap = c->knobs.scatter_wg_per_cu >= 0 ? (uint32_t)c->knobs.scatter_wg_per_cu : ((sdf_only && !split) ? 0u : 2u);
	ScatterArgs sa;
	sa.g12 = T.g12; sa.srec = T.srec; sa.B = B; sa.grid_grad = half ? nullptr : c->grads.p + c->off_grid;
	sa.grid_grad16 = half ? reinterpret_cast<uint32_t*>(c->grads16.p + c->off_grid) : nullptr; // (off_grid is even: half2 entries are 4-byte aligned)
	// cfg.deterministic: the scatter kernels add 64-bit fixed-point integers into grads_fixed, and k_fixed_narrow -- behind each group, on its stream, in front of the
	// group's event -- rounds the group's exact sums once into the gradient vector: everything downstream (optimizer chunks, data-parallel blocks, stage calls) finds the
	// vector it finds in the other modes, at the same points
	const bool fixed = c->fixed_acc();
	sa.grid_fixed = fixed ? c->grads_fixed.p : nullptr;
	sa.prio = side_streams ? (uint32_t)c->knobs.scatter_prio : 0u;
	auto narrow = [&](hipStream_t st, hipEvent_t done, uint32_t l0, uint32_t l1) { // levels [l0, l1)
		const uint64_t lo = (uint64_t)c->grid.offsets[l0] * 2, hi = (uint64_t)c->grid.offsets[l1] * 2;
		if (hi <= lo) { if (done) (void)hipEventRecord(done, st); return; }
		const uint32_t blocks = (uint32_t)std::min<uint64_t>(2048, ((hi - lo) / 2 + 255) / 256);
		LAUNCH_EV(k_fixed_narrow, dim3(blocks), dim3(256), 0, st, done, c->grads_fixed.p, half ? nullptr : c->grads.p + c->off_grid, half ? c->grads16.p + c->off_grid : nullptr, lo, hi);
	};
	// `done` (if any) fires when the group's last kernel has completed; a group without kernels records it the plain way
	const bool dbg_levels = c->prof.on && c->knobs.dbg_scatter_lo >= 0;
	// RNB_SCATTER_ANYORDER: every scatter launch behind the step's first one may start before the kernel in front of it has drained (other levels: no dependency); the first one keeps
	// its barrier (it needs k_fwd_bwd's operands), and so does everything behind the scatter
	bool sc_first = true;
	auto sc_flags = [&]() -> uint32_t { const bool any = c->knobs.scatter_anyorder && side_streams && !sc_first; sc_first = false; return any ? (uint32_t)hipExtAnyOrderLaunch : 0u; };
	auto launch_a = [&](hipStream_t st, hipEvent_t done, uint32_t l0, uint32_t l1) { // levels [l0, l1) of the group
		if (dbg_levels) { l0 = std::max(l0, (uint32_t)c->knobs.dbg_scatter_lo); l1 = std::max(l0, std::min(l1, (uint32_t)c->knobs.dbg_scatter_hi)); }
		const uint32_t n_vb = (B * 4 + 255) / 256, cap = scatter_cap ? std::max(1u, (uint32_t)c->n_cus * scatter_cap / std::max(1u, l1 - l0)) : n_vb;
		if (l1 > l0 && fixed) {
			LAUNCH_EVF(k_grid_scatter_quad_fixed, dim3(std::min(n_vb, cap), l1 - l0), dim3(256), 0, st, nullptr, sc_flags(), c->meta(), sa, l0, n_vb);
			narrow(st, done, l0, l1);
		} else if (l1 > l0 && half && c->knobs.scatter_plain) LAUNCH_EVF(k_grid_scatter_quad_h_per_addend, dim3(std::min(n_vb, cap), l1 - l0), dim3(256), 0, st, done, sc_flags(), c->meta(), sa, l0, n_vb);
		else if (l1 > l0 && half) LAUNCH_EVF(k_grid_scatter_quad_h, dim3(std::min(n_vb, cap), l1 - l0), dim3(256), 0, st, done, sc_flags(), c->meta(), sa, l0, n_vb);
		else if (l1 > l0) LAUNCH_EVF(k_grid_scatter_quad, dim3(std::min(n_vb, cap), l1 - l0), dim3(256), 0, st, done, sc_flags(), c->meta(), sa, l0, n_vb);
		else if (done) (void)hipEventRecord(done, st);
	};
	auto launch_b = [&](hipStream_t st, hipEvent_t done) { // one launch for all these levels, each with the workgroups its run length needs
		uint32_t e_c = sg.e_c, l_fine = sg.l_fine; // (shadow the group's bounds: the measurement aid narrows them)
		if (dbg_levels) { e_c = std::max(e_c, (uint32_t)c->knobs.dbg_scatter_lo); l_fine = std::min(l_fine, (uint32_t)c->knobs.dbg_scatter_hi); }
		if (l_fine <= e_c) { if (done) (void)hipEventRecord(done, st); return; }
		ScatterRlPlan plan;
		plan.n = l_fine - e_c; plan.k_log2 = sg.k_log2 >> (4 * (e_c - sg.e_c));
		uint32_t wg = 0;
		for (uint32_t q = 0; q < plan.n; ++q) { plan.wg_start[q] = wg; wg += (((B + sg.Ks[e_c + q] - 1) / sg.Ks[e_c + q]) * 4 + 255) / 256; }
		plan.wg_start[plan.n] = wg;
		const uint32_t cap_rl = scatter_cap ? (uint32_t)c->n_cus * scatter_cap : wg;
		const bool staged = c->knobs.scatter_rl_staged >= 0 ? c->knobs.scatter_rl_staged != 0 : (c->cur_n_rays != 0 && c->cur_n_rays < c->knobs.march_narrow_from);
		if (fixed) {
			LAUNCH_EVF(k_grid_scatter_quad_rl_fixed, dim3(std::min(wg, cap_rl)), dim3(256), 0, st, nullptr, sc_flags(), c->meta(), sa, e_c, plan);
			narrow(st, done, e_c, l_fine);
		} else if (!staged) {
			if (half) LAUNCH_EVF(k_grid_scatter_quad_rl_direct_h, dim3(std::min(wg, cap_rl)), dim3(256), 0, st, done, sc_flags(), c->meta(), sa, e_c, plan);
			else if (c->knobs.scatter_share) LAUNCH_EVF(k_grid_scatter_quad_rl_direct_share, dim3(std::min(wg, cap_rl)), dim3(256), 0, st, done, sc_flags(), c->meta(), sa, e_c, plan);
			else LAUNCH_EVF(k_grid_scatter_quad_rl_direct, dim3(std::min(wg, cap_rl)), dim3(256), 0, st, done, sc_flags(), c->meta(), sa, e_c, plan);
		} else if (half) LAUNCH_EVF(k_grid_scatter_quad_rl_h, dim3(std::min(wg, cap_rl)), dim3(256), LDS_SCATTER_RL, st, done, sc_flags(), c->meta(), sa, e_c, plan);
		else if (c->knobs.scatter_share) LAUNCH_EVF(k_grid_scatter_quad_rl_share, dim3(std::min(wg, cap_rl)), dim3(256), LDS_SCATTER_RL, st, done, sc_flags(), c->meta(), sa, e_c, plan);
		else LAUNCH_EVF(k_grid_scatter_quad_rl, dim3(std::min(wg, cap_rl)), dim3(256), LDS_SCATTER_RL, st, done, sc_flags(), c->meta(), sa, e_c, plan);
	};
	auto launch_c = [&](hipStream_t st, hipEvent_t done) {
		if (!e_c) { if (done) (void)hipEventRecord(done, st); return; }
		ScatterLdsArgs la; la.a = sa; la.n_levels = e_c;
		const uint32_t wg_cap = 128; // workgroups of the LDS scatter (measured optimum: half the CUs, each zeroing / flushing its private table once)
		const uint32_t n_wg = std::max(1u, std::min<uint32_t>(wg_cap, (B + 1023) / 1024));
		la.samples_per_wg = ((B + n_wg - 1) / n_wg + 3) / 4 * 4;
		if (fixed) {
			LAUNCH_EVF(k_grid_scatter_lds_fixed, dim3(n_wg, 2), dim3(512), (size_t)c->grid.offsets[e_c] * 8, st, nullptr, sc_flags(), c->meta(), la);
			narrow(st, done, 0, e_c);
		} else if (half) LAUNCH_EVF(k_grid_scatter_lds_h, dim3(n_wg), dim3(512), (size_t)c->grid.offsets[e_c] * 8, st, done, sc_flags(), c->meta(), la);
		else LAUNCH_EVF(k_grid_scatter_lds, dim3(n_wg), dim3(512), (size_t)c->grid.offsets[e_c] * 8, st, done, sc_flags(), c->meta(), la);
	};

	if (!side_streams) {
		launch_dw(s, nullptr);
		c->prof.mark(s, P_DW);
		launch_c(s, nullptr); c->prof.mark(s, P_SCATTER_LDS);
		launch_b(s, nullptr); c->prof.mark(s, P_SCATTER_RL);
		launch_a(s, nullptr, l_fine, L); c->prof.mark(s, P_SCATTER_QUAD);
		c->prof.units[P_SCATTER_LDS] += B; c->prof.units[P_SCATTER_RL] += B; c->prof.units[P_SCATTER_QUAD] += B;
	} else {
		// The caller's stream carries the scatter (B, A in two halves, then C), the side stream the GEMMs. After B and each half of A an
		// event lets the optimizer step that group's levels while the rest is still being scattered (optimizer_step); what is left
		// after C is the MLPs' and the coarse levels' parameters.
		hipStream_t sd = c->s_dw;
		HIP_TRY(hipStreamWaitEvent(sd, c->ev_fb, 0));
		c->sc.dp = c->dp_order();
		const bool dw_late = c->knobs.dw_late && !c->sc.dp && !split; // RNB_DW_LATE=1 (A/B): k_dw_finish behind the first scatter group instead of beside it
		if (!dw_late) launch_dw(sd, c->ev_dw);
		const uint32_t a_mid = l_fine + (L - l_fine + 1) / 2;
		// Data parallel: C, B, then A, so that the parameters in front of A's levels (MLPs, C, B: one contiguous block) are final at
		// ev_sc[0] + ev_dw and their exchange runs beside the scatter of A; A's levels + variance are the second block.
		if (c->sc.dp) launch_c(s, nullptr);
		// Order of the groups by batch shape (round 4, ms/step over 160 steps, one box, B-A1-A2-C / A1-A2-B-C / A-B-C): step 1000, 14 k rays: 0.6446 / 0.6473 / 0.6325;
		// 1200, 18.5 k: 0.6435 / 0.6594 / 0.6394; 1400, 24 k: 0.6304 / 0.6324 / 0.6388; 1800, 41 k: 0.6132 / 0.6142 / 0.6161; 6000, 94 k: 0.6452 / 0.6482 / 0.6470
		// (profiles/r04_sweep_scatter_order.txt): the fine levels first and in one launch while the batch is few long rays (the regime of the 16-lanes-per-ray march).
		c->sc.order = c->sc.dp ? 0 : c->knobs.scatter_order >= 0 ? c->knobs.scatter_order : ((c->cur_n_rays < c->knobs.march_narrow_from && !split) ? 2 : 0); // (albedo mode, step 1000: 0.763 with A-B-C vs 0.755 with B-A1-A2-C: its march is held behind the training kernels)
		c->sc.c_early = !c->sc.dp && c->knobs.scatter_c_early && e_c != 0;
		if (c->sc.c_early) { // C beside the first atomic group: LDS + VALU work next to wavefronts that wait for the memory side
			HIP_TRY(hipStreamWaitEvent(c->s_adam, c->ev_fb, 0));
			launch_c(c->s_adam, c->ev_sc[2]);
		}
		if (c->sc.order == 0) { // B, A1, A2 (, C)
			launch_b(s, c->ev_sc[0]);
			if (dw_late) { HIP_TRY(hipStreamWaitEvent(sd, c->ev_sc[0], 0)); launch_dw(sd, c->ev_dw); }
			launch_a(s, c->ev_sc[1], l_fine, a_mid);
			launch_a(s, c->ev_sc[3], a_mid, L);
		} else if (c->sc.order == 1) { // A1, A2, B, C: the fine levels' atomics before any optimizer chunk shares the memory side with them
			launch_a(s, c->ev_sc[1], l_fine, a_mid);
			if (dw_late) { HIP_TRY(hipStreamWaitEvent(sd, c->ev_sc[1], 0)); launch_dw(sd, c->ev_dw); }
			launch_a(s, c->ev_sc[3], a_mid, L);
			launch_b(s, c->ev_sc[0]);
		} else { // A (one launch), B, C
			launch_a(s, c->ev_sc[1], l_fine, L);
			if (dw_late) { HIP_TRY(hipStreamWaitEvent(sd, c->ev_sc[1], 0)); launch_dw(sd, c->ev_dw); }
			launch_b(s, c->ev_sc[0]);
		}
		c->sc.split_mid = c->off_grid + (uint64_t)c->grid.offsets[a_mid] * 2;
		c->sc.c_side = false;
		if (c->sc.c_early) HIP_TRY(hipStreamWaitEvent(s, c->ev_sc[2], 0)); // (every gradient is final when `s` is: rnb_gradient_part_wait, stage calls)
		else if (!c->sc.dp && !join_dw && c->knobs.scatter_c_side && e_c != 0 && !dbg_levels) {
			// the training step (the optimizer follows): group C is handed to the optimizer's launch (rnb_ctx::sc.c_side); flush_c_side launches it here for whoever asks for the gradients first
			const ScatterArgs sa_v = sa; const uint32_t e_c_v = e_c, B_v = B; const bool fixed_v = fixed, half_v = half;
			c->sc.c_launch = [c, sa_v, e_c_v, B_v, fixed_v, half_v](hipStream_t st, hipEvent_t done) {
				ScatterLdsArgs la; la.a = sa_v; la.n_levels = e_c_v;
				const uint32_t n_wg = std::max(1u, std::min<uint32_t>(128u, (B_v + 1023) / 1024));
				la.samples_per_wg = ((B_v + n_wg - 1) / n_wg + 3) / 4 * 4;
				if (fixed_v) {
					LAUNCH_EV(k_grid_scatter_lds_fixed, dim3(n_wg, 2), dim3(512), (size_t)c->grid.offsets[e_c_v] * 8, st, nullptr, c->meta(), la);
					const uint64_t lo = 0, hi = (uint64_t)c->grid.offsets[e_c_v] * 2;
					const uint32_t blocks = (uint32_t)std::min<uint64_t>(2048, ((hi - lo) / 2 + 255) / 256);
					LAUNCH_EV(k_fixed_narrow, dim3(blocks), dim3(256), 0, st, done, c->grads_fixed.p, half_v ? nullptr : c->grads.p + c->off_grid, half_v ? c->grads16.p + c->off_grid : nullptr, lo, hi);
				} else if (half_v) LAUNCH_EV(k_grid_scatter_lds_h, dim3(n_wg), dim3(512), (size_t)c->grid.offsets[e_c_v] * 8, st, done, c->meta(), la);
				else LAUNCH_EV(k_grid_scatter_lds, dim3(n_wg), dim3(512), (size_t)c->grid.offsets[e_c_v] * 8, st, done, c->meta(), la);
			};
			c->sc.c_side = true;
			c->sc.c_after = c->sc.order == 0 ? c->ev_sc[3] : c->ev_sc[0]; // the last atomic group of the order
		} else if (!c->sc.dp) launch_c(s, nullptr); // last: its levels hold 32 k parameters, so almost nothing of the optimizer is left after the scatter (-6 % step time vs. first)
		if (c->sc.dp || join_dw) HIP_TRY(hipStreamWaitEvent(s, c->ev_dw, 0));
		c->sc.dw_joined = c->sc.dp || join_dw; // the training step leaves the join to the optimizer, which continues on the side stream (optimizer_step)
		c->sc.valid = true; // parameter ranges of the groups (grid entries are 2 parameters each)
		c->sc.split[0] = c->off_grid + (uint64_t)c->grid.offsets[l_fine] * 2; // A = [split0, off_var)
		c->sc.split[1] = c->off_grid + (uint64_t)c->grid.offsets[e_c] * 2;    // B = [split1, split0), C = [off_grid, split1)
	}
	c->prof.units[P_DW] += B;
	c->prof.units[P_SCATTER] += B; // (its time: the three kernels' marks above, summed by Profiler::collect -- the profiler runs the serial branch)
	c->backward_stream = s; // rnb_gradient_part_wait records "every gradient is final" there if a caller asks
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

// Once per step: learning-rate schedule, step count, EMA debias terms (exponential_decay.h:61-72, ema.h:116-117).
constexpr uint32_t ADAM_LR_TABLE_N = 1u << 16;
static void fill_lr_table(rnb_ctx* c, hipStream_t s) { // stream-ordered in front of the optimizer kernels that read it
	if (c->lr_table_beta1 == c->cfg.beta1 && c->lr_table_beta2 == c->cfg.beta2) return;
	(void)hipDeviceSynchronize(); // a chunk of the previous step may still be reading the old table (rnb_update_config changed a beta)
	hipLaunchKernelGGL(k_adam_lr_table, dim3(ADAM_LR_TABLE_N / 256), dim3(256), 0, s, c->adam_lr_table.p, ADAM_LR_TABLE_N, c->cfg.beta1, c->cfg.beta2);
	(void)hipStreamSynchronize(s); // the optimizer's chunks run on several streams
	c->lr_table_beta1 = c->cfg.beta1; c->lr_table_beta2 = c->cfg.beta2;
}

static int optimizer_begin(rnb_ctx* c) {
	if (c->opt.begun) return RNB_OK;
	const rnb_config& cfg = c->cfg;
	{ const int rc = ensure_opt_records(c); if (rc != RNB_OK) return rc; }
	c->opt_plain_current = false; // the records move on; the staging views are refreshed when a caller asks for one (rnb_buffer)
	fill_lr_table(c, nullptr);
	const uint32_t step0 = c->optimizer_step_count;
	if (step0 == 0) c->lr_factor = 1.0f;
	if (step0 >= cfg.lr_decay_start && (step0 - cfg.lr_decay_start) % cfg.lr_decay_interval == 0 && step0 <= 10000000u) c->lr_factor *= cfg.lr_decay_base;
	const uint32_t current_step = ++c->optimizer_step_count;
	AdamArgs& a = c->opt.args;
	a.n = c->n_params; a.n_matrix = RNB_N_SDF_MLP_PARAMS + RNB_N_RGB_MLP_PARAMS;
	a.rec = c->opt_rec.p; a.w16 = c->params_fp16.p; a.ema = c->params_ema.p;
	a.grads = c->grads.p; a.grads16 = c->half_acc() ? c->grads16.p : nullptr;
	a.base_lr = cfg.learning_rate * c->lr_factor; a.beta1 = cfg.beta1; a.beta2 = cfg.beta2; a.epsilon = cfg.epsilon; a.l2_reg = cfg.l2_reg;
	a.ema_decay = cfg.ema_decay;
	a.skip_lo = cfg.only_sdf_training ? (uint64_t)c->off_rgb : 0; a.skip_hi = cfg.only_sdf_training ? (uint64_t)c->off_grid : 0;
	a.ema_debias_old = 1 - (float)std::pow(cfg.ema_decay, current_step - 1);
	a.ema_debias_new = 1.0f / (1 - (float)std::pow(cfg.ema_decay, current_step));
	a.lr_table = c->adam_lr_table.p; a.lr_table_n = (uint32_t)c->adam_lr_table.n;
	c->opt.begun = true;
	c->opt.early_done = false;
	return RNB_OK;
}

static void adam_launch(rnb_ctx* c, hipStream_t st, uint64_t lo, uint64_t hi, hipEvent_t done = nullptr) {
	if (hi <= lo) { if (done) (void)hipEventRecord(done, st); return; }
	AdamArgs a = c->opt.args;
	a.begin = lo; a.end = hi;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>(4096, ((hi - lo) / 4 + 255) / 256);
	LAUNCH_EV(k_adam_ema, dim3(blocks), dim3(256), 0, st, done, a);
}
// Optimizer on the early gradient block only (rnb_gradient_parts block 0), on the caller's stream.
int optimizer_step_early(rnb_ctx* c, hipStream_t st) {
	if (!c->sc.valid || c->opt.early_done) return RNB_OK;
	{ const int rc = optimizer_begin(c); if (rc != RNB_OK) return rc; }
	if (c->sc.dp) adam_launch(c, st, 0, c->sc.split[0], c->ev_adam);
	else adam_launch(c, st, c->sc.split[1], c->sc.split[0], c->ev_adam);
	c->opt.early_done = true;
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

// Which kernel scatters which level (see forward_backward).
static void plan_scatter_groups(rnb_ctx* c) {
	rnb_ctx::ScatterGroups& g = c->sg;
	const uint32_t L = c->cfg.n_levels;
	uint32_t l;
	{
		for (l = 0; l < L; ++l) {
			const float run = 590.f / (float)c->grid.resolution[l]; // compacted samples of a ray that share a cell of this level
			g.Ks[l] = run >= 5.f ? 16 : run >= 2.5f ? 8 : run >= 1.2f ? 4 : 1; // below ~1 sample per cell the plain quad kernel is faster (measured)
			// A/B (round 6, face sharing): RNB_SCATTER_KMIN = the shortest walk of a run-length level; RNB_SCATTER_RL_UPTO = levels below it walk (at least 4 samples) even where a cell holds < 1.2 samples
			if (c->knobs.scatter_rl_upto > 0 && l < (uint32_t)c->knobs.scatter_rl_upto && g.Ks[l] == 1) g.Ks[l] = 4;
			if (c->knobs.scatter_kmin > 0 && g.Ks[l] > 1) g.Ks[l] = std::max<uint32_t>(g.Ks[l], (uint32_t)c->knobs.scatter_kmin);
		}
	}
	if (c->knobs.scatter_plain) { // RNB_SCATTER_PLAIN=1: every level through the plain kernel, one atomic per corner and sample -- the reference's own scatter structure (grid.h:366-495)
		for (l = 0; l < L; ++l) g.Ks[l] = 1;
		c->dp_split = c->off_grid; c->dp_mid = c->off_grid + (uint64_t)c->grid.offsets[(L + 1) / 2] * 2;
		return;
	}
	for (l = 0; l < L; ++l) if (l == g.e_c && (size_t)c->grid.offsets[l + 1] * 8 <= 150 * 1024) g.e_c = l + 1; // the coarsest levels whose fp32 gradient tables fit in LDS together
	g.l_fine = g.e_c;
	// (round 5: every atomic level through ONE run-length launch, K = 1 on the fine levels -- 225.8 vs 114.8 + 124.2 us at step 1000, 164.4 vs 94.2 + 74.8 at 2000: the launch boundary and
	// nothing else; the coarse levels' latency-bound walks already overlap inside group B, the groups are bound by their atomic requests. profiles/r05_scatter_per_level.txt)
	for (l = g.e_c; l < L && g.Ks[l] > 1 && l - g.e_c < 16; ++l) {
		g.k_log2 |= (uint64_t)ilog2(g.Ks[l]) << (4 * (l - g.e_c));
		g.l_fine = l + 1;
	}
	c->dp_split = c->off_grid + (uint64_t)c->grid.offsets[g.l_fine] * 2;
	c->dp_mid = c->off_grid + (uint64_t)c->grid.offsets[g.l_fine + (L - g.l_fine + 1) / 2] * 2; // (a_mid of forward_backward)
}

// End of a step's parameter update: the LDS weight images of the next step's kernels, bookkeeping.
static int optimizer_finish(rnb_ctx* c, hipStream_t s, bool images_done = false) {
	c->opt.begun = false;
	c->opt.early_done = false;
	c->sc.valid = false;
	if (!images_done) hipLaunchKernelGGL(k_prepare_weight_images, dim3(WIMG_WGS, 4), dim3(WG), 0, s, c->net(false), c->wimg_fwd.p, c->wimg_fbs.p, c->wimg_train.p, c->wimg_rgb.p, c->half_acc() ? 1 : 0);
	c->wimg_valid = true;
	c->prof.mark(s, P_ADAM);
	c->prof.units[P_ADAM] += (double)c->n_params;
	HIP_TRY(hipGetLastError());
	c->grads_clean = true;
	return RNB_OK;
}

// Blocks of the sharded data-parallel optimizer, in the order their gradients become final: each block is world_size equal
// chunks (multiples of 4 parameters), chunk r belongs to rank r; the last block ends at param_capacity.
static void shard_layout(const rnb_ctx* c, rnb_shard_part parts[RNB_MAX_SHARD_PARTS], uint32_t* n_parts) {
	const uint64_t W = std::max(1u, c->cfg.world_size), r = c->cfg.rank, q = 4 * W;
	uint32_t n = 0;
	// static: the ownership of a parameter must not move between steps. Boundaries rounded DOWN: the parameters between a rounded boundary and the group's first
	// one belong to the later block and were final earlier. Blocks: MLPs + groups C, B | first half of the fine levels (A1) | second half (A2) + variance (round 4:
	// the exchange of A1 runs beside the scatter of A2; rounds 2-3 had A as one block of 16.8 MB behind the scatter)
	const uint64_t m0 = c->dp_split / q * q, m1 = std::max(m0, c->dp_mid / q * q);
	if (m0) { parts[n].lo = 0; parts[n].hi = m0; ++n; }
	if (m1 > m0 && m1 < c->param_capacity) { parts[n].lo = m0; parts[n].hi = m1; ++n; }
	parts[n].lo = (m1 > m0 && m1 < c->param_capacity) ? m1 : m0; parts[n].hi = c->param_capacity; ++n;
	for (uint32_t k = 0; k < n; ++k) {
		const uint64_t chunk = (parts[k].hi - parts[k].lo) / W;
		parts[k].own_lo = parts[k].lo + r * chunk;
		parts[k].own_hi = parts[k].own_lo + chunk;
	}
	*n_parts = n;
}

int optimizer_step(rnb_ctx* c, hipStream_t s) {
	{ const int rc = optimizer_begin(c); if (rc != RNB_OK) return rc; }
	c->prof.mark(s, P_NONE);
	const bool chunked = !c->opt.early_done && c->overlap() && !c->sc.dp && c->sc.valid && !c->sc.exchanged;
	if (c->sc.c_side && !(chunked && !c->sc.dw_joined)) flush_c_side(c, s); // only the path below that continues on the weight-gradient stream takes group C with it
	if (!c->sc.dw_joined && !chunked) {
		HIP_TRY(hipStreamWaitEvent(s, c->ev_dw, 0)); // the paths below step the MLPs on `s`
		c->sc.dw_joined = true;
	}
	bool images_done = false;
	if (c->opt.early_done) {
		// the caller has already stepped the early block (after exchanging it): the rest, then join
		if (!c->sc.dp) adam_launch(c, s, 0, c->sc.split[1]);
		adam_launch(c, s, c->sc.split[0], c->n_params);
		HIP_TRY(hipStreamWaitEvent(s, c->ev_adam, 0));
	} else if (chunked) {
		// The update is independent per parameter, so each scatter group's levels are stepped as soon as that group is done,
		// on the side stream, beside the scatter of the next group; only the last half of group A is left for the end.
		hipStream_t sa = c->s_adam;
		if (c->sc.c_early) adam_launch(c, sa, c->off_grid, c->sc.split[1]); // group C's levels right behind their scatter on this stream
		if (c->sc.order == 0) {
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[0], 0));
			adam_launch(c, sa, c->sc.split[1], c->sc.split[0]);  // group B's levels, beside the scatter of group A
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[1], 0));
			adam_launch(c, sa, c->sc.split[0], c->sc.split_mid);        // group A's levels, first half beside the second half's scatter
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[3], 0));
			adam_launch(c, sa, c->sc.split_mid, c->off_var, c->ev_adam);
		} else if (c->sc.order == 1) {
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[1], 0));
			adam_launch(c, sa, c->sc.split[0], c->sc.split_mid);
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[3], 0));
			adam_launch(c, sa, c->sc.split_mid, c->off_var);
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[0], 0));
			adam_launch(c, sa, c->sc.split[1], c->sc.split[0], c->ev_adam);
		} else {
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[1], 0));
			adam_launch(c, sa, c->sc.split[0], c->off_var);
			HIP_TRY(hipStreamWaitEvent(sa, c->ev_sc[0], 0));
			adam_launch(c, sa, c->sc.split[1], c->sc.split[0], c->ev_adam);
		}
		if (c->sc.dw_joined) {
			adam_launch(c, s, 0, c->sc.c_early ? c->off_grid : c->sc.split[1]); // MLPs + group C's levels (contiguous), variance; s has joined the side stream
			adam_launch(c, s, c->off_var, c->n_params);
		} else {
			// behind the dW GEMMs on their stream: the MLPs' and the variance's parameters, then the LDS weight images of the next
			// step's kernels, all long before the scatter ends
			hipStream_t sd = c->s_dw;
			adam_launch(c, sd, 0, c->off_grid);
			adam_launch(c, sd, c->off_var, c->n_params);
			const bool c_side = c->sc.c_side;
			c->sc.c_side = false;
			LAUNCH_EV(k_prepare_weight_images, dim3(WIMG_WGS, 4), dim3(WG), 0, sd, c_side ? nullptr : c->ev_tail, c->net(false), c->wimg_fwd.p, c->wimg_fbs.p, c->wimg_train.p, c->wimg_rgb.p, c->half_acc() ? 1 : 0);
			images_done = true;
			if (c_side) { // group C + its optimizer chunk end this stream (ev_tail); the critical stream is idle behind its last atomic group
				HIP_TRY(hipStreamWaitEvent(sd, c->sc.c_after, 0));
				c->sc.c_launch(sd, nullptr);
				adam_launch(c, sd, c->off_grid, c->sc.split[1], c->ev_tail);
				// the joins, in the order the events are expected to fire, so that only the last packet's latency is left when the last of them has: the march (queued already?), then by batch shape
				// the optimizer's stream and this one
				if (c->pre.valid) { HIP_TRY(hipStreamWaitEvent(s, c->ev_march, 0)); c->pre.march_joined = true; }
				if (c->sc.order == 2) { HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0)); HIP_TRY(hipStreamWaitEvent(s, c->ev_adam, 0)); }
				else { HIP_TRY(hipStreamWaitEvent(s, c->ev_adam, 0)); HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0)); }
				c->sc.dw_joined = true;
				return optimizer_finish(c, s, images_done);
			}
			if (!c->sc.c_early) adam_launch(c, s, c->off_grid, c->sc.split[1]);
			// the join with the side stream: on the next step's march stream if that march is queued after this call (launch_premarch), else here
			if (c->knobs.join_fold && c->pre.valid) { // the march of the next step is queued already (ev_march is recorded): the weight-gradient stream, idle behind its weight images, takes the joins (rnb_ctx::join_pending)
				HIP_TRY(hipStreamWaitEvent(sd, c->ev_adam, 0));
				HIP_TRY(hipStreamWaitEvent(sd, c->ev_march, 0));
				HIP_TRY(hipEventRecord(c->ev_join, sd));
				c->join_pending = true;
				c->sc.dw_joined = true;
				return optimizer_finish(c, s, images_done);
			}
			if (c->knobs.defer_tail && !c->pre.valid && !prep_due(c->cur_step + 1)) c->tail_pending = true;
			else if (!c->knobs.unsafe_skip_joins) HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0));
			c->sc.dw_joined = true;
		}
		if (!c->knobs.unsafe_skip_joins) HIP_TRY(hipStreamWaitEvent(s, c->ev_adam, 0));
	} else {
		adam_launch(c, s, 0, c->n_params);
	}
	return optimizer_finish(c, s, images_done);
}

// Sharded optimizer (data parallel): block `part` of shard_layout has been reduce-scattered by the caller; step this rank's
// chunk and clear the rest of the block's accumulators (their sums live on the other ranks).
int optimizer_step_shard(rnb_ctx* c, uint32_t part, hipStream_t st) {
	rnb_shard_part parts[RNB_MAX_SHARD_PARTS];
	uint32_t n = 0;
	shard_layout(c, parts, &n);
	if (part >= n) return fail(RNB_ERR_INVALID, "rnb_train_step_apply_shard: no such block");
	const rnb_shard_part& p = parts[part];
	{ const int rc = optimizer_begin(c); if (rc != RNB_OK) return rc; }
	if (!c->sc.dw_joined) HIP_TRY(hipStreamWaitEvent(st, c->ev_dw, 0)); // any block may hold MLP parameters
	if (p.own_lo > p.lo) HIP_TRY(hipMemsetAsync(c->grad_ptr(p.lo), 0, (p.own_lo - p.lo) * c->grad_elem(), st));
	if (p.hi > p.own_hi) HIP_TRY(hipMemsetAsync(c->grad_ptr(p.own_hi), 0, (p.hi - p.own_hi) * c->grad_elem(), st));
	adam_launch(c, st, std::min<uint64_t>(p.own_lo, c->n_params), std::min<uint64_t>(p.own_hi, c->n_params));
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static int update_config_common(rnb_config& dst, const rnb_config* cfg) {
	if (cfg->abi_version != RNB_ABI_VERSION) return fail(RNB_ERR_INVALID, "abi_version mismatch");
	if (cfg->n_levels != dst.n_levels || cfg->log2_hashmap_size != dst.log2_hashmap_size || cfg->base_resolution != dst.base_resolution ||
	    cfg->per_level_scale != dst.per_level_scale || cfg->target_batch_size != dst.target_batch_size || cfg->max_rays_per_batch != dst.max_rays_per_batch ||
	    cfg->aabb_scale != dst.aabb_scale || cfg->seed != dst.seed || cfg->world_size != dst.world_size || cfg->rank != dst.rank || cfg->accumulate != dst.accumulate || cfg->deterministic != dst.deterministic)
		return fail(RNB_ERR_INVALID, "rnb_update_config: geometry fields differ from the context's");
	dst = *cfg;
	return RNB_OK;
}


} // namespace

// =====================================================================================================
extern "C" {
// No exception crosses the C boundary: a std::bad_alloc of a host-side container, or anything else thrown below an entry point, becomes a
// status code and a message for rnb_last_error() -- across ctypes / a C caller it would be std::terminate and SIGABRT.
#define RNB_GUARD                                                                                                   \
	catch (const std::bad_alloc&) { return fail(RNB_ERR_NOMEM, "out of host memory (std::bad_alloc)"); }                \
	catch (const std::exception& e_) { return fail(RNB_ERR_INVALID, std::string("exception: ") + e_.what()); }          \
	catch (...) { return fail(RNB_ERR_INVALID, "unknown exception"); }

const char* rnb_last_error(void) { return g_err.c_str(); }
uint32_t rnb_abi_version(void) { return RNB_ABI_VERSION; }

int rnb_default_config(rnb_config* cfg) try {
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
	cfg->overlap = 1;
	return RNB_OK;
} RNB_GUARD

int rnb_destroy(rnb_ctx* c) try {
	if (!c) return RNB_OK;
	if (c->march_stats.p) { // RNB_MARCH_STATS=1
		unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		(void)hipDeviceSynchronize();
		if (hipMemcpy(h, c->march_stats.p, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[0])
			fprintf(stderr, "k_march_count_skip: %llu wavefronts, %.2f loop iterations each; %llu rays: %.3f skipped at least once, %.3f ended early, %.5f start-overs per ray, %.3f rounds per ray looking for a re-entry cell\n",
			        h[0], (double)h[1] / h[0], h[2], (double)h[3] / h[2], (double)h[6] / h[2], (double)h[4] / h[2], (double)h[5] / h[2]);
		c->march_stats.free();
	}
	c->opt_rec.free(); c->params_fp32.free(); c->grads.free(); c->grads16.free(); c->grads_fixed.free(); c->adam_m.free(); c->adam_v.free(); c->params_fp16.free(); c->params_ema.free(); c->adam_steps.free(); c->adam_lr_table.free();
	c->density_grid.free(); c->density_grid_tmp.free(); c->density_grid_tmp_alt.free(); c->density_mean.free(); c->mean_partial.free(); c->loss_sums.free(); c->bitfield.free(); c->coarse_bits.free(); c->coarse_count.free();
	c->grid_sample_pos.free(); c->grid_sample_idx.free(); c->views.free(); c->pixels.free();
	c->gs_sorted_pos.free(); c->gs_sorted_idx.free(); c->gs_stage_pos.free(); c->gs_stage_idx.free(); c->gs_hist.free(); c->gs_range.free(); c->gs_eval_pos.free(); c->gs_eval_idx.free();
	c->ray_indices.free(); c->numsteps.free(); c->counters.free(); c->rays.free(); c->coords.free(); c->coords_compacted.free();
	c->loss.free(); c->mlp_out.free(); c->chain_rec.free(); c->ray_grad.free(); c->ray_of.free(); c->slot_of.free(); c->wg_partial.free(); c->dloss_dout.free();
	c->wimg_fwd.free(); c->wimg_fbs.free(); c->wimg_train.free(); c->wimg_rgb.free(); c->cin_eval.free(); c->dcin.free(); c->rgb_out_scratch.free(); c->src_slot.free(); c->ray_const.free(); c->ray_base1.free(); c->scan_words.free(); c->idx1.free(); c->idx2.free(); c->fwd_counts.free(); c->unfinished.free();
	c->ray_setup.free(); c->ray_t.free(); c->ray_dunnorm.free(); c->ray_steps.free(); c->ray_base.free(); c->ray_slot.free(); c->ncomp.free(); c->cbase.free(); c->scan_tiles.free(); c->loss_partial.free(); c->ray_loss.free();
	c->mc_table.free(); c->fm.free(); c->g12.free(); c->srec.free(); c->var_partial.free(); c->dw_partial.free();
	c->prof.destroy();
	if (c->s_march) { (void)hipStreamSynchronize(c->s_march); (void)hipStreamDestroy(c->s_march); }
	for (hipStream_t st : {c->s_dw, c->s_adam}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
	for (hipEvent_t e : {c->ev_join, c->ev_loss, c->ev_march, c->ev_fb, c->ev_dw, c->ev_adam, c->ev_tail, c->ev_march_rest, c->ev_all, c->ev_grid, c->ev_gs, c->ev_sc[0], c->ev_sc[1], c->ev_sc[2], c->ev_sc[3]}) if (e) (void)hipEventDestroy(e);
	if (c->host_rb) (void)hipHostFree(c->host_rb);
	if (c->host_coarse) (void)hipHostFree(c->host_coarse);
	delete c;
	return RNB_OK;
} RNB_GUARD

int rnb_create(const rnb_config* cfg, rnb_ctx** out) try {
	if (!cfg || !out) return fail(RNB_ERR_INVALID, "null argument");
	if (cfg->abi_version != RNB_ABI_VERSION) return fail(RNB_ERR_INVALID, "abi_version mismatch");
	if (cfg->n_levels == 0 || cfg->n_levels > 14) return fail(RNB_ERR_INVALID, "n_levels must be in [1,14]");
	if (cfg->base_resolution < 2) return fail(RNB_ERR_INVALID, "base_resolution must be >= 2");
	if (cfg->aabb_scale == 0 || (cfg->aabb_scale & (cfg->aabb_scale - 1)) != 0 || cfg->aabb_scale > 128) return fail(RNB_ERR_INVALID, "aabb_scale must be a power of two <= 128");
	if (cfg->target_batch_size == 0 || cfg->target_batch_size % 128 != 0) return fail(RNB_ERR_INVALID, "target_batch_size must be a positive multiple of 128");
	if (cfg->max_rays_per_batch == 0 || cfg->max_rays_per_batch > (1u << 18)) return fail(RNB_ERR_INVALID, "max_rays_per_batch must be in [1, 2^18]");
	if (cfg->world_size == 0 || cfg->rank >= cfg->world_size) return fail(RNB_ERR_INVALID, "bad rank/world_size");
	if (cfg->accumulate > RNB_ACCUM_HALF) return fail(RNB_ERR_INVALID, "accumulate must be RNB_ACCUM_FP32 or RNB_ACCUM_HALF");
	if (cfg->deterministic > 1) return fail(RNB_ERR_INVALID, "deterministic must be 0 or 1");
	if (cfg->accumulate == RNB_ACCUM_HALF && getenv("RNB_FWD_BWD_GENERIC")) return fail(RNB_ERR_INVALID, "RNB_FWD_BWD_GENERIC (the generic training kernel of rounds 1-3) has no half-accumulate form");
	int dev = 0;
	HIP_TRY(hipGetDevice(&dev));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, dev));
	if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) return fail(RNB_ERR_DEVICE, std::string("librnb_neus2_hip targets gfx950 (MI355X); found ") + prop.gcnArchName);
	rnb_ctx* c = new rnb_ctx();
	// from here on a failing HIP call releases the context (device buffers, streams, events) before returning
#define HIP_TRY_C(expr)                                                                                                  \
	do {                                                                                                                 \
		hipError_t e_ = (expr);                                                                                          \
		if (e_ != hipSuccess) { rnb_destroy(c); return fail(RNB_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
	} while (0)
	c->cfg = *cfg;
	if (const char* e = getenv("RNB_DETERMINISTIC")) c->cfg.deterministic = atoi(e) != 0 ? 1u : 0u; // measurement aid (tools/ab_interleaved.sh): the mode without touching the caller
	c->n_cus = prop.multiProcessorCount;
	build_grid_tables(c);
	c->off_sdf = 0;
	c->off_rgb = RNB_N_SDF_MLP_PARAMS;
	c->off_grid = c->off_rgb + RNB_N_RGB_MLP_PARAMS;
	c->off_var = c->off_grid + c->n_grid_params;
	c->n_params = c->off_var + RNB_N_VARIANCE_PARAMS;
	c->aabb.mn = 0.5f - 0.5f * std::min(1u << (N_CASCADES - 1), cfg->aabb_scale); // testbed_nerf.cu:3198-3214
	c->aabb.mx = 0.5f + 0.5f * std::min(1u << (N_CASCADES - 1), cfg->aabb_scale);
	c->aabb.max_cascade = 0;
	while ((1u << c->aabb.max_cascade) < cfg->aabb_scale) ++c->aabb.max_cascade;
	c->aabb.cone_angle = cfg->aabb_scale <= 1 ? 0.0f : (1.0f / 256.0f);
	const uint32_t B = cfg->target_batch_size, maxr = cfg->max_rays_per_batch;
	const uint32_t n_grid = GRID_CELLS * (c->aabb.max_cascade + 1);
#define ALLOC(buf, count)                                                                                              \
	do {                                                                                                               \
		if ((buf).alloc(count) != hipSuccess) { rnb_destroy(c); return fail(RNB_ERR_NOMEM, "hipMalloc failed for " #buf); } \
	} while (0)
	{ // parameter-shaped arrays, padded to a whole number of 4-parameter groups per data-parallel rank
		const uint64_t q = 4ull * std::max(1u, cfg->world_size);
		c->param_capacity = (c->n_params + q - 1) / q * q;
	}
#define ALLOC_P(buf)                                                                                                   \
	do {                                                                                                               \
		if ((buf).alloc_padded(c->n_params, c->param_capacity) != hipSuccess) { rnb_destroy(c); return fail(RNB_ERR_NOMEM, "hipMalloc failed for " #buf); } \
	} while (0)
	if (getenv("RNB_MARCH_STATS")) { if (c->march_stats.alloc_padded(8, 8) != hipSuccess) { rnb_destroy(c); return fail(RNB_ERR_NOMEM, "hipMalloc failed"); } }
	ALLOC(c->adam_lr_table, ADAM_LR_TABLE_N);
	ALLOC(c->opt_rec, c->param_capacity * 4);
	if (cfg->accumulate == RNB_ACCUM_HALF) ALLOC_P(c->grads16); else ALLOC_P(c->grads);
	if (c->cfg.deterministic) {
		if (c->grads_fixed.alloc_padded(c->n_grid_params, c->n_grid_params) != hipSuccess) { rnb_destroy(c); return fail(RNB_ERR_NOMEM, "hipMalloc failed for the fixed-point gradient accumulators"); }
	}
	ALLOC_P(c->params_fp32); ALLOC_P(c->adam_m); ALLOC_P(c->adam_v); ALLOC_P(c->params_fp16); ALLOC_P(c->params_ema); ALLOC_P(c->adam_steps);
#undef ALLOC_P
	ALLOC(c->density_grid, n_grid); ALLOC(c->density_grid_tmp, n_grid); ALLOC(c->density_grid_tmp_alt, n_grid); ALLOC(c->density_mean, 1); ALLOC(c->mean_partial, 1024); ALLOC(c->loss_sums, 16);
	ALLOC(c->bitfield, (size_t)GRID_CELLS / 8 * N_CASCADES); ALLOC(c->coarse_bits, COARSE_BUF_WORDS); ALLOC(c->coarse_count, 1);
	ALLOC(c->grid_sample_pos, (size_t)n_grid * 3); ALLOC(c->grid_sample_idx, n_grid);
	ALLOC(c->ray_indices, maxr); ALLOC(c->numsteps, (size_t)maxr * 2); ALLOC(c->counters, 4); ALLOC(c->rays, (size_t)maxr * 6);
	ALLOC(c->coords, (size_t)B * 16 * 7); ALLOC(c->mlp_out, (size_t)B * 16 * 16); ALLOC(c->chain_rec, (size_t)B * 16 * CHAIN_REC_FLOATS); ALLOC(c->ray_grad, (size_t)maxr * 16); ALLOC(c->ray_of, B); ALLOC(c->slot_of, B); ALLOC(c->wg_partial, ((size_t)maxr + 15) / 16 * 3); ALLOC(c->dloss_dout, (size_t)B * 16); ALLOC(c->coords_compacted, (size_t)B * 7);
	ALLOC(c->loss, (size_t)maxr * 3); c->ek_loss = c->loss.p + maxr; c->mask_loss = c->loss.p + (size_t)maxr * 2;
	ALLOC(c->ray_setup, (size_t)maxr * 8); ALLOC(c->ray_t, (size_t)maxr * RNB_MAX_STEPS); ALLOC(c->ray_dunnorm, (size_t)maxr * 3); ALLOC(c->ray_steps, maxr); ALLOC(c->ray_base, maxr); ALLOC(c->ray_slot, maxr);
	ALLOC(c->scan_tiles, 256 + 64); ALLOC(c->loss_partial, 64 * 3);
	ALLOC(c->ncomp, maxr); ALLOC(c->cbase, maxr); ALLOC(c->ray_loss, maxr);
	ALLOC(c->wimg_fwd, W_FWD_END); ALLOC(c->wimg_fbs, SWF_END); ALLOC(c->wimg_train, W_TRAIN_END); ALLOC(c->wimg_rgb, RW_END);
	ALLOC(c->ray_const, (size_t)maxr * RAY_CONST_FLOATS); ALLOC(c->ray_base1, maxr); ALLOC(c->scan_words, 3 * 64 * 4); ALLOC(c->unfinished, maxr); ALLOC(c->fwd_counts, 4); ALLOC(c->idx1, (size_t)B * 16); ALLOC(c->idx2, (size_t)B * 16);
	// feature-major operand arrays: h2 h1 z1 dz1 dh2 dh1 dz front (64 rows), cin sdfin ddin (32 rows), dr dso (16 rows)
	ALLOC(c->fm, (size_t)B * (8 * 64 + 3 * 32 + 2 * 16));
	ALLOC(c->g12, (size_t)B * 14 * 2); ALLOC(c->srec, (size_t)B * 8);
	{
		half_t* q = c->fm.p;
		TrainScratch& T = c->ts;
		T.h2 = q; q += (size_t)B * 64; T.h1 = q; q += (size_t)B * 64; T.z1 = q; q += (size_t)B * 64; T.dz1 = q; q += (size_t)B * 64;
		T.dh2 = q; q += (size_t)B * 64; T.dh1 = q; q += (size_t)B * 64; T.dz = q; q += (size_t)B * 64; T.front = q; q += (size_t)B * 64;
		T.cin = q; q += (size_t)B * 32; T.sdfin = q; q += (size_t)B * 32; T.ddin = q; q += (size_t)B * 32;
		T.dr = q; q += (size_t)B * 16; T.dso = q; q += (size_t)B * 16;
		T.g12 = c->g12.p; T.srec = c->srec.p;
	}
	c->fwd_grid = std::min<uint32_t>((B / TILE + WAVES_PER_WG - 1) / WAVES_PER_WG, (uint32_t)c->n_cus);
	ALLOC(c->var_partial, (size_t)c->n_cus * 3 * WAVES_PER_WG);
	c->ts.var_partial = c->var_partial.p;
	{ // split-K geometry of the weight-gradient GEMMs: chunk = B / nwg, a multiple of 128 samples
		const uint32_t units = B / 128;
		uint32_t nwg = 1;
		for (uint32_t d = 1; d <= std::min<uint32_t>(units, 256u); ++d) if (units % d == 0) nwg = d;
		c->dw_nwg = nwg;
		c->dw_chunk = B / nwg;
		const size_t per_wave = (size_t)16 * 64 * 3 + 64 * 64 + (size_t)64 * 32 * 3;
		ALLOC(c->dw_partial, (size_t)std::max<uint32_t>(nwg * WAVES_PER_WG, (uint32_t)c->n_cus * 2) * per_wave); // one slab per producing workgroup: k_dw's, or k_fwd_bwd_sdf's own (<= 2 per CU)
	}
#undef ALLOC
	HIP_TRY_C(hipMemset(c->params_fp32.p, 0, c->params_fp32.bytes()));
	HIP_TRY_C(hipMemset(c->params_fp16.p, 0, c->params_fp16.bytes()));
	HIP_TRY_C(hipMemset(c->params_ema.p, 0, c->params_ema.bytes()));
	HIP_TRY_C(hipMemset(c->density_grid.p, 0, c->density_grid.bytes()));
	HIP_TRY_C(hipMemset(c->density_grid_tmp.p, 0, c->density_grid_tmp.bytes()));
	HIP_TRY_C(hipMemset(c->bitfield.p, 0, c->bitfield.bytes()));
	HIP_TRY_C(hipMemset(c->counters.p, 0, c->counters.bytes()));
	HIP_TRY_C(hipMemset(c->density_mean.p, 0, 4));
	HIP_TRY_C(hipMemset(c->coords.p, 0, c->coords.bytes()));
	HIP_TRY_C(hipMemset(c->coords_compacted.p, 0, c->coords_compacted.bytes()));
	HIP_TRY_C(hipMemset(c->dloss_dout.p, 0, c->dloss_dout.bytes()));
	HIP_TRY_C(hipMemset(c->mlp_out.p, 0, c->mlp_out.bytes()));
	int rc = reset_optimizer_state(c);
	if (rc != RNB_OK) { rnb_destroy(c); return rc; }
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_forward_chained), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FWD2));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_forward_chained_emul), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FWD2));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_forward_chained_emul_pipe<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FWD2));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_forward_chained_pipe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FWD2));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_forward_chained_pipe<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FWD2));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_forward_chained_pipe<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FWD2));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_forward_chained_pipe<4, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FWD2));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TRAIN));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_bwd_sdf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FBS));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_bwd_sdf_full), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FBS_FULL));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rgb_fwd_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_RGB));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_bwd_sdf_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FBS));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_bwd_sdf_hs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FBS));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_bwd_sdf_full_hs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FBS_FULL));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rgb_fwd_bwd_hs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_RGB));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd_bwd_sdf_full_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_FBS_FULL));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rgb_fwd_bwd_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_RGB));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_scatter_lds_h), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
	{
		const int march_lds_max = (int)((2 * COARSE_WORDS + 2 * COARSE_MAX_BLOCKS) * sizeof(uint32_t));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count<true>), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count_wide<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count_wide<64, true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count_skip<256>), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max + (int)(COARSE_WORDS * sizeof(uint32_t))));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count_skip<512>), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max + (int)(COARSE_WORDS * sizeof(uint32_t))));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count_skip<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max + (int)(COARSE_WORDS * sizeof(uint32_t))));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count_skip_narrow), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max + (int)(COARSE_WORDS * sizeof(uint32_t))));
		HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_march_count_bbox), hipFuncAttributeMaxDynamicSharedMemorySize, march_lds_max));
	}
HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_scatter_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
	HIP_TRY_C(hipFuncSetAttribute(reinterpret_cast<const void*>(k_grid_scatter_lds_fixed), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
	// Testbed::reset_network (testbed.cu:2223-2237)
	c->rng = Pcg32{cfg->seed};
	c->density_grid_rng = Pcg32{c->rng.next_uint()};
	(void)c->rng.next_uint(); // tv_loss_rng
	c->rays_per_batch = std::min(cfg->initial_rays_per_batch, cfg->max_rays_per_batch);
	c->training_step = 0;
	c->valid_level = compute_valid_level(c->cfg, 0);
	build_light_dirs(c);
	if (const char* e = getenv("RNB_FWD_K1")) { c->fwd_k1 = (uint32_t)atoi(e); c->fwd_k1_fixed = true; } // a fixed head length of the two-round network evaluation; 0 = one round over all samples
	{
		rnb_ctx::Knobs& k = c->knobs;
		k.march_narrow = getenv("RNB_MARCH_NARROW") != nullptr; k.fwd_bwd_generic = getenv("RNB_FWD_BWD_GENERIC") != nullptr;
		k.loss_wave_per_ray = getenv("RNB_LOSS_WAVE_PER_RAY") != nullptr; // the loss passes with one wavefront per ray whatever the batch (A/B, tests)
		k.dp_order = getenv("RNB_DP_FORCE_COLLECTIVES") != nullptr;
		k.march_late = getenv("RNB_MARCH_LATE") != nullptr;
		k.march_running_sums = getenv("RNB_MARCH_RUNNING_SUMS") != nullptr;
		if (const char* e = getenv("RNB_MARCH_NARROW_FROM")) k.march_narrow_from = (uint32_t)atoi(e);
		if (const char* e = getenv("RNB_SCATTER_WG_PER_CU")) k.scatter_wg_per_cu = std::max(0, atoi(e));
		if (const char* e = getenv("RNB_FBS_WG_PER_CU")) k.fbs_wg_per_cu = (uint32_t)std::max(1, std::min(2, atoi(e)));
		if (const char* e = getenv("RNB_GRID_PRESORT")) k.grid_presort = atoi(e) != 0;
		if (const char* e = getenv("RNB_FUSED_UPDATE")) k.fused_update = atoi(e) != 0;
		if (const char* e = getenv("RNB_POLL_LOSS")) k.poll_loss = atoi(e) != 0;
		if (const char* e = getenv("RNB_DEFER_TAIL")) k.defer_tail = atoi(e) != 0;
		if (const char* e = getenv("RNB_JOIN_FOLD")) k.join_fold = atoi(e) != 0;
		if (const char* e = getenv("RNB_SCATTER_C_SIDE")) k.scatter_c_side = atoi(e) != 0;
		if (const char* e = getenv("RNB_UNSAFE_SKIP_JOINS")) k.unsafe_skip_joins = atoi(e) != 0;
		if (const char* e = getenv("RNB_MARCH_WAVE_PER_RAY_BELOW")) k.march_wave_per_ray_below = (uint32_t)atoi(e);
		if (const char* e = getenv("RNB_SCATTER_ORDER")) k.scatter_order = std::max(-1, std::min(2, atoi(e)));
		if (const char* e = getenv("RNB_SCAN_CHAIN")) k.scan_chain = atoi(e) != 0;
		if (const char* e = getenv("RNB_LOSS_SCAN_FUSED")) { k.loss_scan_fused = atoi(e) != 0; k.loss_scan_fused_always = atoi(e) == 2; }
		if (const char* e = getenv("RNB_LOSS_FLAT")) k.loss_flat = atoi(e) != 0;
		if (const char* e = getenv("RNB_LOSS_CHAIN_RECORDS")) k.loss_chain_records = atoi(e) != 0;
		if (const char* e = getenv("RNB_RAY_CONST_DENSE")) k.ray_const_dense = atoi(e);
		if (const char* e = getenv("RNB_MARCH_WRITE_SPLIT")) k.march_write_split = atoi(e) != 0 ? 1 : 0;
		if (const char* e = getenv("RNB_SCATTER_PLAIN")) k.scatter_plain = atoi(e) != 0;
		if (const char* e = getenv("RNB_MARCH_NARROW_WGS")) { const int w = atoi(e); k.march_narrow_wgs = (w == 64 || w == 256 || w == 512) ? (uint32_t)w : 128u; }
		if (const char* e = getenv("RNB_MARCH_WGS")) { const int w = atoi(e); k.march_wgs = (w == 256 || w == 512) ? w : 1024; }
		if (const char* e = getenv("RNB_MARCH_WRITE_WG")) { const int w = atoi(e); k.march_write_wg = (w == 1024 || w == 128 || w == 64) ? w : 256; }
		if (const char* e = getenv("RNB_CHAIN_PLAIN")) k.chain_plain = atoi(e) != 0;
		if (const char* e = getenv("RNB_POINT_XCD")) k.point_xcd = atoi(e) != 0;
		if (const char* e = getenv("RNB_DW_SLICED")) k.dw_sliced = atoi(e) != 0;
		if (const char* e = getenv("RNB_ENCODE_PAIR")) k.encode_pair = atoi(e) != 0;
		if (const char* e = getenv("RNB_ENCODE_DEPTH")) { const int d = atoi(e); k.encode_depth = (d == 0 || d == 2 || d == 4 || d == 7) ? d : 4; }
		if (const char* e = getenv("RNB_DEBUG_SCATTER_LEVELS")) { int lo = -1, hi = -1; if (sscanf(e, "%d,%d", &lo, &hi) == 2 && lo >= 0 && hi > lo) { k.dbg_scatter_lo = lo; k.dbg_scatter_hi = hi; } }
		if (const char* e = getenv("RNB_SCATTER_C_EARLY")) k.scatter_c_early = atoi(e) != 0;
		if (const char* e = getenv("RNB_SCATTER_RL_STAGED")) k.scatter_rl_staged = atoi(e) != 0 ? 1 : 0;
		if (const char* e = getenv("RNB_SCATTER_ANYORDER")) k.scatter_anyorder = atoi(e) != 0;
		if (const char* e = getenv("RNB_SCATTER_SHARE")) k.scatter_share = atoi(e) != 0;
		if (const char* e = getenv("RNB_MARCH_SKIP")) k.march_skip = std::max(0, std::min(2, atoi(e)));
		if (const char* e = getenv("RNB_MARCH_SKIP_NARROW")) k.march_skip_narrow = atoi(e) != 0;
		if (const char* e = getenv("RNB_SCATTER_PRIO")) k.scatter_prio = std::max(0, std::min(3, atoi(e)));
		if (const char* e = getenv("RNB_MARCH_PRIO")) k.march_prio = std::max(0, std::min(3, atoi(e)));
		if (const char* e = getenv("RNB_MARCH_BBOX")) k.march_bbox = std::max(0, std::min(2, atoi(e)));
		if (const char* e = getenv("RNB_DW_LATE")) k.dw_late = atoi(e) != 0;
		if (const char* e = getenv("RNB_SCATTER_KMIN")) k.scatter_kmin = std::max(0, std::min(16, atoi(e)));
		if (const char* e = getenv("RNB_SCATTER_RL_UPTO")) k.scatter_rl_upto = std::max(0, std::min(14, atoi(e)));
	}
	plan_scatter_groups(c);
	{ // RNB_STREAM_PRIO="march,dw,adam" (A/B): HIP stream priorities of the three side streams (0 = normal, -1 = high, 1 = low)
		int pr[3] = {0, 0, 0};
		if (const char* e = getenv("RNB_STREAM_PRIO")) sscanf(e, "%d,%d,%d", &pr[0], &pr[1], &pr[2]);
		hipStream_t* st[3] = {&c->s_march, &c->s_dw, &c->s_adam};
		for (int i = 0; i < 3; ++i) {
			if (pr[i] == 0) HIP_TRY_C(hipStreamCreateWithFlags(st[i], hipStreamNonBlocking));
			else HIP_TRY_C(hipStreamCreateWithPriority(st[i], hipStreamNonBlocking, pr[i]));
		}
	}
	// ev_loss publishes the step's counters to the HOST (system-scope release). The others only order kernels on this device:
	// without the system-scope fence the queue is spared a cache writeback + invalidate at each of them.
	HIP_TRY_C(hipEventCreateWithFlags(&c->ev_loss, hipEventDisableTiming));
	const unsigned dev_flags = hipEventDisableTiming | (unsigned)hipEventDisableSystemFence;
	for (hipEvent_t* e : {&c->ev_join, &c->ev_march, &c->ev_fb, &c->ev_dw, &c->ev_adam, &c->ev_tail, &c->ev_march_rest, &c->ev_all, &c->ev_grid, &c->ev_gs, &c->ev_sc[0], &c->ev_sc[1], &c->ev_sc[2], &c->ev_sc[3]}) HIP_TRY_C(hipEventCreateWithFlags(e, dev_flags));
	HIP_TRY_C(hipHostMalloc(reinterpret_cast<void**>(&c->host_rb), sizeof(*c->host_rb), hipHostMallocMapped));
	std::memset(c->host_rb, 0, sizeof(*c->host_rb)); // (a recycled pinned block may hold a previous context's sequence word: wait_loss_readback would take it for this context's first step)
	HIP_TRY_C(hipHostGetDevicePointer(&c->host_rb_dev, c->host_rb, 0));
	HIP_TRY_C(hipHostMalloc(reinterpret_cast<void**>(&c->host_coarse), 64, hipHostMallocMapped));
	std::memset(c->host_coarse, 0, 64); // [0] block count of the LDS occupancy, [5] / [6] k_scan_rays_chain / k_scan_compact_chain gave up a wait
	*c->host_coarse = 0xffffffffu;
	HIP_TRY_C(hipMemset(c->scan_words.p, 0, c->scan_words.bytes()));
	HIP_TRY_C(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->host_coarse_dev), c->host_coarse, 0));
	*out = c;
	return RNB_OK;
#undef HIP_TRY_C
} RNB_GUARD

int rnb_update_config(rnb_ctx* c, const rnb_config* cfg) try {
	if (!c || !cfg) return fail(RNB_ERR_INVALID, "null argument");
	discard_premarch(c);
	int rc = update_config_common(c->cfg, cfg);
	if (rc != RNB_OK) return rc;
	build_light_dirs(c);
	return RNB_OK;
} RNB_GUARD

uint64_t rnb_n_params(const rnb_ctx* c) { return c ? c->n_params : 0; }

int rnb_param_layout(const rnb_ctx* c, uint64_t offsets[5]) try {
	if (!c || !offsets) return fail(RNB_ERR_INVALID, "null argument");
	offsets[0] = c->off_sdf; offsets[1] = c->off_rgb; offsets[2] = c->off_grid; offsets[3] = c->off_var; offsets[4] = c->n_params;
	return RNB_OK;
} RNB_GUARD

int rnb_grid_tables(const rnb_ctx* c, uint32_t* offsets, uint32_t* resolution, float* scale) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	for (uint32_t i = 0; i <= c->cfg.n_levels; ++i) if (offsets) offsets[i] = c->grid.offsets[i];
	for (uint32_t i = 0; i < c->cfg.n_levels; ++i) { if (resolution) resolution[i] = c->grid.resolution[i]; if (scale) scale[i] = c->grid.scale[i]; }
	return RNB_OK;
} RNB_GUARD

int rnb_init_params(rnb_ctx* c, const float* sdf_w) try {
	if (!c || !sdf_w) return fail(RNB_ERR_INVALID, "null argument");
	discard_premarch(c);
	HIP_TRY(hipDeviceSynchronize()); // a pipelined step may still be running its optimizer (side streams)
	std::seed_seq seq{c->cfg.seed}; // Trainer ctor, trainer.h:54-61
	std::vector<uint32_t> seeds(2);
	seq.generate(std::begin(seeds), std::end(seeds));
	Pcg32 rnd{seeds.front()};
	std::vector<float> mlp(RNB_N_SDF_MLP_PARAMS + RNB_N_RGB_MLP_PARAMS, 0.f);
	auto xavier = [&](float* dst, int rows, int cols) { // gpu_matrix.h:292-306
		float scale = 1.f;
		scale *= std::sqrt(6.0f / (float)(cols + rows));
		for (int i = 0; i < rows * cols; ++i) dst[i] = (float)(rnd.next_float() * 2.0f * scale - scale);
	};
	xavier(mlp.data(), 64, 32);            // density network: draws consumed, then overwritten (nerf_network.h:629-643)
	xavier(mlp.data() + 64 * 32, 16, 64);
	std::memcpy(mlp.data(), sdf_w, sizeof(float) * RNB_N_SDF_MLP_PARAMS);
	float* rgb = mlp.data() + RNB_N_SDF_MLP_PARAMS;
	xavier(rgb, 64, 48);
	xavier(rgb + 64 * 48, 64, 64);
	xavier(rgb + 64 * 48 + 64 * 64, 16, 64);
	HIP_TRY(hipMemset(c->params_fp32.p, 0, c->params_fp32.bytes()));
	HIP_TRY(hipMemcpy(c->params_fp32.p + c->off_sdf, mlp.data(), mlp.size() * sizeof(float), hipMemcpyHostToDevice));
	{ // hash grid U(-1e-4, 1e-4) on the device (grid.h:1379-1384; random.h:67-93)
		const uint64_t n = c->n_grid_params;
		const uint64_t n_thr = (n + 3) / 4;
		const uint64_t n_threads_total = ((n_thr + 127) / 128) * 128;
		hipLaunchKernelGGL(k_random_uniform, dim3((uint32_t)(n_threads_total / 128)), dim3(128), 0, 0, rnd, n, n_threads_total, -1e-4f, 1e-4f, c->params_fp32.p + c->off_grid);
		HIP_TRY(hipGetLastError());
		rnd.advance((int64_t)n);
	}
	float var[RNB_N_VARIANCE_PARAMS];
	for (int q = 0; q < RNB_N_VARIANCE_PARAMS; ++q) { // nerf_network.h:691-692: pcg32{1337}, U(0.3, 0.3)
		Pcg32 vr{1337};
		vr.advance(q);
		const float val = vr.next_float();
		var[q] = val * (0.300f - 0.300f) + 0.300f;
	}
	HIP_TRY(hipMemcpy(c->params_fp32.p + c->off_var, var, sizeof(var), hipMemcpyHostToDevice));
	c->trainer_rng = rnd;
	c->wimg_valid = false; // the training weights change outside the optimizer
	int rc = derive_half_params(c, 0);
	if (rc != RNB_OK) return rc;
	HIP_TRY(hipMemset(c->params_ema.p, 0, c->params_ema.bytes()));
	rc = reset_optimizer_state(c);
	if (rc != RNB_OK) return rc;
	HIP_TRY(hipDeviceSynchronize());
	return RNB_OK;
} RNB_GUARD

int rnb_set_params(rnb_ctx* c, const float* params) try {
	if (!c || !params) return fail(RNB_ERR_INVALID, "null argument");
	HIP_TRY(hipDeviceSynchronize()); // a pipelined step may still be running its optimizer
	HIP_TRY(hipMemcpy(c->params_fp32.p, params, c->params_fp32.bytes(), hipMemcpyHostToDevice));
	c->wimg_valid = false; // the training weights change outside the optimizer
	int rc = derive_half_params(c, 0);
	if (rc != RNB_OK) return rc;
	HIP_TRY(hipMemcpy(c->params_ema.p, c->params_fp16.p, c->params_fp16.bytes(), hipMemcpyDeviceToDevice));
	rc = reset_optimizer_state(c);
	if (rc != RNB_OK) return rc;
	HIP_TRY(hipDeviceSynchronize());
	return RNB_OK;
} RNB_GUARD

int rnb_buffer(rnb_ctx* c, int id, void** ptr, uint64_t* n_bytes) try {
	if (!c || !ptr || !n_bytes) return fail(RNB_ERR_INVALID, "null argument");
	flush_c_side(c, c->backward_stream); // (a gradient vector asked for between a training step's backward pass and its optimizer)
#define BUF(b) do { *ptr = (void*)(b).p; *n_bytes = (b).bytes(); return RNB_OK; } while (0)
	const bool read_only = (id & RNB_BUF_READONLY) != 0;
	id &= ~RNB_BUF_READONLY;
	if (id == RNB_BUF_PARAMS_FP32 || id == RNB_BUF_ADAM_M || id == RNB_BUF_ADAM_V || id == RNB_BUF_ADAM_STEPS) {
		// staging views of the optimizer records (rnb_ctx::opt_rec): brought up to date here, read back in front of the next optimizer launch
		const int rc = ensure_opt_views(c);
		if (rc != RNB_OK) return rc;
		if (!read_only) c->opt_rec_current = false;
	}
	if (c->join_pending || id == RNB_BUF_PARAMS_FP16 || id == RNB_BUF_PARAMS_EMA) join_tail_host(c); // both are written by the side stream's optimizer launch (ev_tail)
	if (!read_only) { // a possible write before the caller's next call: drop the cached forms now (a kept pointer written later: rnb_params_changed / rnb_bitfield_changed)
		if (id == RNB_BUF_PARAMS_FP16) c->wimg_valid = false;
		else if (id == RNB_BUF_DENSITY_BITFIELD) { discard_premarch(c); c->coarse_valid = false; c->gs_pre.valid = false; c->bitfield_foreign = true; }
		else if (id == RNB_BUF_DENSITY_GRID) c->gs_pre.valid = false;
	}
	switch (id) {
		case RNB_BUF_PARAMS_FP32: BUF(c->params_fp32);
		case RNB_BUF_PARAMS_FP16: BUF(c->params_fp16);
		case RNB_BUF_PARAMS_EMA: BUF(c->params_ema);
		case RNB_BUF_GRADS_FP32: if (c->half_acc()) return fail(RNB_ERR_INVALID, "accumulate = RNB_ACCUM_HALF: the gradient vector is RNB_BUF_GRADS_FP16"); BUF(c->grads);
		case RNB_BUF_GRADS_FP16: if (!c->half_acc()) return fail(RNB_ERR_INVALID, "accumulate = RNB_ACCUM_FP32: the gradient accumulators are RNB_BUF_GRADS_FP32"); BUF(c->grads16);
		case RNB_BUF_ADAM_M: BUF(c->adam_m);
		case RNB_BUF_ADAM_V: BUF(c->adam_v);
		case RNB_BUF_ADAM_STEPS: BUF(c->adam_steps);
		case RNB_BUF_DENSITY_GRID: BUF(c->density_grid);
		case RNB_BUF_DENSITY_BITFIELD: BUF(c->bitfield);
		case RNB_BUF_DENSITY_MEAN: BUF(c->density_mean);
		case RNB_BUF_RAY_INDICES: BUF(c->ray_indices);
		case RNB_BUF_RAYS: BUF(c->rays);
		case RNB_BUF_NUMSTEPS: BUF(c->numsteps);
		case RNB_BUF_COORDS: BUF(c->coords);
		case RNB_BUF_MLP_OUT: BUF(c->mlp_out);
		case RNB_BUF_DLOSS_DOUT: BUF(c->dloss_dout);
		case RNB_BUF_COORDS_COMPACTED: BUF(c->coords_compacted);
		case RNB_BUF_LOSS: *ptr = (void*)c->loss.p; *n_bytes = c->loss.bytes() / 3; return RNB_OK;
		case RNB_BUF_EK_LOSS: *ptr = (void*)c->ek_loss; *n_bytes = c->loss.bytes() / 3; return RNB_OK;
		case RNB_BUF_MASK_LOSS: *ptr = (void*)c->mask_loss; *n_bytes = c->loss.bytes() / 3; return RNB_OK;
		case RNB_BUF_COUNTERS: BUF(c->counters);
		case RNB_BUF_STEP_VECTOR: *ptr = (void*)(c->loss_sums.p + 8); *n_bytes = 7 * sizeof(double); return RNB_OK;
		case RNB_BUF_DENSITY_GRID_TMP: BUF(c->density_grid_tmp);
		case RNB_BUF_GRID_SAMPLE_POS: *ptr = c->grid_sample_pos.p; *n_bytes = (uint64_t)c->n_grid_samples * 12; return RNB_OK;
		case RNB_BUF_GRID_SAMPLE_IDX: *ptr = c->grid_sample_idx.p; *n_bytes = (uint64_t)c->n_grid_samples * 4; return RNB_OK;
		case RNB_BUF_GRID_SAMPLE_POS_EVAL: *ptr = c->gs_eval_pos.p; *n_bytes = c->last_update_sorted ? (uint64_t)c->n_grid_samples * 12 : 0; return RNB_OK;
		case RNB_BUF_GRID_SAMPLE_IDX_EVAL: *ptr = c->gs_eval_idx.p; *n_bytes = c->last_update_sorted ? (uint64_t)c->n_grid_samples * 4 : 0; return RNB_OK;
		default: return fail(RNB_ERR_INVALID, "unknown buffer id");
	}
#undef BUF
} RNB_GUARD

int rnb_params_changed(rnb_ctx* c) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	c->wimg_valid = false;
	return RNB_OK;
} RNB_GUARD

int rnb_bitfield_changed(rnb_ctx* c) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	discard_premarch(c); // a batch generated ahead of time marched through the old bits
	c->coarse_valid = false;
	c->gs_pre.valid = false;
	c->bitfield_foreign = true;
	return RNB_OK;
} RNB_GUARD

int rnb_device_malloc(rnb_ctx* c, uint64_t n_bytes, void** ptr) try {
	if (!c || !ptr) return fail(RNB_ERR_INVALID, "null argument");
	*ptr = nullptr;
	if (n_bytes == 0) return RNB_OK;
	if (hipMalloc(ptr, n_bytes) != hipSuccess) return fail(RNB_ERR_NOMEM, "hipMalloc failed");
	return RNB_OK;
} RNB_GUARD
int rnb_device_free(rnb_ctx* c, void* ptr) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (ptr) HIP_TRY(hipFree(ptr));
	return RNB_OK;
} RNB_GUARD

int rnb_memcpy(rnb_ctx* c, void* dst, const void* src, uint64_t n_bytes, int kind) try {
	if (!dst || !src) return fail(RNB_ERR_INVALID, "null argument");
	if (c && kind != RNB_D2H) { // a write into the occupancy grid: the next update's samples, prepared from the old grid, are generated afresh
		const char *d = static_cast<const char*>(dst), *g = reinterpret_cast<const char*>(c->density_grid.p);
		if (g && d < g + c->density_grid.bytes() && d + n_bytes > g) c->gs_pre.valid = false;
	}
	hipMemcpyKind k = kind == RNB_H2D ? hipMemcpyHostToDevice : kind == RNB_D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
	HIP_TRY(hipDeviceSynchronize());
	if (c) c->tail_pending = false;
	if (c) { const int rc = check_scan_errors(c); if (rc != RNB_OK) return rc; } // stage API: the scans of rnb_generate_training_samples / rnb_compute_loss are read through here
	HIP_TRY(hipMemcpy(dst, src, n_bytes, k));
	return RNB_OK;
} RNB_GUARD

int rnb_set_dataset(rnb_ctx* c, uint32_t n_views, const rnb_view* views, const uint16_t* const* normals, const uint16_t* const* albedos) try {
	if (!c || !views || !normals || !albedos || n_views == 0) return fail(RNB_ERR_INVALID, "bad dataset");
	discard_premarch(c);
	size_t total = 0;
	for (uint32_t v = 0; v < n_views; ++v) {
		if (views[v].width == 0 || views[v].height == 0 || !normals[v] || !albedos[v]) return fail(RNB_ERR_INVALID, "empty view");
		total += (size_t)views[v].width * views[v].height * 4 * 2;
	}
	c->pixels.free(); c->views.free();
	if (c->pixels.alloc(total) != hipSuccess) return fail(RNB_ERR_NOMEM, "hipMalloc failed for the dataset");
	if (c->views.alloc(n_views) != hipSuccess) return fail(RNB_ERR_NOMEM, "hipMalloc failed for view metadata");
	std::vector<ViewDev> vd(n_views);
	size_t off = 0;
	for (uint32_t v = 0; v < n_views; ++v) {
		const size_t n = (size_t)views[v].width * views[v].height * 4;
		vd[v].width = views[v].width; vd[v].height = views[v].height;
		for (int k = 0; k < 2; ++k) { vd[v].focal[k] = views[v].focal_length[k]; vd[v].principal[k] = views[v].principal_point[k]; }
		for (int k = 0; k < 12; ++k) vd[v].xform[k] = views[v].xform[k];
		vd[v].normal = c->pixels.p + off;
		HIP_TRY(hipMemcpy(c->pixels.p + off, normals[v], n * 2, hipMemcpyHostToDevice));
		off += n;
		vd[v].albedo = c->pixels.p + off;
		HIP_TRY(hipMemcpy(c->pixels.p + off, albedos[v], n * 2, hipMemcpyHostToDevice));
		off += n;
	}
	HIP_TRY(hipMemcpy(c->views.p, vd.data(), sizeof(ViewDev) * n_views, hipMemcpyHostToDevice));
	c->n_views = n_views;
	return RNB_OK;
} RNB_GUARD

int rnb_set_training_step(rnb_ctx* c, uint32_t step) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	discard_premarch(c);
	c->training_step = step;
	c->valid_level = compute_valid_level(c->cfg, (int)step);
	return RNB_OK;
} RNB_GUARD
uint32_t rnb_valid_level(const rnb_ctx* c) { return c ? c->valid_level : 0; }

int rnb_update_density_grid(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	discard_premarch(c);
	const int rc = training_prep(c, as_stream(stream));
	if (rc != RNB_OK) return rc;
	return pregenerate_grid_samples(c, as_stream(stream));
} RNB_GUARD
int rnb_set_grid_exchange(rnb_ctx* c, rnb_grid_exchange_fn fn, void* user) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	c->grid_exchange = fn; c->grid_exchange_user = user;
	return RNB_OK;
} RNB_GUARD
int rnb_update_density_grid_begin(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	discard_premarch(c);
	return training_prep_front(c, as_stream(stream), c->cfg.world_size > 1);
} RNB_GUARD
int rnb_update_density_grid_end(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	const int rc = update_density_grid_back(c, as_stream(stream));
	if (rc != RNB_OK) return rc;
	return pregenerate_grid_samples(c, as_stream(stream));
} RNB_GUARD
int rnb_update_density_bitfield(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	discard_premarch(c);
	c->gs_pre.valid = false; // called after the grid was written from outside
	return update_bitfield(c, as_stream(stream));
} RNB_GUARD

int rnb_sdf(rnb_ctx* c, void* stream, const float* xyz, uint32_t n, uint16_t* out, int inference) try {
	if (!c || (!xyz && n) || (!out && n)) return fail(RNB_ERR_INVALID, "null argument");
	return launch_point_query(c, as_stream(stream), xyz, n, reinterpret_cast<half_t*>(out), nullptr, nullptr, 0, inference != 0);
} RNB_GUARD
int rnb_density(rnb_ctx* c, void* stream, const float* xyz, uint32_t n, uint16_t* out, int inference) try {
	if (!c || (!xyz && n) || (!out && n)) return fail(RNB_ERR_INVALID, "null argument");
	return launch_point_query(c, as_stream(stream), xyz, n, reinterpret_cast<half_t*>(out), nullptr, nullptr, 1, inference != 0);
} RNB_GUARD
int rnb_forward_infer(rnb_ctx* c, void* stream, const float* coords, uint32_t n, uint16_t* out, int inference) try {
	if (!c || (!coords && n) || (!out && n)) return fail(RNB_ERR_INVALID, "null argument");
	return launch_forward(c, as_stream(stream), coords, nullptr, n, reinterpret_cast<half_t*>(out), inference != 0);
} RNB_GUARD

// ---- mesh extraction (src/testbed_nerf.cu:4218-4269, src/marching_cubes.cu:276-430, 794-822) ----
int rnb_sdf_lattice(rnb_ctx* c, void* stream, const uint32_t res[3], float lattice_min, float lattice_max, float* out, int inference) try {
	if (!c || !res || !out) return fail(RNB_ERR_INVALID, "null argument");
	if (!res[0] || !res[1] || !res[2]) return fail(RNB_ERR_INVALID, "empty lattice");
	hipStream_t s = as_stream(stream);
	const uint64_t n = (uint64_t)res[0] * res[1] * res[2];
	const uint32_t batch = 1u << 22; // lattice points per network launch (the reference: 2^20 per call, same arithmetic per point)
	float* pos = nullptr;
	half_t* val = nullptr;
	if (hipMalloc((void**)&pos, (size_t)batch * 12) != hipSuccess || hipMalloc((void**)&val, (size_t)batch * 2) != hipSuccess) {
		if (pos) (void)hipFree(pos);
		return fail(RNB_ERR_NOMEM, "hipMalloc failed for the lattice batch");
	}
	const float diag = c->aabb.mx - c->aabb.mn;
	int rc = RNB_OK;
	for (uint64_t off = 0; off < n && rc == RNB_OK; off += batch) {
		const uint32_t nb = (uint32_t)std::min<uint64_t>(batch, n - off);
		hipLaunchKernelGGL(k_lattice_positions, dim3((nb + 255) / 256), dim3(256), 0, s, off, nb, res[0], res[1], res[2], lattice_min, lattice_max - lattice_min, c->aabb.mn, diag, pos);
		rc = launch_point_query(c, s, pos, nb, val, nullptr, nullptr, 0, inference != 0);
		if (rc != RNB_OK) break;
		hipLaunchKernelGGL(k_half_to_float, dim3((nb + 255) / 256), dim3(256), 0, s, val, out + off, nb);
		if (hipGetLastError() != hipSuccess) rc = fail(RNB_ERR_DEVICE, "rnb_sdf_lattice: kernel launch failed");
	}
	hipError_t e = hipStreamSynchronize(s);
	(void)hipFree(pos); (void)hipFree(val);
	if (rc != RNB_OK) return rc;
	if (e != hipSuccess) return fail(RNB_ERR_DEVICE, std::string("rnb_sdf_lattice: ") + hipGetErrorString(e));
	return RNB_OK;
} RNB_GUARD

namespace {
// elements of block-sum scratch scan_exclusive needs for n counts: sum over the levels of ceil(n / 1024^k)
uint64_t scan_scratch_elems(uint64_t n) {
	uint64_t total = 0;
	do { n = (n + 1023) / 1024; total += n; } while (n > 1);
	return total;
}
// in-place exclusive prefix sums of n counts (device), total to *total_out; `scratch` holds scan_scratch_elems(n) uint32 (allocated once per
// rnb_marching_cubes call and shared by its two scans)
int scan_exclusive(uint32_t* data, uint64_t n, hipStream_t s, uint32_t* total_out, uint32_t* scratch) {
	std::vector<uint32_t*> levels{data};
	std::vector<uint64_t> sizes{n};
	while (true) {
		const uint64_t nb = (sizes.back() + 1023) / 1024;
		uint32_t* sums = scratch;
		scratch += nb;
		hipLaunchKernelGGL(k_scan_blocks, dim3((uint32_t)nb), dim3(1024), 0, s, levels.back(), sizes.back(), sums);
		levels.push_back(sums); sizes.push_back(nb);
		if (nb == 1) break;
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(total_out, levels.back(), 4, hipMemcpyDeviceToHost, s));
	// levels.back() holds one value (the total); every level below gets its block offsets added, top-down
	for (size_t k = levels.size() - 1; k >= 2; --k) { /* levels[k-1] was scanned by the launch that produced levels[k]; add it to levels[k-2] */
		hipLaunchKernelGGL(k_scan_add, dim3((uint32_t)sizes[k - 1]), dim3(1024), 0, s, levels[k - 2], sizes[k - 2], levels[k - 1]);
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	return RNB_OK;
}
} // namespace

int rnb_marching_cubes(rnb_ctx* c, void* stream, const float* density, const uint32_t res[3], const float aabb_min[3], const float aabb_max[3], float thresh,
                       float** verts_out, uint32_t** indices_out, uint32_t* n_verts, uint32_t* n_indices) try {
	if (!c || !density || !res || !aabb_min || !aabb_max || !verts_out || !indices_out || !n_verts || !n_indices) return fail(RNB_ERR_INVALID, "null argument");
	*verts_out = nullptr; *indices_out = nullptr; *n_verts = 0; *n_indices = 0;
	const uint64_t res3 = (uint64_t)res[0] * res[1] * res[2];
	if (res3 == 0 || res3 >= (1ull << 32)) return fail(RNB_ERR_INVALID, "lattice must hold 1 .. 2^32-1 points");
	hipStream_t s = as_stream(stream);
	if (!c->mc_table.p) { // the case table of host/mesh.hpp, once
		static const mesh::Tables T;
		std::vector<McTable> h(1);
		for (int m = 0; m < 256; ++m) {
			int k = 0;
			for (; T.tri[m][k] >= 0; ++k) h[0].tri[m][k] = T.tri[m][k];
			h[0].n[m] = (uint8_t)k;
			for (; k < 40; ++k) h[0].tri[m][k] = -1;
		}
		if (c->mc_table.alloc(1) != hipSuccess) return fail(RNB_ERR_NOMEM, "hipMalloc failed for the case table");
		HIP_TRY(hipMemcpy(c->mc_table.p, h.data(), sizeof(McTable), hipMemcpyHostToDevice));
	}
	McArgs a;
	a.density = density; a.rx = res[0]; a.ry = res[1]; a.rz = res[2]; a.thresh = thresh;
	for (int d = 0; d < 3; ++d) { a.sc[d] = (aabb_max[d] - aabb_min[d]) / (float)res[d]; a.mn[d] = aabb_min[d]; }
	const uint64_t n_wg64 = (res3 + MC_WG - 1) / MC_WG;
	const uint32_t n_wg = (uint32_t)n_wg64;
	uint32_t* wg = nullptr;
	uint32_t* scan_scratch = nullptr;
	int32_t* vidx = nullptr;
	float* verts = nullptr;
	uint32_t* indices = nullptr;
	auto cleanup = [&](int rc) { if (wg) (void)hipFree(wg); if (scan_scratch) (void)hipFree(scan_scratch); if (vidx) (void)hipFree(vidx); if (rc != RNB_OK) { if (verts) (void)hipFree(verts); if (indices) (void)hipFree(indices); } return rc; };
	if (hipMalloc((void**)&wg, (size_t)n_wg * 4) != hipSuccess || hipMalloc((void**)&scan_scratch, scan_scratch_elems(n_wg) * 4) != hipSuccess || hipMalloc((void**)&vidx, (size_t)res3 * 3 * 4) != hipSuccess) return cleanup(fail(RNB_ERR_NOMEM, "hipMalloc failed for the marching-cubes scratch (12 bytes per lattice point)"));
	uint32_t nv = 0, ni = 0;
	hipLaunchKernelGGL(k_mc_verts<false>, dim3(n_wg), dim3(MC_WG), 0, s, a, wg, (const uint32_t*)nullptr, (float*)nullptr, (int32_t*)nullptr);
	int rc = scan_exclusive(wg, n_wg, s, &nv, scan_scratch);
	if (rc != RNB_OK) return cleanup(rc);
	if (nv && hipMalloc((void**)&verts, (size_t)nv * 12) != hipSuccess) return cleanup(fail(RNB_ERR_NOMEM, "hipMalloc failed for the vertices"));
	hipLaunchKernelGGL(k_mc_verts<true>, dim3(n_wg), dim3(MC_WG), 0, s, a, (uint32_t*)nullptr, wg, verts, vidx);
	hipLaunchKernelGGL(k_mc_faces<false>, dim3(n_wg), dim3(MC_WG), 0, s, a, c->mc_table.p, wg, (const uint32_t*)nullptr, vidx, (uint32_t*)nullptr);
	rc = scan_exclusive(wg, n_wg, s, &ni, scan_scratch);
	if (rc != RNB_OK) return cleanup(rc);
	if (ni && hipMalloc((void**)&indices, (size_t)ni * 4) != hipSuccess) return cleanup(fail(RNB_ERR_NOMEM, "hipMalloc failed for the indices"));
	hipLaunchKernelGGL(k_mc_faces<true>, dim3(n_wg), dim3(MC_WG), 0, s, a, c->mc_table.p, (uint32_t*)nullptr, wg, vidx, indices);
	if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) return cleanup(fail(RNB_ERR_DEVICE, "rnb_marching_cubes: kernel failure"));
	*verts_out = verts; *indices_out = indices; *n_verts = nv; *n_indices = ni;
	return cleanup(RNB_OK);
} RNB_GUARD

int rnb_generate_training_samples(rnb_ctx* c, void* stream, uint32_t n_rays, uint32_t n_rays_total, uint32_t max_samples) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	discard_premarch(c);
	if (c->n_views == 0) return fail(RNB_ERR_INVALID, "no dataset");
	if (n_rays == 0 || n_rays > c->cfg.max_rays_per_batch) return fail(RNB_ERR_INVALID, "n_rays out of range");
	if (max_samples > c->cfg.target_batch_size * 16) return fail(RNB_ERR_INVALID, "max_samples exceeds 16*target_batch_size");
	c->cin_flow = false; // stage calls never use the rows a (failed) training step may have left behind
	return generate_training_samples(c, as_stream(stream), n_rays, n_rays_total, max_samples);
} RNB_GUARD

int rnb_compute_loss(rnb_ctx* c, void* stream, uint32_t n_rays, uint32_t n_rays_total) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (c->n_views == 0) return fail(RNB_ERR_INVALID, "no dataset");
	if (n_rays == 0 || n_rays > c->cfg.max_rays_per_batch) return fail(RNB_ERR_INVALID, "n_rays out of range");
	c->cin_flow = false;
	return compute_loss(c, as_stream(stream), n_rays, n_rays_total);
} RNB_GUARD

int rnb_forward_backward(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	c->cin_flow = false; // the stage pass evaluates the compacted batch itself for the colour MLP's input rows
	return forward_backward(c, as_stream(stream));
} RNB_GUARD

int rnb_optimizer_step(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	return optimizer_step(c, as_stream(stream));
} RNB_GUARD

// Drops samples generated ahead of time for a step whose inputs have since changed (controller, flags, bitfield ...).
static void discard_premarch(rnb_ctx* c) {
	c->pre.march_joined = false;
	if (!c->pre.valid) return;
	(void)hipStreamSynchronize(c->s_march);
	c->n_rays_total = c->pre.n_rays_total;
	c->pre.valid = false;
	c->pre.loss_cleared = false;
}

static uint32_t next_max_inference(rnb_ctx* c) { // testbed_nerf.cu:3891-3896
	const uint32_t max_samples = c->cfg.target_batch_size * 16;
	if (c->measured_batch_size_before_compaction == 0) { c->measured_batch_size_before_compaction = max_samples; return max_samples; }
	return next_multiple_u32(std::min(c->measured_batch_size_before_compaction, max_samples), 128u);
}


// Occupancy update (when due), ray generation + march, network evaluation of all samples, loss + compaction.
static int step_front(rnb_ctx* c, hipStream_t s) {
	c->valid_level = compute_valid_level(c->cfg, (int)c->training_step); // testbed.cu:2792
	const bool joined = c->join_pending; // ev_join: the optimizer's and the weight images' streams of the previous step AND ev_march
	if (c->join_pending) { HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0)); c->join_pending = false; }
	if (c->tail_pending) { HIP_TRY(hipStreamWaitEvent(s, c->ev_tail, 0)); c->tail_pending = false; } // no march was queued behind the optimizer to take the join
	c->grid_updated = false;
	c->prep_ms = 0.f;
	int rc;
	if (prep_due(c->training_step)) {
		discard_premarch(c); // never generated for such a step; defensive
		const uint32_t n_prep_to_skip = std::min(std::max(c->training_step / 16u, 1u), 16u);
		auto t0 = std::chrono::steady_clock::now();
		rc = training_prep(c, s);
		if (rc != RNB_OK) return rc;
		// testbed.cu:2820 waits here (it times the update). The overlapped schedule does not: the march, the network evaluation and the loss pass of
		// this step are queued behind the update while it runs (the wait left the queue empty for ~0.1 ms after every update, profiles/r03_timeline_update_*);
		// prep_ms is then the time it took to queue the update.
		if (!c->overlap()) HIP_TRY(hipStreamSynchronize(s));
		c->grid_updated = true;
		c->prep_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / n_prep_to_skip;
	}
	c->step_start = std::chrono::steady_clock::now();
	const uint32_t max_inference = next_max_inference(c);
	if (c->training_step == 0) { discard_premarch(c); c->n_rays_total = 0; } // testbed_nerf.cu:3906-3908
	const uint32_t n_rays = c->rays_per_batch;
	if (c->pre.valid && (c->pre.n_rays != n_rays || c->pre.max_inference != max_inference)) discard_premarch(c);
	uint32_t n_rays_total;
	bool join_rest = false; // the second k_march_write of a march generated ahead is still to be waited for (knobs.march_write_split)
	if (c->pre.valid) { // generated beside the previous step's backward pass
		n_rays_total = c->pre.n_rays_total;
		c->pre.valid = false;
		c->cur_k1 = c->pre.k1;
		join_rest = c->pre.split;
		if (!joined && !c->pre.march_joined && !c->knobs.unsafe_skip_joins) HIP_TRY(hipStreamWaitEvent(s, c->ev_march, 0));
		c->pre.march_joined = false;
	} else {
		n_rays_total = c->n_rays_total;
		c->n_rays_total += n_rays * c->cfg.world_size;
		// (Counters::prepare_for_training_steps, testbed_nerf.cu:3519-3530, zeroes the counters here; every one of them is written with a plain store by
		// the scans of this step -- k_scan_rays*: [0] [2] [3], k_scan_compact*: [1] -- so the fill, a launch on the critical path of the update steps, is left out)
		rc = generate_training_samples(c, s, n_rays, n_rays_total, max_inference);
		if (rc != RNB_OK) return rc;
		c->cur_k1 = c->gen_k1;
	}
	c->cur_n_rays = n_rays;
	c->cur_n_rays_total = n_rays_total;
	c->prof.mark(s, P_NONE);
	const bool two_round = c->cur_k1 != 0;
	c->cin_flow = c->rgb_split();
	if (c->cin_flow) { rc = ensure_rgb_buffers(c); if (rc != RNB_OK) return rc; }
	half_t* cin_out = c->cin_flow ? c->cin_eval.p : nullptr;
	if (two_round) rc = launch_forward(c, s, c->coords.p, c->fwd_counts.p, max_inference, c->mlp_out.p, false, c->idx1.p, cin_out);
	else rc = launch_forward(c, s, c->coords.p, c->counters.p + 3, max_inference, c->mlp_out.p, false, nullptr, cin_out);
	if (rc != RNB_OK) return rc;
	c->prof.mark(s, P_FORWARD);
	if (join_rest) HIP_TRY(hipStreamWaitEvent(s, c->ev_march_rest, 0));
	rc = compute_loss(c, s, n_rays, n_rays_total, two_round ? max_inference : 0, true, true);
	if (rc != RNB_OK) return rc;
	return pregenerate_grid_samples(c, s); // after an update: the next one's samples, behind everything this step's front needed from the host
}

static int step_back(rnb_ctx* c, hipStream_t s) {
	int rc = forward_backward(c, s, false);
	if (rc != RNB_OK) return rc;
	c->rng.advance(); // testbed_nerf.cu:4118
	return RNB_OK;
}

// Polling mode (launch_reduce_losses): spin on the sequence word of the pinned readback block until the step's k_reduce_losses has published it.
static int wait_loss_readback(rnb_ctx* c, hipStream_t s) {
	if (c->loss_polled) return RNB_OK;
	const volatile uint32_t* seq = &c->host_rb->seq;
	const auto t0 = std::chrono::steady_clock::now();
	for (uint64_t spins = 0; *seq != c->rb_seq; ++spins) {
		if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) { // the kernel never ran, or failed: let the runtime say why
			HIP_TRY(hipStreamSynchronize(s));
			if (*seq != c->rb_seq) return fail(RNB_ERR_DEVICE, "the loss readback of the step did not arrive");
		}
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	c->loss_polled = true;
	return RNB_OK;
}

static int launch_reduce_losses(rnb_ctx* c, hipStream_t s) {
	if (c->loss_reduced) return RNB_OK; // k_loss_pass2_samples did it (compute_loss)
	c->prof.mark(s, P_NONE);
	// the 48-byte readback goes straight into the pinned host block (no copy kernel, no marker packet); ev_loss is the kernel's completion
	const bool tiled = c->cur_n_rays >= c->knobs.march_narrow_from;
	const uint32_t n_tiles = (c->cur_n_rays + SCAN_TILE - 1) / SCAN_TILE;
	if (tiled) hipLaunchKernelGGL(k_reduce_losses_tiles, dim3(n_tiles), dim3(1024), 0, s, c->cur_n_rays, c->counters.p, c->loss.p, c->ek_loss, c->mask_loss, c->loss_partial.p);
	// Single GPU, overlapped schedule: no completion event (ev_loss carries a system-scope fence and has a waiter on the march's stream: ~5 us between this kernel
	// and k_fwd_bwd, tools/probe_barriers.hip); the host polls the readback's sequence word instead (wait_loss_readback). Data parallel / serial: the event, as before.
	const bool poll = c->poll_loss();
	if (poll) { if (++c->rb_seq == 0) ++c->rb_seq; c->loss_polled = false; }
	LAUNCH_EV(k_reduce_losses_rollover, dim3(1 + 255), dim3(1024), 0, s, poll ? nullptr : c->ev_loss, c->cur_n_rays, c->counters.p, c->loss.p, c->ek_loss, c->mask_loss, c->loss_sums.p, c->fwd_counts.p,
	          reinterpret_cast<double*>(c->host_rb_dev), tiled ? c->loss_partial.p : nullptr, n_tiles, c->cfg.target_batch_size, c->dloss_dout.p, c->coords_compacted.p, c->cin_flow ? c->src_slot.p : nullptr,
	          poll ? c->rb_seq : 0u);
	c->prof.mark(s, P_REDUCE);
	HIP_TRY(hipGetLastError());
	return RNB_OK;
}

// Ray generation + march of the step after the current one, on the side stream, once the current step's loss pass has
// released the sample buffers (ev_loss). Skipped when that step starts with an occupancy update (it changes the bitfield).
static int launch_premarch(rnb_ctx* c) {
	if (!c->overlap() || c->pre.valid || prep_due(c->cur_step + 1)) return RNB_OK; // cur_step + 1: _finish may run before _apply
	const uint32_t n_rays = c->rays_per_batch, max_inference = next_max_inference(c), n_rays_total = c->n_rays_total;
	// The march starts right after the loss pass, i.e. beside k_fwd_bwd (RNB_MARCH_LATE=1: only when k_fwd_bwd is done). History,
	// measured with tools/march_determinism.py: an earlier 16-lanes-per-ray kernel, which carried the index of the next visited
	// position in four more ballot masks (SGPR pairs, spilled through VGPR lanes) and used packed fp32 instructions, came out
	// beside k_fwd_bwd with a wrong direction for a few rays (wavefront lanes 48-63 only) in 2-3 % of the launches, 26-54 % beside
	// the generic kernel of the albedo mode -- a sample set that the same launch on an idle GPU does not produce. Without packed
	// fp32 (rnb-neus2_amd/build.py) and with that index read through one cross-lane shuffle instead: 0 of 2000 launches beside
	// k_fwd_bwd (0 of 1300 for the one-thread-per-ray kernel of the large batches). DESIGN.md section 6.
	if (c->poll_loss()) { const int rc0 = wait_loss_readback(c, c->step_stream); if (rc0 != RNB_OK) return rc0; } // the host has SEEN the loss pass end: nothing to wait for on the device
	else HIP_TRY(hipStreamWaitEvent(c->s_march, c->ev_loss, 0));
	// No fill in front of the march: every step counter is written with a plain store by the scans (k_scan_rays*, k_scan_compact*), and the
	// loss rows are written for every kept ray by k_loss_pass2 (zeros for a ray without compacted samples) -- k_reduce_losses reads nothing
	// else. (Round 2 queued a k_clear_step here: 62 us behind k_fwd_bwd_sdf's workgroups at the head of the march chain.)
	// Albedo mode: always. Its two training kernels (k_rgb_fwd_bwd: one wavefront per SIMD with ~470 registers; k_fwd_bwd_sdf_full: two with 256) and the march
	// exclude each other on a SIMD; a march that has started beside the first keeps the second at half occupancy (253 instead of 113 us, the step 0.79 instead of
	// 0.76 ms). Behind them it runs beside the scatter, as it effectively does with --no-albedo, where k_fwd_bwd_sdf claims the registers first.
	if (c->knobs.march_late || c->rgb_split()) HIP_TRY(hipStreamWaitEvent(c->s_march, c->ev_fb, 0));
	int rc = generate_training_samples(c, c->s_march, n_rays, n_rays_total, max_inference, c->ev_march, c->tail_pending ? c->ev_tail : nullptr, c->ev_march_rest,
	                                   // the loss sums were published from INSIDE k_loss_pass2_samples (workgroup 0), whose other workgroups may still be reading the step's sample buffers when the host
	                                   // gets here: k_march_count touches none of them, everything behind it waits for k_fwd_bwd*'s completion (same stream as the loss pass, behind it)
	                                   c->loss_reduced ? c->ev_fb : nullptr);
	if (rc != RNB_OK) return rc;
	c->tail_pending = false; // the next step reaches ev_tail through ev_march
	c->pre.loss_cleared = true;
	c->n_rays_total += n_rays * c->cfg.world_size;
	c->pre.march_joined = false;
	c->pre.valid = true; c->pre.n_rays = n_rays; c->pre.n_rays_total = n_rays_total; c->pre.max_inference = max_inference; c->pre.k1 = c->gen_k1; c->pre.split = c->gen_split;
	return RNB_OK;
}

int rnb_train_step_begin(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (c->n_views == 0) return fail(RNB_ERR_INVALID, "no dataset");
	hipStream_t s = as_stream(stream);
	c->cur_step = c->training_step;
	c->step_stream = s;
	int rc = step_front(c, s);
	if (rc != RNB_OK) { c->cin_flow = false; return rc; }
	// the step's counters and loss sums are final here: hand them to the host now, so that the ray controller (and the next
	// step's march) does not have to wait for the backward pass
	rc = launch_reduce_losses(c, s);
	if (rc != RNB_OK) return rc;
	return step_back(c, s);
} RNB_GUARD

int rnb_train_step_apply(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	hipStream_t s = as_stream(stream);
	int rc = optimizer_step(c, s);
	if (rc != RNB_OK) return rc;
	++c->training_step;
	return RNB_OK;
} RNB_GUARD

// Counters + loss sums of the step started by the last rnb_train_step_begin. May be called before or after _apply.
int rnb_train_step_local(rnb_ctx* c, void* stream, uint64_t counters_out[4], double loss_sums_out[3]) try {
	if (!c || !counters_out || !loss_sums_out) return fail(RNB_ERR_INVALID, "null argument");
	if (c->poll_loss()) { const int rc = wait_loss_readback(c, as_stream(stream)); if (rc != RNB_OK) return rc; } // the backward pass / optimizer may still be running
	else if (c->overlap()) HIP_TRY(hipEventSynchronize(c->ev_loss));
	else HIP_TRY(hipStreamSynchronize(as_stream(stream)));      // testbed.cu:2866
	{ const int rc = check_scan_errors(c); if (rc != RNB_OK) return rc; }
	const uint32_t* counters = c->host_rb->counters;
	if (c->prof.on) { c->prof.collect(); c->prof.units[P_FORWARD] += c->cur_k1 ? c->host_rb->fwd[0] + c->host_rb->fwd[1] : counters[3]; }
	for (int k = 0; k < 4; ++k) counters_out[k] = counters[k];
	for (int k = 0; k < 3; ++k) loss_sums_out[k] = c->host_rb->sums[k];
	c->local_measured_before = counters[0];
	return RNB_OK;
} RNB_GUARD

// Counters::update_after_training (testbed_nerf.cu:3532-3558) on counters summed over the data-parallel ranks.
int rnb_train_step_finish(rnb_ctx* c, const uint64_t counters[4], const double sums[3], rnb_step_stats* stats) try {
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
		c->measured_batch_size_before_compaction = c->local_measured_before; // bounds this rank's next inference launch
		c->measured_batch_size = (uint32_t)(counters[1] / c->cfg.world_size);
		const float measured = (float)counters[1], target = (float)Bg;
		loss_scalar = (float)sums[0] * measured / target;
		ek_scalar = (float)sums[1] * measured / target;
		mask_scalar = (float)sums[2] * measured / target;
		next_rays = (uint32_t)((float)c->rays_per_batch * target / measured);
		next_rays = std::min(next_multiple_u32(next_rays, 128u), c->cfg.max_rays_per_batch);
	}
	if (stats) {
		stats->training_step = c->cur_step + 1;
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
	if (rc != RNB_OK) return rc;
	return launch_premarch(c);
} RNB_GUARD

int rnb_train_step_end(rnb_ctx* c, void* stream, rnb_step_stats* stats) try {
	if (c && c->overlap()) {
		// Overlapped schedule: the ray controller and the next step's march FIRST, the moment the loss readback arrives, the optimizer's launches behind them (they are not
		// due before the first scatter group has finished, ~0.2 ms later). Measured (round 4, 200-step slices): optimizer first 0.6081 / 0.5953 / 0.6319 ms/step at steps
		// 1000 / 2000 / 6000, march first through four Python calls 0.6028 / 0.5897 / 0.6285. The optimizer runs even when the step produced no samples, as in the reference.
		uint64_t counters[4];
		double sums[3];
		int rc = rnb_train_step_local(c, stream, counters, sums);
		if (rc != RNB_OK) return rc;
		const int rc_finish = rnb_train_step_finish(c, counters, sums, stats);
		rc = rnb_train_step_apply(c, stream);
		return rc != RNB_OK ? rc : rc_finish;
	}
	int rc = rnb_train_step_apply(c, stream);
	if (rc != RNB_OK) return rc;
	uint64_t counters[4];
	double sums[3];
	rc = rnb_train_step_local(c, stream, counters, sums);
	if (rc != RNB_OK) return rc;
	return rnb_train_step_finish(c, counters, sums, stats);
} RNB_GUARD

int rnb_train_step(rnb_ctx* c, void* stream, rnb_step_stats* stats) try {
	// With cfg.overlap the host only waits for the loss pass (rnb_train_step_local), so the call returns once the step's
	// statistics are known and the next step's march has been queued beside the backward pass; the optimizer may still be
	// running. Without it (or while profiling) every wait is a full stream synchronisation, as in the reference.
	int rc = rnb_train_step_begin(c, stream);
	if (rc != RNB_OK) return rc;
	return rnb_train_step_end(c, stream, stats);
} RNB_GUARD

int rnb_profile_enable(rnb_ctx* c, int on) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	c->prof.on = on != 0;
	c->prof.reset();
	return RNB_OK;
} RNB_GUARD
int rnb_profile_count(const rnb_ctx*) { return P_COUNT; }
int rnb_profile_get(const rnb_ctx* c, int idx, const char** name, double* total_ms, uint64_t* launches, double* units) try {
	if (!c || idx < 0 || idx >= P_COUNT) return fail(RNB_ERR_INVALID, "bad profile index");
	if (name) *name = PROF_NAMES[idx];
	if (total_ms) *total_ms = c->prof.total_ms[idx];
	if (launches) *launches = c->prof.launches[idx];
	if (units) *units = c->prof.units[idx];
	return RNB_OK;
} RNB_GUARD

uint32_t rnb_training_step(const rnb_ctx* c) { return c ? c->training_step : 0; }
uint32_t rnb_rays_per_batch(const rnb_ctx* c) { return c ? c->rays_per_batch : 0; }

int rnb_eval_primitives(rnb_ctx* c, int kind, const uint32_t* in_host, uint32_t n_items, uint32_t* out_host) try {
	if (!c || (!in_host && n_items) || (!out_host && n_items)) return fail(RNB_ERR_INVALID, "null argument");
	if (kind < 0 || kind > RNB_PRIM_DW_SLICED) return fail(RNB_ERR_INVALID, "unknown primitive kind");
	if (n_items == 0) return RNB_OK;
	if (kind == RNB_PRIM_PREP_DUE) { // host logic (Testbed::train, src/testbed.cu:2805-2806): does training step in[i] begin with an occupancy update, and the interval it is due at
		for (uint32_t i = 0; i < n_items; ++i) { out_host[2 * i] = prep_due(in_host[i]) ? 1u : 0u; out_host[2 * i + 1] = std::min(std::max(in_host[i] / 16u, 1u), 16u); }
		return RNB_OK;
	}
	const size_t n_in = (size_t)n_items * PRIM_IN_WORDS[kind], n_out = (size_t)n_items * PRIM_OUT_WORDS[kind], n_bf = (size_t)GRID_CELLS / 8 * N_CASCADES;
	uint32_t *in = nullptr, *out = nullptr;
	uint8_t* bf = nullptr;
	int rc = RNB_OK;
	if (hipMalloc((void**)&in, n_in * 4) != hipSuccess || hipMalloc((void**)&out, n_out * 4) != hipSuccess || ((kind == RNB_PRIM_MARCH || kind == RNB_PRIM_MARCH_RAY) && hipMalloc((void**)&bf, n_bf) != hipSuccess))
		rc = fail(RNB_ERR_NOMEM, "hipMalloc failed for the primitive self-test");
	std::vector<uint32_t> own; // RNB_PRIM_RAY_TARGETS: nine words 0xffffffff for the light directions = "the context's own" (build_light_dirs)
	if (kind == RNB_PRIM_RAY_TARGETS) {
		own.assign(in_host, in_host + n_in);
		for (uint32_t i = 0; i < n_items; ++i) {
			uint32_t* ld = own.data() + (size_t)i * PRIM_IN_WORDS[kind] + 26;
			bool all = true;
			for (int k = 0; k < 9; ++k) all = all && ld[k] == 0xffffffffu;
			if (all) std::memcpy(ld, c->light_dirs, 36);
		}
		in_host = own.data();
	}
	if (rc == RNB_OK && hipMemcpy(in, in_host, n_in * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(RNB_ERR_DEVICE, "rnb_eval_primitives: copy in failed");
	if (rc == RNB_OK) {
		if (bf) hipLaunchKernelGGL(k_prim_bitfield, dim3((uint32_t)((n_bf + 255) / 256)), dim3(256), 0, 0, bf, (uint32_t)n_bf);
		if (kind == RNB_PRIM_DW_SLICED) { // the half mode's weight-gradient GEMM (k_dw_sliced + the slice sum of k_dw_finish) on the caller's operands, one 4 x 4 GEMM per item
			const uint32_t S = PRIM_DW_SAMPLES, n_sl = (S + DW_SLICE - 1) / DW_SLICE;
			float* tmp = nullptr;
			if (hipMalloc((void**)&tmp, (size_t)n_sl * 16 * 4) != hipSuccess) rc = fail(RNB_ERR_NOMEM, "hipMalloc failed for the primitive self-test");
			for (uint32_t i = 0; i < n_items && rc == RNB_OK; ++i) {
				const uint32_t* item = in + (size_t)i * PRIM_IN_WORDS[kind];
				DwSlicedArgs q;
				for (uint32_t g = 0; g < 7; ++g) { q.YT[g] = nullptr; q.XT[g] = nullptr; q.out[g] = nullptr; q.n_out[g] = 0; q.n_out_live[g] = 0; q.n_in[g] = 4; q.ones[g] = 0; }
				const bool ones = (in_host[(size_t)i * PRIM_IN_WORDS[kind]] & 1u) != 0u;
				q.YT[0] = reinterpret_cast<const half_t*>(item + 4); q.XT[0] = q.YT[0] + (size_t)4 * S; q.out[0] = tmp;
				q.n_out[0] = 4; q.n_out_live[0] = ones ? 1u : 4u; q.n_in[0] = 4; q.ones[0] = ones ? 1u : 0u;
				q.n = 1; q.B = S; q.n_slices = n_sl; q.first_wg[0] = 0;
				for (uint32_t g = 1; g < 8; ++g) q.first_wg[g] = n_sl;
				hipLaunchKernelGGL(k_dw_sliced, dim3(n_sl), dim3(256), 0, 0, q);
				hipLaunchKernelGGL(k_prim_dw_sliced_total, dim3(1), dim3(64), 0, 0, tmp, 16u, n_sl, reinterpret_cast<float*>(out + (size_t)i * 16));
			}
			(void)hipDeviceSynchronize();
			if (tmp) (void)hipFree(tmp);
		} else if (kind == RNB_PRIM_ENCODE) hipLaunchKernelGGL(k_prim_encode, dim3(n_items), dim3(64), 0, 0, in, n_items, out);
		else hipLaunchKernelGGL(k_primitives, dim3((n_items + 127) / 128), dim3(128), 0, 0, kind, in, n_items, out, bf);
		if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemcpy(out_host, out, n_out * 4, hipMemcpyDeviceToHost) != hipSuccess)
			rc = fail(RNB_ERR_DEVICE, "rnb_eval_primitives: kernel or copy out failed");
	}
	if (in) (void)hipFree(in);
	if (out) (void)hipFree(out);
	if (bf) (void)hipFree(bf);
	return rc;
} RNB_GUARD

int rnb_set_optimizer_step(rnb_ctx* c, uint32_t step) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	HIP_TRY(hipDeviceSynchronize()); // a pipelined step may still be running its optimizer
	c->opt.begun = false;
	c->optimizer_step_count = step;
	c->lr_factor = 1.0f;
	for (uint64_t s0 = c->cfg.lr_decay_start; s0 < step && s0 <= 10000000u; s0 += std::max(1u, c->cfg.lr_decay_interval)) c->lr_factor *= c->cfg.lr_decay_base; // exponential_decay.h:61-72, one factor per event
	return RNB_OK;
} RNB_GUARD

int rnb_set_controller(rnb_ctx* c, uint32_t training_step, uint32_t rays_per_batch, uint32_t measured_before, uint32_t n_rays_total) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (rays_per_batch == 0 || rays_per_batch > c->cfg.max_rays_per_batch) return fail(RNB_ERR_INVALID, "rays_per_batch out of range");
	discard_premarch(c);
	c->training_step = training_step;
	c->valid_level = compute_valid_level(c->cfg, (int)training_step);
	c->rays_per_batch = rays_per_batch;
	c->measured_batch_size_before_compaction = measured_before;
	c->n_rays_total = n_rays_total;
	return RNB_OK;
} RNB_GUARD

int rnb_gradient_parts(rnb_ctx* c, uint64_t ranges[3][2], uint32_t* n_parts) try {
	if (!c || !ranges || !n_parts) return fail(RNB_ERR_INVALID, "null argument");
	flush_c_side(c, c->backward_stream); // (a caller that exchanges gradients: group C belongs to the backward pass's stream again)
	c->sc.exchanged = true; // the caller sums gradients across ranks: the optimizer must not start on a block before its exchange
	if (c->sc.valid && c->sc.dp) { // scatter order C, B, A1, A2: everything in front of A's levels is final first (ev_sc[0]), then A1 (ev_sc[1])
		const bool mid = c->sc.split_mid > c->sc.split[0] && c->sc.split_mid < c->off_var;
		ranges[0][0] = 0;              ranges[0][1] = c->sc.split[0];
		ranges[1][0] = c->sc.split[0]; ranges[1][1] = mid ? c->sc.split_mid : c->n_params;
		if (mid) { ranges[2][0] = c->sc.split_mid; ranges[2][1] = c->n_params; }
		*n_parts = mid ? 3 : 2;
	} else if (c->sc.valid && c->sc.split[1] < c->sc.split[0]) { // scatter order B, A, C: B's levels are final first (ev_sc[0])
		ranges[0][0] = c->sc.split[1]; ranges[0][1] = c->sc.split[0];
		ranges[1][0] = 0;              ranges[1][1] = c->sc.split[1];
		ranges[2][0] = c->sc.split[0]; ranges[2][1] = c->n_params;
		*n_parts = 3;
	} else {
		ranges[0][0] = 0; ranges[0][1] = c->n_params; *n_parts = 1;
	}
	return RNB_OK;
} RNB_GUARD

int rnb_train_step_apply_early(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	flush_c_side(c, c->backward_stream);
	return optimizer_step_early(c, as_stream(stream));
} RNB_GUARD

int rnb_shard_layout(rnb_ctx* c, rnb_shard_part parts[RNB_MAX_SHARD_PARTS], uint32_t* n_parts, uint64_t* capacity) try {
	if (!c || !parts || !n_parts || !capacity) return fail(RNB_ERR_INVALID, "null argument");
	flush_c_side(c, c->backward_stream);
	c->sc.exchanged = true;
	c->sc.sharded = true;
	shard_layout(c, parts, n_parts);
	*capacity = c->param_capacity;
	return RNB_OK;
} RNB_GUARD

int rnb_train_step_apply_shard(rnb_ctx* c, uint32_t part, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	flush_c_side(c, c->backward_stream);
	return optimizer_step_shard(c, part, as_stream(stream));
} RNB_GUARD

int rnb_train_step_apply_done(rnb_ctx* c, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	if (!c->opt.begun) return fail(RNB_ERR_INVALID, "rnb_train_step_apply_done without rnb_train_step_apply_shard");
	int rc = optimizer_finish(c, as_stream(stream));
	if (rc != RNB_OK) return rc;
	++c->training_step;
	return RNB_OK;
} RNB_GUARD

int rnb_gradient_part_wait(rnb_ctx* c, uint32_t part, void* stream) try {
	if (!c) return fail(RNB_ERR_INVALID, "null ctx");
	flush_c_side(c, c->backward_stream);
	// blocks 0 (and, in the data-parallel order, 1 = the first half of the fine levels) of the overlapped schedule have their own events; everything is final at the
	// end of the backward pass
	const bool early = part == 0 && c->sc.valid && (c->sc.dp || !c->sc.sharded);
	bool mid = false;
	if (part == 1 && c->sc.valid && c->sc.dp) { // is block 1 the middle one of three? (the same test as rnb_gradient_parts / shard_layout)
		if (c->sc.sharded) { rnb_shard_part parts[RNB_MAX_SHARD_PARTS]; uint32_t n = 0; shard_layout(c, parts, &n); mid = n == 3; }
		else mid = c->sc.split_mid > c->sc.split[0] && c->sc.split_mid < c->off_var;
	}
	if (!c->sc.dw_joined) HIP_TRY(hipStreamWaitEvent(as_stream(stream), c->ev_dw, 0)); // the weight-gradient GEMMs' side stream has not been joined
	if (!early && !mid && !c->sc.all_final_recorded) { // only data-parallel callers pay for this marker
		HIP_TRY(hipEventRecord(c->ev_all, c->backward_stream));
		c->sc.all_final_recorded = true;
	}
	HIP_TRY(hipStreamWaitEvent(as_stream(stream), early ? c->ev_sc[0] : mid ? c->ev_sc[1] : c->ev_all, 0));
	if (early && c->sc.dp) HIP_TRY(hipStreamWaitEvent(as_stream(stream), c->ev_dw, 0)); // block 0 holds the MLPs' gradients (side stream)
	return RNB_OK;
} RNB_GUARD

} // extern "C"
