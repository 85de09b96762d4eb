// kernels_ray.cuh — occupancy grid, ray marching and loss kernels.
//   k_grid_samples / k_ema_grid / k_mean_* / k_grid_to_bitfield / k_bitfield_max_pool   testbed_nerf.cu:585-740, 3424-3517
//   k_march_count / k_scan_rays / k_march_write                                         testbed_nerf.cu:1216-1387
//   k_loss_pass1 / k_scan_compact / k_loss_pass2 / k_rollover                            testbed_nerf.cu:1396-2097, 4044-4052
// Sample slots are assigned in ray order by prefix sums instead of the reference's atomicAdd order (any order is a
// legal outcome of the reference's race; ray order makes the step reproducible and the sample stream coalesced).
#pragma once
#include "common.cuh"
#include "chain.cuh"

namespace rnb {

struct SceneAabb { float mn, mx, cone_angle; uint32_t max_cascade; };

__device__ __forceinline__ Vec3 warp_position(const SceneAabb& A, const Vec3& p) {
	const float diag = A.mx - A.mn;
	return {(p.x - A.mn) / diag, (p.y - A.mn) / diag, (p.z - A.mn) / diag};
}
__device__ __forceinline__ bool aabb_contains(const SceneAabb& A, const Vec3& p) {
	return p.x >= A.mn && p.x <= A.mx && p.y >= A.mn && p.y <= A.mx && p.z >= A.mn && p.z <= A.mx;
}
__device__ __forceinline__ void ray_intersect(const SceneAabb& A, const Vec3& pos, const Vec3& dir, float* tmin_o, float* tmax_o) { // bounding_box.cuh:163-206
	const float mn = A.mn, mx = A.mx;
	const float FMAX = 3.402823466e+38f;
	float tmin = (mn - pos.x) / dir.x, tmax = (mx - pos.x) / dir.x;
	if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
	float tymin = (mn - pos.y) / dir.y, tymax = (mx - pos.y) / dir.y;
	if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
	if (tmin > tymax || tymin > tmax) { *tmin_o = FMAX; *tmax_o = FMAX; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (mn - pos.z) / dir.z, tzmax = (mx - pos.z) / dir.z;
	if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
	if (tmin > tzmax || tzmin > tmax) { *tmin_o = FMAX; *tmax_o = FMAX; return; }
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	*tmin_o = tmin; *tmax_o = tmax;
}

// ---------------------------------------------------------------------------------------------
// Occupancy grid
// ---------------------------------------------------------------------------------------------
// generate_grid_samples_nerf_nonuniform (testbed_nerf.cu:585-614)
// `hist` (optional): the number of samples per block of 2^key_shift Morton-consecutive cells, for k_grid_samples_place below.
__global__ void k_grid_samples(const uint32_t n_elements, Pcg32 rng, const uint32_t step, const SceneAabb A, const float* __restrict__ grid_in,
                               float* __restrict__ pos_out, uint32_t* __restrict__ idx_out, const float thresh, uint32_t* __restrict__ hist, const uint32_t key_shift) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint32_t n_cascades = A.max_cascade + 1;
	rng.advance((int64_t)i * 4);
	const uint32_t level = (uint32_t)(rng.next_float() * n_cascades) % n_cascades;
	uint32_t idx = 0;
	for (uint32_t j = 0; j < 10; ++j) {
		idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % GRID_CELLS;
		idx += level * GRID_CELLS;
		if (grid_in[idx] > thresh) break;
	}
	const uint32_t pos_idx = idx % GRID_CELLS;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
	const float sc = scalbnf(1.0f, (int)level);
	const Vec3 pos = {(((float)x + rx) / GRIDSIZE - 0.5f) * sc + 0.5f, (((float)y + ry) / GRIDSIZE - 0.5f) * sc + 0.5f, (((float)z + rz) / GRIDSIZE - 0.5f) * sc + 0.5f};
	const Vec3 w = warp_position(A, pos);
	pos_out[(size_t)i * 3 + 0] = w.x; pos_out[(size_t)i * 3 + 1] = w.y; pos_out[(size_t)i * 3 + 2] = w.z;
	idx_out[i] = idx;
	if (hist) atomicAdd(hist + (idx >> key_shift), 1u);
}

// The samples of an occupancy update in cell order (Morton blocks of 2^key_shift cells; any order inside a block). The network is evaluated
// on a SET of points and splatted with atomicMax (testbed_nerf.cu:616-635), so the order of the samples is free: the reference's order
// (a multiplicative congruence of the thread index) sends consecutive lanes to unrelated cells, cell order lets the gathers of the coarse
// and middle levels share cache lines (k_point_query_chained 0.50 -> 0.34 ms). `cursor` = exclusive prefix sums of k_grid_samples'
// histogram, consumed here. Generated a whole update interval ahead on a side stream: the samples depend on the PREVIOUS update's grid and
// on the RNG only, not on the weights.
__global__ void k_grid_samples_place(const uint32_t n, const float* __restrict__ pos_in, const uint32_t* __restrict__ idx_in, uint32_t* __restrict__ cursor, const uint32_t key_shift,
                                     float* __restrict__ pos_out, uint32_t* __restrict__ idx_out) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const uint32_t idx = idx_in[i];
	const uint32_t p = atomicAdd(cursor + (idx >> key_shift), 1u);
	pos_out[(size_t)p * 3 + 0] = pos_in[(size_t)i * 3 + 0]; pos_out[(size_t)p * 3 + 1] = pos_in[(size_t)i * 3 + 1]; pos_out[(size_t)p * 3 + 2] = pos_in[(size_t)i * 3 + 2];
	idx_out[p] = idx;
}

// Data parallel: this rank's share [out[0], out[1]) of the cell-ordered samples. The order INSIDE a block of cells is whatever k_grid_samples_place's atomics
// made of it on this rank, so a share must begin and end at block boundaries, where the membership does not depend on the order: boundary r = the first block
// end at or beyond n r / W (binary search over `ends`, the cursor array as k_grid_samples_place leaves it: ends[k] = number of samples in blocks 0 .. k).
__global__ void k_shard_range(const uint32_t* __restrict__ ends, const uint32_t n_keys, const uint32_t n, const uint32_t world, const uint32_t rank, uint32_t* __restrict__ out) {
	if (threadIdx.x >= 2) return;
	const uint32_t b = rank + threadIdx.x; // boundary index 0 .. world
	uint32_t v;
	if (b == 0) v = 0;
	else if (b >= world) v = n;
	else {
		const uint32_t target = (uint32_t)((uint64_t)n * b / world);
		uint32_t lo = 0, hi = n_keys - 1; // smallest k with ends[k] >= target
		while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (ends[mid] >= target) hi = mid; else lo = mid + 1; }
		v = ends[lo];
	}
	out[threadIdx.x] = v;
}

// ema_grid_samples_nerf (testbed_nerf.cu:655-685)
__global__ void k_ema_grid(const uint32_t n_elements, const float decay, float* __restrict__ grid_out, const float* __restrict__ grid_in) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const float importance = grid_in[i];
	const float prev_val = grid_out[i];
	grid_out[i] = (prev_val < 0.f) ? prev_val : fmaxf(prev_val * decay, importance);
}

// k_ema_grid + k_mean_partial in one launch (round 4: an occupancy update is a chain of small dependent launches on the critical path of every 16th
// step, ~5 us each). Block b owns cells [2048 b, 2048 (b + 1)): the update of k_ema_grid, and -- for the blocks of the first mip -- k_mean_partial's
// sum of exactly those cells in exactly its order (thread t: cells t, t + 256, ...; the same tree), so `partial` holds the same 1024 doubles.
__global__ __launch_bounds__(256) void k_ema_mean(const uint32_t n_elements, const float decay, float* __restrict__ grid_out, const float* __restrict__ grid_in, double* __restrict__ partial) {
	__shared__ double sh[256];
	const uint32_t base = blockIdx.x * 2048u;
	static_assert(GRID_CELLS / 1024u == 2048u, "k_mean_partial's blocks");
	const bool first_mip = base < GRID_CELLS;
	double acc = 0.0;
	for (uint32_t i = threadIdx.x; i < 2048u; i += 256u) {
		const uint32_t idx = base + i;
		if (idx >= n_elements) break;
		const float importance = grid_in[idx];
		const float prev_val = grid_out[idx];
		const float v = (prev_val < 0.f) ? prev_val : fmaxf(prev_val * decay, importance);
		grid_out[idx] = v;
		if (first_mip) acc += (double)(fmaxf(v, 0.f) / (float)GRID_CELLS);
	}
	if (!first_mip) return; // (uniform over the block)
	sh[threadIdx.x] = acc;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
		__syncthreads();
	}
	if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// mean of max(v,0)/n over the first mip (testbed_nerf.cu:3509), summed in fp64 in a fixed order (2 kernels).
__global__ __launch_bounds__(256) void k_mean_partial(const float* __restrict__ grid, double* __restrict__ partial) {
	__shared__ double sh[256];
	const uint32_t per_block = GRID_CELLS / gridDim.x;
	const uint32_t base = blockIdx.x * per_block;
	double acc = 0.0;
	for (uint32_t i = threadIdx.x; i < per_block; i += 256) acc += (double)(fmaxf(grid[base + i], 0.f) / (float)GRID_CELLS);
	sh[threadIdx.x] = acc;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
		__syncthreads();
	}
	if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
// one wavefront: strided partial sums, then a shuffle tree (a fixed order; fp64, so the float result does not depend on it); result in lane 0
__device__ __forceinline__ double mean_final_wave(const double* __restrict__ partial, const uint32_t n, const uint32_t lane) {
	double s = 0.0;
	for (uint32_t i = lane; i < n; i += 64) s += partial[i];
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
	return s;
}
__global__ void k_mean_final(const double* __restrict__ partial, const uint32_t n, float* __restrict__ mean_out) {
	const double s = mean_final_wave(partial, n, threadIdx.x);
	if (threadIdx.x == 0) *mean_out = (float)s;
}
// Occupancy of cascade 0 in a form a workgroup keeps in LDS (the march's loop is one occupancy test per visited cell; as a
// dependent global load it made the thread-per-ray kernel latency-bound: 129 such loads per wavefront, 46 % of its cycles waiting
// alone, and 319 instead of 185 us beside the backward pass):
//   coarse[1024]  one bit per 4x4x4 block of cells, blocks in LINEAR order (x fastest): three shifts of the cell coordinates, no Morton code
//   rank[1024]    number of set bits in front of each coarse word
//   blocks[n]     the 64 cell bits of every non-empty block (the cells are in Morton order, so a block is 8 consecutive bytes of
//                 the bitfield and a cell's bit is the low 6 bits of its Morton index), in coarse-bit order
// 8 KB + 8 B per non-empty block (a surface: 2-3 k blocks). A clear coarse bit answers "not occupied"; a set one reads the block's
// bits from LDS, or -- for blocks beyond the launch's LDS budget -- the bitfield itself. The decisions, and with them the sample
// set, are the reference's bit for bit.
constexpr uint32_t COARSE_WORDS = GRID_CELLS / 64 / 32;
constexpr uint32_t COARSE_MAX_BLOCKS = 4096; // LDS budget of a march workgroup: 8 KB + 32 KB
// Round 6: behind the image, COARSE_WORDS more words: the coarse bits DILATED by one block in every direction (a block's bit is set if it or any of its 26 neighbours is non-empty).
// A point within 1/32 of a non-empty block (in every coordinate) reads a set bit there: k_march_count_skip samples a ray every 1/32 of its length against these words to find,
// conservatively, the stretches of the ray that can hold samples at all.
constexpr uint32_t COARSE_DIL_OFF = 2 * COARSE_WORDS + 2 * COARSE_MAX_BLOCKS;
// ... and 8 words: the bounding box of the non-empty blocks in block coordinates {x0 y0 z0 x1 y1 z1} (inclusive; x0 > x1: the grid is empty). A ray is over once it has left that
// box (dilated by one block): no position behind it can be occupied. The thread-per-ray march of the large batches stops there instead of walking on to the scene box's exit.
constexpr uint32_t COARSE_BBOX_OFF = COARSE_DIL_OFF + COARSE_WORDS;
constexpr uint32_t COARSE_BUF_WORDS = COARSE_BBOX_OFF + 8;
template <uint32_t NW = 16>
__device__ __forceinline__ uint32_t block_exclusive_scan(const uint32_t mine, const uint32_t lane, const uint32_t wave, uint32_t* __restrict__ wsum, uint32_t& total);
// One workgroup of 1024 threads (= COARSE_WORDS): out = coarse | rank | blocks (uint2 each) ; *n_blocks = number of non-empty blocks.
__device__ __forceinline__ void coarse_bitfield_body(const uint8_t* __restrict__ bitfield, uint32_t* __restrict__ out, uint32_t* __restrict__ n_blocks, uint32_t* __restrict__ n_blocks_host, uint32_t* __restrict__ wsum) {
	const uint32_t w = threadIdx.x, lane = w & 63u, wave = w >> 6;
	uint32_t bits = 0;
	for (uint32_t k = 0; k < 32; ++k) {
		const uint32_t b = w * 32 + k; // linear block id: x | y << 5 | z << 10
		const uint2 v = reinterpret_cast<const uint2*>(bitfield)[morton3D(b & 31u, (b >> 5) & 31u, b >> 10)];
		if (v.x | v.y) bits |= 1u << k;
	}
	uint32_t total;
	const uint32_t before = block_exclusive_scan(__popc(bits), lane, wave, wsum, total);
	out[w] = bits;
	out[COARSE_WORDS + w] = before;
	if (w < 8) out[COARSE_BBOX_OFF + w] = w < 3 ? 31u : 0u;
	__syncthreads(); // (every word of `out[0 .. COARSE_WORDS)` is written: a word is one x-row of blocks, w = y | z << 5)
	if (bits) {
		atomicMin(out + COARSE_BBOX_OFF + 0, (uint32_t)__builtin_ctz(bits)); atomicMax(out + COARSE_BBOX_OFF + 3, 31u - (uint32_t)__builtin_clz(bits));
		atomicMin(out + COARSE_BBOX_OFF + 1, w & 31u); atomicMax(out + COARSE_BBOX_OFF + 4, w & 31u);
		atomicMin(out + COARSE_BBOX_OFF + 2, w >> 5); atomicMax(out + COARSE_BBOX_OFF + 5, w >> 5);
	}
	{
		const int y = (int)(w & 31u), z = (int)(w >> 5);
		uint32_t acc = 0;
		for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
			const int y2 = y + dy, z2 = z + dz;
			if (y2 < 0 || y2 > 31 || z2 < 0 || z2 > 31) continue;
			const uint32_t v = out[(uint32_t)y2 | ((uint32_t)z2 << 5)];
			acc |= v | (v << 1) | (v >> 1);
		}
		out[COARSE_DIL_OFF + w] = acc;
	}
	uint2* blocks = reinterpret_cast<uint2*>(out + 2 * COARSE_WORDS);
	uint32_t r = before;
	for (uint32_t k = 0; k < 32; ++k) {
		if (!((bits >> k) & 1u)) continue;
		const uint32_t b = w * 32 + k;
		if (r < COARSE_MAX_BLOCKS) blocks[r] = reinterpret_cast<const uint2*>(bitfield)[morton3D(b & 31u, (b >> 5) & 31u, b >> 10)];
		++r;
	}
	if (w == 0) { *n_blocks = total; if (n_blocks_host) *n_blocks_host = total; } // the host copy sizes the LDS of later march launches (any size is exact)
}
__global__ __launch_bounds__(1024) void k_coarse_bitfield(const uint8_t* __restrict__ bitfield, uint32_t* __restrict__ out, uint32_t* __restrict__ n_blocks, uint32_t* __restrict__ n_blocks_host) {
	__shared__ uint32_t wsum[16];
	coarse_bitfield_body(bitfield, out, n_blocks, n_blocks_host, wsum);
}
// the first 2 * COARSE_WORDS + 2 * n_blocks_lds words of k_coarse_bitfield's output
__device__ __forceinline__ void load_coarse(uint32_t* __restrict__ lds, const uint32_t* __restrict__ g, const uint32_t n_blocks_lds, const uint32_t tid, const uint32_t n_threads) {
	const uint32_t n_words = 2 * COARSE_WORDS + 2 * n_blocks_lds; // a multiple of 2; the source is 16-byte aligned
	for (uint32_t q = tid * 2; q < n_words; q += n_threads * 2) *reinterpret_cast<uint2*>(lds + q) = *reinterpret_cast<const uint2*>(g + q);
}
// density_grid_occupied_at for mip 0 (cascaded_grid_idx_at's arithmetic with mip_scale = 1) from the LDS form
__device__ __forceinline__ bool occupied_mip0(Vec3 pos, const uint8_t* __restrict__ bitfield, const uint32_t* __restrict__ lds, const uint32_t n_blocks_lds) {
	pos = pos - v3(0.5f, 0.5f, 0.5f);
	pos = 1.0f * pos;
	pos = pos + v3(0.5f, 0.5f, 0.5f);
	const int ix = (int)(pos.x * GRIDSIZE), iy = (int)(pos.y * GRIDSIZE), iz = (int)(pos.z * GRIDSIZE);
	const uint32_t x = (uint32_t)min(max(ix, 0), (int)GRIDSIZE - 1), y = (uint32_t)min(max(iy, 0), (int)GRIDSIZE - 1), z = (uint32_t)min(max(iz, 0), (int)GRIDSIZE - 1);
	const uint32_t block = (x >> 2) | ((y >> 2) << 5) | ((z >> 2) << 10);
	const uint32_t word = lds[block >> 5], bit = block & 31u;
	if (!((word >> bit) & 1u)) return false;
	const uint32_t r = lds[COARSE_WORDS + (block >> 5)] + __popc(word & ((1u << bit) - 1u));
	if (r < n_blocks_lds) {
		const uint2 cells = reinterpret_cast<const uint2*>(lds + 2 * COARSE_WORDS)[r];
		const uint32_t c = (x & 1u) | ((y & 1u) << 1) | ((z & 1u) << 2) | ((x & 2u) << 2) | ((y & 2u) << 3) | ((z & 2u) << 4); // low 6 bits of morton3D(x, y, z)
		return (((c & 32u) ? cells.y : cells.x) >> (c & 31u)) & 1u;
	}
	const uint32_t idx = morton3D(x, y, z);
	return bitfield[idx >> 3] & (1 << (idx & 7u));
}

// grid_to_bitfield (testbed_nerf.cu:693-717)
__global__ void k_grid_to_bitfield(const uint32_t n_elements, const uint32_t n_nonzero_elements, const float* __restrict__ grid, uint8_t* __restrict__ bitfield, const float* __restrict__ mean_density_ptr) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	if (i >= n_nonzero_elements) { bitfield[i] = 0; return; }
	uint8_t bits = 0;
	const float thresh = fminf(MIN_OPTICAL_THICKNESS, *mean_density_ptr);
#pragma unroll
	for (uint8_t j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? ((uint8_t)1 << j) : 0;
	bitfield[i] = bits;
}
// Single-cascade scenes (aabb_scale 1: every RNb scene): k_mean_final + k_grid_to_bitfield + the first k_bitfield_max_pool in one launch. Every
// block forms the mean from the 1024 partial sums the way k_mean_final does (same order, same bits), thread i makes byte i of level 0, and the
// 8 consecutive threads of a 64-cell block make its pooled byte of level 1 from one ballot. Levels >= 1 have no bits of their own here and are
// written with plain stores where the pooling reaches (level 1: its central 32^3 bytes, from here; levels 2.. : k_pool_tail_coarse); the rest
// of those levels is zero from the creation of the context on and stays zero -- a caller-written bitfield takes the general kernels once, which
// zero-fill (update_bitfield, rnb_ctx::bitfield_foreign). Same bytes as the three kernels.
__global__ __launch_bounds__(256) void k_bitfield_sc(const float* __restrict__ grid, uint8_t* __restrict__ bitfield, const double* __restrict__ partial, float* __restrict__ mean_out) {
	__shared__ float sh_mean;
	if (threadIdx.x < 64) {
		const double s = mean_final_wave(partial, 1024u, threadIdx.x);
		if (threadIdx.x == 0) { sh_mean = (float)s; if (blockIdx.x == 0) *mean_out = (float)s; }
	}
	__syncthreads();
	const float thresh = fminf(MIN_OPTICAL_THICKNESS, sh_mean);
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; // < GRID_CELLS / 8 (the launch covers level 0 exactly)
	const f4 g0 = reinterpret_cast<const f4*>(grid)[(size_t)i * 2 + 0], g1 = reinterpret_cast<const f4*>(grid)[(size_t)i * 2 + 1];
	uint8_t bits = 0;
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) { bits |= g0[j] > thresh ? (uint8_t)(1u << j) : (uint8_t)0; bits |= g1[j] > thresh ? (uint8_t)(16u << j) : (uint8_t)0; }
	bitfield[i] = bits;
	const unsigned long long nz = __ballot(bits > 0);
	const uint32_t lane = threadIdx.x & 63u;
	if ((lane & 7u) == 0u) {
		const uint8_t pooled = (uint8_t)((nz >> lane) & 0xffull); // bit j = "byte 8 q + j of level 0 is non-zero" (k_bitfield_max_pool)
		const uint32_t q = i >> 3;
		const uint32_t x = morton3D_invert(q >> 0) + GRIDSIZE / 8, y = morton3D_invert(q >> 1) + GRIDSIZE / 8, z = morton3D_invert(q >> 2) + GRIDSIZE / 8;
		bitfield[GRID_CELLS / 8 + morton3D(x, y, z)] = pooled;
	}
}

// bitfield_max_pool (testbed_nerf.cu:719-740)
__global__ void k_bitfield_max_pool(const uint32_t n_elements, const uint8_t* __restrict__ prev_level, uint8_t* __restrict__ next_level) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	uint8_t bits = 0;
#pragma unroll
	for (uint8_t j = 0; j < 8; ++j) bits |= prev_level[(size_t)i * 8 + j] > 0 ? ((uint8_t)1 << j) : 0;
	const uint32_t x = morton3D_invert(i >> 0) + GRIDSIZE / 8;
	const uint32_t y = morton3D_invert(i >> 1) + GRIDSIZE / 8;
	const uint32_t z = morton3D_invert(i >> 2) + GRIDSIZE / 8;
	next_level[morton3D(x, y, z)] |= bits;
}
// The same for levels first_level .. N_CASCADES-1 in ONE launch of one workgroup, when level first_level-1 has no bits of its own (it is above
// the scene's last cascade): such a level is zero outside what was pooled into its central 64^3 cells, its pooled image is zero outside the
// central 32^3 cells of the next level, and so on. In bytes (= 2x2x2-cell blocks, a 64^3 lattice per level) the support of level
// first_level-1 is [16, 48)^3: its first pooled image (16^3 outputs, one 8-byte word each) is kept in LDS and the levels follow each other there -- 4096, 512, 64, 8, 8 ... outputs
// instead of 32768 per level and no trip through memory between them -- so the seven launches of a single-cascade scene (~4.7 us each,
// launch-bound, on the critical path of every occupancy update) become two. What these loops do not write was zero-filled by
// k_grid_to_bitfield and stays zero, exactly as `|= 0` leaves it in the full kernel; what they write had no bits of its own to keep.
__device__ __forceinline__ void max_pool_tail_body(const uint32_t first_level, uint8_t* __restrict__ bitfield, uint8_t (*buf)[16 * 16 * 16]) {
	uint32_t lo = GRIDSIZE / 8, hi = GRIDSIZE / 8 * 3; // support of the level being read, byte coordinates
	uint32_t cur = 0;
	for (uint32_t level = first_level; level < N_CASCADES; ++level) {
		uint8_t* next_level = bitfield + (size_t)(GRID_CELLS / 8) * level;
		const uint32_t wi = hi - lo; // width of the support held in buf[cur] (levels after the first)
		const uint32_t o_lo = lo / 2, o_hi = (hi + 1) / 2, w = o_hi - o_lo; // outputs whose 2x2x2 input bytes touch the support
		for (uint32_t q = threadIdx.x; q < w * w * w; q += blockDim.x) {
			const uint32_t ox = o_lo + q % w, oy = o_lo + (q / w) % w, oz = o_lo + q / (w * w);
			uint8_t bits = 0;
			if (level == first_level) { // from memory: the 8 input bytes i * 8 + j of the full kernel are one aligned word (16^3 outputs, 4 per thread)
				const uint64_t v = reinterpret_cast<const uint64_t*>(bitfield + (size_t)(GRID_CELLS / 8) * (level - 1))[morton3D(ox, oy, oz)];
#pragma unroll
				for (uint32_t j = 0; j < 8; ++j) bits |= ((v >> (8 * j)) & 0xffull) ? (uint8_t)(1u << j) : (uint8_t)0;
			} else {
#pragma unroll
				for (uint32_t j = 0; j < 8; ++j) { // input byte i * 8 + j of the full kernel = Morton neighbour j of (2 ox, 2 oy, 2 oz)
					const uint32_t x = 2 * ox + (j & 1u), y = 2 * oy + ((j >> 1) & 1u), z = 2 * oz + (j >> 2);
					const bool in = x >= lo && x < hi && y >= lo && y < hi && z >= lo && z < hi;
					if (in && buf[cur][(x - lo) + wi * ((y - lo) + wi * (z - lo))] > 0) bits |= (uint8_t)(1u << j);
				}
			}
			next_level[morton3D(ox + GRIDSIZE / 8, oy + GRIDSIZE / 8, oz + GRIDSIZE / 8)] = bits;
			buf[cur ^ 1u][q] = bits; // = index (ox - o_lo) + w ((oy - o_lo) + w (oz - o_lo)) of the next support
		}
		lo = o_lo + GRIDSIZE / 8; hi = o_hi + GRIDSIZE / 8;
		cur ^= 1u;
		__syncthreads();
	}
}
__global__ __launch_bounds__(1024) void k_bitfield_max_pool_tail(const uint32_t first_level, uint8_t* __restrict__ bitfield) {
	__shared__ uint8_t buf[2][16 * 16 * 16];
	max_pool_tail_body(first_level, bitfield, buf);
}
// The upper pool levels and the march kernels' LDS form of level 0 in one launch of two workgroups (each is one workgroup of 1024 threads, and both
// only read what the launch in front of them wrote: level first_level - 1 resp. level 0).
__global__ __launch_bounds__(1024) void k_pool_tail_coarse(const uint32_t first_level, uint8_t* __restrict__ bitfield, uint32_t* __restrict__ coarse_out, uint32_t* __restrict__ n_blocks, uint32_t* __restrict__ n_blocks_host) {
	__shared__ uint8_t buf[2][16 * 16 * 16];
	__shared__ uint32_t wsum[16];
	if (blockIdx.x == 0) coarse_bitfield_body(bitfield, coarse_out, n_blocks, n_blocks_host, wsum);
	else if (first_level < N_CASCADES) max_pool_tail_body(first_level, bitfield, buf);
}

// ---------------------------------------------------------------------------------------------
// Dataset access (common_device.cuh:31-61, 621-700)
// ---------------------------------------------------------------------------------------------
struct ViewDev {
	uint32_t width, height;
	float focal[2], principal[2];
	float xform[12];
	const uint16_t* normal;
	const uint16_t* albedo;
};

__device__ __forceinline__ float srgb_to_linear(float srgb) {
	if (srgb <= 0.04045f) return srgb / 12.92f;
	return powf((srgb + 0.055f) / 1.055f, 2.4f);
}
__device__ __forceinline__ float linear_to_srgb(float linear) {
	if (linear < 0.0031308f) return 12.92f * linear;
	return 1.055f * powf(linear, 0.41666f) - 0.055f;
}
__device__ __forceinline__ void read_rgba(const float xy[2], const ViewDev& m, const uint16_t* __restrict__ pixels, float rgba[4]) {
	const int x = (int)(xy[0] * (float)m.width), y = (int)(xy[1] * (float)m.height);
	const int px = max(min(x, (int)m.width - 1), 0), py = max(min(y, (int)m.height - 1), 0);
	const uint2 raw = *reinterpret_cast<const uint2*>(pixels + ((size_t)px + (size_t)py * m.width) * 4);
	if (raw.x == 0x00FF00FFu && raw.y == 0u) { rgba[0] = rgba[1] = rgba[2] = rgba[3] = -1.f; return; }
	const float v0 = (float)(raw.x & 0xffffu), v1 = (float)(raw.x >> 16), v2 = (float)(raw.y & 0xffffu), v3 = (float)(raw.y >> 16);
	const float alpha = v3 * (1.0f / 65535.0f);
	rgba[0] = srgb_to_linear(v0 * (1.0f / 65535.0f)) * alpha;
	rgba[1] = srgb_to_linear(v1 * (1.0f / 65535.0f)) * alpha;
	rgba[2] = srgb_to_linear(v2 * (1.0f / 65535.0f)) * alpha;
	rgba[3] = alpha;
}
// red channel test of testbed_nerf.cu:1264 without the pow(): linear*alpha <= 0  <=>  v0 == 0 || alpha == 0 (or the -1 sentinel)
__device__ __forceinline__ bool red_is_nonpositive(const float xy[2], const ViewDev& m, const uint16_t* __restrict__ pixels) {
	const int x = (int)(xy[0] * (float)m.width), y = (int)(xy[1] * (float)m.height);
	const int px = max(min(x, (int)m.width - 1), 0), py = max(min(y, (int)m.height - 1), 0);
	const uint2 raw = *reinterpret_cast<const uint2*>(pixels + ((size_t)px + (size_t)py * m.width) * 4);
	if (raw.x == 0x00FF00FFu && raw.y == 0u) return true;
	return (raw.x & 0xffffu) == 0u || (raw.y >> 16) == 0u;
}
__device__ __forceinline__ void random_image_pos(Pcg32& rng, const uint32_t w, const uint32_t h, const bool snap, float xy[2]) { // testbed_nerf.cu:1171-1192
	xy[0] = rng.next_float(); xy[1] = rng.next_float();
	if (snap) {
		const float res[2] = {(float)w, (float)h};
		const float lim[2] = {(float)((int)w - 1), (float)((int)h - 1)};
#pragma unroll
		for (int a = 0; a < 2; ++a) {
			float p = xy[a] * res[a];
			p = fmaxf(p, 0.0f);
			p = fminf(p, lim[a]);
			xy[a] = (p + 0.5f) / res[a];
		}
	}
}
// The ray of an image position (testbed_nerf.cu:1279-1305, no lens distortion, no rolling shutter): origin = the camera matrix's last column, direction = its 3x3 block
// applied to ((x - cx) w / fx, (y - cy) h / fy, 1) -- Eigen's fixed-size product: a row's three terms as x0 + (x1 + x2), esum3 -- and normalised by a division by the norm.
__device__ __forceinline__ void camera_ray(const ViewDev& m, const float xy[2], Vec3& o, Vec3& du, Vec3& dir) {
	o = v3(m.xform[3], m.xform[7], m.xform[11]);
	const Vec3 dcam = {
		(xy[0] - m.principal[0]) * (float)m.width / m.focal[0],
		(xy[1] - m.principal[1]) * (float)m.height / m.focal[1],
		1.0f,
	};
	du = v3(esum3(m.xform[0] * dcam.x, m.xform[1] * dcam.y, m.xform[2] * dcam.z),
	        esum3(m.xform[4] * dcam.x, m.xform[5] * dcam.y, m.xform[6] * dcam.z),
	        esum3(m.xform[8] * dcam.x, m.xform[9] * dcam.y, m.xform[10] * dcam.z));
	dir = normalized(du);
}
__device__ __forceinline__ uint32_t image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_rays_total, uint32_t n_images) { // testbed_nerf.cu:1194-1214
	return (((base_idx + n_rays_total) * n_images) / n_rays) % n_images;
}

// ---------------------------------------------------------------------------------------------
// K6: ray generation + DDA march (testbed_nerf.cu:1216-1387)
// ---------------------------------------------------------------------------------------------
struct LossFlags {
	uint32_t apply_L2, apply_rgbplus, apply_no_albedo, apply_light_opti, apply_relu, apply_bce, snap;
	float mask_loss_weight, ek_loss_weight;
};

constexpr int RAY_CONST_FLOATS = 12; // rgbtarget[4], light[3], mask_certainty, mask_gt, pad[3]

struct MarchArgs {
	uint32_t n_rays;        // this rank's rays
	uint32_t n_rays_global; // n_rays * world_size
	uint32_t ray_offset;    // rank * n_rays
	uint32_t n_rays_total;
	uint32_t max_samples;
	uint32_t n_images;
	uint32_t snap;
	Pcg32 rng;
	SceneAabb A;
	const ViewDev* views;
	const uint8_t* bitfield;
	const uint32_t* coarse; // k_coarse_bitfield of cascade 0
	uint32_t n_blocks_lds;  // how many of its blocks the launch's LDS holds
	uint32_t lattice_ok;    // the march lattice t_{k+1} = fl(t_k + MIN_CONE_STEPSIZE) is linear inside every binade of [0.25, 8) (no rounding ties): closed form allowed
	// per-ray scratch
	float* setup;       // [n_rays][8]: o(3) dir(3) startt alive
	float* ray_t;       // [n_rays][RNB_MAX_STEPS]: t of every sample found by the counting pass
	float* d_unnorm;    // [n_rays][3]
	uint32_t* steps;    // [n_rays]
	uint32_t* base;     // [n_rays]
	uint32_t* slot;     // [n_rays] (0xffffffff = dropped)
	uint32_t* base1;    // [n_rays] offset of the ray's first-round samples in idx1 (two-round network evaluation, see k_loss_pass1)
	uint32_t* idx1;     // sample slots of the first round: the first min(steps, k1) samples of every kept ray
	uint32_t k1;        // 0 = single round
	uint32_t part;      // k_march_write: 0 = everything; 1 = what the first network evaluation reads (idx1, the coordinates of the heads); 2 = the rest
	// per-ray loss constants (target colour, light, masks) depend on the ray and the dataset only: worked out here, off the
	// step's critical path, for the loss passes to pick up (k_march_write -> k_loss_pass1)
	LossFlags F;
	float light_dirs[9];
	float* ray_const;   // [n_rays kept][RAY_CONST_FLOATS]
	// outputs
	uint32_t* ray_indices; float* rays; uint32_t* numsteps; float* coords; uint32_t* counters;
	uint32_t prio;             // RNB_MARCH_PRIO (A/B): s_setprio of the march kernels' wavefronts
	uint32_t use_bbox;         // RNB_MARCH_BBOX (round 6): 0 off; 1 (default): the thread-per-ray march ends where the ray leaves the bounding box of the non-empty blocks; 2: + one jump to that box's entry (k_march_count_bbox)
	unsigned long long* stats; // RNB_MARCH_STATS=1 (measurement aid, k_march_count_skip): [0] wavefronts, [1] loop iterations, [2] rays, [3] rays that skipped, [4] start-overs, [5] rounds spent looking for a re-entry cell, [6] rays ended early
};

// SC: one cascade and no cone (aabb_scale 1, every RNb scene): dt is the constant step and every position inside the box is in
// mip 0 (mip_from_pos clamps to max_cascade = 0, mip_from_dt returns it because dt * 2 * GRIDSIZE < 1), so the per-position
// frexp / scalbn / variable-resolution arithmetic folds into constants. Same values, fewer dependent instructions per voxel.
template <bool SC, typename F>
__device__ __forceinline__ uint32_t march(const SceneAabb& A, const uint8_t* __restrict__ bitfield, const uint32_t* __restrict__ coarse_lds, const uint32_t n_blocks_lds, const Vec3& o, const Vec3& dir, const float startt, const uint32_t max_steps, F&& emit, const float t_stop = 3.0e38f) {
	const Vec3 idir = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	uint32_t j = 0;
	float t = startt;
	Vec3 pos;
	while (aabb_contains(A, pos = o + t * dir) && j < max_steps && t <= t_stop) { // (t_stop: behind it no position can be occupied -- the reference walks on and emits nothing)
		const float dt = SC ? MIN_CONE_STEPSIZE : calc_dt(t, A.cone_angle);
		const uint32_t mip = SC ? 0u : (uint32_t)mip_from_dt(dt, pos);
		if (SC ? occupied_mip0(pos, bitfield, coarse_lds, n_blocks_lds) : density_grid_occupied_at(pos, bitfield, mip)) {
			emit(j, pos, dt, t);
			++j;
			t += dt;
		} else if (SC) {
			const float t_target = t + distance_to_next_voxel(pos, dir, idir, GRIDSIZE);
			do { t += MIN_CONE_STEPSIZE; } while (t < t_target);
		} else {
			const uint32_t res = GRIDSIZE >> mip;
			t = advance_to_next_voxel(t, A.cone_angle, pos, dir, idir, res);
		}
	}
	return j;
}

template <bool SC>
__global__ __launch_bounds__(512) void k_march_count(const MarchArgs a) { // (launched with 128 ... 512 threads: RNB_MARCH_NARROW_WGS)
	if (a.prio == 1u) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2u) __builtin_amdgcn_s_setprio(2); else if (a.prio >= 3u) __builtin_amdgcn_s_setprio(3);
	extern __shared__ __attribute__((aligned(16))) uint32_t coarse_lds[];
	if (SC) { load_coarse(coarse_lds, a.coarse, a.n_blocks_lds, threadIdx.x, blockDim.x); __syncthreads(); }
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= a.n_rays) return;
	const uint32_t gi = a.ray_offset + i;
	const uint32_t img = image_idx(gi, a.n_rays_global, a.n_rays_total, a.n_images);
	const ViewDev m = a.views[img];
	Pcg32 rng = a.rng;
	rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
	float xy[2];
	random_image_pos(rng, m.width, m.height, a.snap != 0, xy);
	uint32_t steps = 0;
	float alive = 0.f;
	Vec3 o = {0, 0, 0}, dir = {0, 0, 1}, du = {0, 0, 1};
	float startt = 0.f;
	bool dead = false;
	if (red_is_nonpositive(xy, m, m.normal)) {
		if (rng.next_float() >= 0.9) dead = true; // testbed_nerf.cu:1264, short-circuit draw
	}
	if (!dead) {
		(void)rng.next_float(); // motionblur_time, testbed_nerf.cu:1270
		camera_ray(m, xy, o, du, dir);
		float tmin, tmax;
		ray_intersect(a.A, o, dir, &tmin, &tmax);
		tmin = fmaxf(tmin, 0.0f);
		startt = tmin;
		startt += calc_dt(startt, a.A.cone_angle) * rng.next_float();
		alive = 1.f;
		float* tt = a.ray_t + (size_t)i * RNB_MAX_STEPS;
		float t_stop = 3.0e38f;
		bool can_hit = true;
		if (SC && a.use_bbox) { // round 6: where the ray leaves the (dilated) bounding box of the non-empty blocks it is over; a ray that misses the box has no sample
			const uint32_t* bb = a.coarse + COARSE_BBOX_OFF;
			const float lo[3] = {((float)bb[0] - 1.0f) * (1.0f / 32.0f), ((float)bb[1] - 1.0f) * (1.0f / 32.0f), ((float)bb[2] - 1.0f) * (1.0f / 32.0f)};
			const float hi[3] = {((float)bb[3] + 2.0f) * (1.0f / 32.0f), ((float)bb[4] + 2.0f) * (1.0f / 32.0f), ((float)bb[5] + 2.0f) * (1.0f / 32.0f)};
			const float oo[3] = {o.x, o.y, o.z}, dd[3] = {dir.x, dir.y, dir.z};
			float t0 = -3.0e38f, t1 = 3.0e38f;
			can_hit = bb[0] <= bb[3];
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				if (fabsf(dd[d]) < 1e-12f) { if (oo[d] < lo[d] || oo[d] > hi[d]) can_hit = false; continue; }
				const float ta = (lo[d] - oo[d]) / dd[d], tb = (hi[d] - oo[d]) / dd[d];
				t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
			}
			if (t0 > t1) can_hit = false;
			t_stop = t1 + 4.0f * MIN_CONE_STEPSIZE; // (the box is already a block = 18 steps wider than the non-empty blocks on every side)
		}
		if (can_hit) steps = march<SC>(a.A, a.bitfield, coarse_lds, a.n_blocks_lds, o, dir, startt, RNB_MAX_STEPS, [&](uint32_t j, const Vec3&, float, float t) { tt[j] = t; }, t_stop);
	}
	float* st = a.setup + (size_t)i * 8;
	st[0] = o.x; st[1] = o.y; st[2] = o.z; st[3] = dir.x; st[4] = dir.y; st[5] = dir.z; st[6] = startt; st[7] = alive;
	a.d_unnorm[(size_t)i * 3 + 0] = du.x; a.d_unnorm[(size_t)i * 3 + 1] = du.y; a.d_unnorm[(size_t)i * 3 + 2] = du.z;
	a.steps[i] = steps;
}

// Same result as k_march_count, 16 lanes per ray. The sequence of march positions t_0 = startt, t_{k+1} = t_k + dt(t_k)
// does not depend on the occupancy (the reference's advance_to_next_voxel steps by the same dt, testbed_nerf.cu:311-323),
// only WHICH positions are visited does. Each round the 16 lanes of a ray test 16 consecutive positions at once (one
// round of dependent bitfield loads instead of 16), exchange the outcomes with ballots, and every lane replays the
// reference's sequential visit order over those outcomes with integer bit operations: occupied -> sample, next
// position; empty -> jump to the first position at or beyond the voxel exit. Visited set and t values are identical.
// MG = lanes per ray (16 or 32): more lanes = fewer dependent rounds per ray, more redundant t-chain work per round.
// WGS = threads per workgroup. (Round 4 measured one WAVEFRONT per ray, <64, true, 1024>, for the march that has the GPU to itself behind an occupancy update:
// 660 us against 184 -- the rounds are not what a ray costs; the sequential replay below is, and every lane of a ray's group executes it.)
// (Round 4 also measured the replay as a fixed, branch-free scan over the round's 16 positions -- every lane reads its group's 16 jump targets back from LDS and
// propagates "visited" through 16 unrolled steps; bit-exact; 166 vs 180 us alone, but 0.642 vs 0.635 ms/step at step 1000: the loop's cost is mostly scalar
// bookkeeping, the scan's is vector instructions, which is what the kernel shares with the scatter beside it. Dropped, profiles/r04_ab_march_scan_replay.txt.)
template <int MG, bool SC, int WGS = 256>
__global__ __launch_bounds__(WGS) void k_march_count_wide(const MarchArgs a) {
	static_assert(MG == 8 || MG == 16 || MG == 32 || MG == 64, "lanes per ray");
	extern __shared__ __attribute__((aligned(16))) uint32_t coarse_lds[];
	if (SC) { load_coarse(coarse_lds, a.coarse, a.n_blocks_lds, threadIdx.x, blockDim.x); __syncthreads(); }
	constexpr uint64_t GM = MG == 64 ? ~0ull : (1ull << (MG & 63)) - 1ull;  // a group's lanes inside a 64-bit ballot
	const uint32_t i = blockIdx.x * (WGS / MG) + (threadIdx.x / MG);
	const int lane = threadIdx.x & 63;
	const int g = lane & (MG - 1);
	const int gb = lane & ~(MG - 1); // first lane of the group inside the wavefront
	const bool ray_exists = i < a.n_rays;
	Vec3 o = {0, 0, 0}, dir = {0, 0, 1}, du = {0, 0, 1}, idir = {0, 0, 1};
	float t_cur = 0.f, startt = 0.f, alive = 0.f;
	bool term = true;
	if (ray_exists) {
		const uint32_t gi = a.ray_offset + i;
		const uint32_t img = image_idx(gi, a.n_rays_global, a.n_rays_total, a.n_images);
		const ViewDev m = a.views[img];
		Pcg32 rng = a.rng;
		rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
		float xy[2];
		random_image_pos(rng, m.width, m.height, a.snap != 0, xy);
		bool dead = false;
		if (red_is_nonpositive(xy, m, m.normal)) {
			if (rng.next_float() >= 0.9) dead = true; // testbed_nerf.cu:1264
		}
		if (!dead) {
			(void)rng.next_float(); // motionblur_time
			camera_ray(m, xy, o, du, dir);
			idir = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
			float tmin, tmax;
			ray_intersect(a.A, o, dir, &tmin, &tmax);
			tmin = fmaxf(tmin, 0.0f);
			startt = tmin;
			startt += calc_dt(startt, a.A.cone_angle) * rng.next_float();
			t_cur = startt;
			alive = 1.f;
			term = false;
		}
	}
	uint32_t j = 0;
	bool have_pending = false;
	float pending_target = 0.f;
	float* tt = a.ray_t + (size_t)(ray_exists ? i : 0) * RNB_MAX_STEPS;
	const float cone = a.A.cone_angle;
	while (__any(!term)) {
		// The next MG positions of the ray. Closed form of the lattice t_{k+1} = fl(t_k + C) (C = the constant step of the single-cascade scenes):
		// while t stays inside one binade [2^(e-1), 2^e) every t_k is a multiple of that binade's ulp, so fl(t_k + C) = t_k + d with
		// d = C rounded to a multiple of the ulp -- the same d for every k (a.lattice_ok: the host has checked that C is not a rounding tie in
		// the binades used, rnb_neus2_hip.hip lattice_is_linear). Then t_{k+m} = t_k + m d EXACTLY (m d and the sum are exact: multiples of the
		// ulp below 2^e), and "the first position at or beyond a target" is a division plus one exact comparison instead of a scan over 16
		// running sums. A round whose 17 positions straddle a binade boundary takes the running sums (below), as does the multi-cascade march.
		// (the running sums of the slow path are formed again wherever they are compared, not kept: as an array they were 65 wave-uniform values of the one-wavefront-per-ray
		// instance, i.e. SGPRs, 211 of which the compiler kept in VGPR lanes -- the kind of spill the side-stream hazard of round 1 was last seen with, rnb_neus2_hip.hip launch_premarch)
		// (one wavefront per ray: rolled loops -- unrolled, the three loops' sums are common subexpressions again and come back as that array)
		constexpr int SLOW_UNROLL = (MG == 64 || SC) ? 1 : MG; // (single-cascade scenes take these loops only where a round straddles a binade boundary; the multi-cascade march takes them every round)
		auto step_from = [&](const float t) { return t + (SC ? MIN_CONE_STEPSIZE : calc_dt(t, cone)); };
		float t_end = t_cur;
		float my_t = t_cur;
		float dlt = 0.f, inv_dlt = 0.f;
		bool fast = false;
		if (SC && a.lattice_ok) {
			int e;
			(void)frexpf(t_cur, &e); // t_cur in [2^(e-1), 2^e)
			const float base = scalbnf(0.5f, e), hi = scalbnf(1.0f, e);
			dlt = (base + MIN_CONE_STEPSIZE) - base;
			const bool ok = term || (t_cur >= 0.25f && t_cur < 8.0f && t_cur + (float)MG * dlt < hi);
			fast = !__any(!ok);
		}
		if (fast) {
			my_t = t_cur + (float)g * dlt;
			t_end = t_cur + (float)MG * dlt;
			inv_dlt = __builtin_amdgcn_rcpf(dlt); // an estimate is enough: the candidate it yields is checked exactly
		} else {
			float run = t_cur; // T_0; T_{m+1} = T_m + dt(T_m)
#pragma unroll SLOW_UNROLL
			for (int m = 0; m < MG; ++m) {
				run = step_from(run);
				if (m + 1 == g) my_t = run;
			}
			t_end = run;
		}
		// my position
		const Vec3 pos = o + my_t * dir;
		const bool inside = !term && aabb_contains(a.A, pos);
		bool occ = false;
		float t_target = 0.f;
		uint32_t nxt = MG; // absolute index of the next visited position if mine is visited and empty
		if (inside) {
			const float dt = SC ? MIN_CONE_STEPSIZE : calc_dt(my_t, cone);
			const uint32_t mip = SC ? 0u : (uint32_t)mip_from_dt(dt, pos);
			occ = SC ? occupied_mip0(pos, a.bitfield, coarse_lds, a.n_blocks_lds) : density_grid_occupied_at(pos, a.bitfield, mip);
			if (!occ) {
				const uint32_t res = GRIDSIZE >> mip;
				t_target = my_t + distance_to_next_voxel(pos, dir, idir, res);
				if (fast) { // smallest m in (g, MG) with t_cur + m d >= t_target, else MG
					int m0 = (int)floorf(fminf((t_target - t_cur) * inv_dlt, 64.f));
					m0 += (t_cur + (float)m0 * dlt < t_target) ? 1 : 0;
					nxt = (uint32_t)min(max(m0, g + 1), MG);
				} else { // the same by the running sums: the first m in (g, MG) with T_m >= t_target
					float run = t_cur;
#pragma unroll SLOW_UNROLL
					for (int m = 1; m < MG; ++m) {
						run = step_from(run);
						if (nxt == (uint32_t)MG && m > g && run >= t_target) nxt = (uint32_t)m;
					}
				}
			}
		}
		const unsigned long long occ_w = __ballot(occ), in_w = __ballot(inside);
		const uint64_t occ16 = ((uint64_t)occ_w >> gb) & GM, in16 = ((uint64_t)in_w >> gb) & GM;
		// replay of the sequential visit order over this round's outcomes
		uint64_t vis = 0;
		const uint32_t j0 = j;
		int pend_src = -1;
		if (!term) {
			int cur = 0;
			if (have_pending) { // continue the reference's do { t += dt } while (t < t_target) across the round boundary
				if (fast) { // smallest m in [0, MG) with t_cur + m d >= pending_target, else MG
					int m0 = (int)floorf(fminf(fmaxf((pending_target - t_cur) * inv_dlt, -2.f), 64.f));
					m0 += (t_cur + (float)m0 * dlt < pending_target) ? 1 : 0;
					cur = min(max(m0, 0), MG);
				} else {
					cur = MG;
					float run = t_cur;
#pragma unroll SLOW_UNROLL
					for (int m = 0; m < MG; ++m) {
						if (cur == MG && run >= pending_target) cur = m;
						run = step_from(run);
					}
				}
				if (cur < MG) have_pending = false;
			}
			while (cur < MG) {
				if (!((in16 >> cur) & 1ull)) { term = true; break; } // left the box (testbed_nerf.cu:1337)
				if ((occ16 >> cur) & 1ull) {
					const uint64_t run_mask = (occ16 & in16) >> cur;
					int run = (MG == 64 && run_mask == ~0ull) ? 64 : __builtin_ctzll(~run_mask); // bits above the group are zero in run_mask: the complement has a set bit unless the group is the whole ballot
					run = min(run, MG - cur);
					const int allowed = min(run, (int)(RNB_MAX_STEPS - j));
					vis |= (allowed >= 64 ? ~0ull : ((1ull << allowed) - 1ull)) << cur;
					j += (uint32_t)allowed;
					cur += allowed;
					if (j >= RNB_MAX_STEPS) { term = true; break; }
				} else {
					// the visited empty lane's own answer, through one cross-lane read (carried in four ballot masks = eight more SGPRs, the
					// kernel spilled masks into VGPR lanes and was not reproducible beside k_fwd_bwd: rnb_neus2_hip.hip, launch_premarch)
					const int nx = (int)__shfl(nxt, gb + cur, 64);
					if (nx >= MG) { have_pending = true; pend_src = cur; }
					cur = nx;
				}
			}
		}
		// the empty lane that jumped beyond the round hands its voxel-exit target to the whole group
		const float tgt = __shfl(t_target, gb + (pend_src < 0 ? 0 : pend_src), 64);
		if (pend_src >= 0) pending_target = tgt;
		if ((vis >> g) & 1ull) tt[j0 + __popcll(vis & ((1ull << g) - 1ull))] = my_t;
		t_cur = t_end;
	}
	if (ray_exists && g == 0) {
		float* st = a.setup + (size_t)i * 8;
		st[0] = o.x; st[1] = o.y; st[2] = o.z; st[3] = dir.x; st[4] = dir.y; st[5] = dir.z; st[6] = startt; st[7] = alive;
		a.d_unnorm[(size_t)i * 3 + 0] = du.x; a.d_unnorm[(size_t)i * 3 + 1] = du.y; a.d_unnorm[(size_t)i * 3 + 2] = du.z;
		a.steps[i] = j;
	}
}

// ---------------------------------------------------------------------------------------------
// Round 6: k_march_count_wide<16, true> that SKIPS the stretches of a ray that cannot hold a sample -- the same sample set and the same t, bit for bit.
//
// What a ray costs above is its ~45 rounds of 16 march positions from the box entry to the box exit, of which 4-5 find occupied cells: the reference's march visits every
// fine voxel on the way (its jump from an empty position lands on the first lattice position at or behind that voxel's exit, computed in fp32 FROM the position it jumped from),
// and a bit-exact replay has to know, at the first occupied cell, which lattice position the chain arrives on. Two facts make the empty stretches skippable all the same:
//  (1) WHERE: a set bit of the DILATED coarse occupancy (COARSE_DIL_OFF) at the 64 points startt + i / 32 of the ray marks, conservatively, every stretch
//      [startt + (i - 1/2) / 32, startt + (i + 1/2) / 32) in which a position can be occupied (a point of the ray inside a non-empty 4^3 block is within 1/64 of a sample point,
//      which then lies in that block or a neighbour). Rounds whose stretch -- and the one behind it -- has no set bit emit nothing; behind the last set bit the ray is over.
//  (2) RE-ENTRY: let A be a cell ALL of whose lattice positions a .. b are in one round, with position b + 1 in another cell, and let every one of a .. b, taken as the visited
//      position, jump to b + 1 (each lane computes exactly that jump for its own position: `nxt`). Then the reference's chain visits b + 1, whatever it did before: its last
//      visited position v < b + 1 is either in A (and jumps to b + 1 by its own vote) or in a cell in front of A, whose exit is at or before position a up to an fp32 rounding
//      of ~1e-6 of a march step -- it lands on a or a + 1, inside A (or on b + 1 itself if A is that one position). So behind a skip the kernel evaluates rounds WITHOUT replaying
//      them until it finds such a cell (cells hold at most 9 positions: almost always the first round), and continues the exact replay from b + 1 with no pending jump.
// A ray that finds no such cell before its next stretch of interest starts over from its box entry without skipping (measured: none in 10^6 rays; the path is tested by forcing it).
// The lattice position behind n skipped steps is exact: inside a binade t + m d is exact (one fma: a multiple of the binade's ulp below the binade's end), across a binade
// boundary the step is the reference's own addition.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lattice_advance(float t, int n) { // t_{k+n} of t_{k+1} = fl(t_k + MIN_CONE_STEPSIZE); t in [0.25, 8), MarchArgs::lattice_ok
	while (n > 0) {
		int e;
		(void)frexpf(t, &e);
		const float base = scalbnf(0.5f, e), hi = scalbnf(1.0f, e);
		const float d = (base + MIN_CONE_STEPSIZE) - base;
		int m = (int)floorf((hi - t) / d); // steps that stay below the binade's end: the largest m with t + m d < hi
		m = max(m - 1, 0);
		while (__builtin_fmaf((float)(m + 1), d, t) < hi) ++m;
		if (n <= m) return __builtin_fmaf((float)n, d, t);
		t = __builtin_fmaf((float)m, d, t);
		n -= m;
		t = t + MIN_CONE_STEPSIZE; // the step across the boundary: the reference's own addition
		n -= 1;
	}
	return t;
}

template <int WGS = 256>
__global__ __launch_bounds__(WGS) void k_march_count_skip(const MarchArgs a) {
	if (a.prio == 1u) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2u) __builtin_amdgcn_s_setprio(2); else if (a.prio >= 3u) __builtin_amdgcn_s_setprio(3);
	constexpr int MG = 16;
	constexpr bool SC = true;
	extern __shared__ __attribute__((aligned(16))) uint32_t coarse_lds[];
	load_coarse(coarse_lds, a.coarse, a.n_blocks_lds, threadIdx.x, blockDim.x);
	uint32_t* dil = coarse_lds + 2 * COARSE_WORDS + 2 * a.n_blocks_lds;
	for (uint32_t q = threadIdx.x; q < COARSE_WORDS; q += blockDim.x) dil[q] = a.coarse[COARSE_DIL_OFF + q];
	__syncthreads();
	constexpr uint64_t GM = (1ull << MG) - 1ull;
	const uint32_t i = blockIdx.x * (WGS / MG) + (threadIdx.x / MG);
	const int lane = threadIdx.x & 63;
	const int g = lane & (MG - 1);
	const int gb = lane & ~(MG - 1);
	const bool ray_exists = i < a.n_rays;
	Vec3 o = {0, 0, 0}, dir = {0, 0, 1}, du = {0, 0, 1}, idir = {0, 0, 1};
	float t_cur = 0.f, startt = 0.f, alive = 0.f, t_exit = 0.f;
	bool term = true;
	if (ray_exists) {
		const uint32_t gi = a.ray_offset + i;
		const uint32_t img = image_idx(gi, a.n_rays_global, a.n_rays_total, a.n_images);
		const ViewDev m = a.views[img];
		Pcg32 rng = a.rng;
		rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
		float xy[2];
		random_image_pos(rng, m.width, m.height, a.snap != 0, xy);
		bool dead = false;
		if (red_is_nonpositive(xy, m, m.normal)) {
			if (rng.next_float() >= 0.9) dead = true; // testbed_nerf.cu:1264
		}
		if (!dead) {
			(void)rng.next_float(); // motionblur_time
			camera_ray(m, xy, o, du, dir);
			idir = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
			float tmin, tmax;
			ray_intersect(a.A, o, dir, &tmin, &tmax);
			tmin = fmaxf(tmin, 0.0f);
			startt = tmin;
			startt += calc_dt(startt, a.A.cone_angle) * rng.next_float();
			t_cur = startt;
			t_exit = tmax;
			alive = 1.f;
			term = false;
		}
	}
	// (1) the ray's stretches of interest: bit i = the dilated coarse occupancy at startt + i / 32 (the point clamped into the box: a sample point just outside the box still
	// answers for the positions just inside it)
	uint64_t interest = 0;
	bool skipping = a.lattice_ok != 0;
	{
#pragma unroll
		for (int it = 0; it < 4; ++it) {
			const int k = g + 16 * it;
			const float ts = startt + (float)k * (1.0f / 32.0f);
			bool hit = false;
			if (!term && ts <= t_exit + (1.5f / 32.0f)) {
				const Vec3 p = o + ts * dir;
				const int bx = min(max((int)(p.x * 32.0f), 0), 31), by = min(max((int)(p.y * 32.0f), 0), 31), bz = min(max((int)(p.z * 32.0f), 0), 31);
				hit = (dil[(uint32_t)by | ((uint32_t)bz << 5)] >> bx) & 1u;
			}
			const unsigned long long w = __ballot(hit);
			interest |= (((uint64_t)w >> gb) & GM) << (16 * it);
		}
		// a ray longer than 63 / 32 (the box diagonal is 55.4 / 32) cannot happen; if the last sample point is still inside the box, do not skip at all
		if (!term && startt + (63.0f / 32.0f) < t_exit) skipping = false;
	}
	uint32_t j = 0;
	bool have_pending = false;
	float pending_target = 0.f;
	bool anchoring = false;      // behind a skip: rounds are evaluated but not replayed until a cell with unanimous votes is found
	float anchor_deadline = 0.f; // ... which has to happen before the round that starts here
	const bool force_restart = (a.lattice_ok & 2u) != 0u; // RNB_MARCH_SKIP=2 (tests): no cell is ever accepted, every skipping ray takes the start-over path
	float* tt = a.ray_t + (size_t)(ray_exists ? i : 0) * RNB_MAX_STEPS;
	float bin_lo = 1.f, bin_hi = 0.f, dlt = 0.f, d2 = 0.f, inv_dlt = 0.f, inv_d2 = 0.f; // the binade of t_cur and its lattice steps (formed again when t_cur leaves it)
	uint32_t st_iters = 0, st_skips = 0, st_restarts = 0, st_anchor_rounds = 0, st_early = 0;
	while (__any(!term)) {
		++st_iters;
		// ---- skip / end decision (group-uniform: every lane of a ray holds the same state) ----
		if (!term && skipping && !anchoring && t_cur >= 0.25f) {
			const int i_first = max((int)floorf((t_cur - startt) * 32.0f) - 1, 0);
			const uint64_t ahead = i_first < 64 ? (interest >> i_first) : 0ull;
			if ((ahead & 0xFull) == 0ull) { // this round's stretch and the two behind it: nothing
				if (ahead == 0ull) { term = true; st_early = 1; } // nothing ever again: the reference marches on to the box exit (or ends at a position outside it) and emits nothing either way
				else {
					const int nb = i_first + __builtin_ctzll(ahead);
					const float t_int = startt + ((float)nb - 0.5f) * (1.0f / 32.0f);
					const int n = (int)floorf((t_int - 52.0f * MIN_CONE_STEPSIZE - t_cur) * (1.0f / MIN_CONE_STEPSIZE)) - 2;
					// Skipped positions are never tested against the box, and the reference ENDS a ray at the first visited position outside it (testbed_nerf.cu:1337) -- which happens
					// at a ray's very first position when startt = tmin + a tiny fraction of a step rounds to just outside the entry face (one ray in 5 million: found by the pinned
					// states' hashes, tests/golden/pinned_states.json). So a stretch is skipped only if both its ends lie inside the box with a margin far above any rounding: the
					// box is convex, every position between them then does too. The first round of a ray starts on the entry face and is therefore always replayed.
					const float t_new = n >= 32 ? lattice_advance(t_cur, n) : t_cur;
					const Vec3 p0 = o + t_cur * dir, p1 = o + t_new * dir;
					const float mlo = a.A.mn + 1e-4f, mhi = a.A.mx - 1e-4f;
					const bool safe = p0.x > mlo && p0.x < mhi && p0.y > mlo && p0.y < mhi && p0.z > mlo && p0.z < mhi && p1.x > mlo && p1.x < mhi && p1.y > mlo && p1.y < mhi && p1.z > mlo && p1.z < mhi;
					if (n >= 32 && safe) {
						t_cur = t_new;
						anchoring = true;
						++st_skips;
						have_pending = false;
						anchor_deadline = t_int - 17.0f * MIN_CONE_STEPSIZE;
					}
				}
			}
		}
		if (!term && anchoring && t_cur > anchor_deadline) { // no unanimous cell in the window: this ray again, from its box entry, every round
			anchoring = false; skipping = false; t_cur = startt; j = 0; have_pending = false;
			++st_restarts;
		}
		if (!term && anchoring) ++st_anchor_rounds;
		// ---- the round's 16 positions: the lattice in closed form, ACROSS a binade boundary too (k_march_count_wide takes 16-step running sums there, wave-wide; with the
		// rays of a wavefront no longer in step that would be up to eight such rounds per wavefront). Positions 0 .. mb lie below the binade's end: t_cur + m d, exact; position
		// mb + 1 is the reference's own addition across the boundary, tx = fl(T_mb + C); the positions behind it are tx + (m - mb - 1) d2 with the next binade's d2, exact.
		if (!(t_cur >= bin_lo && t_cur < bin_hi)) {
			int eb;
			(void)frexpf(t_cur, &eb);
			bin_lo = scalbnf(0.5f, eb); bin_hi = scalbnf(1.0f, eb);
			dlt = (bin_lo + MIN_CONE_STEPSIZE) - bin_lo; d2 = (bin_hi + MIN_CONE_STEPSIZE) - bin_hi;
			inv_dlt = __builtin_amdgcn_rcpf(dlt); inv_d2 = __builtin_amdgcn_rcpf(d2); // estimates: every candidate they yield is checked exactly
		}
		int mb = MG + 1; // the last position below bin_hi (MG + 1: all 17 of them)
		if (!(t_cur + (float)MG * dlt < bin_hi)) {
			mb = (int)floorf((bin_hi - t_cur) * inv_dlt);
			mb = max(mb - 1, 0);
			while (mb < MG && __builtin_fmaf((float)(mb + 1), dlt, t_cur) < bin_hi) ++mb;
		}
		const float tx = __builtin_fmaf((float)min(mb, MG), dlt, t_cur) + MIN_CONE_STEPSIZE;
		auto lattice = [&](const int m) { return m <= mb ? __builtin_fmaf((float)m, dlt, t_cur) : __builtin_fmaf((float)(m - mb - 1), d2, tx); };
		// the smallest m in [lo, MG) with lattice(m) >= target, else MG
		auto first_at_or_after = [&](const float target, const int lo) {
			int m = (int)floorf(fminf(fmaxf((target - t_cur) * inv_dlt, -2.f), 64.f));
			m += (__builtin_fmaf((float)m, dlt, t_cur) < target) ? 1 : 0;
			m = max(m, lo);
			if (m > mb) {
				int q = (int)floorf(fminf(fmaxf((target - tx) * inv_d2, -2.f), 64.f));
				q += (__builtin_fmaf((float)q, d2, tx) < target) ? 1 : 0;
				m = max(mb + 1 + max(q, 0), lo);
			}
			return min(m, MG);
		};
		const float my_t = lattice(g);
		const float t_end = lattice(MG);
		const Vec3 pos = o + my_t * dir;
		const bool inside = !term && aabb_contains(a.A, pos);
		bool occ = false;
		float t_target = 0.f;
		uint32_t nxt = MG;
		if (inside) {
			occ = occupied_mip0(pos, a.bitfield, coarse_lds, a.n_blocks_lds);
			if (!occ) {
				t_target = my_t + distance_to_next_voxel(pos, dir, idir, GRIDSIZE);
				nxt = (uint32_t)first_at_or_after(t_target, g + 1);
			}
		}
		const unsigned long long occ_w = __ballot(occ), in_w = __ballot(inside);
		const uint64_t occ16 = ((uint64_t)occ_w >> gb) & GM, in16 = ((uint64_t)in_w >> gb) & GM;
		// ---- (2) behind a skip: the first cell of this round with all its positions in the round, all empty, all voting for the position behind it ----
		int start_cur = 0;
		bool replay = !term && !anchoring;
		if (__any(!term && anchoring)) { // (wave-uniform: most rounds have no ray looking for its way back in)
			// the cell of the position (cascaded_grid_idx_at's integer coordinates, as occupied_mip0 forms them)
			const int cx = min(max((int)(pos.x * GRIDSIZE), 0), (int)GRIDSIZE - 1), cy = min(max((int)(pos.y * GRIDSIZE), 0), (int)GRIDSIZE - 1), cz = min(max((int)(pos.z * GRIDSIZE), 0), (int)GRIDSIZE - 1);
			const uint32_t cell = (uint32_t)cx | ((uint32_t)cy << 7) | ((uint32_t)cz << 14);
			const uint32_t prev = (uint32_t)__shfl((int)cell, gb + max(g - 1, 0), 64);
			const bool bnd = g > 0 && cell != prev; // a cell begins at this position (position 0: unknown, the cell may have begun in the round before)
			const unsigned long long bnd_w = __ballot(bnd);
			const uint32_t bnd16 = (uint32_t)(((uint64_t)bnd_w >> gb) & GM);
			// my cell's first position a (the highest boundary at or below me) and the position behind its last one, u (the lowest boundary above me)
			const uint32_t below = bnd16 & ((2u << g) - 1u), above = bnd16 & ~((2u << g) - 1u);
			const int u = above ? __builtin_ctz(above) : MG;
			const bool complete = below != 0u && u < MG;
			const bool vote_ok = complete && inside && !occ && nxt == (uint32_t)u && ((in16 >> u) & 1ull);
			const unsigned long long ok_w = __ballot(vote_ok);
			const uint32_t ok16 = (uint32_t)(((uint64_t)ok_w >> gb) & GM);
			if (!term && anchoring) {
				// walk the round's cells in order (at most 15 boundaries): the first one whose positions all voted for the position behind it
				uint32_t rest = bnd16;
				while (rest) {
					const int a0 = __builtin_ctz(rest);
					rest &= rest - 1u;
					if (!rest) break; // the last cell of the round has no end in it
					const int u0 = __builtin_ctz(rest);
					const uint32_t cellmask = ((1u << u0) - 1u) & ~((1u << a0) - 1u);
					if (!force_restart && (ok16 & cellmask) == cellmask) { start_cur = u0; anchoring = false; replay = true; break; }
				}
			}
		}
		// ---- replay of the sequential visit order over this round's outcomes (as k_march_count_wide; start_cur: the position the chain is known to arrive on) ----
		uint64_t vis = 0;
		const uint32_t j0 = j;
		int pend_src = -1;
		if (replay) {
			int cur = start_cur;
			if (have_pending) {
				cur = first_at_or_after(pending_target, 0);
				if (cur < MG) have_pending = false;
			}
			while (cur < MG) {
				if (!((in16 >> cur) & 1ull)) { term = true; break; }
				if ((occ16 >> cur) & 1ull) {
					const uint64_t run_mask = (occ16 & in16) >> cur;
					int run = __builtin_ctzll(~run_mask);
					run = min(run, MG - cur);
					const int allowed = min(run, (int)(RNB_MAX_STEPS - j));
					vis |= ((1ull << allowed) - 1ull) << cur;
					j += (uint32_t)allowed;
					cur += allowed;
					if (j >= RNB_MAX_STEPS) { term = true; break; }
				} else {
					const int nx = (int)__shfl(nxt, gb + cur, 64);
					if (nx >= MG) { have_pending = true; pend_src = cur; }
					cur = nx;
				}
			}
		}
		const float tgt = __shfl(t_target, gb + (pend_src < 0 ? 0 : pend_src), 64);
		if (pend_src >= 0) pending_target = tgt;
		if ((vis >> g) & 1ull) tt[j0 + __popcll(vis & ((1ull << g) - 1ull))] = my_t;
		t_cur = t_end;
	}
	if (a.stats) {
		if (lane == 0) { atomicAdd(a.stats + 0, 1ull); atomicAdd(a.stats + 1, (unsigned long long)st_iters); }
		if (ray_exists && g == 0) { atomicAdd(a.stats + 2, 1ull); atomicAdd(a.stats + 3, (unsigned long long)(st_skips != 0)); atomicAdd(a.stats + 4, (unsigned long long)st_restarts); atomicAdd(a.stats + 5, (unsigned long long)st_anchor_rounds); atomicAdd(a.stats + 6, (unsigned long long)st_early); }
	}
	if (ray_exists && g == 0) {
		float* st = a.setup + (size_t)i * 8;
		st[0] = o.x; st[1] = o.y; st[2] = o.z; st[3] = dir.x; st[4] = dir.y; st[5] = dir.z; st[6] = startt; st[7] = alive;
		a.d_unnorm[(size_t)i * 3 + 0] = du.x; a.d_unnorm[(size_t)i * 3 + 1] = du.y; a.d_unnorm[(size_t)i * 3 + 2] = du.z;
		a.steps[i] = j;
	}
}

// The same two facts for the ONE-THREAD-PER-RAY march (k_march_count<true>, batches from 18 432 rays on: the window's second half and the late regime, where a ray holds ~20 marched
// samples and ~220 visited empty voxels). A thread walks the reference's loop itself, so its chain state is always exact; when the stretch in front of it is clear for at least 32 steps
// (and safely inside the box) it jumps to 52 steps in front of the next stretch of interest and looks, position by position, for the re-entry cell: a cell whose FIRST lattice position
// it has seen (a cell change behind the jump target), all of whose positions are empty and jump to the position behind the cell's last one. If it finds none within 48 positions it has
// changed nothing: it goes on from where it stood, every voxel, and does not try again on this ray.
// The re-entry cell behind a jump to lattice position tq (thread-per-ray forms): walks positions tq, tq + dt, ... (the reference's own additions) until it has seen a cell from its
// FIRST position to its last, all of them inside the box, empty, and jumping to the position behind the cell's last one; that position (a position the reference's chain visits,
// see k_march_count_skip) is returned in t_found. false: none before t_limit (nothing has been changed; the caller walks on from where it stood).
__device__ __forceinline__ bool find_reentry(const MarchArgs& a, const uint32_t* __restrict__ coarse_lds, const Vec3& o, const Vec3& dir, const Vec3& idir, float tq, const float t_limit, float& t_found) {
	auto cell_of = [&](const Vec3& p) {
		const int cx = min(max((int)(p.x * GRIDSIZE), 0), (int)GRIDSIZE - 1), cy = min(max((int)(p.y * GRIDSIZE), 0), (int)GRIDSIZE - 1), cz = min(max((int)(p.z * GRIDSIZE), 0), (int)GRIDSIZE - 1);
		return (uint32_t)cx | ((uint32_t)cy << 7) | ((uint32_t)cz << 14);
	};
	const bool force_fail = (a.lattice_ok & 2u) != 0u; // RNB_MARCH_SKIP=2 (tests): no re-entry cell is ever accepted
	Vec3 pq = o + tq * dir;
	uint32_t prev_cell = cell_of(pq);
	float cell_first = -1.f, t_prev = tq; // cell_first < 0: the first position of the cell being examined was not seen
	bool cell_ok = false;                 // every position of the examined cell so far is inside and empty ...
	float max_target = 0.f, min_target = 3.0e38f; // ... and their jump targets: all must land on the position behind the cell (t_last < target <= t_next for every one of them)
#pragma unroll 1
	for (int k = 1; k < 48; ++k) {
		if (cell_first >= 0.f) { // the vote of position t_prev
			const bool in = aabb_contains(a.A, pq);
			const bool occ = in && occupied_mip0(pq, a.bitfield, coarse_lds, a.n_blocks_lds);
			const float target = t_prev + distance_to_next_voxel(pq, dir, idir, GRIDSIZE);
			cell_ok = cell_ok && in && !occ;
			max_target = fmaxf(max_target, target); min_target = fminf(min_target, target);
		}
		const float t_next = t_prev + MIN_CONE_STEPSIZE;
		const Vec3 p_next = o + t_next * dir;
		const uint32_t c_next = cell_of(p_next);
		if (c_next != prev_cell) {
			if (cell_first >= 0.f && cell_ok && min_target > t_prev && max_target <= t_next && aabb_contains(a.A, p_next) && !force_fail) { t_found = t_next; return true; }
			cell_first = t_next; cell_ok = true; max_target = 0.f; min_target = 3.0e38f; // the next cell starts here: its first position is seen
			prev_cell = c_next;
		}
		t_prev = t_next; pq = p_next;
		if (t_prev > t_limit) break;
	}
	return false;
}

template <typename F>
__device__ __forceinline__ uint32_t march_skip_narrow(const MarchArgs& a, const uint32_t* __restrict__ coarse_lds, const uint32_t* __restrict__ dil, const Vec3& o, const Vec3& dir,
                                                      const float startt, const float t_exit, unsigned long long* __restrict__ stats, F&& emit) {
	const Vec3 idir = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	// bit i: the dilated coarse occupancy at startt + i / 32 (see k_march_count_skip)
	uint64_t interest = 0;
	bool skipping = a.lattice_ok != 0 && !(startt + (63.0f / 32.0f) < t_exit);
	if (skipping) {
		const int n_pts = min(64, (int)((t_exit - startt) * 32.0f) + 3);
#pragma unroll 1
		for (int k = 0; k < n_pts; ++k) {
			const Vec3 p = o + (startt + (float)k * (1.0f / 32.0f)) * dir;
			const int bx = min(max((int)(p.x * 32.0f), 0), 31), by = min(max((int)(p.y * 32.0f), 0), 31), bz = min(max((int)(p.z * 32.0f), 0), 31);
			interest |= (uint64_t)((dil[(uint32_t)by | ((uint32_t)bz << 5)] >> bx) & 1u) << k;
		}
	}
	const float mlo = a.A.mn + 1e-4f, mhi = a.A.mx - 1e-4f;
	auto safely_inside = [&](const Vec3& p) { return p.x > mlo && p.x < mhi && p.y > mlo && p.y < mhi && p.z > mlo && p.z < mhi; };
	uint32_t j = 0, n_skips = 0, n_fail = 0, n_early = 0;
	float t = startt;
	Vec3 pos;
	while (aabb_contains(a.A, pos = o + t * dir) && j < RNB_MAX_STEPS) {
		if (skipping && t >= 0.25f) {
			const int i_first = max((int)floorf((t - startt) * 32.0f) - 1, 0);
			const uint64_t ahead = i_first < 64 ? (interest >> i_first) : 0ull;
			if ((ahead & 0xFull) == 0ull) {
				if (ahead == 0ull) { n_early = 1; break; } // nothing ever again: the reference marches on (or ends at a position outside the box) and emits nothing
				const int nb = i_first + __builtin_ctzll(ahead);
				const float t_int = startt + ((float)nb - 0.5f) * (1.0f / 32.0f);
				const int n = (int)floorf((t_int - 52.0f * MIN_CONE_STEPSIZE - t) * (1.0f / MIN_CONE_STEPSIZE)) - 2;
				if (n >= 32 && safely_inside(pos)) {
					float tq = lattice_advance(t, n);
					Vec3 pq = o + tq * dir;
					if (safely_inside(pq)) {
						float t_re = 0.f;
						const bool found = find_reentry(a, coarse_lds, o, dir, idir, tq, t_int - 17.0f * MIN_CONE_STEPSIZE, t_re);
						if (found) tq = t_re;
						if (found) { t = tq; ++n_skips; continue; } // t is a position the reference's chain visits: go on from it
						skipping = false; ++n_fail;                 // nothing was changed: every voxel from here on
					}
				}
			}
		}
		if (occupied_mip0(pos, a.bitfield, coarse_lds, a.n_blocks_lds)) {
			emit(j, pos, MIN_CONE_STEPSIZE, t);
			++j;
			t += MIN_CONE_STEPSIZE;
		} else {
			const float t_target = t + distance_to_next_voxel(pos, dir, idir, GRIDSIZE);
			do { t += MIN_CONE_STEPSIZE; } while (t < t_target);
		}
	}
	if (stats) { atomicAdd(stats + 2, 1ull); atomicAdd(stats + 3, (unsigned long long)(n_skips != 0)); atomicAdd(stats + 4, (unsigned long long)n_fail); atomicAdd(stats + 6, (unsigned long long)n_early); }
	return j;
}

// Round 6, RNB_MARCH_BBOX=2 (measured, NOT the default: the jump costs what it saves): the thread-per-ray march of the single-cascade scenes (batches from 18 432 rays on): k_march_count<true>'s walk, with the ray cut to the part that can hold a sample by
// the BOUNDING BOX of the non-empty 4^3 blocks (COARSE_BBOX_OFF, dilated by a block): a ray that misses the box has no sample; a ray is over where it leaves the box; and ONE jump -- from
// the first position that lies safely inside the scene box to 52 steps in front of the occupied box's entry, re-entering the reference's visit chain through find_reentry (a ray
// that finds no re-entry cell has changed nothing and walks every voxel). Unlike the per-ray stretch scan of march_skip_narrow every lane of a wavefront does the same thing at the same
// time: first position, jump, re-entry search, walk -- which is what a wavefront of 64 independent rays needs to gain anything. Sample set and t: the reference's, bit for bit.
__global__ __launch_bounds__(128) void k_march_count_bbox(const MarchArgs a) {
	extern __shared__ __attribute__((aligned(16))) uint32_t coarse_lds[];
	load_coarse(coarse_lds, a.coarse, a.n_blocks_lds, threadIdx.x, blockDim.x);
	__syncthreads();
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= a.n_rays) return;
	const uint32_t gi = a.ray_offset + i;
	const uint32_t img = image_idx(gi, a.n_rays_global, a.n_rays_total, a.n_images);
	const ViewDev m = a.views[img];
	Pcg32 rng = a.rng;
	rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
	float xy[2];
	random_image_pos(rng, m.width, m.height, a.snap != 0, xy);
	uint32_t j = 0;
	float alive = 0.f;
	Vec3 o = {0, 0, 0}, dir = {0, 0, 1}, du = {0, 0, 1};
	float startt = 0.f;
	bool dead = false;
	if (red_is_nonpositive(xy, m, m.normal)) {
		if (rng.next_float() >= 0.9) dead = true; // testbed_nerf.cu:1264, short-circuit draw
	}
	if (!dead) {
		(void)rng.next_float(); // motionblur_time, testbed_nerf.cu:1270
		camera_ray(m, xy, o, du, dir);
		float tmin, tmax;
		ray_intersect(a.A, o, dir, &tmin, &tmax);
		tmin = fmaxf(tmin, 0.0f);
		startt = tmin;
		startt += calc_dt(startt, a.A.cone_angle) * rng.next_float();
		alive = 1.f;
		float* tt = a.ray_t + (size_t)i * RNB_MAX_STEPS;
		// the ray against the dilated bounding box of the non-empty blocks: [t_in, t_out]
		const uint32_t* bb = a.coarse + COARSE_BBOX_OFF;
		const float lo[3] = {((float)bb[0] - 1.0f) * (1.0f / 32.0f), ((float)bb[1] - 1.0f) * (1.0f / 32.0f), ((float)bb[2] - 1.0f) * (1.0f / 32.0f)};
		const float hi[3] = {((float)bb[3] + 2.0f) * (1.0f / 32.0f), ((float)bb[4] + 2.0f) * (1.0f / 32.0f), ((float)bb[5] + 2.0f) * (1.0f / 32.0f)};
		const float oo[3] = {o.x, o.y, o.z}, dd[3] = {dir.x, dir.y, dir.z};
		float t_in = -3.0e38f, t_out = 3.0e38f;
		bool can_hit = bb[0] <= bb[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			if (fabsf(dd[d]) < 1e-12f) { if (oo[d] < lo[d] || oo[d] > hi[d]) can_hit = false; continue; }
			const float ta = (lo[d] - oo[d]) / dd[d], tb = (hi[d] - oo[d]) / dd[d];
			t_in = fmaxf(t_in, fminf(ta, tb)); t_out = fminf(t_out, fmaxf(ta, tb));
		}
		if (t_in > t_out) can_hit = false;
		if (can_hit) {
			const float t_stop = t_out + 4.0f * MIN_CONE_STEPSIZE;
			const Vec3 idir = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
			const float mlo = a.A.mn + 1e-4f, mhi = a.A.mx - 1e-4f;
			auto safely_inside = [&](const Vec3& p) { return p.x > mlo && p.x < mhi && p.y > mlo && p.y < mhi && p.z > mlo && p.z < mhi; };
			bool may_jump = (a.lattice_ok & 1u) != 0u && a.use_bbox >= 2u;
			uint32_t jumped = 0, failed = 0;
			float t = startt;
			Vec3 pos;
			while (aabb_contains(a.A, pos = o + t * dir) && j < RNB_MAX_STEPS && t <= t_stop) {
				if (may_jump && t >= 0.25f && safely_inside(pos)) { // (the ray's first position lies ON the entry face: it is always visited the reference's way)
					may_jump = false; // one attempt per ray
					const int n = (int)floorf((t_in - 52.0f * MIN_CONE_STEPSIZE - t) * (1.0f / MIN_CONE_STEPSIZE)) - 2;
					if (n >= 32) {
						const float tq = lattice_advance(t, n);
						float t_re = 0.f;
						if (safely_inside(o + tq * dir) && find_reentry(a, coarse_lds, o, dir, idir, tq, t_in - 17.0f * MIN_CONE_STEPSIZE, t_re)) { t = t_re; jumped = 1; continue; }
						failed = 1;
					}
				}
				if (occupied_mip0(pos, a.bitfield, coarse_lds, a.n_blocks_lds)) {
					tt[j] = t;
					++j;
					t += MIN_CONE_STEPSIZE;
				} else {
					const float t_target = t + distance_to_next_voxel(pos, dir, idir, GRIDSIZE);
					do { t += MIN_CONE_STEPSIZE; } while (t < t_target);
				}
			}
			if (a.stats) { atomicAdd(a.stats + 2, 1ull); atomicAdd(a.stats + 3, (unsigned long long)jumped); atomicAdd(a.stats + 4, (unsigned long long)failed); }
		}
	}
	float* st = a.setup + (size_t)i * 8;
	st[0] = o.x; st[1] = o.y; st[2] = o.z; st[3] = dir.x; st[4] = dir.y; st[5] = dir.z; st[6] = startt; st[7] = alive;
	a.d_unnorm[(size_t)i * 3 + 0] = du.x; a.d_unnorm[(size_t)i * 3 + 1] = du.y; a.d_unnorm[(size_t)i * 3 + 2] = du.z;
	a.steps[i] = j;
}

__global__ __launch_bounds__(128) void k_march_count_skip_narrow(const MarchArgs a) {
	extern __shared__ __attribute__((aligned(16))) uint32_t coarse_lds[];
	load_coarse(coarse_lds, a.coarse, a.n_blocks_lds, threadIdx.x, blockDim.x);
	uint32_t* dil = coarse_lds + 2 * COARSE_WORDS + 2 * a.n_blocks_lds;
	for (uint32_t q = threadIdx.x; q < COARSE_WORDS; q += blockDim.x) dil[q] = a.coarse[COARSE_DIL_OFF + q];
	__syncthreads();
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= a.n_rays) return;
	const uint32_t gi = a.ray_offset + i;
	const uint32_t img = image_idx(gi, a.n_rays_global, a.n_rays_total, a.n_images);
	const ViewDev m = a.views[img];
	Pcg32 rng = a.rng;
	rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
	float xy[2];
	random_image_pos(rng, m.width, m.height, a.snap != 0, xy);
	uint32_t steps = 0;
	float alive = 0.f;
	Vec3 o = {0, 0, 0}, dir = {0, 0, 1}, du = {0, 0, 1};
	float startt = 0.f;
	bool dead = false;
	if (red_is_nonpositive(xy, m, m.normal)) {
		if (rng.next_float() >= 0.9) dead = true; // testbed_nerf.cu:1264, short-circuit draw
	}
	if (!dead) {
		(void)rng.next_float(); // motionblur_time, testbed_nerf.cu:1270
		camera_ray(m, xy, o, du, dir);
		float tmin, tmax;
		ray_intersect(a.A, o, dir, &tmin, &tmax);
		tmin = fmaxf(tmin, 0.0f);
		startt = tmin;
		startt += calc_dt(startt, a.A.cone_angle) * rng.next_float();
		alive = 1.f;
		float* tt = a.ray_t + (size_t)i * RNB_MAX_STEPS;
		steps = march_skip_narrow(a, coarse_lds, dil, o, dir, startt, tmax, a.stats, [&](uint32_t j, const Vec3&, float, float t) { tt[j] = t; });
	}
	float* st = a.setup + (size_t)i * 8;
	st[0] = o.x; st[1] = o.y; st[2] = o.z; st[3] = dir.x; st[4] = dir.y; st[5] = dir.z; st[6] = startt; st[7] = alive;
	a.d_unnorm[(size_t)i * 3 + 0] = du.x; a.d_unnorm[(size_t)i * 3 + 1] = du.y; a.d_unnorm[(size_t)i * 3 + 2] = du.z;
	a.steps[i] = steps;
}

// Single-workgroup exclusive scans over the rays (n <= 2^18): base = scan(steps); survivors = base + steps <= max_samples;
// slot = scan(survivor). counters[0] = sum(steps) (numsteps_counter), [2] = #survivors (ray_counter), [3] = samples written.
// Exclusive prefix sums over the rays, tile by tile (1024 coalesced elements per tile, wave shuffles + one LDS hop):
//   base  = samples before the ray (numsteps prefix, testbed_nerf.cu:1344-1346)          -> counters[0] = total
//   slot  = index among the rays that keep their samples (0xffffffff = dropped)            -> counters[2], counters[3]
//   base1 = offset of the ray's first-round samples in idx1 (two-round network evaluation) -> fwd_counts[0]
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, const uint32_t lane) {
#pragma unroll
	for (uint32_t off = 1; off < 64; off <<= 1) {
		const uint32_t t = __shfl_up(v, off, 64);
		if (lane >= off) v += t;
	}
	return v;
}

__global__ __launch_bounds__(1024) void k_scan_rays(const uint32_t n, const uint32_t max_samples, const uint32_t* __restrict__ steps,
                                                    uint32_t* __restrict__ base, uint32_t* __restrict__ slot, uint32_t* __restrict__ counters,
                                                    const uint32_t k1, uint32_t* __restrict__ base1, uint32_t* __restrict__ fwd_counts) {
	__shared__ uint32_t wsum[4][16];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	constexpr uint32_t E = 4; // consecutive rays per thread (one 16-byte load): 4096 per tile; 16 per thread measured slower (strided lanes)
	uint32_t carry[4] = {0, 0, 0, 0}; // steps, kept rays, kept samples, first-round samples
	// the next tile's counts are requested before this tile's scans: late in training the batch is ~100 k rays = 23 tiles, and one
	// workgroup cannot hide a global load behind anything else
	uint32_t st_next[E];
#pragma unroll
	for (uint32_t e = 0; e < E; ++e) st_next[e] = tid * E + e < n ? steps[tid * E + e] : 0u;
	for (uint32_t t0 = 0; t0 < n; t0 += 1024 * E) {
		const uint32_t i0 = t0 + tid * E;
		uint32_t st[E];
#pragma unroll
		for (uint32_t e = 0; e < E; ++e) st[e] = st_next[e];
#pragma unroll
		for (uint32_t e = 0; e < E; ++e) st_next[e] = i0 + 1024 * E + e < n ? steps[i0 + 1024 * E + e] : 0u;
		// pass A: sample offsets
		uint32_t mine = 0;
#pragma unroll
		for (uint32_t e = 0; e < E; ++e) mine += st[e];
		const uint32_t inc = wave_inclusive_scan(mine, lane);
		if (lane == 63) wsum[0][wave] = inc;
		__syncthreads();
		uint32_t wprefix = 0, ttotal = 0;
#pragma unroll
		for (uint32_t w = 0; w < 16; ++w) { const uint32_t x = wsum[0][w]; wprefix += w < wave ? x : 0u; ttotal += x; }
		uint32_t run = carry[0] + wprefix + inc - mine;
		uint32_t runs[E];
		bool ok[E];
		uint32_t v[3] = {0, 0, 0};
#pragma unroll
		for (uint32_t e = 0; e < E; ++e) {
			runs[e] = run;
			ok[e] = st[e] > 0 && run + st[e] <= max_samples; // testbed_nerf.cu:1348-1355
			run += st[e];
			v[0] += ok[e] ? 1u : 0u; v[1] += ok[e] ? st[e] : 0u; v[2] += ok[e] ? min(st[e], k1) : 0u;
		}
		// pass B: the three sums that depend on `ok`
		uint32_t vi[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) { vi[k] = wave_inclusive_scan(v[k], lane); if (lane == 63) wsum[1 + k][wave] = vi[k]; }
		__syncthreads();
		uint32_t pre[3] = {0, 0, 0}, tot[3] = {0, 0, 0};
#pragma unroll
		for (uint32_t w = 0; w < 16; ++w)
#pragma unroll
			for (int k = 0; k < 3; ++k) { const uint32_t x = wsum[1 + k][w]; pre[k] += w < wave ? x : 0u; tot[k] += x; }
		uint32_t srun = carry[1] + pre[0] + vi[0] - v[0], frun = carry[3] + pre[2] + vi[2] - v[2];
#pragma unroll
		for (uint32_t e = 0; e < E; ++e) {
			if (i0 + e < n) {
				base[i0 + e] = runs[e];
				slot[i0 + e] = ok[e] ? srun : 0xffffffffu;
				if (k1) base1[i0 + e] = frun;
			}
			srun += ok[e] ? 1u : 0u;
			frun += ok[e] ? min(st[e], k1) : 0u;
		}
		carry[0] += ttotal; carry[1] += tot[0]; carry[2] += tot[1]; carry[3] += tot[2];
		__syncthreads(); // wsum is rewritten by the next tile
	}
	if (tid == 0) { counters[0] = carry[0]; counters[2] = carry[1]; counters[3] = carry[2]; fwd_counts[0] = carry[3]; fwd_counts[1] = 0; fwd_counts[2] = 0; fwd_counts[3] = 0; }
}

// The same scans with one workgroup per 4096-ray tile, for batches of tens of thousands of rays (one workgroup walks 23 tiles at
// 94 k rays: 0.13 ms of a chain that the next kernels wait for): (1) tile sums of the sample counts, (2) offsets + which rays
// keep their samples + tile sums of the three dependent counts, (3) slots. Integer work: bit-identical to k_scan_rays.
constexpr uint32_t SCAN_TILE = 4096;
// Workgroups of 256 threads, 16 rays per thread: beside the scatter kernels a CU rarely has the 16 free wavefront slots a
// 1024-thread workgroup needs at once (k_scan_rays_base waited 45 us for them, measured), 4 are found at once.
constexpr uint32_t SCAN_WG = 256, SCAN_EPT = SCAN_TILE / SCAN_WG, SCAN_NW = SCAN_WG / 64;
// exclusive prefix of `mine` over the NW wavefronts of the workgroup; `total` = sum over the workgroup
template <uint32_t NW>
__device__ __forceinline__ uint32_t block_exclusive_scan(const uint32_t mine, const uint32_t lane, const uint32_t wave, uint32_t* __restrict__ wsum, uint32_t& total) {
	const uint32_t inc = wave_inclusive_scan(mine, lane);
	if (lane == 63) wsum[wave] = inc;
	__syncthreads();
	uint32_t pre = 0, tot = 0;
#pragma unroll
	for (uint32_t w = 0; w < NW; ++w) { const uint32_t x = wsum[w]; pre += w < wave ? x : 0u; tot += x; }
	__syncthreads();
	total = tot;
	return pre + inc - mine;
}
// sum of vals[0 .. count) (count <= 64), by the first wavefront; result in every thread of the workgroup
__device__ __forceinline__ uint32_t tile_prefix(const uint32_t* __restrict__ vals, const uint32_t stride, const uint32_t count, const uint32_t tid, uint32_t* __restrict__ sh) {
	if (tid < 64) {
		uint32_t v = tid < count ? vals[(size_t)tid * stride] : 0u;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
		if (tid == 0) *sh = v;
	}
	__syncthreads();
	const uint32_t r = *sh;
	__syncthreads();
	return r;
}

__global__ __launch_bounds__(SCAN_WG) void k_scan_rays_sums(const uint32_t n, const uint32_t* __restrict__ steps, uint32_t* __restrict__ tile_sum) {
	__shared__ uint32_t wsum[16];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t i0 = blockIdx.x * SCAN_TILE + tid * SCAN_EPT;
	uint32_t mine = 0;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) mine += i0 + e < n ? steps[i0 + e] : 0u;
	uint32_t total;
	(void)block_exclusive_scan<SCAN_NW>(mine, lane, wave, wsum, total);
	if (tid == 0) tile_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_WG) void k_scan_rays_base(const uint32_t n, const uint32_t max_samples, const uint32_t k1, const uint32_t* __restrict__ steps,
                                                         const uint32_t* __restrict__ tile_sum, uint32_t* __restrict__ base, uint32_t* __restrict__ tile_v, uint32_t* __restrict__ counters) {
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t sh;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t tile_base = tile_prefix(tile_sum, 1, blockIdx.x, tid, &sh);
	const uint32_t i0 = blockIdx.x * SCAN_TILE + tid * SCAN_EPT;
	uint32_t st[SCAN_EPT], mine = 0;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) { st[e] = i0 + e < n ? steps[i0 + e] : 0u; mine += st[e]; }
	uint32_t total;
	uint32_t run = tile_base + block_exclusive_scan<SCAN_NW>(mine, lane, wave, wsum, total);
	uint32_t v[3] = {0, 0, 0};
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) {
		if (i0 + e < n) base[i0 + e] = run;
		const bool ok = st[e] > 0 && run + st[e] <= max_samples; // testbed_nerf.cu:1348-1355
		run += st[e];
		v[0] += ok ? 1u : 0u; v[1] += ok ? st[e] : 0u; v[2] += ok ? min(st[e], k1) : 0u;
	}
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		uint32_t t;
		(void)block_exclusive_scan<SCAN_NW>(v[k], lane, wave, wsum, t);
		if (tid == 0) tile_v[blockIdx.x * 3 + k] = t;
	}
	if (blockIdx.x == gridDim.x - 1 && tid == 0) counters[0] = tile_base + total; // numsteps_counter
}

__global__ __launch_bounds__(SCAN_WG) void k_scan_rays_slots(const uint32_t n, const uint32_t max_samples, const uint32_t k1, const uint32_t* __restrict__ steps,
                                                          const uint32_t* __restrict__ base, const uint32_t* __restrict__ tile_v, uint32_t* __restrict__ slot,
                                                          uint32_t* __restrict__ base1, uint32_t* __restrict__ counters, uint32_t* __restrict__ fwd_counts) {
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t sh;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	uint32_t pre[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) pre[k] = tile_prefix(tile_v + k, 3, blockIdx.x, tid, &sh);
	const uint32_t i0 = blockIdx.x * SCAN_TILE + tid * SCAN_EPT;
	uint32_t st[SCAN_EPT], v0 = 0, v2 = 0, v1 = 0;
	bool ok[SCAN_EPT];
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) {
		st[e] = i0 + e < n ? steps[i0 + e] : 0u;
		const uint32_t b = i0 + e < n ? base[i0 + e] : 0u;
		ok[e] = st[e] > 0 && b + st[e] <= max_samples;
		v0 += ok[e] ? 1u : 0u; v1 += ok[e] ? st[e] : 0u; v2 += ok[e] ? min(st[e], k1) : 0u;
	}
	uint32_t t0, t1, t2;
	uint32_t srun = pre[0] + block_exclusive_scan<SCAN_NW>(v0, lane, wave, wsum, t0);
	uint32_t frun = pre[2] + block_exclusive_scan<SCAN_NW>(v2, lane, wave, wsum, t2);
	(void)block_exclusive_scan<SCAN_NW>(v1, lane, wave, wsum, t1);
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) {
		if (i0 + e < n) {
			slot[i0 + e] = ok[e] ? srun : 0xffffffffu;
			if (k1) base1[i0 + e] = frun;
		}
		srun += ok[e] ? 1u : 0u;
		frun += ok[e] ? min(st[e], k1) : 0u;
	}
	if (blockIdx.x == gridDim.x - 1 && tid == 0) { counters[2] = pre[0] + t0; counters[3] = pre[1] + t1; fwd_counts[0] = pre[2] + t2; fwd_counts[1] = 0; fwd_counts[2] = 0; fwd_counts[3] = 0; }
}

// The three tiled kernels above in ONE launch. A tile's sums do not depend on the tiles in front of it, so every workgroup publishes them at once
// (value + the launch's ticket in one 64-bit word) and then adds up the words of the tiles in front of it as they arrive -- no chain from tile to
// tile, two such exchanges per launch (the sample offsets first, then the three sums that depend on which rays fit). Workgroups start in index
// order, so a workgroup that waits only waits for workgroups that are already running. The words are polled with returning atomics: a load, agent
// scope included, is answered by the polling XCD's own L2 once the line is there (profiles/r03_ab_tickets.txt). Integer work: the numbers of
// k_scan_rays. Why: the scans sit in the middle of the chain march -> scans -> write that the next step's network evaluation waits for; as one
// 1024-thread workgroup (small batches) the scan waited for 16 free wavefront slots on one CU beside the scatter (65 us for 21 us of work), as three
// launches (large batches) it took 48 us for 23.
struct ScanChainArgs {
	uint32_t n, max_samples, k1;
	const uint32_t* steps;
	uint32_t *base, *slot, *base1, *counters, *fwd_counts;
	unsigned long long* words;  // [n_tiles][4]: ticket << 32 | {samples, rays with samples, first-round samples} of the tile, were no ray dropped
	unsigned long long* words2; // [n_tiles][4]: ticket << 32 | {kept rays, kept samples, kept first-round samples}: tiles at and behind an overflow of max_samples only
	uint32_t ticket;
	uint32_t* error; // mapped host word: a wait gave up
	uint32_t plain;  // RNB_CHAIN_PLAIN (round 6): the tiles' words travel as agent-scope atomic STORES and LOADS (sc1: served at the memory side, seen by every XCD) instead of read-modify-write
	                 // atomics, which queue behind the gradient scatter's backlog of atomics there
};
__device__ __forceinline__ void chain_put(unsigned long long* p, const unsigned long long v, const bool plain) {
	if (plain) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else atomicExch(p, v);
}
__device__ __forceinline__ unsigned long long chain_get(unsigned long long* p, const bool plain) {
	return plain ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : atomicAdd(p, 0ull);
}
// `bad` (in/out, uniform over the workgroup): a wait of this tile or of a tile in front of it gave up. The HIP programming model does not promise that
// lower-numbered workgroups are scheduled first (they are on this hardware, one dispatcher per queue handing out workgroups in order), so the
// spin is bounded; a tile that gives up would otherwise go on with a stale word of an earlier launch. Instead it marks the word it publishes
// (bit 31: sums stay below 2^23), the mark travels with the sums to the tiles behind, the last tile reports zero counters (the step then ends
// with RNB_ERR_NO_SAMPLES instead of training on garbage) and the mapped host word `error` names the launch for the next synchronising call.
constexpr uint32_t CHAIN_POISON = 0x80000000u;
__device__ __forceinline__ uint32_t chain_prefix(unsigned long long* __restrict__ words, const uint32_t k, const uint32_t tile, const uint32_t ticket, const uint32_t mine,
                                                 const uint32_t tid, uint32_t* __restrict__ sh, uint32_t* __restrict__ error, bool& bad) {
	if (tid == 0) atomicExch(words + tile * 4 + k, ((unsigned long long)ticket << 32) | mine | (bad ? CHAIN_POISON : 0u));
	if (tid < 64) { // tiles in front of this one: one lane each (<= 63)
		uint32_t v = 0, flag = bad ? 1u : 0u;
		if (tid < tile) {
			unsigned long long w;
			uint32_t spins = 0;
			do { w = atomicAdd(words + tid * 4 + k, 0ull); if ((uint32_t)(w >> 32) == ticket) break; __builtin_amdgcn_s_sleep(2); } while (++spins < 20000000u); // (bounded: a lost workgroup must not hang the device)
			if ((uint32_t)(w >> 32) != ticket) { flag = 1u; w = 0; if (error) atomicExch(error, 1u); }
			if ((uint32_t)w & CHAIN_POISON) flag = 1u;
			v = (uint32_t)w & ~CHAIN_POISON;
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); flag |= __shfl_xor(flag, off, 64); }
		if (tid == 0) { sh[0] = v; sh[1] = flag; }
	}
	__syncthreads();
	const uint32_t r = sh[0];
	bad = sh[1] != 0u;
	__syncthreads();
	return r;
}
// Round 4: ONE exchange instead of four. Which rays keep their samples depends on the offsets (a ray is dropped if it would overflow max_samples,
// testbed_nerf.cu:1348-1355), so rounds 1-3 exchanged the sample sums first and the three dependent sums afterwards, one after the other -- and every
// exchange is a returning atomic that queues behind the gradient scatter's backlog at the memory side (21 us for 4 tiles beside the scatter, 42 for 23).
// A tile now publishes its sample sum together with the two sums it would have if none of its rays overflowed (rays with samples, first-round samples;
// the kept samples then are the sample sum): three words of one 32-byte block, written by three lanes of one instruction and polled with three atomics
// in flight. A tile whose last offset stays within max_samples has no overflow in or in front of it and is done; only tiles at or behind the first
// overflowing tile (a batch that marched more than 16 x the target: never in a converged run) exchange their true sums a second time (`words2`), taking
// the optimistic sums of the tiles in front of the overflow as they are.
__global__ __launch_bounds__(SCAN_WG) void k_scan_rays_chain(const ScanChainArgs a) {
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t sh[8];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, tile = blockIdx.x;
	const uint32_t i0 = tile * SCAN_TILE + tid * SCAN_EPT;
	uint32_t st[SCAN_EPT], mine = 0, npos = 0, f1 = 0;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) { st[e] = i0 + e < a.n ? a.steps[i0 + e] : 0u; mine += st[e]; npos += st[e] > 0 ? 1u : 0u; f1 += min(st[e], a.k1); }
	uint32_t total, t_pos, t_f1;
	const uint32_t excl = block_exclusive_scan<SCAN_NW>(mine, lane, wave, wsum, total);
	uint32_t e0 = block_exclusive_scan<SCAN_NW>(npos, lane, wave, wsum, t_pos);
	uint32_t e2 = block_exclusive_scan<SCAN_NW>(f1, lane, wave, wsum, t_f1);
	// exchange A (wavefront 0: lane q polls tile q < tile)
	uint32_t fa[3] = {0, 0, 0}, incl_front = 0, flagA = 0; // wavefront 0 keeps the front tiles' sums for exchange B
	if (tid < 3) chain_put(a.words + tile * 4 + tid, ((unsigned long long)a.ticket << 32) | (tid == 0 ? total : tid == 1 ? t_pos : t_f1), a.plain != 0u);
	if (tid < 64) {
		if (tid < tile) {
			unsigned long long w0, w1, w2;
			uint32_t spins = 0;
			bool got;
			do {
				w0 = chain_get(a.words + tid * 4 + 0, a.plain != 0u); w1 = chain_get(a.words + tid * 4 + 1, a.plain != 0u); w2 = chain_get(a.words + tid * 4 + 2, a.plain != 0u);
				got = (uint32_t)(w0 >> 32) == a.ticket && (uint32_t)(w1 >> 32) == a.ticket && (uint32_t)(w2 >> 32) == a.ticket;
				if (got) break;
				__builtin_amdgcn_s_sleep(2);
			} while (++spins < 20000000u); // (bounded: a lost workgroup must not hang the device; see chain_prefix)
			if (!got) { flagA = 1u; w0 = w1 = w2 = 0; if (a.error) atomicExch(a.error, 1u); }
			if ((uint32_t)w0 & CHAIN_POISON) flagA = 1u;
			fa[0] = (uint32_t)w0 & ~CHAIN_POISON; fa[1] = (uint32_t)w1; fa[2] = (uint32_t)w2;
		}
		incl_front = wave_inclusive_scan(fa[0], lane);
		uint32_t r0 = fa[0], r1 = fa[1], r2 = fa[2], fl = flagA;
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) { r0 += __shfl_xor(r0, off, 64); r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64); fl |= __shfl_xor(fl, off, 64); }
		if (tid == 0) { sh[0] = r0; sh[1] = r1; sh[2] = r2; sh[3] = fl; }
		if (tid == 0 && fl) atomicOr(a.words + tile * 4, (unsigned long long)CHAIN_POISON); // best effort for the tiles that poll later; the host word `error` is what reports it
	}
	__syncthreads();
	const uint32_t tile_base = sh[0];
	uint32_t p0 = sh[1], p1 = tile_base, p2 = sh[2];
	bool bad = sh[3] != 0u;
	uint32_t t0 = t_pos, t1 = total, t2 = t_f1;
	uint32_t run = tile_base + excl;
	uint32_t okmask = 0;
	const bool overflow = tile_base + total > a.max_samples; // uniform over the workgroup; true for every tile behind the first such tile too
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) {
		if (i0 + e < a.n) a.base[i0 + e] = run;
		const bool ok = st[e] > 0 && run + st[e] <= a.max_samples; // testbed_nerf.cu:1348-1355
		run += st[e];
		if (ok) okmask |= 1u << e;
	}
	if (overflow) { // exchange B: the true sums of the tiles at and behind the first overflow
		uint32_t v[3] = {0, 0, 0};
#pragma unroll
		for (uint32_t e = 0; e < SCAN_EPT; ++e) if ((okmask >> e) & 1u) { v[0] += 1u; v[1] += st[e]; v[2] += min(st[e], a.k1); }
		__syncthreads(); // sh is rewritten
		e0 = block_exclusive_scan<SCAN_NW>(v[0], lane, wave, wsum, t0);
		e2 = block_exclusive_scan<SCAN_NW>(v[2], lane, wave, wsum, t2);
		(void)block_exclusive_scan<SCAN_NW>(v[1], lane, wave, wsum, t1);
		if (tid < 3) chain_put(a.words2 + tile * 4 + tid, ((unsigned long long)a.ticket << 32) | (tid == 0 ? (t0 | (bad ? CHAIN_POISON : 0u)) : tid == 1 ? t1 : t2), a.plain != 0u);
		if (tid < 64) {
			uint32_t fb[3] = {fa[1], fa[0], fa[2]}, fl = 0; // a tile in front of the overflow: all of its rays with samples are kept
			if (tid < tile && incl_front > a.max_samples) {
				unsigned long long w0, w1, w2;
				uint32_t spins = 0;
				bool got;
				do {
					w0 = chain_get(a.words2 + tid * 4 + 0, a.plain != 0u); w1 = chain_get(a.words2 + tid * 4 + 1, a.plain != 0u); w2 = chain_get(a.words2 + tid * 4 + 2, a.plain != 0u);
					got = (uint32_t)(w0 >> 32) == a.ticket && (uint32_t)(w1 >> 32) == a.ticket && (uint32_t)(w2 >> 32) == a.ticket;
					if (got) break;
					__builtin_amdgcn_s_sleep(2);
				} while (++spins < 20000000u);
				if (!got) { fl = 1u; w0 = w1 = w2 = 0; if (a.error) atomicExch(a.error, 1u); }
				if ((uint32_t)w0 & CHAIN_POISON) fl = 1u;
				fb[0] = (uint32_t)w0 & ~CHAIN_POISON; fb[1] = (uint32_t)w1; fb[2] = (uint32_t)w2;
			}
			uint32_t r0 = fb[0], r1 = fb[1], r2 = fb[2];
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) { r0 += __shfl_xor(r0, off, 64); r1 += __shfl_xor(r1, off, 64); r2 += __shfl_xor(r2, off, 64); fl |= __shfl_xor(fl, off, 64); }
			if (tid == 0) { sh[4] = r0; sh[5] = r1; sh[6] = r2; sh[7] = fl; }
		}
		__syncthreads();
		p0 = sh[4]; p1 = sh[5]; p2 = sh[6];
		bad = bad || sh[7] != 0u;
	}
	uint32_t srun = p0 + e0, frun = p2 + e2;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) {
		const bool ok = (okmask >> e) & 1u;
		if (i0 + e < a.n) {
			a.slot[i0 + e] = ok ? srun : 0xffffffffu;
			if (a.k1) a.base1[i0 + e] = frun;
		}
		srun += ok ? 1u : 0u;
		frun += ok ? min(st[e], a.k1) : 0u;
	}
	if (tile == gridDim.x - 1 && tid == 0) {
		a.counters[0] = bad ? 0u : tile_base + total; a.counters[2] = bad ? 0u : p0 + t0; a.counters[3] = bad ? 0u : p1 + t1; // bad: nothing is evaluated, the step reports no samples
		a.fwd_counts[0] = bad ? 0u : p2 + t2; a.fwd_counts[1] = 0; a.fwd_counts[2] = 0; a.fwd_counts[3] = 0;
	}
}

// The compaction offsets (k_scan_compact_sums + k_scan_compact_offsets) the same way, one exchange: on the critical stream of the large-batch regime.
__global__ __launch_bounds__(SCAN_WG) void k_scan_compact_chain(const uint32_t n, const uint32_t* __restrict__ ncomp, uint32_t* __restrict__ cbase, uint32_t* __restrict__ counters,
                                                             unsigned long long* __restrict__ words, const uint32_t ticket, uint32_t* __restrict__ error) {
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t sh[2];
	bool bad = false;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, tile = blockIdx.x;
	const uint32_t i0 = tile * SCAN_TILE + tid * SCAN_EPT;
	uint32_t v[SCAN_EPT], mine = 0;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) { v[e] = i0 + e < n ? ncomp[i0 + e] : 0u; mine += v[e]; }
	uint32_t total;
	const uint32_t excl = block_exclusive_scan<SCAN_NW>(mine, lane, wave, wsum, total);
	const uint32_t tile_base = chain_prefix(words, 0, tile, ticket, total, tid, sh, error, bad);
	uint32_t run = tile_base + excl;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) { if (i0 + e < n) cbase[i0 + e] = run; run += v[e]; }
	if (tile == gridDim.x - 1 && tid == 0) counters[1] = bad ? 0u : tile_base + total; // numsteps_counter_compacted (bad: the step reports no samples)
}

// ---------------------------------------------------------------------------------------------
// K8: loss + output gradients (testbed_nerf.cu:1396-2097)
// ---------------------------------------------------------------------------------------------

struct RayLoss { // pass 1 -> pass 2
	uint32_t n_comp;
	float rgb_ray[4];
	float weight_sum_raw;
	float T_resume; // transmittance after the samples pass 1 has consumed (two-round evaluation: phase 1 continues from it)
	float ek_resume; // the eikonal sum over those samples (formed in pass 1 when it leaves chain records)
	float rgbtarget[4];
	float light[3];
	float dir[3];
	float mask_certainty, mask_gt;
};

struct LossArgs {
	uint32_t n_rays, n_rays_global, ray_offset, n_rays_total, n_images, B;
	Pcg32 rng;
	SceneAabb A;
	LossFlags F;
	float light_dirs[9];
	const ViewDev* views;
	const uint32_t* counters;   // [2] = rays kept
	const uint32_t* ray_indices;
	const float* rays;
	uint32_t* numsteps;
	const float* coords;
	const half_t* mlp_out;
	RayLoss* ray_loss;
	uint32_t* ncomp;  // [n_rays]
	uint32_t* cbase;  // [n_rays]
	float* coords_compacted;
	uint32_t* src_slot; // optional [B]: the marched-sample slot of every compacted sample (k_rgb_fwd_bwd reads that slot's colour-MLP input row, written by the network evaluation)
	half_t* dloss;
	float *loss, *ek_loss, *mask_loss;
	const float* ray_const; // per-ray constants precomputed by k_march_write (null: compute them here)
	// two-round network evaluation (cap = 0xffffffff: single round)
	uint32_t cap;          // samples per ray evaluated in round 1
	uint32_t phase;        // 0: all rays, at most `cap` samples each; 1: only the rays round 1 could not finish, all their samples
	uint32_t* unfinished;  // [n_rays] list of the rays round 1 left unsettled (count in fwd_counts[3])
	uint32_t* idx2;        // sample slots still to evaluate (round 2)
	uint32_t* fwd_counts;  // [0] = samples of round 1; 8-byte pair [2] = entries of idx2, [3] = entries of `unfinished`
	// chain records (round 4): pass 1 leaves the running values right after every sample it composites -- {weight, T, weight sum, rgb[0]}, {rgb[1..3], eikonal sum} at
	// CHAIN_REC_FLOATS floats per marched-sample slot -- and pass 2 reads them instead of replaying the recurrence (the same function on the same inputs: same bits)
	float* chain_rec;      // null: pass 2 replays
	// pass 2 in two launches for batches of many short rays (k_loss_pass2_rays + k_loss_pass2_samples): per kept ray 16 floats, per compacted sample its ray and its marched slot
	float* ray_grad;
	uint32_t *ray_of, *slot_of;
	// ... which also pad the compacted batch (fill_rollover*, common_device.h:514-535) and -- in the training step -- reduce and publish the step's loss sums
	// (what k_rollover / k_reduce_losses_rollover do behind the one-launch forms): a kernel boundary on the critical stream costs 6-9 us here
	// ... and form the compaction offsets themselves (k_scan_compact* of the one-launch forms): scan_words != null
	unsigned long long* scan_words; // [64]: ticket << 32 | kept samples of a 4096-ray tile, published by the tile's first workgroup
	uint32_t scan_ticket;
	uint32_t* scan_error;  // mapped host word: a wait gave up
	uint32_t* scan_total;  // &counters[1]
	double* wg_partial;    // [workgroups of k_loss_pass2_rays][3]: sums of the three loss rows over the workgroup's 16 rays
	double* red_out;       // null: no reduction (stage API); else the device block of reduce_losses_body
	double* red_host_out;  // its pinned twin
	uint32_t red_host_seq;
};
constexpr uint32_t CHAIN_REC_FLOATS = 8;

__device__ __forceinline__ void albedo_from_output(const LossFlags& F, const half_t* __restrict__ o, float albedo[4]) { // testbed_nerf.cu:1614-1639
	if (F.apply_no_albedo) { albedo[0] = albedo[1] = albedo[2] = 1.f; albedo[3] = 0.f; return; }
	float a[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) a[k] = logistic(h2f(o[k]));
	albedo[0] = a[0]; albedo[1] = a[1]; albedo[2] = a[2];
	if (F.apply_rgbplus) {
		if (F.apply_L2) albedo[3] = sqrtf(fmaxf(0.0f, 3 - a[0] * a[0] - a[1] * a[1] - a[2] * a[2]));
		else albedo[3] = 3 - fabsf(a[0]) - fabsf(a[1]) - fabsf(a[2]);
	} else albedo[3] = 0.f;
}

struct AlphaTerms { float inv_s, sdf_value, true_cos, iter_cos, est_next, p_div_c, alpha; float g[3]; };

__device__ __forceinline__ AlphaTerms alpha_terms(const half_t* __restrict__ o, const float dt, const float dir[3], const float cos_anneal_ratio) { // testbed_nerf.cu:1652-1677
	AlphaTerms a;
	a.inv_s = expf(h2f((half_t)10.f * o[7]));
	a.sdf_value = h2f(o[3]);
	a.g[0] = h2f(o[4]); a.g[1] = h2f(o[5]); a.g[2] = h2f(o[6]);
	a.true_cos = (dir[0] * a.g[0] + dir[1] * a.g[1] + dir[2] * a.g[2]);
	const float r1 = (float)(-a.true_cos * 0.5 + 0.5);
	const float relu1 = relu(r1);
	const float relu2 = relu(-a.true_cos);
	a.iter_cos = (float)-(relu1 * (1.0 - cos_anneal_ratio) + relu2 * cos_anneal_ratio);
	a.est_next = (float)(a.sdf_value + a.iter_cos * dt * 0.5);
	const float est_prev = (float)(a.sdf_value - a.iter_cos * dt * 0.5);
	const float next_cdf = logistic(a.est_next * a.inv_s);
	const float prev_cdf = logistic(est_prev * a.inv_s);
	const float p = prev_cdf - next_cdf;
	const float cc = prev_cdf;
	a.p_div_c = (p + 1e-5f) / (cc + 1e-5f);
	a.alpha = fminf(fmaxf(a.p_div_c, 0.0f), 1.0f);
	return a;
}

__device__ __forceinline__ float bcast(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }
// v of lane q of my group of LR lanes (LR = 64: q is wave-uniform; LR = 16: q may differ between the four rows of the wavefront)
template <int LR>
__device__ __forceinline__ float group_read(const float v, const int q, const int lane64) {
	if (LR == 64) return bcast(v, q);
	return __shfl(v, (lane64 & ~(LR - 1)) + q, 64);
}

__device__ __forceinline__ void load_out16(const half_t* __restrict__ p, half_t o[16]) {
	const h8* src = reinterpret_cast<const h8*>(p);
	const h8 o0 = src[0], o1 = src[1];
#pragma unroll
	for (int j = 0; j < 8; ++j) { o[j] = o0[j]; o[8 + j] = o1[j]; }
}

// The loss kernel's per-ray targets from the two texels (testbed_nerf.cu:1500-1592): target normal (sRGB-encoded, y / z flipped, normalised), target albedo (+ its rgb+
// fourth channel), the light of the step in the camera frame (optionally rotated onto the target normal, apply_light_opti) and in the world frame, the shading target and
// rgbtarget. Every Eigen reduction as Eigen evaluates it (esum3). tests/golden/float_fixtures.json runs the kernel's own statements.
__device__ __forceinline__ void ray_targets(const LossFlags& F, const float (&xform)[12], const float (&tex_normal)[4], const float (&tex_albedo)[4], const float* __restrict__ light_dirs,
                                            const int random_light, float (&rgbtarget)[4], float (&light)[3]) {
	const float exposure_scale = expf(0.6931471805599453f * 0.f);
	float nv[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) nv[k] = linear_to_srgb(exposure_scale * tex_normal[k]) * 2.0f - 1.0f;
	nv[1] *= -1; nv[2] *= -1;
	{ const float nn = sqrtf(esum3(nv[0] * nv[0], nv[1] * nv[1], nv[2] * nv[2])); nv[0] /= nn; nv[1] /= nn; nv[2] /= nn; } // .matrix().norm()
	float albedo_value[4];
	if (F.apply_no_albedo) { albedo_value[0] = albedo_value[1] = albedo_value[2] = 1.f; albedo_value[3] = 0.f; }
	else {
		float al[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) al[k] = linear_to_srgb(exposure_scale * tex_albedo[k]);
		albedo_value[0] = al[0]; albedo_value[1] = al[1]; albedo_value[2] = al[2];
		if (F.apply_rgbplus) {
			if (F.apply_L2) albedo_value[3] = sqrtf(fmaxf(0.0f, 3 - al[0] * al[0] - al[1] * al[1] - al[2] * al[2]));
			else albedo_value[3] = 3 - fabsf(al[0]) - fabsf(al[1]) - fabsf(al[2]);
		} else albedo_value[3] = 0.f;
	}
	float Ld[9];
#pragma unroll
	for (int k = 0; k < 9; ++k) Ld[k] = light_dirs[k];
	if (F.apply_light_opti) { // testbed_nerf.cu:1563-1581
		float k3[3] = {-nv[1], nv[0], 0.f};
		const float kn = sqrtf(esum3(k3[0] * k3[0], k3[1] * k3[1], k3[2] * k3[2]));
		k3[0] /= kn; k3[1] /= kn; k3[2] /= kn;
		const float cos_theta = nv[2];
		const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
		const float K[9] = {0, -k3[2], k3[1], k3[2], 0, -k3[0], -k3[1], k3[0], 0};
		float Rm[9];
		for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q)
			Rm[r * 3 + q] = cos_theta * (r == q ? 1.f : 0.f) + sin_theta * K[r * 3 + q] + (1 - cos_theta) * (k3[r] * k3[q]);
		float outm[9];
		for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q)
			outm[r * 3 + q] = esum3((-Rm[r * 3 + 0]) * Ld[0 * 3 + q], (-Rm[r * 3 + 1]) * Ld[1 * 3 + q], (-Rm[r * 3 + 2]) * Ld[2 * 3 + q]); // -R * light_directions
		for (int k = 0; k < 9; ++k) Ld[k] = outm[k];
	}
	const float light_cam[3] = {Ld[0 * 3 + random_light], Ld[1 * 3 + random_light], Ld[2 * 3 + random_light]};
#pragma unroll
	for (int r = 0; r < 3; ++r) light[r] = esum3(xform[r * 4 + 0] * light_cam[0], xform[r * 4 + 1] * light_cam[1], xform[r * 4 + 2] * light_cam[2]); // Rt * light_cam
	float shading_target = esum3(nv[0] * light_cam[0], nv[1] * light_cam[1], nv[2] * light_cam[2]);                                                          // .dot(light_cam)
	if (F.apply_relu) shading_target = shading_target > 0.f ? shading_target : 0.f;
#pragma unroll
	for (int k = 0; k < 4; ++k) rgbtarget[k] = albedo_value[k] * shading_target;
}

// Per-ray constants of the loss kernel (testbed_nerf.cu:1485-1593): pixel, target normal, light triplet, shading target.
struct RayConstIn { Pcg32 rng; uint32_t ray_offset, n_rays_global, n_rays_total, n_images; const ViewDev* views; LossFlags F; const float* light_dirs; };

template <typename RayOut>
__device__ __forceinline__ void ray_constants_core(const RayConstIn& a, const uint32_t ray_idx, RayOut& R) {
	const uint32_t gi = a.ray_offset + ray_idx;
	Pcg32 rng = a.rng;
	rng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY);
	const uint32_t img = image_idx(gi, a.n_rays_global, a.n_rays_total, a.n_images);
	const ViewDev m = a.views[img];
	float xy[2];
	random_image_pos(rng, m.width, m.height, a.F.snap != 0, xy);
	float tex_albedo[4], tex_normal[4];
	read_rgba(xy, m, m.albedo, tex_albedo);
	read_rgba(xy, m, m.normal, tex_normal);
	Pcg32 lrng = a.rng; // deterministic light pick: draw #7 of the ray's stream (the reference seeds curand with clock64())
	lrng.advance((int64_t)gi * N_MAX_RANDOM_SAMPLES_PER_RAY + 7);
	const int random_light = (int)(lrng.next_uint() % 3u);
	ray_targets(a.F, m.xform, tex_normal, tex_albedo, a.light_dirs, random_light, R.rgbtarget, R.light);
	R.mask_certainty = (float)(tex_albedo[3] > 0.99);
	R.mask_gt = (float)(tex_normal[3] > 0.99);
}

__device__ __forceinline__ void ray_constants(const LossArgs& a, const uint32_t i, RayLoss& R) {
	if (a.ray_const) { // precomputed beside the march
		const float* q = a.ray_const + (size_t)i * RAY_CONST_FLOATS;
#pragma unroll
		for (int k = 0; k < 4; ++k) R.rgbtarget[k] = q[k];
#pragma unroll
		for (int k = 0; k < 3; ++k) R.light[k] = q[4 + k];
		R.mask_certainty = q[7]; R.mask_gt = q[8];
		return;
	}
	RayConstIn in;
	in.rng = a.rng; in.ray_offset = a.ray_offset; in.n_rays_global = a.n_rays_global; in.n_rays_total = a.n_rays_total; in.n_images = a.n_images;
	in.views = a.views; in.F = a.F; in.light_dirs = a.light_dirs;
	ray_constants_core(in, a.ray_indices[i], R);
}

// Second pass of the reference's kernel (testbed_nerf.cu:1366-1380) without re-marching: LR lanes per ray expand the t values
// recorded by the counting pass into NerfCoordinates (pos = o + t*dir is the same expression the march evaluated).
// LR lanes per ray: 64 while rays are few and long (early training: ~30 marched samples per ray), 16 once the batch has grown
// to ~100 k rays with ~8 samples each (a wavefront per ray would leave 7 of 8 lanes idle).
// The per-ray constants of the loss (pixel fetches, two RNG jumps, sRGB, the light triplet: ~2000 instructions) are worked out by
// ONE thread per ray -- the first WGS / LR threads of the workgroup, one for each of its rays -- and not by every
// lane of the ray's group (that was most of this kernel: 65 -> see DESIGN.md section 6).
// WGS (round 6): 256 threads per workgroup. The 1024 of rounds 2-5 dated from the per-ray constants living here; a 16-wavefront workgroup waits for 16 wave slots to fall free at
// once beside the gradient scatter's short workgroups -- 13 us of writing took 60-160 us there (profiles/r06_timeline_*), as k_dw_finish had in round 4. No barrier, no LDS: any size gives the same stores.
template <int LR, int WGS = 256>
__global__ __launch_bounds__(WGS) void k_march_write(const MarchArgs a) {
	constexpr uint32_t RAYS = WGS / LR;
	const bool head = a.part != 2, rest = a.part != 1;
	if (a.ray_const && rest && threadIdx.x < RAYS) { // thread t: the constants of the workgroup's ray t
		const uint32_t i = blockIdx.x * RAYS + threadIdx.x;
		const uint32_t s = i < a.n_rays ? a.slot[i] : 0xffffffffu;
		if (s != 0xffffffffu) {
			RayConstIn in;
			in.rng = a.rng; in.ray_offset = a.ray_offset; in.n_rays_global = a.n_rays_global; in.n_rays_total = a.n_rays_total; in.n_images = a.n_images;
			in.views = a.views; in.F = a.F; in.light_dirs = a.light_dirs;
			struct { float rgbtarget[4], light[3], mask_certainty, mask_gt; } rc;
			ray_constants_core(in, i, rc);
			float* q = a.ray_const + (size_t)s * RAY_CONST_FLOATS;
#pragma unroll
			for (int k = 0; k < 4; ++k) q[k] = rc.rgbtarget[k];
#pragma unroll
			for (int k = 0; k < 3; ++k) q[4 + k] = rc.light[k];
			q[7] = rc.mask_certainty; q[8] = rc.mask_gt;
		}
	}
	const uint32_t i = blockIdx.x * RAYS + threadIdx.x / LR;
	const uint32_t lane = threadIdx.x & (LR - 1);
	if (i >= a.n_rays) return;
	const uint32_t s = a.slot[i];
	if (s == 0xffffffffu) return;
	const float* st = a.setup + (size_t)i * 8;
	const Vec3 o = {st[0], st[1], st[2]}, dir = {st[3], st[4], st[5]};
	const uint32_t steps = a.steps[i], base = a.base[i];
	if (lane == 0 && rest) {
		a.ray_indices[s] = i;
		float* ro = a.rays + (size_t)s * 6;
		ro[0] = o.x; ro[1] = o.y; ro[2] = o.z;
		ro[3] = a.d_unnorm[(size_t)i * 3 + 0]; ro[4] = a.d_unnorm[(size_t)i * 3 + 1]; ro[5] = a.d_unnorm[(size_t)i * 3 + 2];
		a.numsteps[(size_t)s * 2 + 0] = steps;
		a.numsteps[(size_t)s * 2 + 1] = base;
	}
	if (a.k1 && head) {
		const uint32_t b1 = a.base1[i];
		for (uint32_t j = lane; j < min(steps, a.k1); j += LR) a.idx1[b1 + j] = base + j;
	}
	// part 1: samples [0, k1) of the ray; part 2: [k1, steps) (every lane keeps the samples it has in the one-launch form)
	const uint32_t j_lo = a.part == 2 ? min(steps, a.k1) : 0u, j_hi = a.part == 1 ? min(steps, a.k1) : steps;
	if (j_lo >= j_hi) return;
	const Vec3 wd = warp_direction(dir);
	const float* tt = a.ray_t + (size_t)i * RNB_MAX_STEPS;
	float* co = a.coords + (size_t)base * 7;
	for (uint32_t j = j_lo / LR * LR + lane; j < j_hi; j += LR) {
		if (j < j_lo) continue;
		const float t = tt[j];
		const Vec3 pos = o + t * dir;
		const float dt = calc_dt(t, a.A.cone_angle);
		const Vec3 wp = warp_position(a.A, pos);
		float* q = co + (size_t)j * 7;
		q[0] = wp.x; q[1] = wp.y; q[2] = wp.z; q[3] = warp_dt(dt); q[4] = wd.x; q[5] = wd.y; q[6] = wd.z;
	}
}


// The per-ray constants of the loss as a launch of their own, one thread per KEPT ray (round 4). Inside k_march_write the first RAYS threads of a 1024-thread
// workgroup work them out (~2000 dependent instructions, pixel fetches) while its other wavefronts have long finished: the workgroup holds its slots for that
// chain, and the kernel took 50-80 us beside the optimizer for 15-30 us of writing. Needs RAY_INDICES (k_march_write).
__global__ __launch_bounds__(64) void k_ray_constants(const MarchArgs a, float* __restrict__ ray_const) {
	const uint32_t s = blockIdx.x * 64 + threadIdx.x;
	if (s >= min(a.counters[2], a.n_rays)) return;
	RayConstIn in;
	in.rng = a.rng; in.ray_offset = a.ray_offset; in.n_rays_global = a.n_rays_global; in.n_rays_total = a.n_rays_total; in.n_images = a.n_images;
	in.views = a.views; in.F = a.F; in.light_dirs = a.light_dirs;
	struct { float rgbtarget[4], light[3], mask_certainty, mask_gt; } rc;
	ray_constants_core(in, a.ray_indices[s], rc);
	float* q = ray_const + (size_t)s * RAY_CONST_FLOATS;
#pragma unroll
	for (int k = 0; k < 4; ++k) q[k] = rc.rgbtarget[k];
#pragma unroll
	for (int k = 0; k < 3; ++k) q[4 + k] = rc.light[k];
	q[7] = rc.mask_certainty; q[8] = rc.mask_gt;
}

// The sequential part of the compositing loop (testbed_nerf.cu:1608-1697) for up to 64 samples whose per-sample terms sit
// one per lane: the recurrence runs through chain.cuh (every lane ends with the running values right after its sample, in
// the reference's operation order, so the rounding -- and with it the early stop -- is identical); the loop's exit test
// "T < 1e-4 before sample q" then is one ballot over the lanes, and the running values are read from the lane in front
// of the first such sample. Returns true if the ray terminated inside these samples.
template <bool NO_ALBEDO, int LR>
__device__ __forceinline__ bool composite_replay(const int cnt, const int lane, const int lane64, const float alpha, const float shading, const float (&albedo)[4],
                                                 float& T, float (&rgb)[4], float& weight_sum, uint32_t& n, float* __restrict__ rec, const float ekterm, float& ek) {
	const ChainState s = replay_chain<NO_ALBEDO, LR>(cnt, alpha, shading, albedo, ekterm, T, weight_sum, rgb, ek);
	// the transmittance the loop tests before it takes sample `lane`: the lane in front's (wave_shr:1 / row_shr:1), lane 0 keeps the incoming one
	const float T_before = LR == 64 ? __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(T), __float_as_int(s.T), 0x138, 0xf, 0xf, false))
	                                : __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(T), __float_as_int(s.T), 0x111, 0xf, 0xf, false));
	uint64_t stop = __builtin_amdgcn_ballot_w64(lane < cnt && T_before < 1e-4f);
	if (LR != 64) stop = (stop >> (lane64 & ~(LR - 1))) & ((1ull << LR) - 1ull); // my group's lanes
	const int taken = stop ? (int)__builtin_ctzll(stop) : cnt; // samples composited here
	n += (uint32_t)taken;
	if (rec && lane < taken) { // the chain records of the samples this ray keeps (rec: the chunk's first record)
		f4* q = reinterpret_cast<f4*>(rec + (size_t)lane * CHAIN_REC_FLOATS);
		q[0] = f4{s.w, s.T, s.ws, s.rgb[0]};
		q[1] = f4{s.rgb[1], s.rgb[2], s.rgb[3], s.ek};
	}
	const int last = max(taken - 1, 0);
	const float T1 = group_read<LR>(s.T, last, lane64), w1 = group_read<LR>(s.ws, last, lane64), r0 = group_read<LR>(s.rgb[0], last, lane64);
	if (taken > 0) { T = T1; weight_sum = w1; rgb[0] = r0; }
	if (rec) { const float e1 = group_read<LR>(s.ek, last, lane64); if (taken > 0) ek = e1; } // (uniform: every ray of the launch has records or none has)
	if (!NO_ALBEDO) {
#pragma unroll
		for (int k = 1; k < 4; ++k) { const float rk = group_read<LR>(s.rgb[k], last, lane64); if (taken > 0) rgb[k] = rk; }
	}
	return stop != 0ull;
}

// Pass 1 of the reference kernel (testbed_nerf.cu:1608-1697), LR lanes per ray (64: one wavefront per ray; 16: four rays per
// wavefront once the batch holds many short rays): the per-sample terms (alpha, shading) are evaluated by the ray's lanes at
// once; the transmittance recurrence and the early stop at T < 1e-4 then run in the reference's sequential order (identical fp32
// rounding) across the lanes (chain.cuh). `valid` = the ray exists and is kept; every lane of the wavefront stays active.
// Returns (phase 0 of the two-round evaluation) how many samples of the ray are still to be evaluated, 0 if the ray is settled;
// *base_out = the ray's first sample slot.
template <int LR>
__device__ __forceinline__ uint32_t loss_pass1_ray(const LossArgs& a, const uint32_t i, const bool valid, const int lane, const int lane64, uint32_t* base_out = nullptr) {
	const uint32_t numsteps_all = valid ? a.numsteps[(size_t)i * 2 + 0] : 0u;
	const uint32_t numsteps = a.phase == 0 ? min(numsteps_all, a.cap) : numsteps_all;
	const uint32_t base = valid ? a.numsteps[(size_t)i * 2 + 1] : 0u;
	const float* coords_in = a.coords + (size_t)base * 7;
	const half_t* net = a.mlp_out + (size_t)base * 16;
	RayLoss R;
	ray_constants(a, valid ? i : 0u, R);
	float dir[3];
	{ // BENT_DIR (testbed_nerf.cu:1645-1650): the direction the network echoed for the ray's first sample
		half_t o0[16];
		load_out16(net, o0);
		const Vec3 dv = normalized(unwarp_direction(v3(h2f(o0[8]), h2f(o0[9]), h2f(o0[10]))));
		dir[0] = dv.x; dir[1] = dv.y; dir[2] = dv.z;
	}
	float T = 1.f;
	const float EPSILON = 1e-4f;
	float rgb_ray[4] = {0, 0, 0, 0};
	float weight_sum = 0.f;
	uint32_t n = 0;
	bool done = false;
	uint32_t c_begin = 0;
	float ek_run = 0.f; // the eikonal sum of pass 2 (testbed_nerf.cu:1902-1904), formed here in the same order when chain records are left
	if (a.phase == 1) { // continue where phase 0 stopped (it consumed exactly `cap` samples without terminating)
		const RayLoss P = a.ray_loss[valid ? i : 0u];
		T = P.T_resume; weight_sum = P.weight_sum_raw; n = P.n_comp; c_begin = a.cap; ek_run = P.ek_resume;
#pragma unroll
		for (int k = 0; k < 4; ++k) rgb_ray[k] = P.rgb_ray[k];
	}
	for (uint32_t c0 = c_begin; ; c0 += LR) {
		const bool act = c0 < numsteps && !done; // this ray still has samples to composite (LR = 64: the same for the whole wavefront)
		if (!__any(act)) break;
		const uint32_t j = c0 + lane;
		float alpha = 0.f, shading = 0.f, albedo[4] = {1.f, 1.f, 1.f, 0.f}, ekterm = 0.f;
		if (act && j < numsteps) {
			half_t o[16];
			load_out16(net + (size_t)j * 16, o);
			albedo_from_output(a.F, o, albedo);
			const float dt = unwarp_dt(coords_in[(size_t)j * 7 + 3]);
			const AlphaTerms at = alpha_terms(o, dt, dir, 1.0f);
			alpha = at.alpha;
			shading = esum3(at.g[0] * R.light[0], at.g[1] * R.light[1], at.g[2] * R.light[2]); // normal.dot(light)
			if (a.F.apply_relu) shading = shading > 0.f ? shading : 0.f;
			if (a.chain_rec) {
				const float gradient_norm = (float)sqrt((double)(at.g[0] * at.g[0] + at.g[1] * at.g[1] + at.g[2] * at.g[2]) + 1e-6);
				ekterm = (gradient_norm - 1.0f) * (gradient_norm - 1.0f);
			}
		}
		const int cnt = act ? (int)min((uint32_t)LR, numsteps - c0) : 0;
		float* rec = a.chain_rec ? a.chain_rec + ((size_t)base + c0) * CHAIN_REC_FLOATS : nullptr;
		const bool stopped = a.F.apply_no_albedo ? composite_replay<true, LR>(cnt, lane, lane64, alpha, shading, albedo, T, rgb_ray, weight_sum, n, rec, ekterm, ek_run)
		                                         : composite_replay<false, LR>(cnt, lane, lane64, alpha, shading, albedo, T, rgb_ray, weight_sum, n, rec, ekterm, ek_run);
		done = done || stopped;
	}
	if (a.F.apply_no_albedo) { rgb_ray[1] = rgb_ray[0]; rgb_ray[2] = rgb_ray[0]; } // same addends in the same order; channel 3 only ever receives weight * 0
	uint32_t tail = 0;
	if (a.phase == 0 && a.cap != 0xffffffffu) {
		// The samples past the point where the transmittance falls below 1e-4 are never read again (testbed_nerf.cu:1609), so
		// the network is first evaluated on the head of every ray only. A ray is settled if it terminated inside its head, or
		// has no more samples, or would terminate at the very next check; the others queue their tails for round 2 (the caller
		// allots the queue) and are recomputed in phase 1. Same values as a single full pass.
		const bool settled = done || numsteps_all <= a.cap || T < EPSILON;
		if (!settled) tail = numsteps_all - a.cap;
		if (base_out) *base_out = base;
	}
	if (valid && lane == 0) {
		R.n_comp = n;
#pragma unroll
		for (int k = 0; k < 4; ++k) R.rgb_ray[k] = rgb_ray[k];
		R.weight_sum_raw = weight_sum;
		R.T_resume = T;
		R.ek_resume = ek_run;
		R.dir[0] = dir[0]; R.dir[1] = dir[1]; R.dir[2] = dir[2];
		a.ray_loss[i] = R;
		a.ncomp[i] = n;
	}
	return tail;
}

// Phase 0 (all rays; their heads only in the two-round evaluation), LR lanes per ray, 1024 / LR rays per workgroup. The queue of
// round 2 -- sample slots in idx2, rays in `unfinished` -- is allotted ONCE per workgroup with one 64-bit atomic on the
// (entries, rays) pair: returning atomics on one address retire one after the other (≈6 ns each), and two per unsettled ray
// made this kernel 46 us instead of 10 (measured in round 2 by doubling them: +37 us).
constexpr uint32_t LOSS1_WG = 1024;
template <int LR>
__global__ __launch_bounds__(LOSS1_WG) void k_loss_pass1_heads(const LossArgs a) {
	constexpr uint32_t RAYS = LOSS1_WG / LR; // 16 or 64
	__shared__ uint32_t s_tail[RAYS], s_off[RAYS], s_slot[RAYS], s_base[2];
	const uint32_t lane64 = threadIdx.x & 63u, lane = threadIdx.x & (LR - 1), r = threadIdx.x / LR;
	const uint32_t i = blockIdx.x * RAYS + r;
	const bool valid = i < a.n_rays && i < a.counters[2];
	if (i < a.n_rays && !valid && lane == 0) a.ncomp[i] = 0;
	uint32_t base = 0;
	const uint32_t tail = loss_pass1_ray<LR>(a, i, valid, (int)lane, (int)lane64, &base);
	if (a.cap == 0xffffffffu) return; // single round: nothing is ever queued (uniform over the launch)
	if (lane == 0) s_tail[r] = tail;
	__syncthreads();
	if (threadIdx.x < 64) {
		const uint32_t t = lane64 < RAYS ? s_tail[lane64] : 0u;
		const uint32_t u = t ? 1u : 0u;
		const uint32_t it = wave_inclusive_scan(t, lane64), iu = wave_inclusive_scan(u, lane64);
		if (lane64 < RAYS) { s_off[lane64] = it - t; s_slot[lane64] = iu - u; }
		if (lane64 == RAYS - 1 && iu) {
			const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(a.fwd_counts + 2), ((unsigned long long)iu << 32) | it);
			s_base[0] = (uint32_t)old; s_base[1] = (uint32_t)(old >> 32);
		}
	}
	__syncthreads();
	if (tail) {
		const uint32_t off = s_base[0] + s_off[r];
		if (lane == 0) a.unfinished[s_base[1] + s_slot[r]] = i;
		for (uint32_t j = lane; j < tail; j += LR) a.idx2[off + j] = base + a.cap + j;
	}
}

// Phase 1: the rays round 1 left unsettled (long ones), from the list phase 0 built: one wavefront per ray.
__global__ __launch_bounds__(256) void k_loss_pass1(const LossArgs a) {
	const int lane = threadIdx.x & 63;
	const uint32_t n_list = a.fwd_counts[3];
	for (uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6); k < n_list; k += gridDim.x * 4) (void)loss_pass1_ray<64>(a, a.unfinished[k], true, lane, lane);
}

// exclusive scan of ncomp over the kept rays; counters[1] = total (numsteps_counter_compacted)
__global__ __launch_bounds__(1024) void k_scan_compact(const uint32_t n_max, const uint32_t* __restrict__ ncomp, uint32_t* __restrict__ cbase, uint32_t* __restrict__ counters) {
	__shared__ uint32_t wsum[16];
	const uint32_t n = min(counters[2], n_max);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	uint32_t carry = 0;
	uint32_t v_next = tid < n ? ncomp[tid] : 0u;
	for (uint32_t t0 = 0; t0 < n; t0 += 1024) {
		const uint32_t i = t0 + tid;
		const uint32_t v = v_next;
		v_next = i + 1024 < n ? ncomp[i + 1024] : 0u;
		const uint32_t inc = wave_inclusive_scan(v, lane);
		if (lane == 63) wsum[wave] = inc;
		__syncthreads();
		uint32_t pre = 0, tot = 0;
#pragma unroll
		for (uint32_t w = 0; w < 16; ++w) { const uint32_t x = wsum[w]; pre += w < wave ? x : 0u; tot += x; }
		if (i < n) cbase[i] = carry + pre + inc - v;
		carry += tot;
		__syncthreads();
	}
	if (tid == 0) counters[1] = carry;
}

// k_scan_compact with one workgroup per 4096-ray tile (large batches): tile sums, then offsets. ncomp is zero beyond the kept rays.
__global__ __launch_bounds__(SCAN_WG) void k_scan_compact_sums(const uint32_t n, const uint32_t* __restrict__ ncomp, uint32_t* __restrict__ tile_sum) {
	__shared__ uint32_t wsum[16];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t i0 = blockIdx.x * SCAN_TILE + tid * SCAN_EPT;
	uint32_t mine = 0;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) mine += i0 + e < n ? ncomp[i0 + e] : 0u;
	uint32_t total;
	(void)block_exclusive_scan<SCAN_NW>(mine, lane, wave, wsum, total);
	if (tid == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(SCAN_WG) void k_scan_compact_offsets(const uint32_t n, const uint32_t* __restrict__ ncomp, const uint32_t* __restrict__ tile_sum,
                                                               uint32_t* __restrict__ cbase, uint32_t* __restrict__ counters) {
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t sh;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t tile_base = tile_prefix(tile_sum, 1, blockIdx.x, tid, &sh);
	const uint32_t i0 = blockIdx.x * SCAN_TILE + tid * SCAN_EPT;
	uint32_t v[SCAN_EPT], mine = 0;
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) { v[e] = i0 + e < n ? ncomp[i0 + e] : 0u; mine += v[e]; }
	uint32_t total;
	uint32_t run = tile_base + block_exclusive_scan<SCAN_NW>(mine, lane, wave, wsum, total);
#pragma unroll
	for (uint32_t e = 0; e < SCAN_EPT; ++e) { if (i0 + e < n) cbase[i0 + e] = run; run += v[e]; }
	if (blockIdx.x == gridDim.x - 1 && tid == 0) counters[1] = tile_base + total; // numsteps_counter_compacted
}

// Pass 2 (testbed_nerf.cu:1836-2095).
// What a ray contributes to every one of its samples (testbed_nerf.cu:1836-1890): the loss gradient, the mask term, and the ray's three loss values.
struct RayGrad { float grad[4], weight_sum, gradient_weight_sum, light[3], dir[3], rgb_ray[4]; }; // 16 floats
static_assert(sizeof(RayGrad) == 64, "RayGrad is read as four 16-byte words");
__device__ __forceinline__ void pass2_ray_terms(const LossFlags& F, const RayLoss& R, const float gn, RayGrad& G, float& loss_row, float& mask_row) {
	float loss = loss_and_gradient(F.apply_L2 != 0, R.rgbtarget, R.rgb_ray, G.grad);
	if (F.apply_rgbplus) { loss /= 2; for (int k = 0; k < 4; ++k) G.grad[k] /= 2; }
	loss *= R.mask_certainty;
#pragma unroll
	for (int k = 0; k < 4; ++k) G.grad[k] *= R.mask_certainty;
	float weight_sum = R.weight_sum_raw;
	float gradient_weight_sum;
	if (weight_sum >= 1.0 - 1e-4) { weight_sum = (float)(1.0 - 1e-4); gradient_weight_sum = 0.0f; }
	else if (weight_sum <= 1e-4) { weight_sum = 1e-4; gradient_weight_sum = 0.0f; }
	else {
		const float sig = 1.0f / (1.0f + expf(-weight_sum));
		if (F.apply_bce) gradient_weight_sum = ((1 - R.mask_gt) / (1 - weight_sum) - R.mask_gt / weight_sum) * F.mask_loss_weight;
		else gradient_weight_sum = (sig - R.mask_gt) * F.mask_loss_weight;
	}
	G.weight_sum = weight_sum; G.gradient_weight_sum = gradient_weight_sum;
#pragma unroll
	for (int k = 0; k < 3; ++k) { G.light[k] = R.light[k]; G.dir[k] = R.dir[k]; }
#pragma unroll
	for (int k = 0; k < 4; ++k) G.rgb_ray[k] = R.rgb_ray[k];
	loss_row = loss / gn;
	const float sig = 1.0f / (1.0f + expf(-weight_sum));
	if (F.apply_bce) mask_row = -(R.mask_gt * logf(weight_sum) + (1 - R.mask_gt) * logf(1 - weight_sum));
	else mask_row = -(R.mask_gt * logf(sig) + (1 - R.mask_gt) * logf(1 - sig));
}

// dL/d(network output) of one compacted sample (testbed_nerf.cu:1893-2075) from the ray's terms, the sample's network output `o`, its step `dt` and the
// running values of the compositing recurrence right after it: its weight, the transmittance Tj, the weight sum w2, the colour sums rgb2.
// Eigen evaluates `weight * light_albedo * g` and `weight * shading * jac_rgb * g` as fixed-size products of a scaled 3x4 matrix with a 4-vector: every coefficient of the
// scaled matrix first, then a row's four terms as (x0 + x1) + (x2 + x3); the .dot() of two 4-vectors the same way (esum4). `inter` (tests): the float values behind dl.
__device__ __forceinline__ void pass2_sample(const LossFlags& F, const RayGrad& G, const float loss_scale, const half_t (&o)[16], const float dt,
                                             const float my_weight, const float Tj, const float my_w2, const float (&my_rgb2)[4], half_t (&dl)[16], float* inter = nullptr) {
	float albedo[4];
	albedo_from_output(F, o, albedo);
	const float dir[3] = {G.dir[0], G.dir[1], G.dir[2]};
	const AlphaTerms at = alpha_terms(o, dt, dir, 1.0f);
	float shading = esum3(at.g[0] * G.light[0], at.g[1] * G.light[1], at.g[2] * G.light[2]); // normal.dot(light)
	if (F.apply_relu) shading = shading > 0.f ? shading : 0.f;
	const float gradient_norm = (float)sqrt((double)(at.g[0] * at.g[0] + at.g[1] * at.g[1] + at.g[2] * at.g[2]) + 1e-6);
	const float* grad = G.grad;
	const float alpha = at.alpha;
	const float weight = my_weight;
	float suffix[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) suffix[k] = G.rgb_ray[k] - my_rgb2[k];
	float dloss_dn[3]; // weight * (light * albedo^T) * lg.gradient
#pragma unroll
	for (int d = 0; d < 3; ++d)
		dloss_dn[d] = esum4((weight * (G.light[d] * albedo[0])) * grad[0], (weight * (G.light[d] * albedo[1])) * grad[1], (weight * (G.light[d] * albedo[2])) * grad[2], (weight * (G.light[d] * albedo[3])) * grad[3]);
	float J3[3] = {0, 0, 0};
	if (F.apply_rgbplus) {
		if (F.apply_L2) { for (int d = 0; d < 3; ++d) J3[d] = (float)(-2 * albedo[d] / (albedo[3] + 1e-5)); }
		else { for (int d = 0; d < 3; ++d) J3[d] = -sign1(albedo[d]); }
	}
	float drgb[3]; // (weight * shading) * jac_rgb * lg.gradient, jac_rgb = [I | J3]: the zero entries of the scaled matrix are signed zeros and stay in the sums
	const float ws = weight * shading;
#pragma unroll
	for (int d = 0; d < 3; ++d)
		drgb[d] = esum4((ws * (d == 0 ? 1.0f : 0.0f)) * grad[0], (ws * (d == 1 ? 1.0f : 0.0f)) * grad[1], (ws * (d == 2 ? 1.0f : 0.0f)) * grad[2], (ws * J3[d]) * grad[3]);
#pragma unroll
	for (int q = 0; q < 16; ++q) dl[q] = (half_t)0.f;
	const float opti_rgb = F.apply_no_albedo ? 0.0f : 1.0f;
	if (F.apply_no_albedo) { // 0 x loss_scale x (drgb x a factor in (0, 1/4]): a zero with the sign of drgb (round 4: three exp and three divisions per sample for it before)
#pragma unroll
		for (int d = 0; d < 3; ++d) dl[d] = f2h(copysignf(0.f, drgb[d]));
	} else {
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			const float sg = logistic(h2f(o[d]));
			dl[d] = f2h(opti_rgb * loss_scale * (drgb[d] * (sg * (1 - sg))));
		}
	}
	const float sum_weight_suffix = G.weight_sum - my_w2;
	const float dot_term = esum4(grad[0] * (Tj * albedo[0] * shading - suffix[0]), grad[1] * (Tj * albedo[1] * shading - suffix[1]), grad[2] * (Tj * albedo[2] * shading - suffix[2]),
	                             grad[3] * (Tj * albedo[3] * shading - suffix[3])); // lg.gradient.matrix().dot(...)
	const float dloss_dalpha = (float)((dot_term + (G.gradient_weight_sum * (Tj - sum_weight_suffix))) / (1.0f - alpha + 1e-5));
	float dalpha_dE = 0.f, dE_dsdf = 0.f, dE_dinvs = 0.f, dalpha_dEp = 0.f, dEp_dinvs = 0.f, dEp_ditc = 0.f, dE_ditc = 0.f;
	if (!(at.p_div_c <= 0.0f || at.p_div_c >= 1.0f)) { // testbed_nerf.cu:1982-2014
		const float plus_sigmoid_x = at.inv_s * at.iter_cos * dt;
		const float plus_e = expf(plus_sigmoid_x);
		const float e_minus = expf(-at.est_next * at.inv_s);
		dE_dsdf = -at.inv_s * e_minus;
		dE_dinvs = -at.est_next * e_minus;
		const float aa = 1 + e_minus;
		const float bb = 1 + plus_e * e_minus;
		const float cc = (float)(1e-5 + 1 / (1 + plus_e * e_minus));
		const float delta = aa * (bb * bb) * (cc * cc);
		dalpha_dE = -(plus_e / (delta)-1 / (aa * aa * cc));
		dalpha_dEp = -e_minus / (delta);
		dEp_dinvs = plus_e * at.iter_cos * dt;
		dEp_ditc = plus_e * at.inv_s * dt;
		dE_ditc = (float)(-at.inv_s * e_minus * dt * 0.5);
	}
	const float dloss_dinvs = dloss_dalpha * (dalpha_dE * dE_dinvs + dalpha_dEp * dEp_dinvs);
	const float dloss_dvariance = dloss_dinvs * at.inv_s * 10;
	const float d_iter_cos_true_cos = (at.true_cos >= 0) ? 0.0f : 1.0f;
	const float pos_gradient_norm_inv = 1 - 1 / gradient_norm;
	const float dloss_dnormal_norm = dloss_dalpha * (dalpha_dE * dE_ditc + dEp_ditc * dalpha_dEp) * d_iter_cos_true_cos;
	const float dloss_dsdf = dloss_dalpha * dalpha_dE * dE_dsdf;
	dl[3] = f2h(loss_scale * dloss_dsdf);
#pragma unroll
	for (int d = 0; d < 3; ++d) dl[4 + d] = f2h(F.ek_loss_weight * 2 * LOSS_SCALE * pos_gradient_norm_inv * at.g[d]);
	dl[7] = f2h(loss_scale * dloss_dvariance);
#pragma unroll
	for (int d = 0; d < 3; ++d) dl[8 + d] = f2h(loss_scale * (dloss_dn[d] + dloss_dnormal_norm * dir[d]));
	if (inter) {
		for (int d = 0; d < 3; ++d) { inter[d] = drgb[d]; inter[3 + d] = dloss_dn[d]; }
		inter[6] = dloss_dalpha; inter[7] = dloss_dsdf; inter[8] = dloss_dvariance; inter[9] = dloss_dnormal_norm;
	}
}

// One launch, LR lanes per ray (64: one wavefront per ray; 16: four rays per wavefront): lanes own samples. REC: the running values of the recurrence come from
// the chain records pass 1 left (LossArgs::chain_rec), the eikonal sum included; otherwise (rounds 1-3) the recurrence is replayed here (chain.cuh).
// Every lane of the wavefront stays active.
template <int LR, bool REC>
__global__ __launch_bounds__(256) void k_loss_pass2(const LossArgs a) {
	const uint32_t i_raw = blockIdx.x * (256 / LR) + threadIdx.x / LR;
	const int lane = threadIdx.x & (LR - 1), lane64 = threadIdx.x & 63;
	const bool ray_ok = i_raw < a.n_rays && i_raw < a.counters[2];
	const uint32_t i = ray_ok ? i_raw : 0u;
	const RayLoss R = a.ray_loss[i];
	const uint32_t compacted_base = a.cbase[i];
	const uint32_t compacted_numsteps = ray_ok ? min(a.B - min(a.B, compacted_base), R.n_comp) : 0u; // testbed_nerf.cu:1723
	const uint32_t base = a.numsteps[(size_t)i * 2 + 1];
	__builtin_amdgcn_wave_barrier();
	if (ray_ok && lane == 0) {
		a.numsteps[(size_t)i * 2 + 0] = compacted_numsteps;
		a.numsteps[(size_t)i * 2 + 1] = compacted_base;
	}
	const bool ray_live = compacted_numsteps != 0;
	// a kept ray without compacted samples contributes zeros to the three loss rows (the reference's rows are zero-filled before the
	// step, Counters::prepare_for_training_steps testbed_nerf.cu:3527-3529; writing the zeros here spares the step a fill launch in front of the next march)
	if (ray_ok && !ray_live && lane == 0) { a.loss[i] = 0.f; a.ek_loss[i] = 0.f; a.mask_loss[i] = 0.f; }
	if (!__any(ray_live)) return;
	const float* coords_in = a.coords + (size_t)base * 7;
	const half_t* net = a.mlp_out + (size_t)base * 16;
	float* coords_out = a.coords_compacted + (size_t)compacted_base * 7;
	half_t* dloss = a.dloss + (size_t)compacted_base * 16;
	const LossFlags F = a.F;
	const float gn = (float)a.n_rays_global;
	RayGrad G;
	float loss_row, mask_row;
	pass2_ray_terms(F, R, gn, G, loss_row, mask_row);
	if (ray_live && lane == 0) { a.loss[i] = loss_row; a.mask_loss[i] = mask_row; }

	const float loss_scale = LOSS_SCALE / gn; // testbed_nerf.cu:1832
	float rgb_ray2[4] = {0, 0, 0, 0};
	float weight_sum2 = 0.f;
	float T = 1.f;
	float ek = 0.f;
	for (uint32_t c0 = 0; __any(c0 < compacted_numsteps); c0 += LR) {
		const uint32_t j = c0 + lane;
		const bool valid = j < compacted_numsteps;
		half_t o[16];
#pragma unroll
		for (int q = 0; q < 16; ++q) o[q] = (half_t)0.f;
		float dt = MIN_CONE_STEPSIZE;
		if (valid) {
#pragma unroll
			for (int q = 0; q < 7; ++q) coords_out[(size_t)j * 7 + q] = coords_in[(size_t)j * 7 + q];
			if (a.src_slot) a.src_slot[compacted_base + j] = base + j;
			load_out16(net + (size_t)j * 16, o);
			dt = unwarp_dt(coords_in[(size_t)j * 7 + 3]);
		}
		ChainState cs;
		if (REC) {
			cs.w = cs.T = cs.ws = cs.ek = 0.f; cs.rgb[0] = cs.rgb[1] = cs.rgb[2] = cs.rgb[3] = 0.f;
			if (valid) {
				const f4* q = reinterpret_cast<const f4*>(a.chain_rec + ((size_t)base + j) * CHAIN_REC_FLOATS);
				const f4 r0 = q[0];
				cs.w = r0[0]; cs.T = r0[1]; cs.ws = r0[2]; cs.rgb[0] = r0[3];
				if (!F.apply_no_albedo) { const f4 r1 = q[1]; cs.rgb[1] = r1[0]; cs.rgb[2] = r1[1]; cs.rgb[3] = r1[2]; }
			}
		} else {
			// the per-sample inputs of the recurrence, as pass 1 forms them
			float albedo[4] = {1.f, 1.f, 1.f, 0.f};
			float alpha = 0.f, shading = 0.f, gradient_norm = 1.f;
			if (valid) {
				albedo_from_output(F, o, albedo);
				const AlphaTerms at = alpha_terms(o, dt, G.dir, 1.0f);
				alpha = at.alpha;
				shading = esum3(at.g[0] * G.light[0], at.g[1] * G.light[1], at.g[2] * G.light[2]); // normal.dot(light)
				if (F.apply_relu) shading = shading > 0.f ? shading : 0.f;
				gradient_norm = (float)sqrt((double)(at.g[0] * at.g[0] + at.g[1] * at.g[1] + at.g[2] * at.g[2]) + 1e-6);
			}
			const float ekterm = (gradient_norm - 1.0f) * (gradient_norm - 1.0f);
			// the sequential recurrences (chain.cuh): lane q ends with its own weight and the running values right after sample q
			const int cnt = c0 < compacted_numsteps ? (int)min((uint32_t)LR, compacted_numsteps - c0) : 0;
			cs = F.apply_no_albedo ? replay_chain<true, LR>(cnt, alpha, shading, albedo, ekterm, T, weight_sum2, rgb_ray2, ek)
			                       : replay_chain<false, LR>(cnt, alpha, shading, albedo, ekterm, T, weight_sum2, rgb_ray2, ek);
			// the running values after the chunk's last sample (a ray that is through keeps its own)
			const int last = max(cnt - 1, 0);
			const float T1 = group_read<LR>(cs.T, last, lane64), w1 = group_read<LR>(cs.ws, last, lane64), e1 = group_read<LR>(cs.ek, last, lane64);
			if (cnt > 0) { T = T1; weight_sum2 = w1; ek = e1; }
			float tmp[4] = {cs.rgb[0], cs.rgb[1], cs.rgb[2], cs.rgb[3]};
			if (F.apply_no_albedo) { tmp[1] = tmp[0]; tmp[2] = tmp[0]; tmp[3] = rgb_ray2[3]; }
#pragma unroll
			for (int k = 0; k < 4; ++k) { const float rk = group_read<LR>(tmp[k], last, lane64); if (cnt > 0) rgb_ray2[k] = rk; }
		}
		float my_rgb2[4] = {cs.rgb[0], cs.rgb[1], cs.rgb[2], cs.rgb[3]};
		if (F.apply_no_albedo) { my_rgb2[1] = my_rgb2[0]; my_rgb2[2] = my_rgb2[0]; my_rgb2[3] = 0.f; } // albedo = (1,1,1,0): one accumulator serves the three equal colour channels; the fourth only ever receives weight x 0
		if (valid) {
			half_t dl[16];
			pass2_sample(F, G, loss_scale, o, dt, cs.w, cs.T, cs.ws, my_rgb2, dl);
			h8 w0, w1;
#pragma unroll
			for (int q = 0; q < 8; ++q) { w0[q] = dl[q]; w1[q] = dl[8 + q]; }
			h8* dst = reinterpret_cast<h8*>(dloss + (size_t)j * 16);
			dst[0] = w0;
			dst[1] = w1;
		}
	}
	if (REC && ray_live) ek = a.chain_rec[((size_t)base + compacted_numsteps - 1) * CHAIN_REC_FLOATS + 7]; // the sum right after the ray's last kept sample, formed in pass 1
	if (ray_live && lane == 0) a.ek_loss[i] = ek / ((float)compacted_numsteps * gn);
}

// Pass 2 in two launches for batches of many short rays (round 4; needs the chain records). With 16 lanes per ray and four rays per wavefront a wavefront
// walks as many 16-sample chunks as its longest ray has -- half of its lanes idle at 27 kept samples per ray (step 6000) -- and the kernel is bound by the
// instructions it issues (~700 per chunk). Here the rays only write what they contribute to their samples (RayGrad), their loss rows, and for every compacted
// sample its ray and its marched slot; the samples are then worked on one per lane whatever ray they belong to. Same expressions on the same inputs: same bits.
template <int LR>
__global__ __launch_bounds__(256) void k_loss_pass2_rays(const LossArgs a) {
	constexpr uint32_t RPW = 256 / LR, WG_PER_TILE = SCAN_TILE / RPW;
	__shared__ double rows[RPW][3];
	__shared__ uint32_t wsum[16], n_of[RPW], sh_base;
	const uint32_t i_raw = blockIdx.x * RPW + threadIdx.x / LR;
	const int lane = threadIdx.x & (LR - 1);
	const uint32_t kept = min(a.counters[2], a.n_rays);
	const bool ray_ok = i_raw < kept;
	const uint32_t i = ray_ok ? i_raw : 0u;
	uint32_t scanned_base = 0;
	if (a.scan_words) {
		// The compaction offsets (exclusive prefix of ncomp over the kept rays; counters[1] = total), formed here instead of by a launch of their own on the critical stream:
		// the first workgroup of every 4096-ray tile publishes the tile's sum at once (ticketed word, as k_scan_compact_chain); a workgroup adds the sums of the tiles in front
		// of its own (polled), the counts in front of it inside its tile (<= 4080 loads from L2, 16 per thread) and those of its own rays in front of each ray.
		const uint32_t tid = threadIdx.x, lane64 = tid & 63u, wave = tid >> 6;
		const uint32_t tile = blockIdx.x / WG_PER_TILE, tile_start = tile * SCAN_TILE, i0 = blockIdx.x * RPW;
		uint32_t total;
		if (blockIdx.x % WG_PER_TILE == 0) {
			uint32_t mine = 0;
#pragma unroll
			for (uint32_t e = 0; e < SCAN_TILE / 256; ++e) { const uint32_t k = tile_start + e * 256 + tid; mine += k < kept ? a.ncomp[k] : 0u; }
			(void)block_exclusive_scan<4>(mine, lane64, wave, wsum, total);
			if (tid == 0) atomicExch(a.scan_words + tile, ((unsigned long long)a.scan_ticket << 32) | total);
		}
		uint32_t mine = 0;
		const uint32_t stop = min(i0, kept);
#pragma unroll
		for (uint32_t e = 0; e < SCAN_TILE / 256; ++e) { const uint32_t k = tile_start + e * 256 + tid; mine += k < stop ? a.ncomp[k] : 0u; }
		(void)block_exclusive_scan<4>(mine, lane64, wave, wsum, total);
		uint32_t in_front = total;
		if (tid < 64) { // tiles in front: one lane each (<= 63)
			uint32_t v = 0;
			if (tid < tile) {
				unsigned long long w;
				uint32_t spins = 0;
				do { w = atomicAdd(a.scan_words + tid, 0ull); if ((uint32_t)(w >> 32) == a.scan_ticket) break; __builtin_amdgcn_s_sleep(2); } while (++spins < 20000000u); // (bounded, see chain_prefix)
				if ((uint32_t)(w >> 32) != a.scan_ticket) { // gave up: report, and poison the step -- k_loss_pass2_samples sees the word and compacts nothing (zero counter), as the chain scans do
					w = 0;
					if (a.scan_error) atomicExch(a.scan_error, 1u);
					atomicExch(a.scan_words + 64, ((unsigned long long)a.scan_ticket << 32) | 1ull);
				}
				v = (uint32_t)w;
			}
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
			if (tid == 0) sh_base = v + in_front;
		}
		if (lane == 0) n_of[threadIdx.x / LR] = ray_ok ? a.ncomp[i] : 0u;
		__syncthreads();
		scanned_base = sh_base;
		for (uint32_t r = 0; r < threadIdx.x / LR; ++r) scanned_base += n_of[r];
		if (ray_ok && lane == 0) a.cbase[i] = scanned_base;
		if (threadIdx.x == 0 && (kept == 0 ? blockIdx.x == 0 : (kept - 1) / RPW == blockIdx.x)) { // the workgroup of the last kept ray: numsteps_counter_compacted
			uint32_t t = sh_base;
			for (uint32_t r = 0; r < RPW; ++r) t += n_of[r];
			*a.scan_total = t;
		}
	}
	float loss_row = 0.f, mask_row = 0.f, ek_row = 0.f;
	if (ray_ok) {
		const RayLoss R = a.ray_loss[i];
		const uint32_t compacted_base = a.scan_words ? scanned_base : a.cbase[i];
		const uint32_t compacted_numsteps = min(a.B - min(a.B, compacted_base), R.n_comp); // testbed_nerf.cu:1723
		const uint32_t base = a.numsteps[(size_t)i * 2 + 1];
		__builtin_amdgcn_wave_barrier();
		if (lane == 0) {
			a.numsteps[(size_t)i * 2 + 0] = compacted_numsteps;
			a.numsteps[(size_t)i * 2 + 1] = compacted_base;
		}
		if (compacted_numsteps != 0) {
			const float gn = (float)a.n_rays_global;
			RayGrad G;
			pass2_ray_terms(a.F, R, gn, G, loss_row, mask_row);
			ek_row = a.chain_rec[((size_t)base + compacted_numsteps - 1) * CHAIN_REC_FLOATS + 7] / ((float)compacted_numsteps * gn);
			if (lane == 0) {
				f4* g = reinterpret_cast<f4*>(a.ray_grad + (size_t)i * 16);
				g[0] = f4{G.grad[0], G.grad[1], G.grad[2], G.grad[3]};
				g[1] = f4{G.weight_sum, G.gradient_weight_sum, G.light[0], G.light[1]};
				g[2] = f4{G.light[2], G.dir[0], G.dir[1], G.dir[2]};
				g[3] = f4{G.rgb_ray[0], G.rgb_ray[1], G.rgb_ray[2], G.rgb_ray[3]};
			}
			for (uint32_t j = lane; j < compacted_numsteps; j += LR) {
				a.ray_of[compacted_base + j] = i;
				a.slot_of[compacted_base + j] = base + j;
				if (a.src_slot) a.src_slot[compacted_base + j] = base + j;
			}
		}
		if (lane == 0) { a.loss[i] = loss_row; a.ek_loss[i] = ek_row; a.mask_loss[i] = mask_row; } // zeros for a kept ray without compacted samples (see k_loss_pass2)
	}
	// the workgroup's share of the step's loss sums (fp64 sums of the fp32 rows in ray order: deterministic), for the reduction inside k_loss_pass2_samples
	if (lane == 0) { rows[threadIdx.x / LR][0] = loss_row; rows[threadIdx.x / LR][1] = ek_row; rows[threadIdx.x / LR][2] = mask_row; }
	__syncthreads();
	if (threadIdx.x < 3) {
		double sum = 0;
#pragma unroll
		for (int r = 0; r < 256 / LR; ++r) sum += rows[r][threadIdx.x];
		a.wg_partial[(size_t)blockIdx.x * 3 + threadIdx.x] = sum;
	}
}
// the step's 48-byte readback from the three sums (the tail of reduce_losses_body)
__device__ __forceinline__ void publish_losses(const double s0, const double s1, const double s2, const uint32_t* __restrict__ counters, const uint32_t* __restrict__ fwd_counts,
                                               double* __restrict__ out, double* __restrict__ host_out, const uint32_t host_seq) {
	const uint32_t t = threadIdx.x;
	if (t == 0) { out[0] = s0; out[1] = s1; out[2] = s2; out[12] = s0; out[13] = s1; out[14] = s2; }
	if (t < 4) { reinterpret_cast<uint32_t*>(out + 3)[t] = counters[t]; out[8 + t] = (double)counters[t]; }
	if (t < 2) reinterpret_cast<uint32_t*>(out + 5)[t] = fwd_counts ? fwd_counts[t * 2] : 0u;
	if (host_out) {
		if (t == 0) { host_out[0] = s0; host_out[1] = s1; host_out[2] = s2; }
		if (t < 4) reinterpret_cast<uint32_t*>(host_out + 3)[t] = counters[t];
		if (t < 2) reinterpret_cast<uint32_t*>(host_out + 5)[t] = fwd_counts ? fwd_counts[t * 2] : 0u;
		if (host_seq) { // (see reduce_losses_body)
			__syncthreads();
			if (t == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(host_out + 6), host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}
// Workgroup 0: the step's loss sums from the partial sums of k_loss_pass2_rays (fixed order) and their publication; the others: one lane per compacted sample,
// which also writes the sample's wrapped copies behind the compacted batch when that is shorter than B (fill_rollover_and_rescale<half> + fill_rollover<float>).
__global__ __launch_bounds__(256) void k_loss_pass2_samples(const LossArgs a) {
	// a wait of the compaction scan inside k_loss_pass2_rays gave up (word 64 of its block carries this launch's ticket): the offsets are wrong, nothing is compacted
	const bool poisoned = a.scan_words && (uint32_t)(a.scan_words[64] >> 32) == a.scan_ticket;
	if (blockIdx.x == 0) {
		if (poisoned) { if (threadIdx.x == 0) *a.scan_total = 0u; __syncthreads(); }
		if (!a.red_out) return;
		__shared__ double sh[4][3];
		const uint32_t n_part = (min(a.counters[2], a.n_rays) + 15u) / 16u; // workgroups of k_loss_pass2_rays<16> that hold kept rays
		double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll 4
		for (uint32_t p = threadIdx.x; p < n_part; p += 256) { s0 += a.wg_partial[(size_t)p * 3 + 0]; s1 += a.wg_partial[(size_t)p * 3 + 1]; s2 += a.wg_partial[(size_t)p * 3 + 2]; }
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
		if ((threadIdx.x & 63u) == 0) { sh[threadIdx.x >> 6][0] = s0; sh[threadIdx.x >> 6][1] = s1; sh[threadIdx.x >> 6][2] = s2; }
		__syncthreads();
		s0 = ((sh[0][0] + sh[1][0]) + sh[2][0]) + sh[3][0]; s1 = ((sh[0][1] + sh[1][1]) + sh[2][1]) + sh[3][1]; s2 = ((sh[0][2] + sh[1][2]) + sh[2][2]) + sh[3][2];
		publish_losses(s0, s1, s2, a.counters, a.fwd_counts, a.red_out, a.red_host_out, a.red_host_seq); // loss, eikonal, mask: the order of reduce_losses_body's rows
		return;
	}
	const uint32_t n_in = poisoned ? 0u : a.counters[1];
	const uint32_t n = min(n_in, a.B);
	const uint32_t q = (blockIdx.x - 1) * 256 + threadIdx.x;
	if (q >= n) return;
	const uint32_t slot = a.slot_of[q];
	RayGrad G;
	{
		const f4* g = reinterpret_cast<const f4*>(a.ray_grad + (size_t)a.ray_of[q] * 16);
		const f4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
		G.grad[0] = g0[0]; G.grad[1] = g0[1]; G.grad[2] = g0[2]; G.grad[3] = g0[3];
		G.weight_sum = g1[0]; G.gradient_weight_sum = g1[1]; G.light[0] = g1[2]; G.light[1] = g1[3];
		G.light[2] = g2[0]; G.dir[0] = g2[1]; G.dir[1] = g2[2]; G.dir[2] = g2[3];
		G.rgb_ray[0] = g3[0]; G.rgb_ray[1] = g3[1]; G.rgb_ray[2] = g3[2]; G.rgb_ray[3] = g3[3];
	}
	const float* ci = a.coords + (size_t)slot * 7;
	float* co = a.coords_compacted + (size_t)q * 7;
	float cv[7];
#pragma unroll
	for (int k = 0; k < 7; ++k) cv[k] = ci[k];
#pragma unroll
	for (int k = 0; k < 7; ++k) co[k] = cv[k];
	half_t o[16];
	load_out16(a.mlp_out + (size_t)slot * 16, o);
	const float dt = unwarp_dt(cv[3]);
	const f4* rq = reinterpret_cast<const f4*>(a.chain_rec + (size_t)slot * CHAIN_REC_FLOATS);
	const f4 r0 = rq[0];
	float my_rgb2[4] = {r0[3], r0[3], r0[3], 0.f};
	if (!a.F.apply_no_albedo) { const f4 r1 = rq[1]; my_rgb2[1] = r1[0]; my_rgb2[2] = r1[1]; my_rgb2[3] = r1[2]; }
	half_t dl[16];
	pass2_sample(a.F, G, LOSS_SCALE / (float)a.n_rays_global, o, dt, r0[0], r0[1], r0[2], my_rgb2, dl);
	h8 w0, w1;
#pragma unroll
	for (int k = 0; k < 8; ++k) { w0[k] = dl[k]; w1[k] = dl[8 + k]; }
	// the wrapped copies below must start from the BITS stored here: without the barrier the compiler forms some of these halfs twice, once as multiply + convert and once
	// as v_fma_mixlo_f16 with a +0 addend, which turns a -0 product into +0 (seen in the ISA; the one-launch forms read the stored row back, k_rollover)
	asm volatile("" : "+v"(w0), "+v"(w1));
	h8* dst = reinterpret_cast<h8*>(a.dloss + (size_t)q * 16);
	dst[0] = w0;
	dst[1] = w1;
	// the batch wraps: sample q again at q + n_in, q + 2 n_in ... < B, its loss gradient rescaled (common_device.h:514-535; same expressions as rollover_body)
	for (uint32_t q2 = q + n_in; q2 < a.B; q2 += n_in) {
		float* co2 = a.coords_compacted + (size_t)q2 * 7;
#pragma unroll
		for (int k = 0; k < 7; ++k) co2[k] = cv[k];
		h8 v0, v1;
#pragma unroll
		for (int k = 0; k < 8; ++k) { v0[k] = f2h(h2f(w0[k]) * n_in / a.B); v1[k] = f2h(h2f(w1[k]) * n_in / a.B); }
		h8* d2 = reinterpret_cast<h8*>(a.dloss + (size_t)q2 * 16);
		d2[0] = v0;
		d2[1] = v1;
		if (a.src_slot) a.src_slot[q2] = slot;
	}
}

// fill_rollover_and_rescale<half> + fill_rollover<float> (common_device.h:514-535; testbed_nerf.cu:4044-4049)
// fill_rollover_and_rescale / fill_rollover (common_device.h:514-535): pad the compacted batch to B by wrapping.
__device__ __forceinline__ void rollover_body(const uint32_t B, const uint32_t* __restrict__ counters, half_t* __restrict__ dloss, float* __restrict__ coords,
                                              const uint64_t first, const uint64_t stride, uint32_t* __restrict__ src_slot = nullptr) {
	const uint32_t n_in = counters[1];
	if (n_in == 0 || n_in >= B) return;
	const uint64_t n_out16 = (uint64_t)B * 16, n_in16 = (uint64_t)n_in * 16;
	const uint64_t n_out7 = (uint64_t)B * 7, n_in7 = (uint64_t)n_in * 7;
	for (uint64_t q = first; q < n_out16; q += stride) {
		if (q >= n_in16) {
			const float v = h2f(dloss[q % n_in16]);
			dloss[q] = f2h(v * n_in / B);
		}
		if (q >= n_in7 && q < n_out7) coords[q] = coords[q % n_in7];
		if (src_slot && q >= n_in && q < B) src_slot[q] = src_slot[q % n_in]; // the wrapped samples are the same samples: same input rows
	}
}
__global__ void k_rollover(const uint32_t B, const uint32_t* __restrict__ counters, half_t* __restrict__ dloss, float* __restrict__ coords, uint32_t* __restrict__ src_slot) {
	rollover_body(B, counters, dloss, coords, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x, src_slot);
}

// per-tile (4096 rays) fp64 sums of the three loss rows, for batches too large for one workgroup to walk (fixed order: deterministic)
__global__ __launch_bounds__(1024) void k_reduce_losses_tiles(const uint32_t n_max, const uint32_t* __restrict__ counters, const float* __restrict__ l0, const float* __restrict__ l1,
                                                              const float* __restrict__ l2, double* __restrict__ partial) {
	__shared__ double sh[3][1024];
	const uint32_t n = min(counters[2], n_max);
	const uint32_t i0 = blockIdx.x * SCAN_TILE + threadIdx.x;
	double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
	for (uint32_t e = 0; e < 4; ++e) { const uint32_t i = i0 + e * 1024; if (i < n) { s0 += l0[i]; s1 += l1[i]; s2 += l2[i]; } }
	sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
	__syncthreads();
	for (int off = 512; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; sh[2][threadIdx.x] += sh[2][threadIdx.x + off]; }
		__syncthreads();
	}
	if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

// loss scalars of Counters::update_after_training (testbed_nerf.cu:3549-3551): fp64 sums over the kept rays
__device__ __forceinline__ void reduce_losses_body(const uint32_t n_max, const uint32_t* __restrict__ counters, const float* __restrict__ l0, const float* __restrict__ l1, const float* __restrict__ l2, double* __restrict__ out, const uint32_t* __restrict__ fwd_counts, double* __restrict__ host_out,
                                                        const double* __restrict__ partial, const uint32_t n_partial, const uint32_t host_seq = 0) {
	__shared__ double sh[3][1024];
	const uint32_t n = min(counters[2], n_max);
	double s0 = 0, s1 = 0, s2 = 0;
	if (partial) { // large batches: per-tile sums of k_reduce_losses_tiles
		for (uint32_t i = threadIdx.x; i < n_partial; i += 1024) { s0 += partial[i * 3 + 0]; s1 += partial[i * 3 + 1]; s2 += partial[i * 3 + 2]; }
	} else {
		for (uint32_t i = threadIdx.x; i < n; i += 1024) { s0 += l0[i]; s1 += l1[i]; s2 += l2[i]; }
	}
	sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
	__syncthreads();
	for (int off = 512; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; sh[2][threadIdx.x] += sh[2][threadIdx.x + off]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { out[0] = sh[0][0]; out[1] = sh[1][0]; out[2] = sh[2][0]; }
	if (threadIdx.x < 4) reinterpret_cast<uint32_t*>(out + 3)[threadIdx.x] = counters[threadIdx.x]; // one 48-byte readback: sums + counters + evaluated samples
	if (threadIdx.x < 2) reinterpret_cast<uint32_t*>(out + 5)[threadIdx.x] = fwd_counts ? fwd_counts[threadIdx.x * 2] : 0u; // samples of round 1, round 2
	// the same numbers as one double[7] {counters, sums}: what data-parallel ranks all-reduce (RNB_BUF_STEP_VECTOR)
	if (threadIdx.x < 4) out[8 + threadIdx.x] = (double)counters[threadIdx.x];
	if (threadIdx.x == 0) { out[12] = sh[0][0]; out[13] = sh[1][0]; out[14] = sh[2][0]; }
	if (host_out) { // the same 48 bytes into host memory (visible to the host once the kernel's completion event has fired)
		if (threadIdx.x == 0) { host_out[0] = sh[0][0]; host_out[1] = sh[1][0]; host_out[2] = sh[2][0]; }
		if (threadIdx.x < 4) reinterpret_cast<uint32_t*>(host_out + 3)[threadIdx.x] = counters[threadIdx.x];
		if (threadIdx.x < 2) reinterpret_cast<uint32_t*>(host_out + 5)[threadIdx.x] = fwd_counts ? fwd_counts[threadIdx.x * 2] : 0u;
		// host_seq != 0: the host does not wait for this kernel's completion but polls the word behind the 56 bytes for the step's sequence number
		// (rnb_train_step_local): no completion event with a system-scope fence on the critical stream, and everything this workgroup reads
		// (counters, loss rows, fwd_counts) has been read by then, so the next step's march may overwrite it.
		if (host_seq) {
			__syncthreads();
			if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(host_out + 6), host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}


// The training step's form: workgroup 0 reduces the losses (the step's 48-byte readback), the others pad the compacted batch --
// both need nothing but the second loss pass, and one launch on the critical stream instead of two saves a kernel boundary.
__global__ __launch_bounds__(1024) void k_reduce_losses_rollover(const uint32_t n_max, const uint32_t* __restrict__ counters, const float* __restrict__ l0, const float* __restrict__ l1,
                                                                 const float* __restrict__ l2, double* __restrict__ out, const uint32_t* __restrict__ fwd_counts, double* __restrict__ host_out,
                                                                 const double* __restrict__ partial, const uint32_t n_partial, const uint32_t B, half_t* __restrict__ dloss, float* __restrict__ coords, uint32_t* __restrict__ src_slot,
                                                                 const uint32_t host_seq) {
	if (blockIdx.x == 0) reduce_losses_body(n_max, counters, l0, l1, l2, out, fwd_counts, host_out, partial, n_partial, host_seq);
	else rollover_body(B, counters, dloss, coords, (uint64_t)(blockIdx.x - 1) * blockDim.x + threadIdx.x, (uint64_t)(gridDim.x - 1) * blockDim.x, src_slot);
}

// ---------------------------------------------------------------------------------------------
// rnb_eval_primitives: the integer / index primitives above, one thread per item (include/rnb_neus2.h); tests/golden/int_fixtures.json
// holds what the reference's own host-compilable fragments return for the same items.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t PRIM_DW_SAMPLES = 8256; // RNB_PRIM_DW_SLICED: two whole slices and a short one
constexpr uint32_t PRIM_IN_WORDS[20] = {6, 3, 1, 8, 9, 1, 9, 9, 9, 7, 32, 20, 35, 37, 16, 263, 10, 2, 1, 4 + 8 * PRIM_DW_SAMPLES / 2}, PRIM_OUT_WORDS[20] = {4, 4, 2, 3, 7, 3, 11, 5, 3, 3, 5, 9, 7, 28, 9, 16, 23, 1, 2, 16};
__global__ void k_prim_bitfield(uint8_t* __restrict__ bitfield, const uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	Pcg32 q{5};
	q.advance((int64_t)i);
	bitfield[i] = (uint8_t)(q.next_uint() >> 24);
}
// RNB_PRIM_ENCODE: one (sample, level) of the hash-grid encoding from a table of <= 256 entries that travels with the item -- through encode_level_core (the training kernels' form)
// AND through level_issue / level_consume (the evaluation kernels' pipelined form). One WAVEFRONT per item: both read their level's constants with readfirstlane.
__global__ __launch_bounds__(64) void k_prim_encode(const uint32_t* __restrict__ in, const uint32_t n, uint32_t* __restrict__ out) {
	__shared__ LevelMeta lm[RNB_MAX_LEVELS];
	const uint32_t i = blockIdx.x;
	if (i >= n) return;
	const uint32_t* a = in + (size_t)i * PRIM_IN_WORDS[RNB_PRIM_ENCODE];
	GridMeta G{};
	G.n_levels = 1; G.valid_level = 0;
	for (uint32_t l = 1; l <= RNB_MAX_LEVELS; ++l) G.offsets[l] = a[0];
	G.resolution[0] = a[1]; G.scale[0] = __uint_as_float(a[2]);
	fill_level_meta(lm, G, (int)threadIdx.x);
	__syncthreads();
	const float x = __uint_as_float(a[3]), y = __uint_as_float(a[4]), z = __uint_as_float(a[5]);
	const uint32_t* table = a + 6; // half2 per entry; entry `size` is readable (257 words)
	half_t f0, f1, g0, g1;
	float dy0[3], dy1[3], ey0[3], ey1[3];
	encode_level_core<true>(table, a[0], a[1], __uint_as_float(a[2]), x, y, z, f0, f1, dy0, dy1);
	uint32_t v[8];
	level_issue(lm, table, 0, x, y, z, v);
	level_consume<true>(lm, 0, x, y, z, v, g0, g1, ey0, ey1);
	if (threadIdx.x == 0) {
		uint32_t* o = out + (size_t)i * PRIM_OUT_WORDS[RNB_PRIM_ENCODE];
		o[0] = (uint32_t)__builtin_bit_cast(uint16_t, f0); o[1] = (uint32_t)__builtin_bit_cast(uint16_t, f1);
		o[8] = (uint32_t)__builtin_bit_cast(uint16_t, g0); o[9] = (uint32_t)__builtin_bit_cast(uint16_t, g1);
		for (int d = 0; d < 3; ++d) { o[2 + d] = __float_as_uint(dy0[d]); o[5 + d] = __float_as_uint(dy1[d]); o[10 + d] = __float_as_uint(ey0[d]); o[13 + d] = __float_as_uint(ey1[d]); }
	}
}
__global__ void k_primitives(const int kind, const uint32_t* __restrict__ in, const uint32_t n, uint32_t* __restrict__ out, const uint8_t* __restrict__ bitfield) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t* a = in + (size_t)i * PRIM_IN_WORDS[kind];
	uint32_t* o = out + (size_t)i * PRIM_OUT_WORDS[kind];
	auto f = [](uint32_t u) { return __uint_as_float(u); };
	auto u = [](float v) { return __float_as_uint(v); };
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
		SceneAabb A; A.mn = f(a[0]); A.mx = f(a[1]); A.cone_angle = 0.f; A.max_cascade = 0;
		const Vec3 p = {f(a[2]), f(a[3]), f(a[4])}, d = {f(a[5]), f(a[6]), f(a[7])};
		float t0, t1;
		ray_intersect(A, p, d, &t0, &t1);
		o[0] = u(t0); o[1] = u(t1); o[2] = aabb_contains(A, p) ? 1u : 0u;
	} else if (kind == RNB_PRIM_ACTIVATION) {
		const float l = logistic(f(a[0]));
		o[0] = u(relu(f(a[0]))); o[1] = u(l); o[2] = u(l * (1 - l));
	} else if (kind == RNB_PRIM_WARP) {
		SceneAabb A; A.mn = f(a[0]); A.mx = f(a[1]); A.cone_angle = 0.f; A.max_cascade = 0;
		const Vec3 wp = warp_position(A, {f(a[2]), f(a[3]), f(a[4])}), wd = warp_direction({f(a[5]), f(a[6]), f(a[7])}), ud = unwarp_direction(wd);
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
		ViewDev m{};
		m.width = a[0]; m.height = a[1];
		const float xy[2] = {f(a[2]), f(a[3])};
		float c[4];
		read_rgba(xy, m, reinterpret_cast<const uint16_t*>(a + 4), c); // the item's own words are the image: RGBA16, two words per pixel
		o[0] = u(c[0]); o[1] = u(c[1]); o[2] = u(c[2]); o[3] = u(c[3]);
		o[4] = red_is_nonpositive(xy, m, reinterpret_cast<const uint16_t*>(a + 4)) ? 1u : 0u;
	} else if (kind == RNB_PRIM_CAMERA_RAY) {
		ViewDev m{};
		m.width = a[0]; m.height = a[1]; m.focal[0] = f(a[2]); m.focal[1] = f(a[3]); m.principal[0] = f(a[4]); m.principal[1] = f(a[5]);
		for (int k = 0; k < 12; ++k) m.xform[k] = f(a[8 + k]);
		const float xy[2] = {f(a[6]), f(a[7])};
		Vec3 ro, du, dir;
		camera_ray(m, xy, ro, du, dir);
		o[0] = u(ro.x); o[1] = u(ro.y); o[2] = u(ro.z); o[3] = u(du.x); o[4] = u(du.y); o[5] = u(du.z); o[6] = u(dir.x); o[7] = u(dir.y); o[8] = u(dir.z);
	} else if (kind == RNB_PRIM_RAY_TARGETS) {
		LossFlags F{};
		F.apply_no_albedo = a[0]; F.apply_rgbplus = a[1]; F.apply_L2 = a[2]; F.apply_light_opti = a[3]; F.apply_relu = a[4];
		float X[12], tn[4], ta[4], ld[9], tgt[4], lw[3];
		for (int k = 0; k < 12; ++k) X[k] = f(a[6 + k]);
		for (int k = 0; k < 4; ++k) { tn[k] = f(a[18 + k]); ta[k] = f(a[22 + k]); }
		for (int k = 0; k < 9; ++k) ld[k] = f(a[26 + k]);
		ray_targets(F, X, tn, ta, ld, (int)a[5], tgt, lw);
		o[0] = u(tgt[0]); o[1] = u(tgt[1]); o[2] = u(tgt[2]); o[3] = u(tgt[3]); o[4] = u(lw[0]); o[5] = u(lw[1]); o[6] = u(lw[2]);
	} else if (kind == RNB_PRIM_LOSS_SAMPLE) {
		LossFlags F{};
		F.apply_no_albedo = a[0]; F.apply_rgbplus = a[1]; F.apply_L2 = a[2]; F.apply_relu = a[3];
		half_t oh[16];
		for (int k = 0; k < 8; ++k) { const h2 p2 = unpack_h2(a[4 + k]); oh[2 * k] = p2[0]; oh[2 * k + 1] = p2[1]; }
		const float dt = f(a[12]);
		RayGrad G;
		for (int k = 0; k < 3; ++k) { G.dir[k] = f(a[13 + k]); G.light[k] = f(a[16 + k]); }
		for (int k = 0; k < 4; ++k) { G.grad[k] = f(a[19 + k]); G.rgb_ray[k] = f(a[23 + k]); }
		float rgb2[4] = {f(a[27]), f(a[28]), f(a[29]), f(a[30])};
		G.weight_sum = f(a[31]);
		float w2 = f(a[32]), T = f(a[33]);
		G.gradient_weight_sum = f(a[34]);
		const float loss_scale = f(a[35]);
		F.ek_loss_weight = f(a[36]);
		// the forward part of the loop body (testbed_nerf.cu:1866-1917) with the compositing recurrence as replay_chain states it (chain.cuh)
		float albedo[4];
		albedo_from_output(F, oh, albedo);
		const AlphaTerms at = alpha_terms(oh, dt, G.dir, 1.0f);
		float shading = esum3(at.g[0] * G.light[0], at.g[1] * G.light[1], at.g[2] * G.light[2]);
		if (F.apply_relu) shading = shading > 0.f ? shading : 0.f;
		const float w = at.alpha * T;
		T = T * (1.f - at.alpha);
		w2 = w2 + w;
		for (int k = 0; k < 4; ++k) rgb2[k] = rgb2[k] + w * albedo[k] * shading;
		half_t dl[16];
		float inter[10];
		pass2_sample(F, G, loss_scale, oh, dt, w, T, w2, rgb2, dl, inter);
		for (int k = 0; k < 10; ++k) o[18 + k] = u(inter[k]);
		o[0] = u(at.alpha); o[1] = u(T); o[2] = u(w2);
		for (int k = 0; k < 4; ++k) o[3 + k] = u(rgb2[k]);
		for (int k = 0; k < 11; ++k) o[7 + k] = (uint32_t)__builtin_bit_cast(uint16_t, dl[k]);
	} else if (kind == RNB_PRIM_RAY_LOSS) {
		LossFlags F{};
		F.apply_L2 = a[0]; F.apply_rgbplus = a[1]; F.apply_bce = a[2]; F.mask_loss_weight = f(a[3]);
		RayLoss R{};
		for (int k = 0; k < 4; ++k) { R.rgbtarget[k] = f(a[5 + k]); R.rgb_ray[k] = f(a[9 + k]); }
		R.mask_certainty = (float)(f(a[13]) > 0.99); R.mask_gt = (float)(f(a[14]) > 0.99); // as ray_constants_core forms them from the texels' alpha
		R.weight_sum_raw = f(a[15]);
		RayGrad G;
		float lrow, mrow;
		pass2_ray_terms(F, R, (float)a[4], G, lrow, mrow);
		o[0] = u(lrow * (float)a[4]); // (the ray's loss itself is not kept by the kernels: row x n_rays, compared as such)
		o[1] = u(G.grad[0]); o[2] = u(G.grad[1]); o[3] = u(G.grad[2]); o[4] = u(G.grad[3]); o[5] = u(G.weight_sum); o[6] = u(G.gradient_weight_sum); o[7] = u(lrow); o[8] = u(mrow);
	} else if (kind == RNB_PRIM_MARCH_RAY) {
		SceneAabb A; A.mn = f(a[0]); A.mx = f(a[1]); A.cone_angle = f(a[2]); A.max_cascade = 0;
		const Vec3 ro = {f(a[3]), f(a[4]), f(a[5])}, dir = {f(a[6]), f(a[7]), f(a[8])};
		const Vec3 wd = warp_direction(dir);
		uint32_t chk = 0;
		for (int q = 0; q < 23; ++q) o[q] = 0u;
		uint32_t last[7] = {0, 0, 0, 0, 0, 0, 0};
		const uint32_t n = march<false>(A, bitfield, nullptr, 0u, ro, dir, f(a[9]), RNB_MAX_STEPS, [&](uint32_t j, const Vec3& pos, float dt, float) {
			const Vec3 wp = warp_position(A, pos);
			const uint32_t c7[7] = {u(wp.x), u(wp.y), u(wp.z), u(warp_dt(dt)), u(wd.x), u(wd.y), u(wd.z)}; // a NerfCoordinate (nerf.h:76-104): position, dt, direction
			for (int q = 0; q < 7; ++q) { chk += c7[q]; last[q] = c7[q]; if (j < 2) o[2 + j * 7 + q] = c7[q]; }
		});
		o[0] = n; o[1] = chk;
		for (int q = 0; q < 7; ++q) o[16 + q] = last[q];
	} else if (kind == RNB_PRIM_SDF_DENSITY) {
		o[0] = (uint32_t)__builtin_bit_cast(uint16_t, sdf_to_density(__builtin_bit_cast(half_t, (uint16_t)a[0]), __builtin_bit_cast(half_t, (uint16_t)a[1])));
	} else if (kind == RNB_PRIM_GRID) {
		float pos; uint32_t cell;
		pos_fract(f(a[5]), f(a[6]), &pos, &cell);
		o[0] = grid_entry(a[0], a[1], a[2], a[3], a[4]); o[1] = u(pos); o[2] = cell;
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
		o[4] = density_grid_occupied_at(p, bitfield, (uint32_t)mip) ? 1u : 0u;
		o[5] = u(distance_to_next_voxel(p, d, idir, res)); o[6] = u(advance_to_next_voxel(t, cone, p, d, idir, res));
	}
}

} // namespace rnb
