// nr3d_lib_amd/csrc/compact.h -- compaction of the non-empty packs of a pack table in one exclusive scan (gfx950); shared by
// the glue entry points (ray_glue.hip) and the marcher's fused count + hit-ray compaction (occ_grid.hip).
#pragma once
#include "common.h"
#include "scan.h"

namespace nr3d {
namespace glue {

// ------------------------------------------------------------------------------------------------
// compaction of flagged packs: value(i) = count_i | (count_i > 0) << 36, one exclusive scan gives the new begin of
// every pack (low 36 bits) and its rank among the non-empty ones (high 28 bits).
// Range (round-3 advisor finding; checked in compact_packs): fewer than 2^28 packs per call, and the counts must sum to less
// than 2^36 -- which 288 GB of HBM guarantee: a sample of these tables carries >= 16 bytes (depth, delta, ray index, ...), so a
// device holds < 2^35 of them.  (Round 3 split 40 / 24: 2^24 rays in one call -- configs[4]'s size on ONE device -- overflowed.)
// ------------------------------------------------------------------------------------------------
constexpr int kShift = 36;
constexpr uint64_t kMaxPacks = 1ull << (64 - kShift);
constexpr uint64_t kLow = (1ull << kShift) - 1ull;

template <typename TCnt, int STRIDE>    // counts[i * STRIDE] (STRIDE 2: the count column of an [n, 2] pack table, offset applied by the caller)
__device__ __forceinline__ uint64_t cval(const TCnt *__restrict__ counts, uint64_t i) {
	const uint64_t c = (uint64_t)counts[i * STRIDE];
	return c | ((c ? 1ull : 0ull) << kShift);
}

template <typename TCnt, int STRIDE>
__global__ __launch_bounds__(scan::kThreads) void k_c_tile_sums(uint64_t n, const TCnt *__restrict__ counts,
                                                                uint64_t *__restrict__ tile_sums) {
	__shared__ uint64_t lds[4];
	const uint64_t first = (uint64_t)blockIdx.x * scan::kTile + (uint64_t)threadIdx.x * scan::kItems;
	uint64_t s = 0;
#pragma unroll
	for (int k = 0; k < scan::kItems; ++k)
		if (first + k < n) s += cval<TCnt, STRIDE>(counts, first + k);
	uint64_t tot;
	scan::block_exclusive(s, tot, lds);
	if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the tile sums in place, totals[0] = sum of counts, totals[1] = non-empty packs
static __global__ __launch_bounds__(scan::kThreads) void k_c_scan_tiles(uint32_t n_tiles, uint64_t *__restrict__ tile_sums,
                                                                 int64_t *__restrict__ totals) {
	__shared__ uint64_t lds[4];
	uint64_t carry = 0;
	for (uint32_t base = 0; base < n_tiles; base += scan::kThreads) {
		const uint32_t i = base + threadIdx.x;
		const uint64_t v = i < n_tiles ? tile_sums[i] : 0;
		uint64_t tot;
		const uint64_t ex = scan::block_exclusive(v, tot, lds);
		if (i < n_tiles) tile_sums[i] = carry + ex;
		carry += tot;
	}
	if (threadIdx.x == 0) {
		__hip_atomic_store(&totals[0], (int64_t)(carry & kLow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&totals[1], (int64_t)(carry >> kShift), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// what a pack writes once its exclusive prefix is known
template <typename TCnt, int STRIDE>
struct PackWriter {
	const TCnt *counts;
	const int64_t *tag_in;       // optional per-pack tag carried along (e.g. the pack's ray index); NULL: the pack index itself
	int64_t *begin_all;          // [n] new begin of EVERY pack, or NULL
	int64_t *idx_out;            // [n_hit] index (or tag) of the non-empty packs
	int64_t *pack_infos_out;     // [n_hit, 2]
	int32_t *info32 = nullptr;   // [n, 2] (begin, count) of EVERY pack as int32 (the marcher's packed_info), or NULL
	__device__ __forceinline__ void operator()(uint64_t i, uint64_t ex) const {
		const uint64_t c = (uint64_t)counts[i * STRIDE];
		const uint64_t begin = ex & kLow, rank = ex >> kShift;
		if (begin_all) begin_all[i] = (int64_t)begin;
		if (info32) { info32[2 * i] = (int32_t)begin; info32[2 * i + 1] = (int32_t)c; }
		if (c) {
			if (idx_out) idx_out[rank] = tag_in ? tag_in[i] : (int64_t)i;
			pack_infos_out[2 * rank] = (int64_t)begin;
			pack_infos_out[2 * rank + 1] = (int64_t)c;
		}
	}
};

template <typename TCnt, int STRIDE>
__global__ __launch_bounds__(scan::kThreads) void k_c_write(uint64_t n, const uint64_t *__restrict__ tile_prefix,
                                                            PackWriter<TCnt, STRIDE> w) {
	__shared__ uint64_t lds[4];
	const uint64_t first = (uint64_t)blockIdx.x * scan::kTile + (uint64_t)threadIdx.x * scan::kItems;
	uint64_t v[scan::kItems], s = 0;
#pragma unroll
	for (int k = 0; k < scan::kItems; ++k) {
		v[k] = (first + k < n) ? cval<TCnt, STRIDE>(w.counts, first + k) : 0;
		s += v[k];
	}
	uint64_t tot;
	uint64_t run = tile_prefix[blockIdx.x] + scan::block_exclusive(s, tot, lds);
#pragma unroll
	for (int k = 0; k < scan::kItems; ++k) {
		if (first + k < n) w(first + k, run);
		run += v[k];
	}
}

// n <= scan::kSmallMax: one workgroup, one launch.  Round 6: 1024 threads, every thread's PER consecutive counts loaded ONCE into
// registers (independent loads, all in flight together) and written from there -- the 256-thread version walked 16 counts per thread
// through two dependent load loops: 13 us for the 4096 rays of configs[2], on the critical path of a launch-bound op.
constexpr int kSmallThreads = 1024;
template <typename TCnt, int STRIDE, int PER>
__global__ __launch_bounds__(kSmallThreads) void k_c_small(uint32_t n, PackWriter<TCnt, STRIDE> w, int64_t *__restrict__ totals) {
	__shared__ uint64_t lds[kSmallThreads / 64];
	const uint32_t first = threadIdx.x * PER;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint64_t v[PER], s = 0;
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		v[k] = (first + k < n) ? cval<TCnt, STRIDE>(w.counts, first + k) : 0;
		s += v[k];
	}
	const uint64_t inc = scan::wave_inclusive(s, lane);
	if (lane == 63) lds[wave] = inc;
	__syncthreads();
	uint64_t wave_off = 0, tot = 0;
#pragma unroll
	for (int q = 0; q < kSmallThreads / 64; ++q) {
		const uint64_t t = lds[q];
		if (q < wave) wave_off += t;
		tot += t;
	}
	uint64_t run = wave_off + inc - s;
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		if (first + k < n) w(first + k, run);
		run += v[k];
	}
	// system scope: `totals` is normally pinned host memory a host thread may be polling (nr3d_wait_host_words)
	if (threadIdx.x == 0) {
		__hip_atomic_store(&totals[0], (int64_t)(tot & kLow), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&totals[1], (int64_t)(tot >> kShift), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

template <typename TCnt, int STRIDE>
static int compact_packs(uint64_t n, const PackWriter<TCnt, STRIDE> &w, int64_t *totals, void *tmp, hipStream_t st) {
	NR3D_CHECK(n < kMaxPacks, "pack compaction: %llu packs in one call, the limit is 2^28 - 1 (split the batch)", (unsigned long long)n);
	if (n == 0) {
		NR3D_HIP_CHECK(hipMemsetAsync(totals, 0, 2 * sizeof(int64_t), st));
		return 0;
	}
	if (n <= scan::kSmallMax) {
		const uint32_t per = (uint32_t)((n + kSmallThreads - 1) / kSmallThreads);
#define NR3D_C_SMALL(PER) hipLaunchKernelGGL((k_c_small<TCnt, STRIDE, PER>), dim3(1), dim3(kSmallThreads), 0, st, (uint32_t)n, w, totals)
		if (per <= 1) NR3D_C_SMALL(1); else if (per <= 2) NR3D_C_SMALL(2); else if (per <= 4) NR3D_C_SMALL(4);
		else if (per <= 8) NR3D_C_SMALL(8); else if (per <= 16) NR3D_C_SMALL(16); else NR3D_C_SMALL(32);
#undef NR3D_C_SMALL
		NR3D_LAUNCH_CHECK();
		return 0;
	}
	const uint32_t n_tiles = (uint32_t)((n + scan::kTile - 1) / scan::kTile);
	uint64_t *tile_sums = (uint64_t *)tmp;
	hipLaunchKernelGGL((k_c_tile_sums<TCnt, STRIDE>), dim3(n_tiles), dim3(scan::kThreads), 0, st, n, w.counts, tile_sums);
	hipLaunchKernelGGL(k_c_scan_tiles, dim3(1), dim3(scan::kThreads), 0, st, n_tiles, tile_sums, totals);
	hipLaunchKernelGGL((k_c_write<TCnt, STRIDE>), dim3(n_tiles), dim3(scan::kThreads), 0, st, n, tile_sums, w);
	NR3D_LAUNCH_CHECK();
	return 0;
}

}  // namespace glue
}  // namespace nr3d
