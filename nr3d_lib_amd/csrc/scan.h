// nr3d_lib_amd/csrc/scan.h -- device-wide exclusive scan of per-pack counts into (begin, length)
// pack descriptors, plus the grand total.  Replaces the reference's host-side
// `cumsum` + `stack` + `.item()` sequence of every two-phase op (e.g. ray_marching.cu:205-209,
// pack_ops_cuda.cu:584-586): here the scan stays on the device and the caller does ONE readback.
//
// Three launches (reduce tiles -> scan tile sums -> rescan + write); wave64 shuffles inside a tile.  Up to 32768 counts:
// one launch of one workgroup.
#pragma once
#include "common.h"

namespace nr3d {
namespace scan {

constexpr int kThreads = 256;
constexpr int kItems = 8;                       // per thread
constexpr int kTile = kThreads * kItems;        // 2048 counts per workgroup

__device__ __forceinline__ uint64_t wave_inclusive(uint64_t v, int lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint64_t t = __shfl_up(v, off, 64);
		if (lane >= off) v += t;
	}
	return v;
}

// exclusive prefix of `v` across the 256-thread block; returns block total in `total`
__device__ __forceinline__ uint64_t block_exclusive(uint64_t v, uint64_t &total, uint64_t *lds /*[4]*/) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t inc = wave_inclusive(v, lane);
	if (lane == 63) lds[wave] = inc;
	__syncthreads();
	uint64_t wave_off = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < kThreads / 64; ++w) {
		const uint64_t s = lds[w];
		if (w < wave) wave_off += s;
		tot += s;
	}
	__syncthreads();
	total = tot;
	return wave_off + inc - v;
}

template <typename TIn>
__global__ __launch_bounds__(kThreads) void k_tile_sums(uint64_t n, const TIn *__restrict__ counts,
                                                        uint64_t *__restrict__ tile_sums) {
	__shared__ uint64_t lds[4];
	const uint64_t base = (uint64_t)blockIdx.x * kTile;
	uint64_t s = 0;
#pragma unroll
	for (int k = 0; k < kItems; ++k) {
		const uint64_t i = base + (uint64_t)k * kThreads + threadIdx.x;
		if (i < n) s += (uint64_t)counts[i];
	}
	uint64_t tot;
	block_exclusive(s, tot, lds);
	if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single workgroup: in-place exclusive scan of the tile sums, total -> *total_out (int64)
static __global__ __launch_bounds__(kThreads) void k_scan_tile_sums(uint32_t n_tiles, uint64_t *__restrict__ tile_sums,
                                                             int64_t *__restrict__ total_out) {
	__shared__ uint64_t lds[4];
	uint64_t carry = 0;
	for (uint32_t base = 0; base < n_tiles; base += kThreads) {
		const uint32_t i = base + threadIdx.x;
		const uint64_t v = i < n_tiles ? tile_sums[i] : 0;
		uint64_t tot;
		const uint64_t ex = block_exclusive(v, tot, lds);
		if (i < n_tiles) tile_sums[i] = carry + ex;
		carry += tot;
	}
	if (threadIdx.x == 0) __hip_atomic_store(total_out, (int64_t)carry, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // a host thread may be polling it
}

// rescan each tile, add its prefix and write (begin, length) pairs.  Thread t owns kItems CONSECUTIVE counts.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void k_write_pack_infos(uint64_t n, const TIn *__restrict__ counts,
                                                               const uint64_t *__restrict__ tile_prefix,
                                                               TOut *__restrict__ pack_infos) {
	__shared__ uint64_t lds[4];
	const uint64_t first = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kItems;
	uint64_t c[kItems], s = 0;
#pragma unroll
	for (int k = 0; k < kItems; ++k) {
		c[k] = (first + k < n) ? (uint64_t)counts[first + k] : 0;
		s += c[k];
	}
	uint64_t tot;
	uint64_t run = tile_prefix[blockIdx.x] + block_exclusive(s, tot, lds);
#pragma unroll
	for (int k = 0; k < kItems; ++k) {
		if (first + k < n) {
			pack_infos[2 * (first + k)] = (TOut)run;
			pack_infos[2 * (first + k) + 1] = (TOut)c[k];
		}
		run += c[k];
	}
}

// n <= kSmallMax: ONE workgroup does the whole job (the three launches above cost ~14 us of launch and drain for 4096
// counts, this one ~4): thread t owns `per` consecutive counts, sums them, one block scan, then re-reads and writes.
constexpr uint64_t kSmallMax = 32768;
// (round 6: 1024 threads, a thread's PER counts loaded once into registers -- see compact.h k_c_small)
constexpr int kSmallThreads = 1024;
template <typename TIn, typename TOut, int PER>
__global__ __launch_bounds__(kSmallThreads) void k_pack_infos_small(uint32_t n, const TIn *__restrict__ counts,
                                                                    TOut *__restrict__ pack_infos, int64_t *__restrict__ total_out) {
	__shared__ uint64_t lds[kSmallThreads / 64];
	const uint32_t first = threadIdx.x * PER;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint64_t v[PER], s = 0;
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		v[k] = (first + k < n) ? (uint64_t)counts[first + k] : 0;
		s += v[k];
	}
	const uint64_t inc = wave_inclusive(s, lane);
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
		if (first + k < n) {
			pack_infos[2 * (size_t)(first + k)] = (TOut)run;
			pack_infos[2 * (size_t)(first + k) + 1] = (TOut)v[k];
		}
		run += v[k];
	}
	if (threadIdx.x == 0) __hip_atomic_store(total_out, (int64_t)tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);     // a host thread may be polling it
}

static inline uint64_t tmp_bytes(uint64_t n) { return ((n + kTile - 1) / kTile + 1) * sizeof(uint64_t); }

// counts[n] -> pack_infos[n,2] (TOut) and total[0] (int64); tmp >= tmp_bytes(n)
template <typename TIn, typename TOut>
static int pack_infos_from_counts(uint64_t n, const TIn *counts, TOut *pack_infos, int64_t *total, void *tmp,
                                  hipStream_t st) {
	if (n == 0) {
		NR3D_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(int64_t), st));
		return 0;
	}
	if (n <= kSmallMax) {
		const uint32_t per = (uint32_t)((n + kSmallThreads - 1) / kSmallThreads);
#define NR3D_PI_SMALL(PER) hipLaunchKernelGGL((k_pack_infos_small<TIn, TOut, PER>), dim3(1), dim3(kSmallThreads), 0, st, (uint32_t)n, counts, pack_infos, total)
		if (per <= 1) NR3D_PI_SMALL(1); else if (per <= 2) NR3D_PI_SMALL(2); else if (per <= 4) NR3D_PI_SMALL(4);
		else if (per <= 8) NR3D_PI_SMALL(8); else if (per <= 16) NR3D_PI_SMALL(16); else NR3D_PI_SMALL(32);
#undef NR3D_PI_SMALL
		NR3D_LAUNCH_CHECK();
		return 0;
	}
	const uint32_t n_tiles = (uint32_t)((n + kTile - 1) / kTile);
	uint64_t *tile_sums = (uint64_t *)tmp;
	hipLaunchKernelGGL(k_tile_sums<TIn>, dim3(n_tiles), dim3(kThreads), 0, st, n, counts, tile_sums);
	hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kThreads), 0, st, n_tiles, tile_sums, total);
	hipLaunchKernelGGL((k_write_pack_infos<TIn, TOut>), dim3(n_tiles), dim3(kThreads), 0, st, n, counts, tile_sums,
	                   pack_infos);
	NR3D_LAUNCH_CHECK();
	return 0;
}

}  // namespace scan
}  // namespace nr3d
