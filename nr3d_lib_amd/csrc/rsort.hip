// nr3d_lib_amd/csrc/rsort.hip -- the library's own device sort (round 5; replaces the hipCUB radix sort of round 4, the one piece of
// third-party device code libnr3d_hip.so carried): a stable LSD radix sort of (u32 key, u32 value) pairs for gfx950, up to two
// independent sorts of the same length per launch (grid.y), element count optionally read from device memory.
//
// What it is for: lotd_sorted.hip orders the POINTS of a dL/dparam pass by (table block, cell row) -- keys with few significant
// bits (17 for the reference's forest workload), so the sort is two 9-bit passes, not the five a 36-bit key needs.  The reference
// has no library sort on its default path either (pack_ops_cuda.cu:2621-2629 compiles thrust out; :2634-2720 is its own kernel).
//
// One pass = three launches:
//   k_hist     workgroup = tile of 6 144 elements: digit histogram in LDS -> hist[digit][tile]
//   k_scan     workgroup = one digit: exclusive scan of its row over the tiles (in place), digit total
//   k_scatter  workgroup = tile: stable rank of every element among the tile's elements with the same digit -- per wave by matching
//              digits across lanes with ballots (one LDS counter row per wave, bumped by the leader lane of each match group: no
//              atomics, ranks follow element order), then across waves by a scan of the 16 counter rows -- elements staged in LDS
//              in digit order and written out as runs (a digit's run of the tile is contiguous: one 128-byte request serves
//              32 elements where a direct scatter issues 32).
// Nothing here depends on the key distribution; the result is the stable order, identical on every run.
#include "common.h"

namespace nr3d {
namespace rsort {

// 6 elements per thread: 66-80 KB of LDS per workgroup, two workgroups per CU -- one loads / ranks while the other writes out (12 per thread,
// one workgroup per CU: 3.65 M pairs x 2, 15 bits: 144 -> 113 us; 2^20 pairs, 32 bits: 113 -> 83 us; tools/bench_rsort.py)
constexpr uint32_t kThreads = 1024, kWaves = kThreads / 64, kItems = 6, kTile = kThreads * kItems;

struct Job { const uint32_t *kin, *vin; uint32_t *kout, *vout; };
struct Args {
	Job job[2];
	const uint32_t *n_dev;       // element count on the device (<= n_max), or NULL: n_max
	uint32_t n_max, n_tiles, shift, mask;   // digit of a key: (key >> shift) & mask (the last pass may be narrower than the digit width)
	uint32_t *hist, *totals;     // [batch][bins][n_tiles], [batch][bins]
};

__device__ __forceinline__ uint32_t count_of(const Args &a) {
	if (!a.n_dev) return a.n_max;
	const uint32_t n = *a.n_dev;
	return n < a.n_max ? n : a.n_max;
}

template <int DB>
__global__ __launch_bounds__(kThreads) void k_hist(Args a) {
	constexpr uint32_t BINS = 1u << DB;
	__shared__ uint32_t h[BINS];
	const uint32_t n = count_of(a), tile = blockIdx.x, base = tile * kTile;
	for (uint32_t i = threadIdx.x; i < BINS; i += kThreads) h[i] = 0u;
	__syncthreads();
	const uint32_t *__restrict__ k = a.job[blockIdx.y].kin;
#pragma unroll
	for (uint32_t j = 0; j < kItems; ++j) {
		const uint32_t i = base + j * kThreads + threadIdx.x;
		if (i < n) atomicAdd(&h[(k[i] >> a.shift) & a.mask], 1u);
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < BINS; i += kThreads) a.hist[((size_t)blockIdx.y * BINS + i) * a.n_tiles + tile] = h[i];
}

__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v, uint32_t lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off, 64); if ((int)lane >= off) v += t; }
	return v;
}

// grid (bins, batch): the digit's row of tile counts -> exclusive prefix (in place), its sum -> totals
template <int DB>
__global__ __launch_bounds__(256) void k_scan(Args a) {
	constexpr uint32_t BINS = 1u << DB;
	__shared__ uint32_t wt[4];
	uint32_t *row = a.hist + ((size_t)blockIdx.y * BINS + blockIdx.x) * a.n_tiles;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < a.n_tiles; base += 256u) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < a.n_tiles ? row[i] : 0u;
		const uint32_t inc = wave_inclusive(v, lane);
		if (lane == 63u) wt[wave] = inc;
		__syncthreads();
		uint32_t off = 0, tot = 0;
#pragma unroll
		for (uint32_t w = 0; w < 4u; ++w) { const uint32_t t = wt[w]; if (w < wave) off += t; tot += t; }
		if (i < a.n_tiles) row[i] = carry + off + inc - v;
		carry += tot;
		__syncthreads();
	}
	if (threadIdx.x == 0) a.totals[blockIdx.y * BINS + blockIdx.x] = carry;
}

// exclusive scan over the workgroup's threads (callers put 0 beyond the bins); wt: 16 words of LDS
__device__ __forceinline__ uint32_t group_exclusive(uint32_t v, uint32_t *wt) {
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t inc = wave_inclusive(v, lane);
	if (lane == 63u) wt[wave] = inc;
	__syncthreads();
	uint32_t off = 0;
	for (uint32_t w = 0; w < wave; ++w) off += wt[w];
	__syncthreads();
	return off + inc - v;
}

template <int DB>
__global__ __launch_bounds__(kThreads) void k_scatter(Args a) {
	constexpr uint32_t BINS = 1u << DB;
	extern __shared__ __attribute__((aligned(16))) uint32_t rs_lds[];
	uint32_t *cnt = rs_lds;                                // [kWaves][BINS] per-wave digit counters, then the wave's base inside the digit's run
	uint32_t *stage_k = cnt + kWaves * BINS, *stage_v = stage_k + kTile;
	uint32_t *lbase = stage_v + kTile, *gofs = lbase + BINS, *wt = gofs + BINS;
	const uint32_t n = count_of(a), tile = blockIdx.x, base = tile * kTile;
	if (base >= n) return;
	const Job jb = a.job[blockIdx.y];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (uint32_t i = threadIdx.x; i < kWaves * BINS; i += kThreads) cnt[i] = 0u;
	// element (wave, item, lane) = tile position wave * 64 * kItems + item * 64 + lane: a wave owns a contiguous piece, in order
	uint32_t key[kItems], val[kItems], rank[kItems];
	const uint32_t first = base + wave * 64u * kItems + lane;
#pragma unroll
	for (uint32_t j = 0; j < kItems; ++j) {
		const uint32_t i = first + j * 64u;
		key[j] = i < n ? jb.kin[i] : 0u;
		val[j] = i < n ? (jb.vin ? jb.vin[i] : i) : 0u;
	}
	__syncthreads();
	volatile uint32_t *cw = cnt + wave * BINS;
	const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
	for (uint32_t j = 0; j < kItems; ++j) {
		const bool valid = first + j * 64u < n;
		const uint32_t d = (key[j] >> a.shift) & a.mask;
		uint64_t m = __ballot(valid);
#pragma unroll
		for (int b = 0; b < DB; ++b) {
			const bool bit = (d >> b) & 1u;
			const uint64_t bal = __ballot(bit);
			m &= bit ? bal : ~bal;
		}
		// m: the valid lanes holding my digit (me included when valid); its lowest lane speaks for the group
		const int leader = valid ? __ffsll((long long)m) - 1 : (int)lane;
		uint32_t old = 0u;
		if (valid && (int)lane == leader) { old = cw[d]; cw[d] = old + (uint32_t)__popcll(m); }
		old = __shfl(old, leader, 64);
		rank[j] = old + (uint32_t)__popcll(m & below);
	}
	__syncthreads();
	// per digit: the waves' counts -> each wave's offset inside the digit's run; the tile's count of the digit
	uint32_t tile_cnt = 0u;
	if (threadIdx.x < BINS) {
		for (uint32_t w = 0; w < kWaves; ++w) { const uint32_t c = cnt[w * BINS + threadIdx.x]; cnt[w * BINS + threadIdx.x] = tile_cnt; tile_cnt += c; }
	}
	const uint32_t lb = group_exclusive(tile_cnt, wt);                                        // where the digit's run starts in the staged tile
	const uint32_t dtot = threadIdx.x < BINS ? a.totals[blockIdx.y * BINS + threadIdx.x] : 0u;
	const uint32_t db = group_exclusive(dtot, wt);                                            // where the digit starts in the output
	if (threadIdx.x < BINS) {
		lbase[threadIdx.x] = lb;
		gofs[threadIdx.x] = db + a.hist[((size_t)blockIdx.y * BINS + threadIdx.x) * a.n_tiles + tile] - lb;
	}
	__syncthreads();
#pragma unroll
	for (uint32_t j = 0; j < kItems; ++j) {
		if (first + j * 64u < n) {
			const uint32_t d = (key[j] >> a.shift) & a.mask;
			const uint32_t p = lbase[d] + cnt[wave * BINS + d] + rank[j];
			stage_k[p] = key[j];
			stage_v[p] = val[j];
		}
	}
	__syncthreads();
	const uint32_t tile_n = (n - base) < kTile ? (n - base) : kTile;
	for (uint32_t i = threadIdx.x; i < tile_n; i += kThreads) {
		const uint32_t k = stage_k[i];
		const uint32_t o = gofs[(k >> a.shift) & a.mask] + i;
		jb.kout[o] = k;
		jb.vout[o] = stage_v[i];
	}
}

template <int DB> constexpr size_t scatter_lds() { return (size_t)(kWaves * (1u << DB) + 2u * kTile + 2u * (1u << DB) + 16u) * 4u; }

static inline uint32_t tiles_of(uint32_t n) { return n ? (n + kTile - 1u) / kTile : 1u; }
static inline size_t align256(size_t b) { return (b + 255u) / 256u * 256u; }

// scratch: histograms + digit totals of one pass, and one pair of ping-pong buffers per sort
size_t tmp_bytes(uint32_t n_max, int batch) {
	return align256((size_t)batch * 512u * tiles_of(n_max) * 4u) + align256((size_t)batch * 512u * 4u) + (size_t)batch * 2u * align256(4ull * n_max);
}

// Sorts `batch` (1 or 2) independent arrays of pairs by key bits [0, bits): kin[b] / vin[b] -> kout[b] / vout[b], stable.
// vin[b] == NULL: the values are the element indices 0, 1, 2 ...  n_dev (optional): element count in device memory, <= n_max.
// The inputs are left untouched; kout / vout must not alias them.
int sort_pairs(void *tmp, int batch, const uint32_t *const *kin, const uint32_t *const *vin, uint32_t *const *kout, uint32_t *const *vout,
               uint32_t n_max, const uint32_t *n_dev, int bits, hipStream_t st) {
	NR3D_CHECK(batch == 1 || batch == 2, "rsort: batch %d", batch);
	NR3D_CHECK(bits >= 0 && bits <= 32, "rsort: %d key bits", bits);
	NR3D_CHECK(n_max < (1u << 31), "rsort: %u elements in one call, the limit is 2^31 - 1 (32-bit positions, a tile may overhang)", n_max);
	if (n_max == 0) return 0;
	bits = bits < 1 ? 1 : bits;
	// 8-bit digits when they need no more passes than 9-bit ones (half the counters)
	const int passes9 = (bits + 8) / 9, passes8 = (bits + 7) / 8;
	const int DB = passes8 == passes9 ? 8 : 9, passes = passes9;
	char *p = (char *)tmp;
	Args a;
	a.n_dev = n_dev; a.n_max = n_max; a.n_tiles = tiles_of(n_max);
	a.hist = (uint32_t *)p; p += align256((size_t)batch * 512u * a.n_tiles * 4u);
	a.totals = (uint32_t *)p; p += align256((size_t)batch * 512u * 4u);
	uint32_t *tk[2], *tv[2];
	for (int b = 0; b < batch; ++b) { tk[b] = (uint32_t *)p; p += align256(4ull * n_max); tv[b] = (uint32_t *)p; p += align256(4ull * n_max); }
	static bool attr[64] = {};
	int dev_id = 0;
	NR3D_HIP_CHECK(hipGetDevice(&dev_id));
	if (!attr[dev_id & 63]) {
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_scatter<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds<8>()));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_scatter<9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds<9>()));
		attr[dev_id & 63] = true;
	}
	for (int ps = 0; ps < passes; ++ps) {
		const bool to_out = ((passes - 1 - ps) & 1) == 0;          // the last pass lands in kout / vout
		for (int b = 0; b < batch; ++b) {
			a.job[b].kin = ps == 0 ? kin[b] : (to_out ? tk[b] : kout[b]);
			a.job[b].vin = ps == 0 ? vin[b] : (to_out ? tv[b] : vout[b]);
			a.job[b].kout = to_out ? kout[b] : tk[b];
			a.job[b].vout = to_out ? vout[b] : tv[b];
		}
		a.shift = (uint32_t)(ps * DB);
		a.mask = (1u << (bits - ps * DB < DB ? bits - ps * DB : DB)) - 1u;
		const dim3 gt(a.n_tiles, batch);
		if (DB == 8) {
			hipLaunchKernelGGL(k_hist<8>, gt, dim3(kThreads), 0, st, a);
			hipLaunchKernelGGL(k_scan<8>, dim3(256, batch), dim3(256), 0, st, a);
			hipLaunchKernelGGL(k_scatter<8>, gt, dim3(kThreads), scatter_lds<8>(), st, a);
		} else {
			hipLaunchKernelGGL(k_hist<9>, gt, dim3(kThreads), 0, st, a);
			hipLaunchKernelGGL(k_scan<9>, dim3(512, batch), dim3(256), 0, st, a);
			hipLaunchKernelGGL(k_scatter<9>, gt, dim3(kThreads), scatter_lds<9>(), st, a);
		}
		NR3D_LAUNCH_CHECK();
	}
	return 0;
}

}  // namespace rsort
}  // namespace nr3d
