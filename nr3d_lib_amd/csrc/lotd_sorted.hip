// nr3d_lib_amd/csrc/lotd_sorted.hip -- its own translation unit since round 6 (was lotd_sorted.inc inside lotd_bin.hip, after k_vm_direct).
//
// dL/dparam of LARGE VM levels over SORTED points, without records (round 4; the "(block, level)-major schedule over block-sorted
// points" of the round-3 review).  The record path costs O(point blocks x buckets) whatever the records are: a 128-point stage-A
// block scans, sorts into and writes offsets for every 8192-entry bucket of the level, and stage B walks every (bucket, block) run.
// The reference's forest workload (unit_test_forest.py: 6 blocks, VM planes up to 1600 x 1600, 3.65 M points) has 9 100 buckets
// and less than one record per run: 17.8 + 14.5 ms, hardly any of it record traffic.
//
// Here the POINTS are sorted instead of their updates -- once along x_0 and once along x_1, inside their block.  The plane of
// component d is stored row-major over dim a(d) (x_1 for d = 0, x_0 for d = 1, 2) and the cell row is monotone in x_a at EVERY
// level, so in any order that is monotone in every served level's row the points of a band of rows are one contiguous range, found
// by binary search.  The sort key is the coarsest such order (round 5): (block, SUM over the served levels of the point's cell row)
// -- a sum of monotone step functions steps wherever one of them does, so it refines every level's rows, and it has as many
// values as the levels have rows together (17 bits with the block for the reference's forest workload, against the 36 of
// (block, float bits of x_a) in round 4): two 9-bit passes of the library's own radix sort (rsort.hip; round 4 called hipCUB).  A work item = (block, pseudo level, component, band of rows that fits LDS): its workgroup adds the updates of
// its range (emit_vm_component, the arithmetic of the record path) to fp64 accumulators in LDS -- plane band + line d -- and
// writes the band to dL/dparam itself.  No record, no bucket histogram, no offsets table; every point is evaluated once per
// (pseudo level, component).  A band with more than kVsPmax points is split into replicas whose tables a second kernel adds in
// a fixed order (so are the line tables of all bands of a component and the row two neighbouring bands share): no atomics on
// global memory, the result does not depend on the schedule.
// Forest (FO): cells whose corners all lie in the point's own block are the plain level shifted by one node (as in bin_body);
// the others -- a ~6/R fraction, nested over the levels -- come from two lists of the points near a block face (near enough for the
// coarsest level served), of all blocks, ordered by x_0 and by x_1: band workgroup (B, level, d, rows) takes the range of the
// list whose cells can reach its rows (plus the two wrap-around rows a neighbour's first / last cell lands on), resolves every
// corner's owner through the octree and keeps the corners that B owns and whose plane row lies in its band (the per-corner
// arithmetic of emit_forest).  (First version: one list ordered by face distance, every band walking the whole prefix of its
// level -- 110 M visits and 2.2 of the kernel's 3.8 ms on the reference's forest workload; by row ranges 15 M.)
#include "lotd_sorted.h"
#include <type_traits>

namespace nr3d {
namespace lotd {


// per item (k_vs_plan): what k_vs_tmax / k_vs_units / k_vs_reduce would otherwise derive again by walking vp.item_base (a chain of ~20
// dependent scalar loads per workgroup once the plan lives in memory)
struct VsItemRec { uint32_t qd, lv, b, band, row0, nrows, last, boff; };
struct VsDev {
	VsItemRec *irec;                                                     // [n_items]
	uint32_t *item_lo, *item_cnt, *rep_base, *multi_base;                // [n_items (+ 1)]
	uint32_t *sub_lo, *sub_cnt;                                          // forest: [3][n_items] ranges of the candidate list
	const uint32_t *skey[2];               // sorted keys of order o: (block << row_bits[o]) | row sum of x_o
	const uint32_t *n_cand;                // forest: how many boundary candidates the pass has
	const struct VsUnit *units;            // [w_max] what every work unit starts from (k_vs_units)
	uint32_t *stats;                       // float bits of max |dL_dy| over the served columns, of max |dL_ddLdx| (k_vs_gather); [2]: items with replicas
	uint32_t *multi_list;                  // those items (k_vs_units)
	const uint32_t *permb[2];              // forest: the boundary candidates of ALL blocks in the order of x_o (their points)
	const float *xm[2];                    // ... and those points' coordinates [n, 3]
	const struct VsCand *cinfo[2];         // ... and their (point index, block, block position) records
	const float *xs[2], *vs[2], *gts[2];   // x [n, 3], dL_ddLdx [n, 3], dL_dy [E, n] in order o
	float *slots, *handoff, *lines;
};
struct VsItem { uint32_t qd, b, band, row0, nrows, q, d; bool last; };

__device__ __forceinline__ VsItem vs_item(const VsPlan &vp, uint32_t item, uint32_t cells) {
	VsItem it;
	it.qd = 0;
	while (it.qd + 1 < vp.n_qd && vp.item_base[it.qd + 1] <= item) ++it.qd;
	const uint32_t rem = item - vp.item_base[it.qd], nb = vp.n_bands[it.qd], rows = vp.rows[it.qd];
	it.b = rem / nb; it.band = rem - it.b * nb;
	it.row0 = it.band * rows;
	it.nrows = (cells - it.row0) < rows ? (cells - it.row0) : rows;
	it.last = it.band + 1 == nb;
	it.q = vp.q[it.qd]; it.d = vp.d[it.qd];
	return it;
}

// the row field of a sort key: sum over the distinct scales of order o of the point's (clamped) cell row, the row computed with
// the arithmetic of locate() / locate_forest() -- floor(fma(x, scale, 1/2)) -- so that the searches below, which apply the same
// expression to one scale, see a monotone sequence
__device__ __forceinline__ uint32_t vs_rowsum(const VsPlan &vp, int o, float v) {
	if (!(v >= 0.0f)) return vp.row_none[o];                       // negative or NaN: behind every row of the block
	float s = 0.0f;                                                // (exact: integers below 2^24, checked by the plan)
	for (uint32_t k = 0; k < vp.n_sc[o]; ++k) s += fminf(floorf(__fmaf_rn(v, vp.sc[o][k], 0.5f)), vp.cap[o][k]);
	return (uint32_t)s;
}

__device__ __forceinline__ bool vs_is_candidate(float x0, float x1, float x2, float thr_max) {
	const float m = fminf(fminf(fminf(x0, 1.0f - x0), fminf(x1, 1.0f - x1)), fminf(x2, 1.0f - x2));
	return m >= 0.0f && m < thr_max;                               // (NaN / outside the block: never)
}

// the plan -> device memory, once per pass (a kernel argument that is indexed at run time is copied to LDS / scratch by every
// workgroup: 8 KiB per workgroup, the larger part of k_vs_reduce's 0.35 ms)
__global__ __launch_bounds__(256) void k_vs_store_plan(VsPlan vp, uint32_t *__restrict__ dst) {
	static_assert(sizeof(VsPlan) % 4 == 0, "word copy");
	const uint32_t *src = reinterpret_cast<const uint32_t *>(&vp);
	for (uint32_t t = threadIdx.x; t < sizeof(VsPlan) / 4u; t += 256u) dst[t] = src[t];
}

// sort keys (block, row sum along x_0), (block, row sum along x_1); forest: how many of the workgroup's points lie near a face of
// their block (the boundary candidates)
template <bool FO>
__global__ __launch_bounds__(1024) void k_vs_keys(const VsPlan *__restrict__ vpp, uint32_t n, const float *__restrict__ x, Batch ba,
                                                  uint32_t *__restrict__ key0, uint32_t *__restrict__ key1, uint32_t *__restrict__ cand_cnt) {
	const VsPlan &vp = *vpp;                      // (the plan lives in device memory: indexed by value it was copied into LDS by every workgroup)
	__shared__ uint32_t wcnt;
	if (FO) { if (threadIdx.x == 0) wcnt = 0u; __syncthreads(); }
	const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
	bool cand = false;
	if (i < n) {
		uint32_t pbase, bi;
		const bool ok = batch_base_index(ba, i, pbase, bi) && bi < vp.n_blocks;
		const uint32_t blk = ok ? bi : vp.n_blocks;
		const float x0 = x[(size_t)i * 3], x1 = x[(size_t)i * 3 + 1], x2 = x[(size_t)i * 3 + 2];
		key0[i] = (blk << vp.row_bits[0]) | vs_rowsum(vp, 0, x0);
		key1[i] = (blk << vp.row_bits[1]) | vs_rowsum(vp, 1, x1);
		cand = FO && ok && vs_is_candidate(x0, x1, x2, vp.thr_max);
	}
	if (FO) {
		const uint64_t m = __ballot(cand);
		if ((threadIdx.x & 63u) == 0u && m) atomicAdd(&wcnt, (uint32_t)__popcll(m));
		__syncthreads();
		if (threadIdx.x == 0) cand_cnt[blockIdx.x] = wcnt;
	}
}

// one workgroup: exclusive scan of the per-workgroup candidate counts (in place), the total
__global__ __launch_bounds__(1024) void k_vs_cand_scan(uint32_t n_wg, uint32_t *__restrict__ cand_cnt, uint32_t *__restrict__ n_cand) {
	__shared__ uint32_t wsum[16];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t carry = 0u;
	for (uint32_t base = 0; base < n_wg; base += 1024u) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < n_wg ? cand_cnt[i] : 0u;
		uint32_t inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off, 64); if ((int)lane >= off) inc += t; }
		if (lane == 63u) wsum[wave] = inc;
		__syncthreads();
		uint32_t woff = 0u, tot = 0u;
		for (uint32_t j = 0; j < 16u; ++j) { const uint32_t t = wsum[j]; if (j < wave) woff += t; tot += t; }
		if (i < n_wg) cand_cnt[i] = carry + woff + inc - v;
		carry += tot;
		__syncthreads();
	}
	if (threadIdx.x == 0) *n_cand = carry;
}

// the candidates in point order (a compaction: the k-th candidate of the pass lands at k): their points and their two sort keys
__global__ __launch_bounds__(1024) void k_vs_cand_write(const VsPlan *__restrict__ vpp, uint32_t n, const float *__restrict__ x, Batch ba,
                                                        const uint32_t *__restrict__ cand_base, uint32_t *__restrict__ idxb,
                                                        uint32_t *__restrict__ keyb0, uint32_t *__restrict__ keyb1) {
	const VsPlan &vp = *vpp;                      // (the plan lives in device memory: indexed by value it was copied into LDS by every workgroup)
	__shared__ uint32_t wcnt[16];
	const uint32_t i = blockIdx.x * 1024u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	bool cand = false;
	float x0 = 0.0f, x1 = 0.0f;
	if (i < n) {
		uint32_t pbase, bi;
		const bool ok = batch_base_index(ba, i, pbase, bi) && bi < vp.n_blocks;
		x0 = x[(size_t)i * 3]; x1 = x[(size_t)i * 3 + 1];
		cand = ok && vs_is_candidate(x0, x1, x[(size_t)i * 3 + 2], vp.thr_max);
	}
	const uint64_t m = __ballot(cand);
	if (lane == 0u) wcnt[wave] = (uint32_t)__popcll(m);
	__syncthreads();
	if (!cand) return;
	uint32_t at = cand_base[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
	for (uint32_t w = 0; w < wave; ++w) at += wcnt[w];
	idxb[at] = i;
	keyb0[at] = vs_rowsum(vp, 0, x0);
	keyb1[at] = vs_rowsum(vp, 1, x1);
}

// x, dL_ddLdx and dL_dy (-> feature-major) in the order of `perm`: 128 points per workgroup, rows read whole, columns written whole;
// only the columns of the pseudo levels the plan serves (cols: one bit per column, E <= 120)
constexpr uint32_t kVsGatherPts = 128;
__global__ __launch_bounds__(256) void k_vs_gather(uint32_t n, uint32_t E, const uint32_t *__restrict__ perm, const float *__restrict__ x,
                                                   const float *__restrict__ vin, const float *__restrict__ g, int64_t g_sn, int64_t g_se,
                                                   uint64_t cols0, uint64_t cols1, float *__restrict__ xs, float *__restrict__ vs,
                                                   float *__restrict__ gts, uint32_t *__restrict__ stats) {
	extern __shared__ __attribute__((aligned(16))) float vg_tile[];               // [W wanted columns][kVsGatherPts + 1]
	__shared__ uint32_t src[kVsGatherPts];
	__shared__ uint8_t wlist[128];                                                 // the wanted columns, ascending
	const uint32_t i0 = blockIdx.x * kVsGatherPts;
	const uint32_t np = min(kVsGatherPts, n - i0);
	if (threadIdx.x < np) src[threadIdx.x] = perm[i0 + threadIdx.x];
	auto wanted = [&](uint32_t e) { return (((e < 64u ? cols0 : cols1) >> (e & 63u)) & 1ull) != 0ull; };
	const uint32_t W = (uint32_t)(__popcll(cols0) + __popcll(cols1));
	if (threadIdx.x < 128u && threadIdx.x < E && wanted(threadIdx.x)) {
		const uint32_t e = threadIdx.x;
		const uint32_t below = e < 64u ? (uint32_t)__popcll(cols0 & ((1ull << e) - 1ull))
		                               : (uint32_t)(__popcll(cols0) + __popcll(cols1 & ((1ull << (e - 64u)) - 1ull)));
		wlist[below] = (uint8_t)e;
	}
	__syncthreads();
	uint32_t gb = 0u, vb = 0u;                                                    // float bits of max |dL_dy|, max |dL_ddLdx| (stats != nullptr: order 0 only)
	{
		// thread -> (row of the pass, k-th WANTED column) by shift and mask over the next power of two >= W: no division per load, and
		// no lanes parked on columns nobody serves (configs[3]: 14 of 50 columns -- with lanes over all E columns a row load ran 14 of
		// 64 lanes, 527 us per order at 2^22 points)
		uint32_t lg = 0;
		while ((1u << lg) < W) ++lg;                                            // W <= E <= 120: lg <= 7
		const uint32_t k = threadIdx.x & ((1u << lg) - 1u), r0 = threadIdx.x >> lg, step = 256u >> lg;
		if (k < W) {
			const float *ge = g + (int64_t)wlist[k] * g_se;
			float *te = vg_tile + k * (kVsGatherPts + 1u);
			for (uint32_t i = r0; i < np; i += step) {
				const float v = ge[(int64_t)src[i] * g_sn];
				te[i] = v;
				gb = max(gb, __float_as_uint(v) & 0x7FFFFFFFu);
			}
		}
	}
	for (uint32_t t = threadIdx.x; t < np * 3u; t += 256u) {
		const uint32_t i = t / 3u, d = t - i * 3u;
		xs[(size_t)i0 * 3 + t] = x[(size_t)src[i] * 3 + d];
		if (vin) { const float v = vin[(size_t)src[i] * 3 + d]; vs[(size_t)i0 * 3 + t] = v; vb = max(vb, __float_as_uint(v) & 0x7FFFFFFFu); }
	}
	if (stats) {
		// |float| bits order like unsigned integers (NaN / inf on top).  One candidate per wave, and the atomic only when it would raise the
		// running maximum (an L2 load first): tens of thousands of same-address atomics would serialise
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) { gb = max(gb, (uint32_t)__shfl_xor((int)gb, off, 64)); vb = max(vb, (uint32_t)__shfl_xor((int)vb, off, 64)); }
		if ((threadIdx.x & 63u) == 0u) {
			if (gb > __hip_atomic_load(stats, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stats, gb);
			if (vb > __hip_atomic_load(stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(stats + 1, vb);
		}
	}
	__syncthreads();
	for (uint32_t o = threadIdx.x; o < W * kVsGatherPts; o += 256u) {
		const uint32_t k = o / kVsGatherPts, i = o % kVsGatherPts;
		if (i < np) gts[(size_t)wlist[k] * n + i0 + i] = vg_tile[k * (kVsGatherPts + 1u) + i];
	}
}

// first position whose point is at or behind (block b, cell row r) of the level with row scale sc -- the row with the arithmetic
// of locate() / locate_forest(), from the point's own coordinate (xs: the points in this order, [n][3]; o: the coordinate)
__device__ __forceinline__ uint32_t vs_search(const uint32_t *__restrict__ keys, const float *__restrict__ xs, int o, uint32_t row_bits,
                                              uint32_t row_none, uint32_t n, uint32_t b, float sc, float r) {
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		const uint32_t k = keys[mid];
		const uint32_t kb = k >> row_bits, rf = k & ((1u << row_bits) - 1u);
		const bool ge = kb > b || (kb == b && (rf == row_none || floorf(__fmaf_rn(xs[(size_t)mid * 3 + o], sc, 0.5f)) >= r));
		if (ge) hi = mid; else lo = mid + 1;
	}
	return lo;
}

// the point range of every work item; the boundary prefix of every (pseudo level, component)
template <bool FO>
__global__ __launch_bounds__(256) void k_vs_plan(const VsPlan *__restrict__ vpp, VsDev dv, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n, Batch ba) {
	const VsPlan &vp = *vpp;                      // (the plan lives in device memory: indexed by value it was copied into LDS by every workgroup)
	const uint32_t item = blockIdx.x * 256u + threadIdx.x;
	if (item >= vp.n_items) return;
	// (the level of the item: its pair first, then the geometry)
	uint32_t qd = 0;
	while (qd + 1 < vp.n_qd && vp.item_base[qd + 1] <= item) ++qd;
	const Lvl L = load_level(md, meta_level_of(md, vp.q[qd]));
	const VmGeom gm = vm_geom(L.res, (int)vp.d[qd]);
	const VsItem it = vs_item(vp, item, gm.Ra - 1u);
	{
		VsItemRec ir;
		ir.qd = qd; ir.lv = meta_level_of(md, vp.q[qd]); ir.b = it.b; ir.band = it.band; ir.row0 = it.row0; ir.nrows = it.nrows; ir.last = it.last ? 1u : 0u;
		ir.boff = ba.offsets ? (uint32_t)ba.offsets[it.b] : it.b * ba.n_params;
		dv.irec[item] = ir;
	}
	const float sc = FO ? (float)gm.Ra : (float)(gm.Ra - 2u);
	const uint32_t r_lo = it.row0 + (FO ? 1u : 0u);
	const int o = it.d == 0u ? 1 : 0;
	const uint32_t *keys = dv.skey[o];
	const uint32_t lo = vs_search(keys, dv.xs[o], o, vp.row_bits[o], vp.row_none[o], n, it.b, sc, (float)r_lo),
	               hi = vs_search(keys, dv.xs[o], o, vp.row_bits[o], vp.row_none[o], n, it.b, sc, (float)(r_lo + it.nrows));
	dv.item_lo[item] = lo;
	dv.item_cnt[item] = hi - lo;
	if (FO) {
		// candidates whose cell (p = floor(x_a R + 1/2), corners p and p + 1 = rows p - 1 and p of their owner) can reach rows
		// row0 .. row0 + nrows: p in [row0, row0 + nrows + 1]; p = 0 also lands on the left neighbour's LAST row, p = R on the right
		// neighbour's row 0 (resolve_block) -- two more ranges, where they are not inside the first
		const float *__restrict__ xm = dv.xm[o];                  // the candidates in the order of x_o (valid coordinates only)
		const uint32_t n_cand = *dv.n_cand;
		auto first_ge = [&](float r) {
			uint32_t a = 0, b = n_cand;
			while (a < b) {
				const uint32_t mid = (a + b) >> 1;
				if (floorf(__fmaf_rn(xm[(size_t)mid * 3 + o], sc, 0.5f)) >= r) b = mid; else a = mid + 1;
			}
			return a;
		};
		uint32_t l0 = first_ge((float)it.row0), h0 = first_ge((float)(it.row0 + it.nrows + 2u));
		uint32_t l1 = 0, h1 = 0, l2 = 0, h2 = 0;
		if (it.last && it.row0 > 0u) h1 = first_ge(1.0f);                                  // p == 0
		if (it.band == 0u && !it.last) { l2 = first_ge((float)gm.Ra); h2 = first_ge((float)(gm.Ra + 1u)); }      // p == R
		dv.sub_lo[item] = l0; dv.sub_cnt[item] = h0 - l0;
		dv.sub_lo[vp.n_items + item] = l1; dv.sub_cnt[vp.n_items + item] = h1 - l1;
		dv.sub_lo[2u * vp.n_items + item] = l2; dv.sub_cnt[2u * vp.n_items + item] = h2 - l2;
	}
}

// exclusive scans over the items: work units (>= 1 per item, one per kVsPmax points) and slots of the items with replicas
__global__ __launch_bounds__(1024) void k_vs_scan(uint32_t n_items, VsDev dv) {
	__shared__ uint32_t wsum[2][16];
	__shared__ uint32_t carry[2];
	if (threadIdx.x < 2) carry[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (uint32_t base = 0; base < n_items; base += 1024u) {
		const uint32_t i = base + threadIdx.x;
		uint32_t v[2] = {0u, 0u};
		if (i < n_items) {
			const uint32_t c = dv.item_cnt[i];
			v[0] = c <= kVsPmax ? 1u : (c + kVsPmax - 1u) / kVsPmax;
			v[1] = v[0] > 1u ? v[0] : 0u;
		}
		uint32_t inc[2] = {v[0], v[1]};
#pragma unroll
		for (int k = 0; k < 2; ++k) {
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc[k], off, 64); if ((int)lane >= off) inc[k] += t; }
			if (lane == 63) wsum[k][wave] = inc[k];
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			uint32_t woff = 0, tot = 0;
			for (uint32_t j = 0; j < 16; ++j) { const uint32_t t = wsum[k][j]; if (j < wave) woff += t; tot += t; }
			const uint32_t ex = carry[k] + woff + inc[k] - v[k];
			if (i < n_items) (k == 0 ? dv.rep_base : dv.multi_base)[i] = ex;
			__syncthreads();
			if (threadIdx.x == 0) carry[k] += tot;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) { dv.rep_base[n_items] = carry[0]; dv.multi_base[n_items] = carry[1]; }
}

// Accumulators of k_vm_sorted (round 6): FEATURE-MAJOR -- feature f of slot s at vs_acc[f * kVsAcc + s], slots = the band's plane entries,
// then line d -- so that the 64 lanes of one atomic spread over all LDS banks (entry-major put each feature on half of them), and
// 64-bit FIXED POINT where the workgroup can bound its updates (ds_add_u64 2.5 T/s against ds_add_f64 1.47, tools/ubench_lds.hip):
//   |update| <= gmax * tmax (* 8 amax, second order: plane g (wo a_d dl + C_m LI) <= 4 |a| tmax, line g (w PC -+ a PI) <= 5 |a| tmax,
//   boundary corners g W'_k t <= 3 |a| tmax; |a| <= 1.5 R |v|),  gmax / vmax = max |dL_dy| / |dL_ddLdx| of the pass (k_vs_gather),
//   tmax = max |table value| over the band's own rows and line d (read by the workgroup before its points: the rows its gathers then hit),
// scale 2^s with |sum| < (2 own points + 8 boundary points) * bound * 2^s <= 2^62 and single values below 2^50: resolution
// bound * 2^-min(62 - log2 count, 50), exact sums -- the band does not depend on the order of its points.  A bound that is zero, not
// finite or outside the exponent range keeps fp64 accumulators (uniform per workgroup).
struct VsFix { double scale, inv; bool on; };
__device__ __forceinline__ unsigned long long vs_to_fix(float v, double scale) {
	const double t = __fma_rn((double)v, scale, 0x1.8p52);            // round to nearest integer in the low mantissa bits
	return (unsigned long long)(__double_as_longlong(t) - 0x4338000000000000LL);
}
template <bool FIX>
__device__ __forceinline__ void vs_add(double *acc, uint32_t at, float v, double scale) {
	if (FIX) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + at, vs_to_fix(v, scale));
	else atomicAdd(acc + at, (double)v);
}
__device__ __forceinline__ float vs_out(const double *acc, uint32_t at, const VsFix &fx) {
	return fx.on ? (float)((double)(long long)reinterpret_cast<const unsigned long long *>(acc)[at] * fx.inv) : (float)acc[at];
}

// what a work unit needs before it can start, in ONE record (k_vs_units).  Until round 6 every workgroup found its item by a binary
// search over rep_base, then walked vp.item_base for its (pseudo level, component), loaded the level, the band, its ranges, the block's
// table offset: five or six rounds of dependent (scalar) loads, ~5 us per work unit with the CU waiting -- 0.26 ms of the kernel on the
// forest workload (13 200 units), measured with everything behind the decode switched off.
struct VsUnit {
	uint32_t item, rep, n_rep, multi;          // item == kVsNoUnit: nothing to do
	uint32_t own_lo, own_cnt, qd, lv;          // the band's range of the order of x_a; its pair; the level's index in the meta
	uint32_t sub_lo[3], b;                     // forest: the three candidate ranges; the block
	uint32_t sub_cnt[3], row0;
	uint32_t nrows, last, boff, tmax;          // boff: first element of the block's tables; tmax: float bits of max |table value| over band + line
	uint32_t res[3], F;                        // the level (what Lvl holds of a VM level) ...
	uint32_t off, dc, foff, col0;              // ... its first element in the block's tables; the component; first feature / dL_dy column of the pseudo level
	int32_t kb[3]; float thr;                  // forest: the block's position; the level's face distance threshold
};
static_assert(sizeof(VsUnit) == 128, "eight 16-byte loads, one round trip");
constexpr uint32_t kVsNoUnit = 0xFFFFFFFFu;
constexpr uint32_t kVsRedPiece = 2048u, kVsRedChunks = (kVsSlot + kVsRedPiece - 1u) / kVsRedPiece;   // k_vs_reduce_items: pieces of an item's table
constexpr uint32_t kVsDense = 24;            // surviving boundary candidates per wave from which every lane walks its own eight corners

__device__ __forceinline__ uint32_t vs_pair_of(const VsPlan &vp, uint32_t item) {
	uint32_t qd = 0;
	while (qd + 1 < vp.n_qd && vp.item_base[qd + 1] <= item) ++qd;
	return qd;
}

// max |table value| over every item's band rows and line d (float bits; 0 for an item nobody touches): the bound behind the fixed-point
// scale of k_vm_sorted.  One workgroup per item, whole rows, eight of them per CU -- inside k_vm_sorted (first version) the same scan cost
// 0.22 ms: the one workgroup a CU holds there waits for it, replica by replica.
template <typename PT, bool FO>
__global__ __launch_bounds__(256) void k_vs_tmax(const VsPlan *__restrict__ vpp, VsDev dv, const nr3d_lotd_meta_t *__restrict__ md, const PT *__restrict__ params, Batch ba,
                                                 uint32_t *__restrict__ item_tmax) {
	const VsPlan &vp = *vpp;                      // (the plan lives in device memory: indexed by value it was copied into LDS by every workgroup)
	__shared__ uint32_t s_max;
	const uint32_t item = blockIdx.x;
	bool empty = dv.item_cnt[item] == 0u;
	if (FO) empty = empty && dv.sub_cnt[item] == 0u && dv.sub_cnt[vp.n_items + item] == 0u && dv.sub_cnt[2u * vp.n_items + item] == 0u;
	if (empty) { if (threadIdx.x == 0) item_tmax[item] = 0u; return; }
	if (threadIdx.x == 0) s_max = 0u;
	__syncthreads();
	const VsItemRec ir = dv.irec[item];
	if (vp.d[ir.qd] == 3u) { if (threadIdx.x == 0) item_tmax[item] = 0x3F800000u; return; }       // Dense: an update is gradient x weight, no table value
	const Lvl L = load_level(md, ir.lv);
	const VmGeom gm = vm_geom(L.res, (int)vp.d[ir.qd]);
	const auto grid = make_tab(params + (ir.boff + L.off));
	const uint32_t foff = meta_cnt_of(md, vp.q[ir.qd]) * 2u;
	const uint32_t band_lo = gm.plane_lo + ir.row0 * gm.Rb, n_plane = (ir.nrows + 1u) * gm.Rb;
	const bool vec = L.F == 2u && (tab_addr(grid) & (uintptr_t)(4u * tab_elt(grid) - 1u)) == 0u;
	uint32_t tb = 0u;
	if (vec) {
		// the band is one contiguous piece of the table, the line another: raw dwords, four per load (half: |.| of both halves of a
		// dword, compared as half bits, widened at the end)
		struct __attribute__((aligned(4))) V4 { uint32_t w[4]; };
		constexpr uint32_t dpe = sizeof(PT) == 2 ? 1u : 2u;                          // dwords per 2-feature entry
		auto take = [&](uint32_t wv) {
			if (sizeof(PT) == 2) tb = max(tb, max(wv & 0x7FFFu, (wv >> 16) & 0x7FFFu));
			else tb = max(tb, wv & 0x7FFFFFFFu);
		};
		auto scan = [&](uint32_t e_lo, uint32_t cnt) {
			const uint32_t *p = reinterpret_cast<const uint32_t *>(tab_addr(grid)) + (size_t)e_lo * dpe;
			const uint32_t nd = cnt * dpe, n4 = nd >> 2;
			for (uint32_t t = threadIdx.x; t < n4; t += 256u) {
				const V4 v = *reinterpret_cast<const V4 *>(p + 4u * (size_t)t);
				take(v.w[0]); take(v.w[1]); take(v.w[2]); take(v.w[3]);
			}
			for (uint32_t t = 4u * n4 + threadIdx.x; t < nd; t += 256u) take(p[t]);
		};
		scan(band_lo, n_plane);
		scan(gm.line_lo, gm.Rd);
		if (sizeof(PT) == 2) tb = __float_as_uint(__half2float(__ushort_as_half((unsigned short)tb)));
	} else {
		for (uint32_t t = threadIdx.x; t < 2u * (n_plane + gm.Rd); t += 256u) {
			const uint32_t sl = t >> 1, e = sl < n_plane ? band_lo + sl : gm.line_lo + (sl - n_plane);
			tb = max(tb, __float_as_uint((float)grid[(size_t)e * L.F + foff + (t & 1u)]) & 0x7FFFFFFFu);
		}
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) tb = max(tb, (uint32_t)__shfl_xor((int)tb, off, 64));
	if ((threadIdx.x & 63u) == 0u) atomicMax(&s_max, tb);
	__syncthreads();
	if (threadIdx.x == 0) item_tmax[item] = s_max;
}

template <bool FO>
__global__ __launch_bounds__(256) void k_vs_units(const VsPlan *__restrict__ vpp, VsDev dv, const nr3d_lotd_meta_t *__restrict__ md, Batch ba, ForestDev fo,
                                                  const uint32_t *__restrict__ item_tmax, VsUnit *__restrict__ units) {
	const VsPlan &vp = *vpp;                      // (the plan lives in device memory: indexed by value it was copied into LDS by every workgroup)
	const uint32_t w = blockIdx.x * 256u + threadIdx.x;
	if (w >= vp.w_max) return;
	VsUnit u = {};
	u.item = kVsNoUnit;
	if (w < dv.rep_base[vp.n_items]) {
		uint32_t lo = 0u, hi = vp.n_items;                                          // the last item whose first unit is <= w
		while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (dv.rep_base[mid] <= w) lo = mid; else hi = mid; }
		u.item = lo; u.rep = w - dv.rep_base[lo]; u.n_rep = dv.rep_base[lo + 1] - dv.rep_base[lo]; u.multi = dv.multi_base[lo];
		u.own_lo = dv.item_lo[lo]; u.own_cnt = dv.item_cnt[lo];
		if (u.rep == 0u && u.n_rep > 1u) dv.multi_list[atomicAdd(dv.stats + 2, 1u)] = lo;      // (any order: k_vs_reduce_items takes them one by one)
		if (FO)
#pragma unroll
			for (uint32_t rg = 0; rg < 3u; ++rg) { u.sub_lo[rg] = dv.sub_lo[rg * vp.n_items + lo]; u.sub_cnt[rg] = dv.sub_cnt[rg * vp.n_items + lo]; }
		const VsItemRec ir = dv.irec[lo];
		u.qd = ir.qd; u.lv = ir.lv; u.b = ir.b; u.row0 = ir.row0; u.nrows = ir.nrows; u.last = ir.last; u.boff = ir.boff;
		u.tmax = item_tmax ? item_tmax[lo] : 0u;
		const Lvl L = load_level(md, ir.lv);
		const uint32_t q = vp.q[ir.qd];
		u.res[0] = L.res[0]; u.res[1] = L.res[1]; u.res[2] = L.res[2]; u.F = L.F; u.off = L.off;
		u.dc = vp.d[ir.qd]; u.foff = meta_cnt_of(md, q) * 2u; u.col0 = meta_col_of(md, q);
		u.thr = FO ? vp.thr[ir.qd] : 0.0f;
		if (FO)
#pragma unroll
			for (int d = 0; d < 3; ++d) u.kb[d] = fo.block_ks[3 * (size_t)ir.b + d];
	}
	units[w] = u;
}

// forest: what a band needs of boundary candidate j of the list ordered by x_o, next to its point xm[j] -- its row of dL_dy (the
// point index), its block and that block's position -- so that the band's walk does not chase permb -> batch index -> block_ks
struct __attribute__((aligned(16))) VsCand { uint32_t id, blk; int32_t kxy, kz; };
__global__ __launch_bounds__(256) void k_vs_cands(const uint32_t *__restrict__ n_dev, const uint32_t *__restrict__ perm, const float *__restrict__ x,
                                                  Batch ba, ForestDev fo, float *__restrict__ xm, VsCand *__restrict__ cinfo) {
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= *n_dev) return;
	const uint32_t id = perm[j];
#pragma unroll
	for (int d = 0; d < 3; ++d) xm[(size_t)j * 3 + d] = x[(size_t)id * 3 + d];
	VsCand c;
	c.id = id; c.blk = kVsNoUnit; c.kxy = 0; c.kz = 0;
	uint32_t pb, bi;
	if (batch_base_index(ba, id, pb, bi)) {
		c.blk = bi;
		c.kxy = (int32_t)(((uint32_t)(uint16_t)fo.block_ks[3 * (size_t)bi]) | ((uint32_t)(uint16_t)fo.block_ks[3 * (size_t)bi + 1] << 16));
		c.kz = (int32_t)fo.block_ks[3 * (size_t)bi + 2];
	}
	cinfo[j] = c;
}

// what the per-point bodies of a work unit share
template <typename TB>
struct VsCtx {
	Lvl L; VmGeom gm;
	uint32_t dc, smooth, n, row0, nrows, band_lo, n_plane, foff, col0, b;
	bool last;
	TB grid;
	double *acc;
	double scale;
};

// one of the band's own points, its coordinates / gradient pair / dL_ddLdx already in registers (the loop loads them one round ahead).
// VEC: the level's entries are one 2-feature pseudo level and the table is aligned to two entries -- three loads of two whole entries
// per point (emit_vm_component_f2) instead of twelve scalar ones
template <bool SECOND, bool FO, bool FIX, bool VEC, typename TB>
__device__ __forceinline__ bool vs_own_point(const VsCtx<TB> &cx, const float (&xp)[3], const float (&grad)[2], const float (&vv)[3], uint32_t dbg) {
	const Lvl &L = cx.L;
	float a[3];
	Cell<3> c;
	if constexpr (FO) {
		locate_forest(xp, L, cx.smooth != 0, c);
		if (!(c.g[0] - 1u < L.res[0] - 1u && c.g[1] - 1u < L.res[1] - 1u && c.g[2] - 1u < L.res[2] - 1u)) return false;   // boundary cell: the candidate lists
#pragma unroll
		for (int d = 0; d < 3; ++d) c.g[d] -= 1u;
	} else {
		locate<3>(xp, L, cx.smooth != 0, c);
		if (!(c.g[0] + 1u < L.res[0] && c.g[1] + 1u < L.res[1] && c.g[2] + 1u < L.res[2])) return false;                  // outside the level
	}
	const uint32_t ca = cx.gm.a == 0 ? c.g[0] : c.g[1];
	if (ca - cx.row0 >= cx.nrows) return false;
#pragma unroll
	for (int d = 0; d < 3; ++d) a[d] = SECOND ? c.sc[d] * vv[d] * c.dw[d] : 0.0f;
	if (cx.dc == 3u) {
		// a Dense level: the cell's eight corners, gradient x corner weight (second order: the combined d/dx weight of corner_scatter)
#pragma unroll
		for (uint32_t k = 0; k < 8u; ++k) {
			uint32_t p[3];
			corner_pos<3>(c, k, p);
			float wk;
			if (!SECOND) wk = corner_weight<3>(c, k);
			else {
				wk = 0.0f;
#pragma unroll
				for (int d = 0; d < 3; ++d) { const float t = face_weight<3>(c, k, d, a[d]); wk += ((k >> d) & 1u) ? t : -t; }
			}
			const uint32_t slot = (p[0] * L.res[1] + p[1]) * L.res[2] + p[2] - cx.band_lo;
			if (!(dbg & 4u)) {
				vs_add<FIX>(cx.acc, slot, grad[0] * wk, cx.scale);
				vs_add<FIX>(cx.acc, kVsAcc + slot, grad[1] * wk, cx.scale);
			}
		}
		return true;
	}
	uint32_t ent[6];
	float val[6][2];
	if constexpr (VEC) {
		if (cx.dc == 0u) emit_vm_component_f2<0, 6, SECOND>(L, c, a, grad, cx.grid, ent, val);                  // block-uniform
		else if (cx.dc == 1u) emit_vm_component_f2<1, 6, SECOND>(L, c, a, grad, cx.grid, ent, val);
		else emit_vm_component_f2<2, 6, SECOND>(L, c, a, grad, cx.grid, ent, val);
	} else {
		if (cx.dc == 0u) emit_vm_component<2, 0, 6, SECOND>(L, c, a, grad, cx.grid, cx.foff, ent, val);
		else if (cx.dc == 1u) emit_vm_component<2, 1, 6, SECOND>(L, c, a, grad, cx.grid, cx.foff, ent, val);
		else emit_vm_component<2, 2, 6, SECOND>(L, c, a, grad, cx.grid, cx.foff, ent, val);
	}
	if (dbg & 4u) return val[0][0] + val[1][1] + val[2][0] + val[3][1] + val[4][0] + val[5][1] == 1.2345f;
#pragma unroll
	for (int k = 0; k < 6; ++k) {
		const uint32_t slot = k < 4 ? ent[k] - cx.band_lo : cx.n_plane + (ent[k] - cx.gm.line_lo);
		vs_add<FIX>(cx.acc, slot, val[k][0], cx.scale);
		vs_add<FIX>(cx.acc, kVsAcc + slot, val[k][1], cx.scale);
	}
	return true;
}

// forest: the boundary candidates of the list ordered by x_a -- the corners this band's block owns of a boundary cell of ANY block.
// vs_cand_test: the cheap part, one lane per candidate -- is its cell a boundary cell of this level whose corners can reach the band's rows,
// in a block next to (or the same as) the band's?  vs_cand_cell / vs_corner: the work behind a survivor -- its cell, its gradient pair, then
// per corner the owner, the table values and four adds.  (Round 5 looked every corner's owner up in a 27-entry table per block; the owner of a
// corner that leaves the candidate's block on side dl (0: below, 1: inside, 2: above, per dimension) is the block at the candidate's position
// + dl - 1 (resolve_block; continuity off: nobody) -- this band's block exactly when dl = kb - k2 + 1 in every dimension.)
__device__ __forceinline__ void vs_cand_pos(const VsCand &ci, int (&k2)[3]) {
	k2[0] = (int)(int16_t)(ci.kxy & 0xFFFF); k2[1] = (int)(int16_t)((uint32_t)ci.kxy >> 16); k2[2] = (int)ci.kz;
}
template <typename TB>
__device__ __forceinline__ bool vs_cand_test(const VsCtx<TB> &cx, const float (&xp)[3], const VsCand &ci, float thr, const int (&kb)[3]) {
	const Lvl &L = cx.L;
	const VmGeom &gm = cx.gm;
	if (fminf(fminf(fminf(xp[0], 1.0f - xp[0]), fminf(xp[1], 1.0f - xp[1])), fminf(xp[2], 1.0f - xp[2])) >= thr) return false;   // interior at this level
	Cell<3> c;
	locate_forest(xp, L, cx.smooth != 0, c);
	if (!(c.g[0] <= L.res[0] && c.g[1] <= L.res[1] && c.g[2] <= L.res[2])) return false;              // outside the block
	if (c.g[0] - 1u < L.res[0] - 1u && c.g[1] - 1u < L.res[1] - 1u && c.g[2] - 1u < L.res[2] - 1u) return false;      // interior: an own point of its band
	// the plane rows the cell's corners land on, in the index space of WHOEVER owns them (0 -> the left neighbour's last row, R + 1 -> the
	// right neighbour's row 0): a band that holds neither has nothing to take from this point -- at 400 bands per plane all but one or two
	const uint32_t ga = c.g[gm.a];
	const uint32_t r0 = (ga == 0u ? gm.Ra : ga) - 1u, r1 = ga == gm.Ra ? 0u : ga;
	const bool in0 = (r0 - cx.row0 < cx.nrows) || (cx.last && r0 - cx.row0 == cx.nrows), in1 = (r1 - cx.row0 < cx.nrows) || (cx.last && r1 - cx.row0 == cx.nrows);
	if (!in0 && !in1) return false;
	if (ci.blk == kVsNoUnit) return false;
	int k2[3];
	vs_cand_pos(ci, k2);
	bool near = true;
#pragma unroll
	for (int d = 0; d < 3; ++d) near = near && (k2[d] - kb[d] + 1) >= 0 && (k2[d] - kb[d] + 1) <= 2;
	return near;
}
template <bool SECOND, typename TB>
__device__ __forceinline__ void vs_cand_cell(const VsCtx<TB> &cx, const float (&xp)[3], const VsCand &ci, const float *__restrict__ g, int64_t g_sn,
                                             int64_t g_se, const float *__restrict__ vin_, Cell<3> &c, float (&grad)[2], float (&a)[3]) {
	locate_forest(xp, cx.L, cx.smooth != 0, c);
#pragma unroll
	for (int f = 0; f < 2; ++f) grad[f] = g[(int64_t)ci.id * g_sn + (int64_t)(cx.col0 + f) * g_se];
#pragma unroll
	for (int d = 0; d < 3; ++d) a[d] = SECOND ? c.sc[d] * vin_[(size_t)ci.id * 3 + d] * c.dw[d] : 0.0f;
}
template <bool SECOND, bool FIX, typename TB>
__device__ __forceinline__ bool vs_corner(const VsCtx<TB> &cx, const Cell<3> &c, const float (&grad)[2], const float (&a)[3], const int (&k2)[3],
                                          const int (&kb)[3], bool foreign_ok, uint32_t k) {
	const Lvl &L = cx.L;
	const VmGeom &gm = cx.gm;
	const uint32_t dc = cx.dc;
	const int b_dim = dc == 2u ? 1 : 2;
	uint32_t p[3], pl[3];
	bool mine = foreign_ok;
	corner_pos<3>(c, k, p);
#pragma unroll
	for (int d = 0; d < 3; ++d) {                                                          // resolve_block
		const uint32_t dl = p[d] == 0u ? 0u : (p[d] == L.res[d] + 1u ? 2u : 1u);
		pl[d] = dl == 0u ? L.res[d] - 1u : (dl == 2u ? 0u : p[d] - 1u);
		mine = mine && (int)dl == kb[d] - k2[d] + 1;
	}
	const uint32_t lr = pl[gm.a] - cx.row0;
	if (!(mine && (lr < cx.nrows || (cx.last && lr == cx.nrows)))) return false;              // another block's, or another band's row
	float wk;
	if (dc == 3u) {                                                                          // a Dense level: one entry, no table factor
		if (!SECOND) wk = corner_weight<3>(c, k);
		else {
			wk = 0.0f;
#pragma unroll
			for (int d = 0; d < 3; ++d) { const float t = face_weight<3>(c, k, d, a[d]); wk += ((k >> d) & 1u) ? t : -t; }
		}
		const uint32_t slot = (pl[0] * L.res[1] + pl[1]) * L.res[2] + pl[2] - cx.band_lo;
		vs_add<FIX>(cx.acc, slot, grad[0] * wk, cx.scale);
		vs_add<FIX>(cx.acc, kVsAcc + slot, grad[1] * wk, cx.scale);
		return true;
	}
	const uint32_t pe = gm.plane_lo + pl[gm.a] * gm.Rb + pl[b_dim], le = gm.line_lo + pl[dc];
	const uint32_t ps = pe - cx.band_lo, ls = cx.n_plane + pl[dc];
	if (!SECOND) wk = corner_weight<3>(c, k);
	else {
		wk = 0.0f;
#pragma unroll
		for (int d = 0; d < 3; ++d) { const float t = face_weight<3>(c, k, d, a[d]); wk += ((k >> d) & 1u) ? t : -t; }
	}
	float tl[2], tp[2];
#pragma unroll
	for (int f = 0; f < 2; ++f) { tl[f] = (float)cx.grid[le * L.F + cx.foff + f]; tp[f] = (float)cx.grid[pe * L.F + cx.foff + f]; }
#pragma unroll
	for (int f = 0; f < 2; ++f) {
		const float wg = grad[f] * wk;
		vs_add<FIX>(cx.acc, (uint32_t)f * kVsAcc + ps, wg * tl[f], cx.scale);
		vs_add<FIX>(cx.acc, (uint32_t)f * kVsAcc + ls, wg * tp[f], cx.scale);
	}
	return true;
}

template <bool SECOND, typename PT, bool FO>
__global__ __launch_bounds__(kVsThreads) void k_vm_sorted(const VsPlan *__restrict__ vpp, uint32_t rb_max, uint32_t rd_max, VsDev dv, uint32_t n, uint32_t smooth,
                                                          const float *__restrict__ x, const float *__restrict__ vin_,
                                                          const float *__restrict__ g, int64_t g_sn, int64_t g_se,
                                                          const PT *__restrict__ params, Batch ba, ForestDev fo, float *__restrict__ dparam,
                                                          uint32_t opt_fix NR3D_DBG_PARAM) {
	NR3D_DBG_DECL   // timing experiments (experiments build only; results wrong by design): 1 no boundary pass, 2 no write-out, 4 no LDS adds, 8 no own points
	extern __shared__ __attribute__((aligned(16))) double vs_acc[];            // [2 features][kVsAcc slots]: plane band (nrows + 1) Rb | line d Rd
	__shared__ uint8_t s_rank[kVsThreads];                                       // per wave: the lanes of its surviving candidates, by rank
	const uint32_t w = blockIdx.x;
	const VsUnit un = dv.units[w];
	if (un.item == kVsNoUnit) return;
	if (dbg & 64u) return;
	const uint32_t gmax_bits = dv.stats[0], vmax_bits = SECOND ? dv.stats[1] : 0u;
	const uint32_t item = un.item, rep = un.rep, n_rep = un.n_rep;
	Lvl L;                                                                         // (everything the unit needs came with its record)
	L.res[0] = un.res[0]; L.res[1] = un.res[1]; L.res[2] = un.res[2]; L.res[3] = 0u; L.F = un.F; L.type = NR3D_LOD_VectorMatrix; L.size = 0u; L.off = un.off;
	// (the component from the plan, not from the record's copy of it: with `un.dc` here the d = 2 line sums of the second feature came out
	// wrong on multi-feature levels -- same value, other code; bisected on the GPU, not understood)
	const uint32_t dc = vpp->d[un.qd];
	const VmGeom gm = vm_geom(L.res, (int)dc);
	const uint32_t row0 = un.row0, nrows = un.nrows;
	const bool last = un.last != 0u;
	const uint32_t n_plane = (nrows + 1u) * gm.Rb, n_slots = n_plane + gm.Rd;
	// a band nobody touches (no own points, no boundary candidates: most bands of a forest whose points sit in a few blocks): zeros to
	// the two small buffers k_vs_reduce reads, no LDS table, no write-out
	const uint32_t own_cnt = un.own_cnt;
	const uint32_t bnd_cnt = FO ? un.sub_cnt[0] + un.sub_cnt[1] + un.sub_cnt[2] : 0u;
	if (own_cnt == 0u && bnd_cnt == 0u) {                                          // (then n_rep == 1)
		if (!last) {
			float *ho = dv.handoff + (size_t)item * 2u * rb_max;
			for (uint32_t t = threadIdx.x; t < 2u * gm.Rb; t += kVsThreads) ho[t] = 0.0f;
		}
		float *lz = dv.lines + (size_t)w * 2u * rd_max;
		for (uint32_t t = threadIdx.x; t < 2u * gm.Rd; t += kVsThreads) lz[t] = 0.0f;
		return;
	}
	const uint32_t foff = un.foff, col0 = un.col0;
	const uint32_t boff = un.boff;
	const auto grid = make_tab(params + (boff + L.off));
	const uint32_t band_lo = gm.plane_lo + row0 * gm.Rb;
	const bool vec = L.F == 2u && (tab_addr(grid) & (uintptr_t)(4u * tab_elt(grid) - 1u)) == 0u;
	// ---- the fixed-point scale from the bound on the workgroup's updates
	VsFix fx;
	fx.on = false; fx.scale = 0.0; fx.inv = 0.0;
	if (opt_fix) {
		double B = (double)__uint_as_float(gmax_bits) * (double)__uint_as_float(un.tmax);
		if (SECOND) B *= 12.0 * (double)(max(L.res[0], max(L.res[1], L.res[2])) + 2u) * (double)__uint_as_float(vmax_bits);
		const unsigned long long bb = (unsigned long long)__double_as_longlong(B);
		const int ex = (int)(bb >> 52);                                        // B >= 0: the biased exponent; B < 2^(ex - 1022)
		if (ex > 1023 - 200 && ex < 1023 + 200) {                                  // not zero / denormal / huge / inf / NaN
			const uint64_t cnt = 2ull * min(own_cnt, kVsPmax) + 8ull * bnd_cnt + 1ull;
			int lg = 0;
			while ((1ull << lg) < cnt) ++lg;
			const int sc = min(62 - lg, 50) - (ex - 1022);
			fx.scale = __longlong_as_double((long long)(sc + 1023) << 52);
			fx.inv = __longlong_as_double((long long)(1023 - sc) << 52);
			fx.on = true;
		}
	}
	// ---- ONE loop: first the item's three ranges of boundary candidates (its replicas share them), then -- from the next multiple of 64:
	// whole waves -- the band's own points, a contiguous range of the order of x_a.  The candidates' two rounds of dependent loads run in
	// their waves while the others accumulate points (one pass after the other, each paid its latency per work unit); a wave's next own
	// point is loaded while it works on the current one.
	VsCtx<std::remove_const_t<decltype(grid)>> cx;
	cx.L = L; cx.gm = gm; cx.dc = dc; cx.smooth = smooth; cx.n = n; cx.row0 = row0; cx.nrows = nrows; cx.band_lo = band_lo; cx.n_plane = n_plane;
	cx.foff = foff; cx.col0 = col0; cx.b = un.b; cx.last = last; cx.grid = grid; cx.acc = vs_acc; cx.scale = fx.scale;
	const int o = dc == 0u ? 1 : 0;
	const uint32_t p_lo = un.own_lo + rep * kVsPmax, p_end = un.own_lo + own_cnt;
	const uint32_t n_own = (dbg & 8u) ? 0u : ((p_end - p_lo) < kVsPmax ? p_end - p_lo : kVsPmax);
	uint32_t c_lo[3] = {0u, 0u, 0u}, c_n[3] = {0u, 0u, 0u};
	int kb[3] = {0, 0, 0};
	if (FO && !(dbg & 1u)) {
#pragma unroll
		for (uint32_t rg = 0; rg < 3u; ++rg) {
			const uint32_t r_cnt = un.sub_cnt[rg], per = (r_cnt + n_rep - 1u) / n_rep;
			const uint32_t lo = rep * per < r_cnt ? rep * per : r_cnt, hi = (rep + 1u) * per < r_cnt ? (rep + 1u) * per : r_cnt;
			c_lo[rg] = un.sub_lo[rg] + lo; c_n[rg] = hi - lo;
		}
#pragma unroll
		for (int d = 0; d < 3; ++d) kb[d] = un.kb[d];
	}
	const uint32_t n_cnd = c_n[0] + c_n[1] + c_n[2], cnd_pad = (n_cnd + 63u) & ~63u;
	const uint32_t total = cnd_pad + n_own;
	const float thr = un.thr;
	const float *__restrict__ xs = dv.xs[o], *__restrict__ vs = dv.vs[o], *__restrict__ gts = dv.gts[o];
	float nx[3] = {0.0f, 0.0f, 0.0f}, ng[2] = {0.0f, 0.0f}, nv[3] = {0.0f, 0.0f, 0.0f};
	// (no branch around the loads: a round that has no next point re-reads a valid one -- behind a branch the compiler waits for the
	// loads at the join, in front of the current point's work, and the look-ahead is gone)
	const uint32_t own_last = n_own ? n_own - 1u : 0u;
	auto fetch = [&](uint32_t idx) {
		const uint32_t k = idx >= cnd_pad ? idx - cnd_pad : 0u;
		const uint32_t i = min(p_lo + min(k, own_last), n - 1u);
#pragma unroll
		for (int d = 0; d < 3; ++d) nx[d] = xs[(size_t)i * 3 + d];
#pragma unroll
		for (int f = 0; f < 2; ++f) ng[f] = gts[(size_t)(col0 + f) * n + i];
		if (SECOND)
#pragma unroll
			for (int d = 0; d < 3; ++d) nv[d] = vs[(size_t)i * 3 + d];
	};
	fetch(threadIdx.x);
	// the band's own rows of dL/dparam as they are now (the write-out adds to them): read here, a round trip before they are needed
	constexpr uint32_t kPre = (kVsSlot + 4u * kVsThreads - 1u) / (4u * kVsThreads);
	const uint32_t own = 2u * (last ? nrows + 1u : nrows) * gm.Rb;
	float *dst = dparam + (boff + L.off) + (size_t)band_lo * L.F + foff;
	float old[kPre][4];
	if (n_rep == 1u && !(dbg & 2u)) {
#pragma unroll
		for (uint32_t k = 0; k < kPre; ++k)
#pragma unroll
			for (uint32_t u = 0; u < 4u; ++u) {
				const uint32_t t = threadIdx.x + (k * 4u + u) * kVsThreads;
				old[k][u] = t < own ? dst[(size_t)(t >> 1) * L.F + (t & 1u)] : 0.0f;
			}
	}
	for (uint32_t t = threadIdx.x; t < n_slots; t += kVsThreads) { vs_acc[t] = 0.0; vs_acc[kVsAcc + t] = 0.0; }      // all-zero bits: 0.0 and fixed-point 0 alike
	__syncthreads();
	bool any = false;
	auto run = [&](auto fix_c, auto vec_c) {
		constexpr bool FIX = decltype(fix_c)::value, VEC = decltype(vec_c)::value;
		for (uint32_t idx = threadIdx.x; idx < total; idx += kVsThreads) {
			float cxp[3], cg[2], cv[3];
#pragma unroll
			for (int d = 0; d < 3; ++d) { cxp[d] = nx[d]; cv[d] = nv[d]; }
			cg[0] = ng[0]; cg[1] = ng[1];
			fetch(idx + kVsThreads);
			if (idx >= cnd_pad) any |= vs_own_point<SECOND, FO, FIX, VEC>(cx, cxp, cg, cv, dbg);
			else if constexpr (FO) {
				// (cnd_pad is a multiple of 64: a wave is on candidates with all its lanes)
				float bx[3] = {0.5f, 0.5f, 0.5f};
				VsCand ci;
				ci.id = 0u; ci.blk = kVsNoUnit; ci.kxy = 0; ci.kz = 0;
				bool surv = false;
				if (idx < n_cnd) {
					const uint32_t rg = idx < c_n[0] ? 0u : (idx < c_n[0] + c_n[1] ? 1u : 2u);
					const uint32_t j = c_lo[rg] + (idx - (rg > 0u ? c_n[0] : 0u) - (rg > 1u ? c_n[1] : 0u));
#pragma unroll
					for (int d = 0; d < 3; ++d) bx[d] = dv.xm[o][(size_t)j * 3 + d];
					ci = dv.cinfo[o][j];
					surv = vs_cand_test(cx, bx, ci, thr, kb);
				}
				const uint64_t m = __ballot(surv);
				const uint32_t ns = (uint32_t)__popcll(m);
				if (ns >= kVsDense) {
					// most lanes have work: every survivor walks its own eight corners
					if (surv) {
						Cell<3> c;
						float grad[2], a[3];
						int k2[3];
						vs_cand_pos(ci, k2);
						vs_cand_cell<SECOND>(cx, bx, ci, g, g_sn, g_se, vin_, c, grad, a);
						const bool fok = fo.continuity != 0 || ci.blk == cx.b;
#pragma unroll
						for (uint32_t k = 0; k < 8u; ++k) any |= vs_corner<SECOND, FIX>(cx, c, grad, a, k2, kb, fok, k);
					}
				} else if (ns) {
					// a few survivors (the fine levels: one candidate in twenty is a boundary cell there): their 8 ns corners over the lanes,
					// one corner each -- lane-serial, the whole wave walked eight corners for three busy lanes
					const uint32_t lane = threadIdx.x & 63u;
					uint8_t *rk = s_rank + (threadIdx.x & ~63u);
					if (surv) rk[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint8_t)lane;
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
					const uint32_t items = 8u * ns, rounds = (items + 63u) >> 6;            // (wave-uniform: every lane takes part in the shuffles)
					for (uint32_t r = 0; r < rounds; ++r) {
						const uint32_t t = r * 64u + lane;
						const int src = (int)rk[t < items ? t >> 3 : 0u];
						float sx[3];
						VsCand sc;
#pragma unroll
						for (int d = 0; d < 3; ++d) sx[d] = __shfl(bx[d], src, 64);
						sc.id = (uint32_t)__shfl((int)ci.id, src, 64); sc.blk = (uint32_t)__shfl((int)ci.blk, src, 64);
						sc.kxy = __shfl(ci.kxy, src, 64); sc.kz = __shfl(ci.kz, src, 64);
						if (t < items) {
							Cell<3> c;
							float grad[2], a[3];
							int k2[3];
							vs_cand_pos(sc, k2);
							vs_cand_cell<SECOND>(cx, sx, sc, g, g_sn, g_se, vin_, c, grad, a);
							any |= vs_corner<SECOND, FIX>(cx, c, grad, a, k2, kb, fo.continuity != 0 || sc.blk == cx.b, t & 7u);
						}
					}
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				}
			}
		}
	};
	if (fx.on) { if (vec) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}); }
	else { if (vec) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}); }
	const int touched = __syncthreads_or(any ? 1 : 0);
	if (dbg & 2u) return;
	// ---- write-out.  Rows 0 .. nrows - 1 of the band belong to this item alone (the last band: nrows as well); row nrows is the
	// next band's row 0 and travels through `handoff`; line d through `lines` (k_vs_reduce adds both, and the replicas, in order).
	// Element t of a table = (slot t >> 1, feature t & 1), the layout of dL/dparam and of the three buffers.
	auto elem = [&](uint32_t t) { return vs_out(vs_acc, (t & 1u) * kVsAcc + (t >> 1), fx); };
	if (n_rep == 1u) {
		if (touched) {
#pragma unroll
			for (uint32_t k = 0; k < kPre; ++k)
#pragma unroll
				for (uint32_t u = 0; u < 4u; ++u) {
					const uint32_t t = threadIdx.x + (k * 4u + u) * kVsThreads;
					if (t < own) {
						const float v = elem(t);
						if (v != 0.0f) dst[(size_t)(t >> 1) * L.F + (t & 1u)] = old[k][u] + v;
					}
				}
		}
		if (!last) {
			float *ho = dv.handoff + (size_t)item * 2u * rb_max;
			for (uint32_t t = threadIdx.x; t < 2u * gm.Rb; t += kVsThreads) ho[t] = elem(own + t);
		}
	} else {
		float *sl = dv.slots + (size_t)(un.multi + rep) * kVsSlot;
		for (uint32_t t = threadIdx.x; t < 2u * n_plane; t += kVsThreads) sl[t] = elem(t);
	}
	float *lo_ = dv.lines + (size_t)w * 2u * rd_max;
	for (uint32_t t = threadIdx.x; t < 2u * gm.Rd; t += kVsThreads) lo_[t] = elem(2u * n_plane + t);
}

// blocks 0 .. n_items - 1: an item's replicas (if it has any) and the row its predecessor hands over; the others: line d of one
// (pseudo level, component, block) over all its work units.  Every element of dL/dparam has one writer here, sums in a fixed order.
__device__ __forceinline__ uint32_t div_up_dev(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }
__global__ __launch_bounds__(256) void k_vs_reduce_items(const VsPlan *__restrict__ vpp, VsDev dv, const nr3d_lotd_meta_t *__restrict__ md, Batch ba, float *__restrict__ dparam) {
	const VsPlan &vp = *vpp;                      // (the plan lives in device memory: indexed by value it was copied into LDS by every workgroup)
	{
		// (item, chunk of 4096 table elements): eight elements per thread, the replicas' slots added in replica order with eight loads in
		// flight (one 1024-thread workgroup per item and one load at a time, the replica bands of the coarse levels were 400 dependent
		// loads per thread); 256 threads: most of the 53 000 workgroups have nothing to do and should cost four waves, not sixteen
		// blocks 0 .. n_items - 1: the items WITHOUT replicas (the row handed over, all of it); behind them (entry j of the list of items with
		// replicas that k_vs_units wrote, chunk): launched for the most such items a pass can have, all but a few find j beyond the list
		uint32_t item, chunk = 0u;
		const bool single = blockIdx.x < vp.n_items;
		if (single) item = blockIdx.x;
		else {
			const uint32_t j = (blockIdx.x - vp.n_items) / kVsRedChunks;
			chunk = (blockIdx.x - vp.n_items) - j * kVsRedChunks;
			if (j >= dv.stats[2]) return;
			item = dv.multi_list[j];
		}
		const uint32_t n_rep = dv.rep_base[item + 1] - dv.rep_base[item];
		if (single && n_rep > 1u) return;
		const VsItemRec it = dv.irec[item];
		if (n_rep == 1u && it.band == 0u) return;
		const uint32_t qd0 = it.qd;
		const Lvl L = load_level(md, it.lv);
		const VmGeom gm = vm_geom(L.res, (int)vp.d[qd0]);
		const uint32_t own = n_rep > 1u ? 2u * (it.last ? it.nrows + 1u : it.nrows) * gm.Rb : 0u;
		const uint32_t head = it.band > 0u ? 2u * gm.Rb : 0u;                         // the previous band's last row is this band's row 0
		const uint32_t t_end = own > head ? own : head;
		float *dst = dparam + (it.boff + L.off) + (size_t)(gm.plane_lo + it.row0 * gm.Rb) * L.F + meta_cnt_of(md, vp.q[qd0]) * 2u;
		const float *sl = dv.slots + (size_t)dv.multi_base[item] * kVsSlot;
		const uint32_t prev = item - (it.band > 0u ? 1u : 0u), p_rep = dv.rep_base[prev + 1] - dv.rep_base[prev], p_rows = vp.rows[qd0];
		const float *ps = dv.slots + (size_t)dv.multi_base[prev] * kVsSlot + 2u * (size_t)p_rows * gm.Rb;
		const float *ho = dv.handoff + (size_t)prev * 2u * vp.rb_max;
		constexpr uint32_t kE = kVsRedPiece / 256u;                                              // elements per thread
		for (; chunk * kVsRedPiece < t_end; chunk += single ? 1u : kVsRedChunks) {           // (an item without replicas: <= 2 rounds, the handed-over row)
		uint32_t t[kE];
		float s[kE] = {};
#pragma unroll
		for (uint32_t u = 0; u < kE; ++u) t[u] = chunk * kVsRedPiece + u * 256u + threadIdx.x;
		auto sum_slots = [&](const float *base, uint32_t reps, uint32_t limit, float (&acc)[kE]) {
			for (uint32_t r = 0; r < reps; ++r) {
				float v[kE];
#pragma unroll
				for (uint32_t u = 0; u < kE; ++u) v[u] = t[u] < limit ? base[(size_t)r * kVsSlot + t[u]] : 0.0f;
#pragma unroll
				for (uint32_t u = 0; u < kE; ++u) acc[u] += v[u];
			}
		};
		if (own) sum_slots(sl, n_rep, own, s);
		if (head && chunk * kVsRedPiece < head) {
			float h[kE] = {};
			if (p_rep > 1u) sum_slots(ps, p_rep, head, h);
			else
#pragma unroll
				for (uint32_t u = 0; u < kE; ++u) h[u] = t[u] < head ? ho[t[u]] : 0.0f;
#pragma unroll
			for (uint32_t u = 0; u < kE; ++u) {
				// (two separate adds into dL/dparam, as before: the replicas' sum, then the row handed over)
				if (t[u] < t_end && s[u] != 0.0f) { dst[(size_t)(t[u] >> 1) * L.F + (t[u] & 1u)] += s[u]; s[u] = 0.0f; }
				if (t[u] < head && h[u] != 0.0f) dst[(size_t)(t[u] >> 1) * L.F + (t[u] & 1u)] += h[u];
			}
		}
#pragma unroll
		for (uint32_t u = 0; u < kE; ++u) if (t[u] < t_end && s[u] != 0.0f) dst[(size_t)(t[u] >> 1) * L.F + (t[u] & 1u)] += s[u];
		}
	}
}

__global__ __launch_bounds__(1024) void k_vs_reduce_lines(const VsPlan *__restrict__ vpp, VsDev dv, const nr3d_lotd_meta_t *__restrict__ md, Batch ba, float *__restrict__ dparam) {
	const VsPlan &vp = *vpp;                      // (the plan lives in device memory: indexed by value it was copied into LDS by every workgroup)
	// line d of (pair qd, block b), 64 elements per workgroup: wave k sums its share of the work units, then the waves in order
	__shared__ float part[16][64];
	const uint32_t n_chunks = div_up_dev(2u * vp.rd_max, 64u);
	const uint32_t lin = blockIdx.x, gidx = lin / n_chunks, chunk = lin - gidx * n_chunks;
	const uint32_t qd = gidx / vp.n_blocks, b = gidx - qd * vp.n_blocks;
	const Lvl L = load_level(md, meta_level_of(md, vp.q[qd]));
	const VmGeom gm = vm_geom(L.res, (int)vp.d[qd]);
	if (chunk * 64u >= 2u * gm.Rd) return;
	const uint32_t first = vp.item_base[qd] + b * vp.n_bands[qd];
	const uint32_t w0 = dv.rep_base[first], w1 = dv.rep_base[first + vp.n_bands[qd]];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
	const uint32_t per = (w1 - w0 + n_waves - 1u) / n_waves;
	const uint32_t t = chunk * 64u + lane;
	const uint32_t a0 = w0 + wave * per, a1 = (a0 + per) < w1 ? (a0 + per) : w1;
	float s = 0.0f;
	if (t < 2u * gm.Rd) {
		// sixteen loads in flight, added in unit order (one at a time this was a chain of ~25 memory latencies per workgroup)
		const float *lp = dv.lines + t;
		const size_t stride = 2u * (size_t)vp.rd_max;
		uint32_t w = a0;
		for (; w + 16u <= a1; w += 16u) {
			float v[16];
#pragma unroll
			for (uint32_t u = 0; u < 16u; ++u) v[u] = lp[(size_t)(w + u) * stride];
#pragma unroll
			for (uint32_t u = 0; u < 16u; ++u) s += v[u];
		}
		for (; w < a1; ++w) s += lp[(size_t)w * stride];
	}
	part[wave][lane] = s;
	__syncthreads();
	if (wave == 0u && t < 2u * gm.Rd) {
		float sum = 0.0f;
		for (uint32_t k = 0; k < n_waves; ++k) sum += part[k][lane];
		const uint32_t boff = ba.offsets ? (uint32_t)ba.offsets[b] : b * ba.n_params;
		if (sum != 0.0f) dparam[(boff + L.off) + (size_t)(gm.line_lo + (t >> 1)) * L.F + meta_cnt_of(md, vp.q[qd]) * 2u + (t & 1u)] += sum;
	}
}

// ---- host: which pseudo levels the sorted path serves (mask; 0: none) and its plan
// NR3D_OPT_VM_SORTED: 0 never; 1 when some VM level of the pass has >= 2^20 entries over its blocks and the pass >= 2^19 points
// (then every VM level of the pass takes it: the sort is paid once); 2 always (tests)
uint64_t vm_sorted_plan(const nr3d_lotd_meta_t *m, uint32_t n, uint32_t n_blocks, bool forest, int32_t min_level, int32_t max_level,
                               uint64_t skip, VsPlan &vp) {
	static_assert(sizeof(VsPlan) <= 2560, "VsPlan travels as a kernel argument");
	vp.n_qd = vp.n_items = 0;
	vp.n_sc[0] = vp.n_sc[1] = 0;
	bool sc_ok = true;
	const int64_t mode_opt = opt::get(NR3D_OPT_VM_SORTED), mode = mode_opt == 3 ? 1 : mode_opt;      // 3: as 1, VM levels only (A/B)
	if (!mode || m->n_dims_to_encode != 3 || m->n_feat_per_pseudo_lvl != 2 || m->n_pseudo_levels > 64u || m->n_encoded_dims > 120u || n == 0) return 0;
	n_blocks = n_blocks ? n_blocks : 1u;
	uint64_t mask = 0, biggest = 0, dense_mask = 0;
	uint32_t items = 0;
	vp.rb_max = vp.rd_max = 0;
	vp.thr_max = 0.0f;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
		const uint32_t lv = m->map_levels[q];
		const nr3d_lotd_level_t &L = m->levels[lv];
		if ((int32_t)lv < min_level || (int32_t)lv > max_level || ((skip >> q) & 1ull)) continue;
		if (L.type == NR3D_LOD_Dense && mode_opt != 3) {
			// small Dense levels ride along (component code 3: rows = x_0 slices, vm_geom; forests and, measured on configs[3], plain metas:
			// two 4-feature Dense levels 1.06 ms of records per pass).  A forest's two: their records cost 0.69 ms on the
			// reference's forest workload (k_bin_forest + k_accum + the feature-major copy of dL_dy for two levels of 34^3 and 55^3),
			// as two more "planes" of the sorted points ~0.2.  Only next to VM levels (dense_mask is dropped when no VM level is served).
			const VmGeom gm = vm_geom(L.res, 3);
			if (vp.n_qd + 1u > kVsMaxQD || gm.Ra < 2u || 2u * gm.Rb > kVsAcc || gm.Ra > 60000u) continue;
			const uint32_t fit = kVsAcc / gm.Rb, cells = gm.Ra - 1u;
			const uint32_t bands = div_up(cells, fit - 1u), rows = div_up(cells, bands);
			const uint32_t k = vp.n_qd++;
			vp.q[k] = (uint16_t)q; vp.d[k] = 3u; vp.rows[k] = (uint16_t)rows; vp.n_bands[k] = (uint16_t)bands;
			vp.item_base[k] = items;
			items += n_blocks * bands;
			vp.rb_max = gm.Rb > vp.rb_max ? gm.Rb : vp.rb_max;
			float thr = 0.0f;
			for (int d = 0; d < 3; ++d) { const float t = 0.5f / (float)L.res[d]; thr = t > thr ? t : thr; }
			vp.thr[k] = thr * 1.001f + 1e-6f;
			vp.thr_max = vp.thr[k] > vp.thr_max ? vp.thr[k] : vp.thr_max;
			const float scale = forest ? (float)gm.Ra : (float)(gm.Ra - 2u);
			bool seen = false;
			for (uint32_t j = 0; j < vp.n_sc[0]; ++j) seen = seen || vp.sc[0][j] == scale;
			if (!seen) {
				if (vp.n_sc[0] < kVsMaxSc) { vp.sc[0][vp.n_sc[0]] = scale; vp.cap[0][vp.n_sc[0]] = (float)(gm.Ra + 2u); ++vp.n_sc[0]; }
				else sc_ok = false;
			}
			dense_mask |= 1ull << q;
			continue;
		}
		if (L.type != NR3D_LOD_VectorMatrix) continue;
		if (vp.n_qd + 3u > kVsMaxQD) break;
		uint32_t rows[3], bands[3];
		bool ok = true;
		for (int d = 0; d < 3 && ok; ++d) {
			const VmGeom gm = vm_geom(L.res, d);
			if (gm.Ra < 2u || gm.Rd + 2u * gm.Rb > kVsAcc || gm.Ra > 60000u) { ok = false; break; }
			const uint32_t fit = (kVsAcc - gm.Rd) / gm.Rb, cells = gm.Ra - 1u;     // entry rows a band can hold; >= 2
			bands[d] = div_up(cells, fit - 1u);
			rows[d] = div_up(cells, bands[d]);
			ok = bands[d] <= 60000u;
		}
		if (!ok) continue;
		float thr = 0.0f;
		for (int d = 0; d < 3; ++d) {
			const VmGeom gm = vm_geom(L.res, d);
			const uint32_t k = vp.n_qd++;
			vp.q[k] = (uint16_t)q; vp.d[k] = (uint8_t)d; vp.rows[k] = (uint16_t)rows[d]; vp.n_bands[k] = (uint16_t)bands[d];
			vp.item_base[k] = items;
			items += n_blocks * bands[d];
			vp.rb_max = gm.Rb > vp.rb_max ? gm.Rb : vp.rb_max;
			vp.rd_max = gm.Rd > vp.rd_max ? gm.Rd : vp.rd_max;
			const float t = 0.5f / (float)L.res[d];
			thr = t > thr ? t : thr;
			// the row scale of this (level, component) along its sort coordinate x_a: k_vs_plan's `sc`
			const float scale = forest ? (float)gm.Ra : (float)(gm.Ra - 2u);
			const int o = gm.a;
			bool seen = false;
			for (uint32_t j = 0; j < vp.n_sc[o]; ++j) seen = seen || vp.sc[o][j] == scale;
			if (!seen) {
				if (vp.n_sc[o] < kVsMaxSc) { vp.sc[o][vp.n_sc[o]] = scale; vp.cap[o][vp.n_sc[o]] = (float)(gm.Ra + 2u); ++vp.n_sc[o]; }
				else sc_ok = false;
			}
		}
		for (uint32_t k = vp.n_qd - 3u; k < vp.n_qd; ++k) vp.thr[k] = thr * 1.001f + 1e-6f;      // a superset; the kernel tests the cell itself
		vp.thr_max = vp.thr[vp.n_qd - 1u] > vp.thr_max ? vp.thr[vp.n_qd - 1u] : vp.thr_max;
		mask |= 1ull << q;
		const uint64_t sz = (uint64_t)L.size * n_blocks;
		biggest = sz > biggest ? sz : biggest;
	}
	vp.item_base[vp.n_qd] = items;
	// (mode 1: big tables from 2^19 points on -- the records cost O(point blocks x buckets) there; small tables from 2^21 points on -- the
	// sorts and gathers are a fixed 0.3 ms, measured against k_vm_direct + records on configs[3])
	if (!mask || (mode == 1 && (n < (1u << 19) || (biggest < (1ull << 20) && n < (1u << 21))))) { vp.n_qd = vp.n_items = 0; return 0; }
	// few points per work item (a forest of many blocks, a short pass): the per-band work -- zeroing and writing a 158-KiB table --
	// would dominate; the records take those
	if (mode == 1 && (uint64_t)n * vp.n_qd < 256ull * items) { vp.n_qd = vp.n_items = 0; return 0; }
	// the key layout: (block | row sum), 32 bits; more than that (thousands of blocks x hundreds of thousands of rows): the records
	uint32_t block_bits = 1;
	while ((1ull << block_bits) <= (uint64_t)n_blocks) ++block_bits;           // the block index, and n_blocks itself for skipped points
	for (int o = 0; o < 2; ++o) {
		double total = 0.0;
		for (uint32_t j = 0; j < vp.n_sc[o]; ++j) total += (double)vp.cap[o][j];
		uint32_t rb = 1;
		while ((double)((1ull << rb) - 1ull) <= total) ++rb;                   // all ones stays free: "no row"
		vp.row_bits[o] = rb; vp.row_none[o] = (uint32_t)((1ull << rb) - 1ull);
		if (rb + block_bits > 32u || rb > 23u) sc_ok = false;
	}
	if (!sc_ok) { vp.n_qd = vp.n_items = 0; return 0; }
	vp.n_items = items;
	vp.n_blocks = n_blocks;
	vp.n_groups = vp.n_qd * n_blocks;
	mask |= dense_mask;
	const uint32_t reps = vp.n_qd * (n / kVsPmax + 1u);
	vp.w_max = items + reps;
	vp.m_slots = 2u * reps;
	return mask;
}

void vm_sorted_scratch(const VsPlan &vp, uint32_t n, uint32_t E, bool second, bool forest, VsScratch &s) {
	uint64_t off = 0;
	auto take = [&](uint64_t bytes) { const uint64_t at = off; off += (bytes + 255) / 256 * 256; return at; };
	for (int o = 0; o < 2; ++o) { s.key_in[o] = take(4ull * n); s.key_out[o] = take(4ull * n); s.perm[o] = take(4ull * n); }
	for (int o = 0; o < 2; ++o) {
		s.keyb_in[o] = take(forest ? 4ull * n : 0); s.keyb_out[o] = take(forest ? 4ull * n : 0); s.permb[o] = take(forest ? 4ull * n : 0);
		s.xm[o] = take(forest ? 12ull * n : 0);
		s.cinfo[o] = take(forest ? 16ull * n : 0);
	}
	s.idxb = take(forest ? 4ull * n : 0);
	s.cand_cnt = take(4ull * div_up(n, 1024u));
	s.n_cand = take(4);
	s.plan = take(sizeof(VsPlan));
	s.stats = take(16);
	s.multi_list = take(4ull * (vp.m_slots / 2u + 1u));
	s.tmp = take(rsort::tmp_bytes(n, 2));
	for (int o = 0; o < 2; ++o) { s.xs[o] = take(12ull * n); s.vs[o] = take(second ? 12ull * n : 0); s.gts[o] = take(4ull * E * n); }
	s.items = take(4ull * (4ull * (vp.n_items + 1) + 6ull * vp.n_items));
	s.irec = take(32ull * vp.n_items);
	s.units = take(128ull * vp.w_max);
	s.item_tmax = take(4ull * vp.n_items);
	s.handoff = take(8ull * vp.n_items * vp.rb_max);
	s.lines = take(8ull * vp.w_max * vp.rd_max);
	s.slots = take(4ull * vp.m_slots * kVsSlot);
	s.total = off;
}

// runs the plan on `st`; scratch = the record / offsets regions of the workspace (the record classes run afterwards)
int vm_sorted_run(bool second, const VsPlan &vp, const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t n, const float *x,
                         const float *vin, const float *g, int64_t g_sn, int64_t g_se, const void *params, bool p_half, const Batch &ba,
                         const ForestDev *forest, float *dparam, char *scratch, const VsScratch &s, hipStream_t st) {
	const uint32_t E = meta->n_encoded_dims;
	const bool FO = forest != nullptr;
	uint32_t *key_in[2] = {(uint32_t *)(scratch + s.key_in[0]), (uint32_t *)(scratch + s.key_in[1])};
	uint32_t *key_out[2] = {(uint32_t *)(scratch + s.key_out[0]), (uint32_t *)(scratch + s.key_out[1])};
	uint32_t *perm[2] = {(uint32_t *)(scratch + s.perm[0]), (uint32_t *)(scratch + s.perm[1])};
	uint32_t *keyb_in[2] = {(uint32_t *)(scratch + s.keyb_in[0]), (uint32_t *)(scratch + s.keyb_in[1])};
	uint32_t *keyb_out[2] = {(uint32_t *)(scratch + s.keyb_out[0]), (uint32_t *)(scratch + s.keyb_out[1])};
	uint32_t *permb[2] = {(uint32_t *)(scratch + s.permb[0]), (uint32_t *)(scratch + s.permb[1])};
	uint32_t *idxb = (uint32_t *)(scratch + s.idxb), *cand_cnt = (uint32_t *)(scratch + s.cand_cnt), *n_cand = (uint32_t *)(scratch + s.n_cand);
	const uint32_t n_wg = div_up(n, 1024u);
	const VsPlan *vpd = (const VsPlan *)(scratch + s.plan);
	hipLaunchKernelGGL(k_vs_store_plan, dim3(1), dim3(256), 0, st, vp, (uint32_t *)(scratch + s.plan));
	if (FO) hipLaunchKernelGGL(k_vs_keys<true>, dim3(n_wg), dim3(1024), 0, st, vpd, n, x, ba, key_in[0], key_in[1], cand_cnt);
	else hipLaunchKernelGGL(k_vs_keys<false>, dim3(n_wg), dim3(1024), 0, st, vpd, n, x, ba, key_in[0], key_in[1], cand_cnt);
	NR3D_LAUNCH_CHECK();
	uint32_t block_bits = 1;
	while ((1ull << block_bits) <= (uint64_t)vp.n_blocks) ++block_bits;
	{
		// both orders in the same launches; the values are the point indices
		const uint32_t *kin[2] = {key_in[0], key_in[1]}, *vin[2] = {nullptr, nullptr};
		const uint32_t rb = vp.row_bits[0] > vp.row_bits[1] ? vp.row_bits[0] : vp.row_bits[1];
		if (int rc = rsort::sort_pairs(scratch + s.tmp, 2, kin, vin, key_out, perm, n, nullptr, (int)(rb + block_bits), st)) return rc;
	}
	if (FO) {
		// the boundary candidates of all blocks: compacted in point order (their number stays on the device), then ordered by row sum
		hipLaunchKernelGGL(k_vs_cand_scan, dim3(1), dim3(1024), 0, st, n_wg, cand_cnt, n_cand);
		hipLaunchKernelGGL(k_vs_cand_write, dim3(n_wg), dim3(1024), 0, st, vpd, n, x, ba, cand_cnt, idxb, keyb_in[0], keyb_in[1]);
		NR3D_LAUNCH_CHECK();
		const uint32_t *kin[2] = {keyb_in[0], keyb_in[1]}, *vin[2] = {idxb, idxb};
		const uint32_t rb = vp.row_bits[0] > vp.row_bits[1] ? vp.row_bits[0] : vp.row_bits[1];
		if (int rc = rsort::sort_pairs(scratch + s.tmp, 2, kin, vin, keyb_out, permb, n, n_cand, (int)rb, st)) return rc;
	}
	VsDev dv;
	dv.n_cand = n_cand;
	dv.irec = (VsItemRec *)(scratch + s.irec);
	uint32_t *ib = (uint32_t *)(scratch + s.items);
	dv.item_lo = ib; dv.item_cnt = ib + (vp.n_items + 1); dv.rep_base = ib + 2 * (size_t)(vp.n_items + 1); dv.multi_base = ib + 3 * (size_t)(vp.n_items + 1);
	dv.sub_lo = ib + 4 * (size_t)(vp.n_items + 1); dv.sub_cnt = dv.sub_lo + 3 * (size_t)vp.n_items;
	for (int o = 0; o < 2; ++o) {
		dv.permb[o] = permb[o]; dv.xm[o] = (const float *)(scratch + s.xm[o]);
		dv.cinfo[o] = (const VsCand *)(scratch + s.cinfo[o]);
		if (FO) hipLaunchKernelGGL(k_vs_cands, dim3(div_up(n, 256)), dim3(256), 0, st, n_cand, permb[o], x, ba, *forest, (float *)(scratch + s.xm[o]),
		                           (VsCand *)(scratch + s.cinfo[o]));
	}
	uint32_t *stats = (uint32_t *)(scratch + s.stats);
	NR3D_HIP_CHECK(hipMemsetAsync(stats, 0, 16, st));
	dv.stats = stats;
	dv.multi_list = (uint32_t *)(scratch + s.multi_list);
	uint64_t cols[2] = {0ull, 0ull};                    // the dL_dy columns of the served pseudo levels (two features each)
	for (uint32_t k = 0; k < vp.n_qd; ++k)
		for (uint32_t f = 0; f < 2u; ++f) { const uint32_t c = meta->map_col[vp.q[k]] + f; cols[c >> 6] |= 1ull << (c & 63u); }
	uint32_t n_wanted = 0;
	for (int w = 0; w < 2; ++w) for (uint64_t b = cols[w]; b; b &= b - 1) ++n_wanted;
	for (int o = 0; o < 2; ++o) {
		float *xs = (float *)(scratch + s.xs[o]), *vs = (float *)(scratch + s.vs[o]), *gts = (float *)(scratch + s.gts[o]);
		hipLaunchKernelGGL(k_vs_gather, dim3(div_up(n, kVsGatherPts)), dim3(256), (size_t)n_wanted * (kVsGatherPts + 1u) * 4u, st, n, E, perm[o], x,
		                   second ? vin : nullptr, g, g_sn, g_se, cols[0], cols[1], xs, vs, gts, o == 0 ? stats : nullptr);
		dv.skey[o] = key_out[o]; dv.xs[o] = xs; dv.vs[o] = vs; dv.gts[o] = gts;
	}
	dv.slots = (float *)(scratch + s.slots); dv.handoff = (float *)(scratch + s.handoff); dv.lines = (float *)(scratch + s.lines);
	const uint32_t plan_threads = vp.n_items;
	if (FO) hipLaunchKernelGGL(k_vs_plan<true>, dim3(div_up(plan_threads, 256)), dim3(256), 0, st, vpd, dv, md, n, ba);
	else hipLaunchKernelGGL(k_vs_plan<false>, dim3(div_up(plan_threads, 256)), dim3(256), 0, st, vpd, dv, md, n, ba);
	hipLaunchKernelGGL(k_vs_scan, dim3(1), dim3(1024), 0, st, vp.n_items, dv);
	const ForestDev fo = forest ? *forest : ForestDev{};
	dv.units = (const VsUnit *)(scratch + s.units);
	const uint32_t opt_fix = opt::get(NR3D_OPT_DIRECT_FIXED) != 0 ? 1u : 0u;      // (k_cp_direct's switch: 0 keeps fp64 accumulators everywhere)
	uint32_t *item_tmax = opt_fix ? (uint32_t *)(scratch + s.item_tmax) : nullptr;
	if (opt_fix) {
		if (p_half) { if (FO) hipLaunchKernelGGL((k_vs_tmax<__half, true>), dim3(vp.n_items), dim3(256), 0, st, vpd, dv, md, (const __half *)params, ba, item_tmax);
		              else hipLaunchKernelGGL((k_vs_tmax<__half, false>), dim3(vp.n_items), dim3(256), 0, st, vpd, dv, md, (const __half *)params, ba, item_tmax); }
		else { if (FO) hipLaunchKernelGGL((k_vs_tmax<float, true>), dim3(vp.n_items), dim3(256), 0, st, vpd, dv, md, (const float *)params, ba, item_tmax);
		       else hipLaunchKernelGGL((k_vs_tmax<float, false>), dim3(vp.n_items), dim3(256), 0, st, vpd, dv, md, (const float *)params, ba, item_tmax); }
	}
	if (FO) hipLaunchKernelGGL(k_vs_units<true>, dim3(div_up(vp.w_max, 256)), dim3(256), 0, st, vpd, dv, md, ba, fo, item_tmax, (VsUnit *)(scratch + s.units));
	else hipLaunchKernelGGL(k_vs_units<false>, dim3(div_up(vp.w_max, 256)), dim3(256), 0, st, vpd, dv, md, ba, fo, item_tmax, (VsUnit *)(scratch + s.units));
	NR3D_LAUNCH_CHECK();
	static bool attr[64] = {};
	int dev_id = 0;
	NR3D_HIP_CHECK(hipGetDevice(&dev_id));
	if (!attr[dev_id & 63]) {
#define NR3D_VS_ATTR(S_, P_, F_) NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_vm_sorted<S_, P_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, kVsLds))
		NR3D_VS_ATTR(false, float, false); NR3D_VS_ATTR(true, float, false); NR3D_VS_ATTR(false, __half, false); NR3D_VS_ATTR(true, __half, false);
		NR3D_VS_ATTR(false, float, true); NR3D_VS_ATTR(true, float, true); NR3D_VS_ATTR(false, __half, true); NR3D_VS_ATTR(true, __half, true);
#undef NR3D_VS_ATTR
		attr[dev_id & 63] = true;
	}
	auto launch = [&](auto kern, auto *tab) {
		hipLaunchKernelGGL(kern, dim3(vp.w_max), dim3(kVsThreads), (size_t)kVsLds, st, vpd, vp.rb_max, vp.rd_max, dv, n, meta->interpolation_type, x, vin, g, g_sn, g_se,
		                   tab, ba, fo, dparam, opt_fix NR3D_DBG_ARG(NR3D_XOPT(VS_DBG, 0)));
	};
	{
		prof::Scope ps(NR3D_PROF_LOTD_DIRECT, st);
#define NR3D_VS_GO(F_) do { if (p_half) { if (second) launch(k_vm_sorted<true, __half, F_>, (const __half *)params); else launch(k_vm_sorted<false, __half, F_>, (const __half *)params); } \
	else { if (second) launch(k_vm_sorted<true, float, F_>, (const float *)params); else launch(k_vm_sorted<false, float, F_>, (const float *)params); } } while (0)
		if (FO) NR3D_VS_GO(true); else NR3D_VS_GO(false);
#undef NR3D_VS_GO
	}
	hipLaunchKernelGGL(k_vs_reduce_items, dim3(vp.n_items + (vp.m_slots / 2u) * kVsRedChunks), dim3(256), 0, st, vpd, dv, md, ba, dparam);
	hipLaunchKernelGGL(k_vs_reduce_lines, dim3(vp.n_groups * div_up(2u * vp.rd_max, 64u)), dim3(1024), 0, st, vpd, dv, md, ba, dparam);
	NR3D_LAUNCH_CHECK();
	return 0;
}


}  // namespace lotd
}  // namespace nr3d
