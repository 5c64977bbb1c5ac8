// nr3d_lib_amd/csrc/lotd_pair.hip -- dL/dparam of 3-D Dense/Hash metas with 2-feature pseudo levels through PAIR records.
//
// Same two-stage, atomic-free organisation as lotd_bin.hip (stage A bins the updates of a block of points by table
// bucket, stage B accumulates a bucket in fp64 LDS and adds the slice to dL/dparam), but the unit that travels through
// HBM is not one corner update {entry, w_c*g0, w_c*g1} (12 B, 8 per point and level) but one PAIR of corners that land
// in the same bucket by construction:
//     Dense: the two corners along the contiguous last dim (entries e, e + 1; buckets are whole rows of the table);
//     Hash : the two corners along dim 0 (prime 1): hash(x0 + 1, ..) and hash(x0, ..) differ only in the low bits that
//            x0 -> x0 + 1 flips, so they share the 8192-entry bucket (power-of-two table, resolution <= 8192);
// written as {idx0 | idx1 << 13 | flags, w_pair, A0, A1} (16 B, 4 per point and level), A_f = g_f * prod of the other
// two dims' weights.  Stage B applies (1 - w_pair) * A_f to idx0 and w_pair * A_f to idx1.  Per (point, level) 64 B
// instead of 96 B cross HBM in each direction, half as many records are ranked and staged through LDS, and a record is
// one aligned 16-byte access everywhere.  Reference kernel replaced: kernel_lod_hashonly_backward_grid
// (csrc/lotd/include/lotd/lotd_hash_only.h:380-470); the per-corner value g_f * w_c is formed as
// (g_f * w_other) * w_pair instead of g_f * ((w_x * w_y) * w_z): same real number, <= 2 ulp apart.
//
// Coherent inputs (consecutive points in the same cell, e.g. samples along a ray) are merged like in lotd_bin.hip:
// the lanes of a run are summed into its first lane, which then emits two SINGLE records per pair (flags say which
// half is valid, weight 1) -- at most as many records as the unmerged lanes would have written.
#include "lotd_device.h"
#include <stdlib.h>


namespace nr3d {
namespace lotd {

constexpr uint32_t kPEpbMax = 8192;                  // accumulator entries per bucket: 2 features x 8192 x 8 B = 128 KiB (1 workgroup
                                                     // per CU) or x 4096 = 64 KiB (2 per CU); plan.lg = log2 of it, NR3D_PAIR_EPB_LOG2
constexpr int kPAccThreads = 1024;
constexpr int kPLdsMax = 2 * (int)kPEpbMax;          // fp64 accumulators per stage-B workgroup (upper bound)
constexpr int kPMaxLv = 32;                          // pseudo levels per plan
constexpr uint32_t kPMaxNb = 511;                    // buckets per pseudo level (one scan pass of the smallest block)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct PairPlan {
	uint32_t qmap[kPMaxLv];             // pseudo level of the meta
	uint32_t nb[kPMaxLv];               // buckets
	uint32_t epb[kPMaxLv];              // entries per bucket (Dense: rows per bucket x Rz; Hash: 8192)
	uint32_t shift[kPMaxLv];            // Dense: log2(rows per bucket); Hash: 13
	uint32_t bucket_base[kPMaxLv + 1];  // flat index of the level's first bucket
	uint32_t offs_base[kPMaxLv];        // start of the level's offset table (uint32 units)
	uint32_t n_blk, n_pseudo;
	uint32_t cap;                       // records per (pseudo level, point block) slot = 4 x points per block
	uint32_t lg;                        // log2 of the accumulator entries per bucket (12 or 13)
	uint32_t sum_log2;                  // updates per accumulator <= 2^sum_log2 (8 corners x points of the pass)
};

// Levels with at most kDirectNb buckets skip the records altogether (k_pair_direct): every workgroup of such a level
// reads (x, dL_dy columns) of a share of the points and accumulates the updates of ITS bucket straight into LDS --
// nb x 20 bytes per point instead of 64 written + ~79 read as records.  C2: levels 0 (1 bucket) and 1 (4 buckets).
constexpr uint32_t kDirectNb = 4, kDirectMaxLv = 8, kDirectMaxWg = 640;
struct DirectPlan {
	uint32_t n;                               // pseudo levels served (0: none)
	uint32_t qmap[kDirectMaxLv], nb[kDirectMaxLv], epb[kDirectMaxLv], shift[kDirectMaxLv];
	uint32_t bucket_base[kDirectMaxLv + 1];
	uint32_t lg, sum_log2, R, pts_per_rep;    // replicas per bucket, points per replica
	uint32_t merge_min;                       // lanes continuing their neighbour's cell from which a wave rotates its update order (experiments build: knob)
};


// Measurement knobs (only a -DNR3D_EXPERIMENTS build reads them, options.h): points per stage-A workgroup 512 | 768 | 1024,
// log2 of the entries per bucket 12 | 13, loads in flight per wave in stage B 4 | 8, stage-B work items (measured, NGP config,
// 2^20 points, 64 KiB buckets, backward ms: 768: 0.679, 1024: 0.669, 1536: 0.657, 2048: 0.671), and the timing experiment
// PAIR_DEBUG (results wrong by design: bit 0 = no LDS atomics, bit 1 = every lane re-reads one record).
static uint32_t pair_bp() { const int64_t x = NR3D_XOPT(PAIR_BP, 1024); return (x == 512 || x == 768) ? (uint32_t)x : 1024u; }
static uint32_t pair_lg() { return NR3D_XOPT(PAIR_EPB_LOG2, 12) == 13 ? 13u : 12u; }
static uint32_t pair_unroll() { return NR3D_XOPT(PAIR_UNROLL, 4) == 8 ? 8u : 4u; }
static uint32_t pair_units() { const int64_t x = NR3D_XOPT(PAIR_UNITS, 1536); return (uint32_t)(x < 256 ? 256 : (x > 8192 ? 8192 : x)); }
// selectable paths (nr3d_set_option; the tests compare both forms in one process)
static bool pair_quad_enabled() { return opt::on(NR3D_OPT_PAIR_QUAD); }
static uint32_t pair_fixed() { return opt::on(NR3D_OPT_PAIR_FIXED) ? 1u : 0u; }      // stage-B accumulators: 64-bit fixed point | fp64

// -------------------------------------------------------------------------------------------------
// Stage A
// -------------------------------------------------------------------------------------------------
// The four pair records of one (point, pseudo level): bucket, bucket-local indices of the two entries (hdr = i0 | i1 << 13),
// A_f = g_f x the weights of the two dims the pair does not run along, and the pair's own weight wp[m] (the same for the
// four records in the first-order form): the entries get (1 - wp) A_f and wp A_f.
// Second order (vin != NULL: d(dL/dx)/dparam, kernel_lod_hashonly_backward_input_backward_grid, lotd_hash_only.h:472-574):
// the combined weight of corner k is W'_k = sum_d a_d s_d(k) prod_{j != d} w_j(k_j), a_d = scale_d vin_d w'_d, s = -1 / +1 for
// the lower / upper corner.  With P the pair dim, (A, B) the other two and m their corner:
//     W'(lower) = (1 - wp) C_m - E_m,   W'(upper) = wp C_m + E_m,     C_m = a_A s_A w_B + a_B s_B w_A,  E_m = a_P w_A w_B
// which is the SAME record form with A_f = g_f C_m and wp'_m = wp + E_m / C_m (one more weight per record, no fifth word).
// C_m = 0 exactly (or 1e18 times smaller than E_m) is replaced by a C of that size: the products A_f (1 - wp') and A_f wp'
// then still give -+ g_f E_m to fp32 rounding, and what is lost of g_f C_m is below 1e-18 of it.
__device__ __forceinline__ float pair_wp2(float wp, float Cm, float Em, float &Cuse) {
	const float floor_c = fmaxf(1e-18f * fabsf(Em), 1e-30f);
	Cuse = fabsf(Cm) >= floor_c ? Cm : copysignf(floor_c, Cm);
	return __fmaf_rn(Em, __builtin_amdgcn_rcpf(Cuse), wp);   // v_rcp_f32 (1 ulp) + fma, no division sequence: 1e-7 of wp', far inside the contract
}
// Quad records (first order, Dense levels, quad_on): the pairs (x, y) and (x, y + 1) of a Dense level are neighbours in memory
// as well -- rows r and r + 1 of the table -- so ONE 16-byte record can carry the four entries e, e + 1, e + Rz, e + Rz + 1 with
// A_f = g_f w_x and both weights (w_y, w_z) as 24-bit fractions: half the record bytes of a Dense level (qmask bit bx: rows r,
// r + 1 of x-corner bx lie in one bucket; wy = w_y; g_f w_x is the sum of the two pairs' A_f = g_f w_x (1 - w_y), g_f w_x w_y).
template <bool SECOND = false>
__device__ __forceinline__ void pair_records(const Lvl &L, uint32_t sh, uint32_t epb, uint32_t lg, const float (&xp)[3], float g0,
                                             float g1, bool smooth, uint32_t (&hdr)[4], uint32_t (&bkt)[4], float (&A)[4][2],
                                             float (&wpm)[SECOND ? 4 : 1], uint32_t (&cell)[3], uint32_t nb, bool &valid,
                                             const float *__restrict__ vin, uint32_t &qmask, float &wy, bool quad_on) {
	// wpm: one pair weight per record in the second-order form, ONE for all four in the first-order form (the first-order
	// stage-A kernel sits exactly at its 64-register budget: four copies of the same value spilled)
	// valid: every pair lies inside ONE bucket of the level.  True by construction for x in [0, 1] (what the Python layer
	// clamps to, lotd.py:68); a point outside (or NaN) would index the LDS histogram / stage out of bounds -- it is dropped.
	Cell<3> c;
	locate<3>(xp, L, smooth, c);
	valid = true;
	float a[3] = {0.0f, 0.0f, 0.0f};
	if constexpr (SECOND) {
#pragma unroll
		for (int d = 0; d < 3; ++d) a[d] = c.sc[d] * vin[d] * c.dw[d];
	}
	if constexpr (!SECOND) { qmask = 0; wy = c.w[1]; }
	if (L.type == NR3D_LOD_Dense) {
		const float wp = c.w[2];
#pragma unroll
		for (uint32_t m = 0; m < 4; ++m) {
			const uint32_t bx = m & 1u, by = m >> 1;
			const uint32_t row = (c.g[0] + bx) * L.res[1] + (c.g[1] + by);
			const uint32_t e0 = row * L.res[2] + c.g[2];
			const uint32_t b = row >> sh;
			const uint32_t i0 = e0 - b * epb;
			valid = valid && b < nb && c.g[2] + 1u < L.res[2];
			bkt[m] = b;
			hdr[m] = i0 | ((i0 + 1u) << 13);
			const float wx = bx ? c.w[0] : 1.0f - c.w[0], wy = by ? c.w[1] : 1.0f - c.w[1];
			const float wo = wx * wy;
			if constexpr (!SECOND) { A[m][0] = g0 * wo; A[m][1] = g1 * wo; wpm[0] = wp; }
			else {
				const float Cm = __fmaf_rn(bx ? a[0] : -a[0], wy, (by ? a[1] : -a[1]) * wx);
				float Cuse;
				wpm[m] = pair_wp2(wp, Cm, a[2] * wo, Cuse);
				A[m][0] = g0 * Cuse; A[m][1] = g1 * Cuse;
			}
		}
		if constexpr (!SECOND) {
#pragma unroll
			for (uint32_t bx = 0; bx < 2; ++bx)       // rows r (slot bx) and r + 1 (slot bx + 2) in one bucket, indices within 13 bits
				if (quad_on && bkt[bx] == bkt[bx + 2u] && (hdr[bx] & 8191u) + L.res[2] + 1u < 8192u) qmask |= 1u << bx;
		}
	} else {
		const float wp = c.w[0];
		const bool pow2 = (L.size & (L.size - 1u)) == 0u;
		const uint32_t emask = (1u << lg) - 1u;
#pragma unroll
		for (uint32_t m = 0; m < 4; ++m) {
			const uint32_t by = m & 1u, bz = m >> 1;
			const uint32_t K = ((c.g[1] + by) * kPrimes[1]) ^ ((c.g[2] + bz) * kPrimes[2]);
			const uint32_t h0 = c.g[0] ^ K, h1 = (c.g[0] + 1u) ^ K;
			const uint32_t e0 = pow2 ? (h0 & (L.size - 1u)) : (h0 % L.size);
			const uint32_t e1 = pow2 ? (h1 & (L.size - 1u)) : (h1 % L.size);
			bkt[m] = e0 >> lg;                         // == e1 >> lg (plan conditions) for x in [0, 1]
			valid = valid && (e1 >> lg) == bkt[m] && bkt[m] < nb;
			hdr[m] = (e0 & emask) | ((e1 & emask) << 13);
			const float wy = by ? c.w[1] : 1.0f - c.w[1], wz = bz ? c.w[2] : 1.0f - c.w[2];
			const float wo = wy * wz;
			if constexpr (!SECOND) { A[m][0] = g0 * wo; A[m][1] = g1 * wo; wpm[0] = wp; }
			else {
				const float Cm = __fmaf_rn(by ? a[1] : -a[1], wz, (bz ? a[2] : -a[2]) * wy);
				float Cuse;
				wpm[m] = pair_wp2(wp, Cm, a[0] * wo, Cuse);
				A[m][0] = g0 * Cuse; A[m][1] = g1 * Cuse;
			}
		}
	}
#pragma unroll
	for (int d = 0; d < 3; ++d) cell[d] = c.g[d];
}

// the buckets of the four pair records alone (pair_records' bkt): cell location + row / hash, nothing else
__device__ __forceinline__ void pair_buckets(const Lvl &L, uint32_t sh, uint32_t lg, const float (&xp)[3], bool smooth, uint32_t (&bkt)[4]) {
	Cell<3> c;
	locate<3>(xp, L, smooth, c);
	if (L.type == NR3D_LOD_Dense) {
#pragma unroll
		for (uint32_t m = 0; m < 4; ++m) bkt[m] = ((c.g[0] + (m & 1u)) * L.res[1] + (c.g[1] + (m >> 1))) >> sh;
	} else {
		const bool pow2 = (L.size & (L.size - 1u)) == 0u;
#pragma unroll
		for (uint32_t m = 0; m < 4; ++m) {
			const uint32_t h0 = c.g[0] ^ ((c.g[1] + (m & 1u)) * kPrimes[1]) ^ ((c.g[2] + (m >> 1)) * kPrimes[2]);
			bkt[m] = (pow2 ? (h0 & (L.size - 1u)) : (h0 % L.size)) >> lg;
		}
	}
}

// One pseudo level of one block of kPBP points: pair records -> rank inside the bucket -> counting sort in LDS ->
// coalesced write-out of the slot + its bucket offsets.  `hist` [nb + 1] must be zero on entry (and that visible: a
// barrier behind the zeroing); `zero_next` (optional) is zeroed for the following call.  Four barriers (five with zero_next).
template <int kPBP, bool SECOND = false>
__device__ __forceinline__ void pair_level(const PairPlan &plan, uint32_t ql, const Lvl &L, bool active, const float (&xp)[3],
                                           float g0, float g1, uint32_t smooth, u32x4 *__restrict__ stage,
                                           uint32_t *__restrict__ hist, uint32_t *__restrict__ zero_next, uint32_t *scan_lds,
                                           u32x4 *__restrict__ dst, uint32_t *__restrict__ ob, uint32_t ob_stride,
                                           const float *__restrict__ vin = nullptr, bool quad_on = false) {
	const uint32_t nb = plan.nb[ql];
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t hdr[4], bkt[4], cell[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
	constexpr int kW = SECOND ? 3 : 0;                   // wp[m & kW]: per record (second order) or shared
	constexpr bool QUADS = !SECOND;                      // qmask stays 0 for Hash levels and with quad_on off
	uint32_t qmask = 0;
	float wy = 0.0f;
	float A[4][2], wp[SECOND ? 4 : 1];
#pragma unroll
	for (int m = 0; m < 4; ++m) { hdr[m] = 0; bkt[m] = 0; A[m][0] = 0.0f; A[m][1] = 0.0f; wp[m & kW] = 0.0f; }
	if (zero_next)
		for (uint32_t b = threadIdx.x; b <= kPMaxNb; b += kPBP) zero_next[b] = 0;
	if (active) {
		bool valid;
		pair_records<SECOND>(L, plan.shift[ql], plan.epb[ql], plan.lg, xp, g0, g1, smooth != 0, hdr, bkt, A, wp, cell, nb, valid, vin, qmask, wy,
		                     quad_on);
		active = valid;
	}

	// ---- coherent inputs: lanes that continue the previous lane's cell are summed into the head of their run ----
	bool emit = active, split = false;
	float Hh[4][2];
#pragma unroll
	for (int m = 0; m < 4; ++m) { Hh[m][0] = 0.0f; Hh[m][1] = 0.0f; }
	{
		auto prev = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); };
		bool same = active && lane > 0;
#pragma unroll
		for (int d = 0; d < 3; ++d) same = same && (prev(cell[d]) == cell[d]);
		same = same && (prev((uint32_t)active) != 0u);
		const unsigned long long cont = __ballot(same);
		if (__popcll(cont) >= 16) {
			float lo[4][2], hi[4][2];
#pragma unroll
			for (int m = 0; m < 4; ++m)
#pragma unroll
				for (int f = 0; f < 2; ++f) { lo[m][f] = (1.0f - wp[m & kW]) * A[m][f]; hi[m][f] = wp[m & kW] * A[m][f]; }
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const unsigned long long need = (1ull << off) - 1ull;
				const bool take = (lane + off < 64) && (((cont >> (lane + 1)) & need) == need);
#pragma unroll
				for (int m = 0; m < 4; ++m)
#pragma unroll
					for (int f = 0; f < 2; ++f) {
						const float tl = __shfl_down(lo[m][f], off, 64), th = __shfl_down(hi[m][f], off, 64);
						if (take) { lo[m][f] += tl; hi[m][f] += th; }
					}
			}
			const bool head_multi = active && !same && lane < 63 && ((cont >> (lane + 1)) & 1ull);
			if (same) emit = false;
			if (head_multi) {
				split = true;
#pragma unroll
				for (int m = 0; m < 4; ++m)
#pragma unroll
					for (int f = 0; f < 2; ++f) { A[m][f] = lo[m][f]; Hh[m][f] = hi[m][f]; }
			}
		}
	}

	// ---- rank inside the bucket (split lanes take two consecutive slots per pair) ----
	uint32_t rank[4] = {0, 0, 0, 0};
	const uint32_t cnt = emit ? (split ? 2u : 1u) : 0u;      // per slot that holds a record
	// slot m holds a record unless its pair travels inside the quad of slot m - 2 (quads only from unmerged lanes)
	auto emits = [&](int m) {
		if constexpr (QUADS) return emit && !(!split && m >= 2 && ((qmask >> (m - 2)) & 1u));
		else return emit;
	};
	if (nb <= 4) {
		// one to four buckets: every record of the block would hit the same histogram counters -> rank through ballots
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const bool e_m = emits(m);
			const uint32_t bv = e_m ? bkt[m] : 0xFFFFFFFFu;
			unsigned long long todo = __ballot(e_m);
			while (todo) {
				const int leader = __ffsll((long long)todo) - 1;
				const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)bv, leader);
				const bool mine = e_m && bv == v;
				const unsigned long long m1 = __ballot(mine), m2 = __ballot(mine && split);
				uint32_t first = 0;
				if ((int)lane == leader) first = atomicAdd(&hist[v], (uint32_t)(__popcll(m1) + __popcll(m2)));
				first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
				const unsigned long long below = (1ull << lane) - 1ull;
				if (mine) rank[m] = first + (uint32_t)(__popcll(m1 & below) + __popcll(m2 & below));
				todo &= ~m1;
			}
		}
	} else {
		// Many buckets.  Spread-out points hit different counters and one LDS atomic per record is fine; samples along rays
		// put most of a wave into ONE bucket and those atomics would serialise on one address.  So the bucket of the wave's
		// first record is ranked through a ballot (one atomic for the whole group) when at least 8 lanes share it; whoever
		// is left uses its own atomic.
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const bool e_m = emits(m);
			const uint32_t bv = e_m ? bkt[m] : 0xFFFFFFFFu;
			const unsigned long long em = __ballot(e_m);
			bool done = !e_m;
			if (em) {
				const int leader = __ffsll((long long)em) - 1;
				const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)bv, leader);
				const bool mine = e_m && bv == v;
				const unsigned long long m1 = __ballot(mine);
				if (__popcll(m1) >= 8) {
					const unsigned long long m2 = __ballot(mine && split);
					uint32_t first = 0;
					if ((int)lane == leader) first = atomicAdd(&hist[v], (uint32_t)(__popcll(m1) + __popcll(m2)));
					first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
					const unsigned long long below = (1ull << lane) - 1ull;
					if (mine) { rank[m] = first + (uint32_t)(__popcll(m1 & below) + __popcll(m2 & below)); done = true; }
				}
			}
			if (!done) rank[m] = atomicAdd(&hist[bv], cnt);
		}
	}
	__syncthreads();

	// ---- exclusive scan of the histogram, hist[nb] = total (nb + 1 <= kPBP) ----
	{
		const uint32_t b = threadIdx.x;
		const uint32_t v = (b < nb) ? hist[b] : 0u;
		uint32_t inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint32_t t = __shfl_up(inc, off, 64);
			if ((int)lane >= off) inc += t;
		}
		if (lane == 63) scan_lds[threadIdx.x >> 6] = inc;
		__syncthreads();
		uint32_t wave_off = 0;
#pragma unroll
		for (int k = 0; k < kPBP / 64; ++k) { const uint32_t t = scan_lds[k]; if (k < (int)(threadIdx.x >> 6)) wave_off += t; }
		if (b <= nb) hist[b] = wave_off + inc - v;
		__syncthreads();
	}

	// ---- counting sort into the LDS staging area ----
	if (emit) {
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			if constexpr (QUADS) { if (!emits(m)) continue; }
			const uint32_t pos = hist[bkt[m]] + rank[m];
			if (QUADS && !split && m < 2 && ((qmask >> m) & 1u)) {
				// quad record: x = i0 | w_y bits 8..20 << 13 | 4 << 26 | w_y bits 21..23 << 29, y = w_z (24 bits) | w_y bits 0..7 << 24,
				// z / w = g_f w_x; stage B adds (1 - w_y)(1 - w_z), (1 - w_y) w_z, w_y (1 - w_z), w_y w_z times that to the entries
				// i0, i0 + 1, i0 + Rz, i0 + Rz + 1
				const uint32_t wzq = (uint32_t)__float2uint_rn(wp[0] * 16777215.0f), wyq = (uint32_t)__float2uint_rn(wy * 16777215.0f);
				stage[pos] = u32x4{(hdr[m] & 8191u) | (((wyq >> 8) & 8191u) << 13) | (4u << 26) | ((wyq >> 21) << 29),
				                   wzq | ((wyq & 255u) << 24), __float_as_uint(A[m][0] + A[m + 2][0]), __float_as_uint(A[m][1] + A[m + 2][1])};
			} else if (!split) {
				stage[pos] = u32x4{hdr[m] | (3u << 26), __float_as_uint(wp[m & kW]), __float_as_uint(A[m][0]), __float_as_uint(A[m][1])};
			} else {
				stage[pos] = u32x4{(hdr[m] & 0x1FFFu) | (1u << 26), 0u, __float_as_uint(A[m][0]), __float_as_uint(A[m][1])};
				stage[pos + 1] = u32x4{(hdr[m] & 0x3FFE000u) | (2u << 26), 0u, __float_as_uint(Hh[m][0]), __float_as_uint(Hh[m][1])};
			}
		}
	}
	__syncthreads();

	// ---- coalesced write-out (written once, read once by stage B: non-temporal) ----
	const uint32_t total = hist[nb];
	for (uint32_t v = threadIdx.x; v < total; v += kPBP) __builtin_nontemporal_store(stage[v], dst + v);
	for (uint32_t b = threadIdx.x; b <= nb; b += kPBP) ob[(size_t)b * ob_stride] = hist[b];
	// the next level's call zeroes THIS call's `hist` (its zero_next) right away: every read above must be done first
	if (zero_next) __syncthreads();
}

// max |dL/dy| of the workgroup as float bits -> gmax (the fixed-point scale of stage B).  |float| bits order like
// unsigned integers (NaN / inf on top).  One candidate per workgroup, and the atomic only when it would raise the running
// maximum (an L2 load first): 2^18 same-address atomics would serialise for ms
template <int kPBP>
__device__ __forceinline__ void pair_gmax(uint32_t gbits, uint32_t *scan_lds, uint32_t *__restrict__ gmax) {
	const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) gbits = max(gbits, (uint32_t)__shfl_xor((int)gbits, off, 64));
	__syncthreads();                                       // scan_lds is free
	if (lane == 0) scan_lds[threadIdx.x >> 6] = gbits;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t m = 0;
#pragma unroll
		for (int k = 0; k < kPBP / 64; ++k) m = max(m, scan_lds[k]);
		if (m > __hip_atomic_load(gmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(gmax, m);
	}
}

// one workgroup = kPBP points x ONE pseudo level; dL_dy given feature-major (coalesced columns) or with any strides
template <int kPBP, bool SECOND = false>
__global__ __launch_bounds__(kPBP, kPBP == 768 ? 6 : 8) /* <= 64 VGPRs: two 64 KiB workgroups per CU */ void k_pair_bin(PairPlan plan, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n,
                                                   int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                   const float *__restrict__ g, int64_t g_sn, int64_t g_se,
                                                   u32x4 *__restrict__ rec, uint32_t *__restrict__ offs_g,
                                                   uint32_t *__restrict__ gmax, DirectPlan dp, const float *__restrict__ vin_,
                                                   uint32_t quad_on) {
	// SECOND: d(dL/dx)/dparam for dL_ddLdx = vin_ (pair_records); else vin_ is unused
	constexpr uint32_t kPCap = (uint32_t)kPBP * 4u;
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];    // stage[kPCap] records | hist[nb + 1]
	__shared__ uint32_t scan_lds[kPBP / 64];
	u32x4 *stage = reinterpret_cast<u32x4 *>(smem);
	uint32_t *hist = smem + (size_t)kPCap * 4;
	const uint32_t blk = blockIdx.x, ql = blockIdx.y;
	const uint32_t q = plan.qmap[ql], nb = plan.nb[ql];
	const uint32_t level = meta_level_of(md, q);
	const Lvl L = load_level(md, level);
	const uint32_t i = blk * kPBP + threadIdx.x;
	for (uint32_t b = threadIdx.x; b <= nb; b += kPBP) hist[b] = 0;
	__syncthreads();
	const bool active = (i < n) && ((int32_t)level <= max_level);
	float xp[3] = {0.0f, 0.0f, 0.0f}, vin[3] = {0.0f, 0.0f, 0.0f}, g0 = 0.0f, g1 = 0.0f;
	if (active) {
#pragma unroll
		for (int d = 0; d < 3; ++d) xp[d] = x[(size_t)i * 3 + d];
		g0 = g[(int64_t)i * g_sn + (int64_t)(q * 2) * g_se];
		g1 = g[(int64_t)i * g_sn + (int64_t)(q * 2 + 1) * g_se];
		if constexpr (SECOND) {
#pragma unroll
			for (int d = 0; d < 3; ++d) vin[d] = vin_[(size_t)i * 3 + d];
		}
	}
	pair_level<kPBP, SECOND>(plan, ql, L, active, xp, g0, g1, smooth, stage, hist, nullptr, scan_lds,
	                         rec + ((size_t)ql * plan.n_blk + blk) * (size_t)plan.cap, offs_g + plan.offs_base[ql] + blk, plan.n_blk, vin,
	                         quad_on != 0u);
	if (gmax) {
		// bound of a single update: first order |g| (a weight in [0, 1] times a gradient); second order |g| sum_d |a_d| with
		// |a_d| <= 1.5 scale_d |vin_d| (w' <= 1.5 for the smoothstep, 1 linear)
		uint32_t gbits;
		if constexpr (!SECOND) {
			gbits = max(__float_as_uint(g0) & 0x7FFFFFFFu, __float_as_uint(g1) & 0x7FFFFFFFu);
			// the levels that bypass the records (k_pair_direct) share the fixed-point scale: their columns of dL_dy count too
			if (ql == 0 && i < n)
				for (uint32_t e = 0; e < dp.n; ++e) {
					const uint32_t qd = dp.qmap[e];
					if ((int32_t)meta_level_of(md, qd) > max_level) continue;
					gbits = max(gbits, __float_as_uint(g[(int64_t)i * g_sn + (int64_t)(qd * 2) * g_se]) & 0x7FFFFFFFu);
					gbits = max(gbits, __float_as_uint(g[(int64_t)i * g_sn + (int64_t)(qd * 2 + 1) * g_se]) & 0x7FFFFFFFu);
				}
		} else {
			auto bound = [&](const Lvl &Lq, float ga, float gb) {
				const float m = fmaxf(fabsf(ga), fabsf(gb)) * 1.5f *
				                ((float)(Lq.res[0] - 2u) * fabsf(vin[0]) + (float)(Lq.res[1] - 2u) * fabsf(vin[1]) + (float)(Lq.res[2] - 2u) * fabsf(vin[2]));
				return __float_as_uint(m) & 0x7FFFFFFFu;
			};
			gbits = bound(L, g0, g1);
			if (ql == 0 && i < n) {
				if (!active) {
#pragma unroll
					for (int d = 0; d < 3; ++d) vin[d] = vin_[(size_t)i * 3 + d];
				}
				for (uint32_t e = 0; e < dp.n; ++e) {
					const uint32_t qd = dp.qmap[e];
					if ((int32_t)meta_level_of(md, qd) > max_level) continue;
					const Lvl Ld = load_level(md, meta_level_of(md, qd));
					gbits = max(gbits, bound(Ld, g[(int64_t)i * g_sn + (int64_t)(qd * 2) * g_se], g[(int64_t)i * g_sn + (int64_t)(qd * 2 + 1) * g_se]));
				}
			}
		}
		pair_gmax<kPBP>(gbits, scan_lds, gmax);
	}
}

// Records per bucket over all point blocks (one wave per bucket), and -- in the LAST workgroup to finish -- the stage-B work
// plan: a bucket with more than total / n_units records is split into replicas over its point blocks, empty buckets
// get no workgroup (item_start = exclusive prefix of the replica counts).  One launch instead of a totals kernel and a
// single-workgroup planning kernel; `ticket` (zeroed with gmax) counts finished workgroups.
__global__ __launch_bounds__(1024) void k_pair_plan(PairPlan plan, const uint32_t *__restrict__ offs_g, uint32_t n_units,
                                                    uint32_t *__restrict__ tot, uint32_t *__restrict__ rep,
                                                    uint32_t *__restrict__ item_start, uint32_t *__restrict__ ticket) {
	__shared__ uint64_t red[16];
	__shared__ uint64_t carry_s;
	__shared__ uint32_t last_s;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t NB = plan.bucket_base[plan.n_pseudo];
	{
		const uint32_t fb = blockIdx.x * 16 + wave;
		if (fb < NB) {
			uint32_t q = 0;
			while (q + 1 < plan.n_pseudo && plan.bucket_base[q + 1] <= fb) ++q;
			const uint32_t *ob0 = offs_g + plan.offs_base[q] + (size_t)(fb - plan.bucket_base[q]) * plan.n_blk;
			const uint32_t *ob1 = ob0 + plan.n_blk;
			uint32_t sum = 0;
			for (uint32_t blk = lane; blk < plan.n_blk; blk += 64) sum += ob1[blk] - ob0[blk];
#pragma unroll
			for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
			if (lane == 0) __hip_atomic_store(tot + fb, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2-visible
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		last_s = (atomicAdd(ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	__syncthreads();
	if (!last_s) return;
	auto ld_tot = [&](uint32_t fb) { return (uint64_t)__hip_atomic_load(tot + fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
	uint64_t part = 0;
	for (uint32_t fb = threadIdx.x; fb < NB; fb += 1024) part += ld_tot(fb);
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
	if (lane == 0) red[wave] = part;
	__syncthreads();
	uint64_t total = 0;
#pragma unroll
	for (int w = 0; w < 16; ++w) total += red[w];
	const uint64_t unit = total / n_units > 0 ? total / n_units : 1;
	if (threadIdx.x == 0) carry_s = 0;
	__syncthreads();
	for (uint32_t base = 0; base < NB; base += 1024) {
		const uint32_t fb = base + threadIdx.x;
		uint32_t r = 0;
		if (fb < NB) {
			const uint64_t t = ld_tot(fb);
			r = t == 0 ? 0u : (uint32_t)((t + unit / 2) / unit);
			if (t != 0 && r < 1) r = 1;
			if (r > plan.n_blk) r = plan.n_blk;
			rep[fb] = r;
		}
		uint64_t inc = r;                      // inclusive scan over the 1024 threads
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) { const uint64_t t2 = __shfl_up(inc, off, 64); if ((int)lane >= off) inc += t2; }
		__syncthreads();
		if (lane == 63) red[wave] = inc;
		__syncthreads();
		uint64_t woff = 0, tsum = 0;
#pragma unroll
		for (int w = 0; w < 16; ++w) { if (w < (int)wave) woff += red[w]; tsum += red[w]; }
		const uint64_t c = carry_s;
		if (fb < NB) item_start[fb] = (uint32_t)(c + woff + inc - r);
		__syncthreads();
		if (threadIdx.x == 0) carry_s = c + tsum;
		__syncthreads();
	}
	if (threadIdx.x == 0) item_start[NB] = (uint32_t)carry_s;
}

// accumulator slot t (feature-major: t = f * 2^lg + el) of bucket b -> element of dL/dparam, nullptr outside the level
__device__ __forceinline__ float *pair_target(const Lvl &L, uint32_t epb, uint32_t lg, uint32_t foff0, uint32_t b, uint32_t t,
                                              float *__restrict__ dparam, bool out_half = false) {
	const uint32_t f = t >> lg, el = t & ((1u << lg) - 1u);
	const uint64_t entry = (uint64_t)b * epb + el;
	if (el >= epb || entry >= L.size) return nullptr;
	const uint64_t at = L.off + (entry * L.F + foff0 + f);
	// half output ((float, half, float) type combination): the same element index in a __half buffer
	return out_half ? reinterpret_cast<float *>(reinterpret_cast<__half *>(dparam) + at) : dparam + at;
}
// dparam[p] += v for either storage type (p from pair_target)
__device__ __forceinline__ float pair_ld(const float *p, bool out_half) {
	return out_half ? __half2float(*reinterpret_cast<const __half *>(p)) : *p;
}
__device__ __forceinline__ void pair_st(float *p, float v, bool out_half) {
	if (out_half) *reinterpret_cast<__half *>(p) = __float2half(v); else *p = v;
}

// fp64 LDS atomics run at ~1.3-1.5 T/s chip-wide, 64-bit integer ones at ~2.5 T/s (tools/ubench_lds): with FIX the
// accumulators are 64-bit fixed point.  Scale 2^s from the largest |dL/dy| of the call (stage A's gmax; every update is
// a weight in [0, 1] times a gradient) and the number of points: |sum| < 8 n max|g| 2^s <= 2^62, resolution
// max|g| * 2^-min(59 - log2 n, 44) -- far below an fp32 ulp of any non-negligible entry -- and the sum is exact, so the result
// does not depend on the order of the updates at all.  Non-finite gradients fall back to fp64 accumulation (uniform).
struct PairFix { double scale, inv; bool on; };
__device__ __forceinline__ PairFix pair_fix(const uint32_t *__restrict__ gmax, uint32_t sum_log2) {
	PairFix f;
	const uint32_t bits = ((cu32_t)gmax)[0];
	f.on = bits < 0x7F800000u;
	const int e = max((int)(bits >> 23), 1) - 126;                    // max|g| < 2^e
	// single values stay below 2^51 (rounding trick below) -- and a "single value" may be the fp32 sum of up to 64 merged
	// lanes (coherent points: stage A's split singles, k_pair_direct's run heads): 6 bits of headroom for those.
	// Resolution max|g| * 2^-44, still far below an fp32 ulp of any entry that matters.
	const int lim = min(62 - (int)sum_log2, 50 - 6);
	const int sc = max(min(lim - e, 1000), -1000);
	f.scale = __longlong_as_double((long long)(sc + 1023) << 52);
	f.inv = __longlong_as_double((long long)(1023 - sc) << 52);
	return f;
}
__device__ __forceinline__ unsigned long long to_fix(float v, double scale) {
	const double t = __fma_rn((double)v, scale, 0x1.8p52);            // round to nearest integer in the low mantissa bits
	return (unsigned long long)(__double_as_longlong(t) - 0x4338000000000000LL);
}

// -------------------------------------------------------------------------------------------------
// Stage B: one bucket (x replica) -> LDS accumulation -> slice of dL/dparam
// -------------------------------------------------------------------------------------------------
template <int kUnroll, bool FIX>
__global__ __launch_bounds__(kPAccThreads, 8) /* 8 waves per SIMD: two 64 KiB workgroups per CU */ void k_pair_accum(PairPlan plan, const nr3d_lotd_meta_t *__restrict__ md,
                                                             const u32x4 *__restrict__ rec,
                                                             const uint32_t *__restrict__ offs_g,
                                                             const uint32_t *__restrict__ rep_g,
                                                             const uint32_t *__restrict__ item_start,
                                                             const uint32_t *__restrict__ gmax,
                                                             float *__restrict__ partial, float *__restrict__ dparam NR3D_DBG_PARAM,
                                                             uint32_t out_half) {
	NR3D_DBG_DECL
	extern __shared__ __attribute__((aligned(16))) unsigned long long acc_raw[];   // [2][2^lg] 8-byte accumulators, feature-major
	double *acc = reinterpret_cast<double *>(acc_raw);
	const uint32_t NB = plan.bucket_base[plan.n_pseudo];
	const cu32_t istart = (cu32_t)item_start, irep = (cu32_t)rep_g;
	if (blockIdx.x >= istart[NB]) return;
	uint32_t lo = 0, hi = NB;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (istart[mid] <= blockIdx.x) lo = mid; else hi = mid; }
	const uint32_t fb = lo, R = irep[fb], r = blockIdx.x - istart[fb];
	uint32_t q = 0;
	while (q + 1 < plan.n_pseudo && plan.bucket_base[q + 1] <= fb) ++q;
	const uint32_t b = fb - plan.bucket_base[q];
	const uint32_t qg = plan.qmap[q];
	const Lvl L = load_level(md, meta_level_of(md, qg));
	const uint32_t foff0 = meta_cnt_of(md, qg) * 2u, epb = plan.epb[q];
	PairFix fx = {1.0, 1.0, false};
	if constexpr (FIX) fx = pair_fix(gmax, plan.sum_log2);
	const bool fix = FIX && fx.on;

	const uint32_t kPEpb = 1u << plan.lg, kPLds = 2u << plan.lg;
	for (uint32_t t = threadIdx.x; t < kPLds; t += kPAccThreads) acc_raw[t] = 0ull;       // +0.0 as a double, too
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr uint32_t n_waves = kPAccThreads / 64;
	// Replica r takes the point blocks r, r + R, r + 2R, ... (round 6; was: one contiguous range of blocks).  The plan gives a bucket
	// replicas by its record COUNT; with ordered inputs (samples along rays / a Morton curve) those records sit in a few neighbouring
	// blocks, so a contiguous range handed one replica all of them and the others none.  Sums are exact (fixed point): the split does
	// not change the result.  blk_lo / blk_hi below count the replica's blocks, block k of it is r + R k.
	const uint32_t blk_lo = 0u, blk_hi = (plan.n_blk > r) ? (plan.n_blk - r + R - 1u) / R : 0u;
	const uint32_t *ob0 = offs_g + plan.offs_base[q] + (size_t)b * plan.n_blk;
	const uint32_t *ob1 = ob0 + plan.n_blk;
	const uint32_t kPCap = plan.cap;
	const u32x4 *rec_q = rec + (size_t)q * plan.n_blk * (size_t)kPCap;
	const uint32_t per_wave = (blk_hi - blk_lo + n_waves - 1) / n_waves;
	const uint32_t w_lo = min(blk_lo + wave * per_wave, blk_hi), w_hi = min(w_lo + per_wave, blk_hi);
	// The point blocks of this replica are split evenly over the waves.  Per step a wave takes up to 64 of its blocks
	// (lane t fetches the run [start, end) of block blk0 + t, coalesced) and walks them in groups of kGroup runs: the
	// runs of a group are treated as ONE stream of `total` records and lane l of pass p takes record 64 p + l of it,
	// whichever run that is (run boundaries are wave-uniform: kGroup - 1 compare / select pairs per load).  Every pass but
	// the last is full, the kUnroll loads of a chunk are independent, and a run a little longer than 64 records does
	// not cost a second, nearly empty, dependent pass.
	constexpr int kGroup = 8;
	for (uint32_t blk0 = w_lo; blk0 < w_hi; blk0 += 64) {
		const uint32_t mb = blk0 + lane;
		const uint32_t s_l = (mb < w_hi) ? ob0[r + R * mb] : 0u;
		const uint32_t e_l = (mb < w_hi) ? ob1[r + R * mb] : 0u;
		const uint32_t n_run = min(64u, w_hi - blk0);
		const u32x4 *rec_b = rec_q + (size_t)(r + R * blk0) * kPCap;
		for (uint32_t j0 = 0; j0 < n_run; j0 += kGroup) {
			uint32_t pre[kGroup + 1], rbase[kGroup];
			pre[0] = 0;
#pragma unroll
			for (int u = 0; u < kGroup; ++u) {
				const uint32_t j = min(j0 + (uint32_t)u, 63u);
				const uint32_t s_j = __builtin_amdgcn_readlane(s_l, j), e_j = __builtin_amdgcn_readlane(e_l, j);
				const uint32_t nrec = (j0 + (uint32_t)u < n_run) ? e_j - s_j : 0u;
				rbase[u] = s_j + j * R * kPCap - pre[u];      // record index of stream position p inside run u: rbase[u] + p
				pre[u + 1] = pre[u] + nrec;
			}
			const uint32_t total = pre[kGroup];
			for (uint32_t p0 = 0; p0 < total; p0 += 64u * kUnroll) {
				u32x4 rv[kUnroll];
#pragma unroll
				for (int v = 0; v < kUnroll; ++v) {
					const uint32_t pos = p0 + 64u * (uint32_t)v + lane;
					uint32_t at = rbase[0];
#pragma unroll
					for (int u = 1; u < kGroup; ++u) at = (pos >= pre[u]) ? rbase[u] : at;
					// branch-free (a load under an `if` makes the compiler drain all outstanding loads at the join): lanes
					// past the end re-read the group's first slot and drop it
					rv[v] = __builtin_nontemporal_load(rec_b + (size_t)((pos < total && !(dbg & 2u)) ? at + pos : rbase[0]));
				}
#pragma unroll
				for (int v = 0; v < kUnroll; ++v)
					if (p0 + 64u * (uint32_t)v + lane < total && !(dbg & 1u)) {
						const uint32_t h = rv[v].x, fl = (h >> 26) & 7u;
						const uint32_t i0 = h & 8191u, i1 = (h >> 13) & 8191u;
						const float w = __uint_as_float(rv[v].y), a0 = __uint_as_float(rv[v].z), a1 = __uint_as_float(rv[v].w);
						if (fl == 4u) {
							// quad record of a Dense level (pair_level): entries i0, i0 + 1 of row r and i0 + Rz, i0 + Rz + 1 of row r + 1
							const uint32_t wyq = (((h >> 13) & 8191u) << 8) | (rv[v].y >> 24) | ((h >> 29) << 21), wzq = rv[v].y & 0xFFFFFFu;
							const float wy = (float)wyq * (1.0f / 16777215.0f), wz = (float)wzq * (1.0f / 16777215.0f);
							const float q00 = (1.0f - wy) * (1.0f - wz), q01 = (1.0f - wy) * wz, q10 = wy * (1.0f - wz), q11 = wy * wz;
							const uint32_t i2 = i0 + L.res[2];
							if (fix) {
								atomicAdd(&acc_raw[i0], to_fix(q00 * a0, fx.scale)); atomicAdd(&acc_raw[kPEpb + i0], to_fix(q00 * a1, fx.scale));
								atomicAdd(&acc_raw[i0 + 1u], to_fix(q01 * a0, fx.scale)); atomicAdd(&acc_raw[kPEpb + i0 + 1u], to_fix(q01 * a1, fx.scale));
								atomicAdd(&acc_raw[i2], to_fix(q10 * a0, fx.scale)); atomicAdd(&acc_raw[kPEpb + i2], to_fix(q10 * a1, fx.scale));
								atomicAdd(&acc_raw[i2 + 1u], to_fix(q11 * a0, fx.scale)); atomicAdd(&acc_raw[kPEpb + i2 + 1u], to_fix(q11 * a1, fx.scale));
							} else {
								atomicAdd(&acc[i0], (double)(q00 * a0)); atomicAdd(&acc[kPEpb + i0], (double)(q00 * a1));
								atomicAdd(&acc[i0 + 1u], (double)(q01 * a0)); atomicAdd(&acc[kPEpb + i0 + 1u], (double)(q01 * a1));
								atomicAdd(&acc[i2], (double)(q10 * a0)); atomicAdd(&acc[kPEpb + i2], (double)(q10 * a1));
								atomicAdd(&acc[i2 + 1u], (double)(q11 * a0)); atomicAdd(&acc[kPEpb + i2 + 1u], (double)(q11 * a1));
							}
							continue;
						}
						const bool pair = fl == 3u;
						const float wl = pair ? 1.0f - w : 1.0f, wh = pair ? w : 1.0f;
						if (fix) {
							if (fl & 1u) { atomicAdd(&acc_raw[i0], to_fix(wl * a0, fx.scale)); atomicAdd(&acc_raw[kPEpb + i0], to_fix(wl * a1, fx.scale)); }
							if (fl & 2u) { atomicAdd(&acc_raw[i1], to_fix(wh * a0, fx.scale)); atomicAdd(&acc_raw[kPEpb + i1], to_fix(wh * a1, fx.scale)); }
						} else {
							if (fl & 1u) { atomicAdd(&acc[i0], (double)(wl * a0)); atomicAdd(&acc[kPEpb + i0], (double)(wl * a1)); }
							if (fl & 2u) { atomicAdd(&acc[i1], (double)(wh * a0)); atomicAdd(&acc[kPEpb + i1], (double)(wh * a1)); }
						}
					}
			}
		}
	}
	__syncthreads();

	auto value = [&](uint32_t t) -> float {
		return fix ? (float)((double)(long long)acc_raw[t] * fx.inv) : (float)acc[t];
	};
	// flush: the only workgroup of a bucket adds its slice to dL/dparam itself; replicas store fp32 partial tables that
	// k_pair_reduce adds in replica order (no global atomic anywhere)
	if (R > 1) {
		float *mine = partial + (size_t)blockIdx.x * kPLds;
		for (uint32_t t = threadIdx.x; t < kPLds; t += kPAccThreads) mine[t] = value(t);
		return;
	}
	constexpr int kFlush = 8;
	for (uint32_t tb = threadIdx.x; tb < kPLds; tb += kPAccThreads * kFlush) {
		float *p[kFlush];
		float v[kFlush], old[kFlush];
#pragma unroll
		for (int k = 0; k < kFlush; ++k) {
			const uint32_t t = tb + (uint32_t)k * kPAccThreads;
			p[k] = (t < kPLds) ? pair_target(L, epb, plan.lg, foff0, b, t, dparam, (out_half & 1u) != 0) : nullptr;
			v[k] = p[k] ? value(t) : 0.0f;
		}
		const bool half_out = (out_half & 1u) != 0, assign = (out_half & 2u) != 0;     // assign: dparam holds no previous value
#pragma unroll
		for (int k = 0; k < kFlush; ++k) old[k] = (p[k] && !assign) ? pair_ld(p[k], half_out) : 0.0f;
#pragma unroll
		for (int k = 0; k < kFlush; ++k) if (p[k]) pair_st(p[k], old[k] + v[k], half_out);
	}
}

// ---- levels without records: (x, dL_dy) -> LDS accumulators of one bucket, one replica per share of the points ----
// Same per-update arithmetic as stage A + stage B ((g * w_other) * {1 - w_pair, w_pair}, fixed point at the call's
// scale), so with FIX the result is the record path's, bit for bit, for inputs whose lanes stage A does not merge.
template <bool FIX, bool SECOND = false>
__global__ __launch_bounds__(kPAccThreads, 8) void k_pair_direct(DirectPlan dp, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n,
                                                                 int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                                 const float *__restrict__ g, int64_t g_sn, int64_t g_se,
                                                                 const uint32_t *__restrict__ gmax, float *__restrict__ partial,
                                                                 const float *__restrict__ vin_) {
	extern __shared__ __attribute__((aligned(16))) unsigned long long acc_raw[];   // [2][2^lg]
	__shared__ uint32_t queue[kPAccThreads / 64][128];                              // per wave: points waiting for the full arithmetic
	double *acc = reinterpret_cast<double *>(acc_raw);
	const uint32_t r = blockIdx.x, fb = blockIdx.y;
	uint32_t e = 0;
	while (e + 1 < dp.n && dp.bucket_base[e + 1] <= fb) ++e;
	const uint32_t b = fb - dp.bucket_base[e], q = dp.qmap[e];
	const uint32_t level = meta_level_of(md, q);
	const Lvl L = load_level(md, level);
	const uint32_t kPEpb = 1u << dp.lg, kPLds = 2u << dp.lg;
	PairFix fx = {1.0, 1.0, false};
	if constexpr (FIX) fx = pair_fix(gmax, dp.sum_log2);
	const bool fix = FIX && fx.on;
	for (uint32_t t = threadIdx.x; t < kPLds; t += kPAccThreads) acc_raw[t] = 0ull;
	__syncthreads();
	if ((int32_t)level <= max_level) {
		const uint32_t p_lo = r * dp.pts_per_rep;
		auto process = [&](uint32_t i) {
			float xp[3];
#pragma unroll
			for (int d = 0; d < 3; ++d) xp[d] = x[(size_t)i * 3 + d];
			const float g0 = g[(int64_t)i * g_sn + (int64_t)(q * 2) * g_se], g1 = g[(int64_t)i * g_sn + (int64_t)(q * 2 + 1) * g_se];
			uint32_t hdr[4], bkt[4], cell[3];
			constexpr int kW = SECOND ? 3 : 0;
			float A[4][2], wp[SECOND ? 4 : 1], vin[3] = {0.0f, 0.0f, 0.0f};
			bool valid;
			if constexpr (SECOND) {
#pragma unroll
				for (int d = 0; d < 3; ++d) vin[d] = vin_[(size_t)i * 3 + d];
			}
			uint32_t qmask;
			float wy_unused;
			pair_records<SECOND>(L, dp.shift[e], dp.epb[e], dp.lg, xp, g0, g1, smooth != 0, hdr, bkt, A, wp, cell, dp.nb[e], valid, vin, qmask, wy_unused, false);
			float lo[4][2], hi[4][2];
#pragma unroll
			for (int m = 0; m < 4; ++m)
#pragma unroll
				for (int f = 0; f < 2; ++f) { lo[m][f] = (1.0f - wp[m & kW]) * A[m][f]; hi[m][f] = wp[m & kW] * A[m][f]; }
			// coherent inputs (samples along a ray or a Morton curve sit in one cell of these coarse levels for many consecutive points):
			// all 64 lanes of an LDS atomic would hit the same accumulator and serialise (no measures: 206 us for 1.67 M ordered points
			// against 53 us for random ones).  Rounds 3-5 summed the lanes of a run into its head with 96 ds_bpermute per wave (163 us);
			// round 6: when >= 8 lanes continue their neighbour's cell, every lane walks its 16 updates in ITS OWN order instead
			// (below) -- 81 us, random points untouched (tools/exp_pair_direct_order.py)
			bool rotate = false;
			{
				const uint32_t ln = threadIdx.x & 63u;
				auto prev = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); };
				bool same = ln > 0;
#pragma unroll
				for (int d = 0; d < 3; ++d) same = same && (prev(cell[d]) == cell[d]);
				same = same && (prev(1u) != 0u);
				rotate = (uint32_t)__popcll(__ballot(same)) >= dp.merge_min;
			}
			if (!valid) return;
			if (rotate) {
				// pair (step + lane) mod 4, entry and feature order by lane bits 2 and 3: the lanes of one atomic that share a cell spread
				// over its 16 accumulators
				const uint32_t ln = threadIdx.x & 63u;
				const bool sw = (ln & 4u) != 0u, fs = (ln & 8u) != 0u;
#pragma unroll
				for (uint32_t step = 0; step < 4u; ++step) {
					const uint32_t m = (step + ln) & 3u;
					auto sel = [&](auto a0, auto a1, auto a2, auto a3) { return m == 0u ? a0 : (m == 1u ? a1 : (m == 2u ? a2 : a3)); };
					if (sel(bkt[0], bkt[1], bkt[2], bkt[3]) != b) continue;
					const uint32_t hd = sel(hdr[0], hdr[1], hdr[2], hdr[3]);
					const float l0 = sel(lo[0][0], lo[1][0], lo[2][0], lo[3][0]), l1 = sel(lo[0][1], lo[1][1], lo[2][1], lo[3][1]);
					const float h0 = sel(hi[0][0], hi[1][0], hi[2][0], hi[3][0]), h1 = sel(hi[0][1], hi[1][1], hi[2][1], hi[3][1]);
					const uint32_t i0 = hd & 8191u, i1 = (hd >> 13) & 8191u;
					const uint32_t ea = sw ? i1 : i0, eb = sw ? i0 : i1;
					const float a0 = sw ? h0 : l0, a1 = sw ? h1 : l1, b0 = sw ? l0 : h0, b1 = sw ? l1 : h1;
					const uint32_t fa = fs ? kPEpb : 0u, fb2 = fs ? 0u : kPEpb;
					if (fix) {
						atomicAdd(&acc_raw[fa + ea], to_fix(fs ? a1 : a0, fx.scale)); atomicAdd(&acc_raw[fb2 + ea], to_fix(fs ? a0 : a1, fx.scale));
						atomicAdd(&acc_raw[fa + eb], to_fix(fs ? b1 : b0, fx.scale)); atomicAdd(&acc_raw[fb2 + eb], to_fix(fs ? b0 : b1, fx.scale));
					} else {
						atomicAdd(&acc[fa + ea], (double)(fs ? a1 : a0)); atomicAdd(&acc[fb2 + ea], (double)(fs ? a0 : a1));
						atomicAdd(&acc[fa + eb], (double)(fs ? b1 : b0)); atomicAdd(&acc[fb2 + eb], (double)(fs ? b0 : b1));
					}
				}
				return;
			}
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				if (bkt[m] != b) continue;
				const uint32_t i0 = hdr[m] & 8191u, i1 = (hdr[m] >> 13) & 8191u;
				if (fix) {
					atomicAdd(&acc_raw[i0], to_fix(lo[m][0], fx.scale)); atomicAdd(&acc_raw[kPEpb + i0], to_fix(lo[m][1], fx.scale));
					atomicAdd(&acc_raw[i1], to_fix(hi[m][0], fx.scale)); atomicAdd(&acc_raw[kPEpb + i1], to_fix(hi[m][1], fx.scale));
				} else {
					atomicAdd(&acc[i0], (double)lo[m][0]); atomicAdd(&acc[kPEpb + i0], (double)lo[m][1]);
					atomicAdd(&acc[i1], (double)hi[m][0]); atomicAdd(&acc[kPEpb + i1], (double)hi[m][1]);
				}
			}
		};
		// Replica r takes every R-th block of 1024 points, not one contiguous share (round 6): ordered inputs (samples along rays, or along a
		// Morton curve) put a contiguous share into ONE bucket of a multi-bucket level, so one of its nb workgroups did all the work and the
		// others only scanned -- 185 us for the three direct levels of the full loop against 28 us on random points.  (Sums are exact in
		// fixed point: which replica adds a point does not change the result.)
		const uint32_t p_step = dp.R * (uint32_t)kPAccThreads, p_hi = n;
		(void)p_lo;
		if (dp.nb[e] == 1) {
			for (uint32_t i = r * (uint32_t)kPAccThreads + threadIdx.x; i < p_hi; i += p_step) process(i);
		} else {
			// several buckets: most points touch one of them, so a wave first finds the points that concern ITS bucket
			// (cell location and bucket indices only), queues them, and runs the full update arithmetic on dense waves of
			// queued points -- the per-point cost of a pass drops from ~250 to ~60 instructions for the others
			const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
			uint32_t *qw = queue[wave];
			uint32_t qn = 0;
			for (uint32_t i0 = r * (uint32_t)kPAccThreads + wave * 64u; i0 < p_hi; i0 += p_step) {
				const uint32_t i = i0 + lane;
				bool match = false;
				if (i < p_hi) {
					float xp[3];
#pragma unroll
					for (int d = 0; d < 3; ++d) xp[d] = x[(size_t)i * 3 + d];
					uint32_t bk[4];
					pair_buckets(L, dp.shift[e], dp.lg, xp, smooth != 0, bk);
					match = bk[0] == b || bk[1] == b || bk[2] == b || bk[3] == b;
				}
				const unsigned long long mm = __ballot(match);
				if (match) qw[qn + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = i;
				qn += (uint32_t)__popcll(mm);
				__builtin_amdgcn_wave_barrier();
				if (qn >= 64u) {
					process(qw[lane]);
					const uint32_t rest = qn - 64u;
					const uint32_t t = lane < rest ? qw[64u + lane] : 0u;
					__builtin_amdgcn_wave_barrier();
					if (lane < rest) qw[lane] = t;
					qn = rest;
					__builtin_amdgcn_wave_barrier();
				}
			}
			if (lane < qn) process(qw[lane]);
		}
	}
	__syncthreads();
	float *mine = partial + ((size_t)fb * dp.R + r) * kPLds;
	for (uint32_t t = threadIdx.x; t < kPLds; t += kPAccThreads)
		mine[t] = fix ? (float)((double)(long long)acc_raw[t] * fx.inv) : (float)acc[t];
}

// dL/dparam slice of a direct bucket = (assign) or += the sum of its replicas' tables, replica 0 first
__global__ __launch_bounds__(kPAccThreads) void k_pair_direct_reduce(DirectPlan dp, const nr3d_lotd_meta_t *__restrict__ md,
                                                                     const float *__restrict__ partial, float *__restrict__ dparam,
                                                                     uint32_t out_half) {
	const uint32_t fb = blockIdx.x;
	const bool half_out = (out_half & 1u) != 0, assign = (out_half & 2u) != 0;
	uint32_t e = 0;
	while (e + 1 < dp.n && dp.bucket_base[e + 1] <= fb) ++e;
	const uint32_t b = fb - dp.bucket_base[e], q = dp.qmap[e];
	const Lvl L = load_level(md, meta_level_of(md, q));
	const uint32_t foff0 = meta_cnt_of(md, q) * 2u, kPLds = 2u << dp.lg;
	const uint32_t t = blockIdx.y * kPAccThreads + threadIdx.x;
	if (t >= kPLds) return;
	float *p = pair_target(L, dp.epb[e], dp.lg, foff0, b, t, dparam, half_out);
	if (!p) return;
	const float *part0 = partial + (size_t)fb * dp.R * kPLds + t;
	float sum = 0.0f;
	uint32_t r0 = 0;
	for (; r0 + 8 <= dp.R; r0 += 8) {
		float v[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) v[j] = part0[(size_t)(r0 + j) * kPLds];
#pragma unroll
		for (int j = 0; j < 8; ++j) sum += v[j];
	}
	for (; r0 < dp.R; ++r0) sum += part0[(size_t)r0 * kPLds];
	pair_st(p, (assign ? 0.0f : pair_ld(p, half_out)) + sum, half_out);
}

constexpr uint32_t kRedRows = 4;
// dL/dparam slice of a replicated bucket += sum of the replicas' partial tables, replica 0 first
// ... and, in the same launch (blockIdx.x >= NB), the direct levels' buckets = sum of their replicas' tables (what
// k_pair_direct_reduce does on its own when no level takes the record path)
__global__ __launch_bounds__(kPAccThreads) void k_pair_reduce(PairPlan plan, const nr3d_lotd_meta_t *__restrict__ md,
                                                              const uint32_t *__restrict__ rep_g,
                                                              const uint32_t *__restrict__ item_start,
                                                              const float *__restrict__ partial, float *__restrict__ dparam,
                                                              uint32_t out_half, uint32_t NB, DirectPlan dp,
                                                              const float *__restrict__ dpart) {
	const bool half_out = (out_half & 1u) != 0, assign = (out_half & 2u) != 0;
	if (blockIdx.x >= NB) {
		// a direct bucket has ~100 replicas to add per element: one 256-element row per workgroup (kRedRows workgroups in x per
		// bucket and grid row), not kRedRows rows in sequence like the sparsely replicated buckets below
		const uint32_t fb = (blockIdx.x - NB) / kRedRows, sub = (blockIdx.x - NB) % kRedRows;
		uint32_t e = 0;
		while (e + 1 < dp.n && dp.bucket_base[e + 1] <= fb) ++e;
		const uint32_t b = fb - dp.bucket_base[e], q = dp.qmap[e];
		const Lvl L = load_level(md, meta_level_of(md, q));
		const uint32_t foff0 = meta_cnt_of(md, q) * 2u, kPLds = 2u << dp.lg;
		const uint32_t t = (blockIdx.y * kRedRows + sub) * kPAccThreads + threadIdx.x;
		if (t >= kPLds) return;
		float *p = pair_target(L, dp.epb[e], dp.lg, foff0, b, t, dparam, half_out);
		if (!p) return;
		const float *part0 = dpart + (size_t)fb * dp.R * kPLds + t;
		float sum = 0.0f;
		uint32_t r0 = 0;
		for (; r0 + 8 <= dp.R; r0 += 8) {
			float v[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) v[j] = part0[(size_t)(r0 + j) * kPLds];
#pragma unroll
			for (int j = 0; j < 8; ++j) sum += v[j];
		}
		for (; r0 < dp.R; ++r0) sum += part0[(size_t)r0 * kPLds];
		pair_st(p, (assign ? 0.0f : pair_ld(p, half_out)) + sum, half_out);
		return;
	}
	const uint32_t fb = blockIdx.x;
	const uint32_t R = rep_g[fb];
	if (R == 1 || (R == 0 && !assign)) return;           // assign mode: a bucket without records still has to be written (zeros)
	uint32_t q = 0;
	while (q + 1 < plan.n_pseudo && plan.bucket_base[q + 1] <= fb) ++q;
	const uint32_t b = fb - plan.bucket_base[q], qg = plan.qmap[q];
	const Lvl L = load_level(md, meta_level_of(md, qg));
	const uint32_t foff0 = meta_cnt_of(md, qg) * 2u;
	const uint32_t kPLds = 2u << plan.lg;
	const float *part0 = partial + (size_t)item_start[fb] * kPLds;
	// kRedRows rows of 1024 accumulators per workgroup: most buckets have one work item and leave at the top, so fewer,
	// fatter workgroups cut the dispatch time of this nearly empty launch
	for (uint32_t row = 0; row < kRedRows; ++row) {
		const uint32_t t = (blockIdx.y * kRedRows + row) * kPAccThreads + threadIdx.x;
		if (t >= kPLds) return;
		float *p = pair_target(L, plan.epb[q], plan.lg, foff0, b, t, dparam, half_out);
		if (!p) continue;
		float sum = 0.0f;
		uint32_t r0 = 0;
		for (; r0 + 8 <= R; r0 += 8) {
			float v[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) v[j] = part0[(size_t)(r0 + j) * kPLds + t];
#pragma unroll
			for (int j = 0; j < 8; ++j) sum += v[j];
		}
		for (; r0 < R; ++r0) sum += part0[(size_t)r0 * kPLds + t];
		pair_st(p, (assign ? 0.0f : pair_ld(p, half_out)) + sum, half_out);
	}
}

// -------------------------------------------------------------------------------------------------
// Host side
// -------------------------------------------------------------------------------------------------
static bool pair_enabled() { return opt::on(NR3D_OPT_LOTD_PAIR); }

bool pair_applies(const nr3d_lotd_meta_t *m) {
	if (!pair_enabled()) return false;
	if (!m || m->n_dims_to_encode != 3 || m->n_feat_per_pseudo_lvl != 2 || !m->c_hash_only) return false;
	if (m->n_pseudo_levels > (uint32_t)kPMaxLv) return false;
	for (uint32_t l = 0; l < m->n_levels; ++l) {
		const nr3d_lotd_level_t &L = m->levels[l];
		const uint32_t kPEpb = 1u << pair_lg();
		if (L.type == NR3D_LOD_Dense) {
			if (L.res[2] > kPEpb) return false;
			const uint64_t rows = (uint64_t)L.res[0] * L.res[1];
			uint32_t lg = 0;
			while ((2ull << lg) * L.res[2] <= kPEpb) ++lg;
			if (((rows + (1ull << lg) - 1) >> lg) > kPMaxNb) return false;
		} else if (L.type == NR3D_LOD_Hash) {
			if (L.size <= kPEpb) continue;
			if ((L.size & (L.size - 1u)) != 0u || L.res[0] > kPEpb) return false;
			if ((L.size >> pair_lg()) > kPMaxNb) return false;
		} else {
			return false;
		}
	}
	return true;
}

// (Rounds 2-3 carried an all-levels stage A here -- k_pair_bin_all / nr3d_lotd_bwd_fused: dL_dy read once, dL/dx folded in --
// measured 6 % slower on the backward of configs[1] (1551 vs 457 + 953 us per 2^22 points; 119 VGPRs for the row + 132 KB of
// LDS left one workgroup per CU against two of k_pair_bin).  Removed in round 4; DESIGN.md section 5b keeps the numbers.)

static void pair_plan(const nr3d_lotd_meta_t *m, uint32_t n_chunk, int32_t min_level, int32_t max_level, PairPlan &plan,
                      uint64_t &offs_words, uint64_t skip_pseudo = 0) {
	plan.n_blk = div_up(n_chunk, pair_bp());
	plan.cap = pair_bp() * 4u;
	plan.lg = pair_lg();
	plan.sum_log2 = 3;
	while ((1ull << (plan.sum_log2 - 3)) < n_chunk) ++plan.sum_log2;
	const uint32_t kPEpb = 1u << plan.lg;
	uint32_t nq = 0;
	uint64_t base = 0;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
		const int32_t lv = (int32_t)m->map_levels[q];
		if (lv < min_level || lv > max_level || ((skip_pseudo >> q) & 1ull)) continue;
		const nr3d_lotd_level_t &L = m->levels[lv];
		uint32_t nb, epb, sh;
		if (L.type == NR3D_LOD_Dense) {
			sh = 0;
			while ((2ull << sh) * L.res[2] <= kPEpb) ++sh;
			epb = (1u << sh) * L.res[2];
			nb = (uint32_t)((((uint64_t)L.res[0] * L.res[1]) + (1ull << sh) - 1) >> sh);
		} else {
			sh = plan.lg; epb = kPEpb;
			nb = L.size <= kPEpb ? 1u : (L.size >> plan.lg);
		}
		plan.qmap[nq] = q; plan.nb[nq] = nb; plan.epb[nq] = epb; plan.shift[nq] = sh;
		plan.bucket_base[nq] = nq ? plan.bucket_base[nq - 1] + plan.nb[nq - 1] : 0u;
		plan.offs_base[nq] = (uint32_t)base;
		base += (uint64_t)(nb + 1) * plan.n_blk;
		++nq;
	}
	plan.n_pseudo = nq;
	plan.bucket_base[nq] = nq ? plan.bucket_base[nq - 1] + plan.nb[nq - 1] : 0u;
	offs_words = base;
}

// NR3D_OPT_PAIR_DIRECT = 0: every level goes through records
static bool pair_direct_enabled() { return opt::on(NR3D_OPT_PAIR_DIRECT); }
// the pseudo levels of `full` with few buckets become the direct plan; returns the mask of those levels (0: none -- also
// when nothing would be left for the record path, whose stage A carries the fixed-point scale)
static uint64_t pair_direct_plan(const PairPlan &full, uint32_t n, DirectPlan &dp) {
	dp.n = 0; dp.lg = full.lg; dp.sum_log2 = full.sum_log2; dp.R = 1; dp.pts_per_rep = n;
	dp.merge_min = (uint32_t)NR3D_XOPT(PAIR_DIRECT_ROTATE, 8);
	dp.bucket_base[0] = 0;
	if (!pair_direct_enabled()) return 0;
	uint64_t mask = 0;
	const int64_t nb_x = NR3D_XOPT(PAIR_DIRECT_NB, kDirectNb);  // buckets up to which a level goes direct (knob: experiments build)
	const uint32_t nb_lim = nb_x < 0 ? 0u : (uint32_t)nb_x;
	for (uint32_t ql = 0; ql < full.n_pseudo && dp.n < kDirectMaxLv; ++ql) {
		if (full.nb[ql] > nb_lim || full.qmap[ql] >= 64u) continue;
		if (dp.bucket_base[dp.n] + full.nb[ql] > kDirectMaxWg) break;      // pair_layout reserves kDirectMaxWg partial tables
		const uint32_t e = dp.n++;
		dp.qmap[e] = full.qmap[ql]; dp.nb[e] = full.nb[ql]; dp.epb[e] = full.epb[ql]; dp.shift[e] = full.shift[ql];
		dp.bucket_base[e + 1] = dp.bucket_base[e] + full.nb[ql];
		mask |= 1ull << full.qmap[ql];
	}
	if (dp.n == 0 || dp.n == full.n_pseudo) { dp.n = 0; return 0; }
	const uint32_t nbk = dp.bucket_base[dp.n];
	uint32_t R = kDirectMaxWg / nbk;                         // ~ two workgroups per CU over all direct buckets
	R = R > 128u ? 128u : R;
	const uint32_t by_points = div_up(n, 2048u);              // >= 2048 points per replica
	R = R > by_points ? by_points : R;
	dp.R = R < 1u ? 1u : R;
	dp.pts_per_rep = div_up(n, dp.R);
	return mask;
}

uint32_t pair_direct_levels(const nr3d_lotd_meta_t *m, uint32_t n_points) {
	PairPlan plan;
	uint64_t ow;
	pair_plan(m, n_points, 0, 0x7fffffff, plan, ow);
	DirectPlan dp;
	pair_direct_plan(plan, n_points, dp);
	return dp.n;
}

// workspace needs of the pair path for a chunk of n_chunk points (regions as in lotd_bin.hip's layout)
void pair_layout(const nr3d_lotd_meta_t *m, uint32_t n_chunk, uint32_t units, uint64_t &rec_bytes, uint64_t &offs_bytes,
                 uint64_t &plan_bytes, uint64_t &part_bytes) {
	PairPlan plan;
	uint64_t ow;
	pair_plan(m, n_chunk, 0, 0x7fffffff, plan, ow);
	const uint32_t NB = plan.bucket_base[plan.n_pseudo];
	rec_bytes = (uint64_t)plan.n_pseudo * plan.n_blk * plan.cap * 16;
	offs_bytes = ((ow * 4 + 255) / 256) * 256;
	plan_bytes = (((uint64_t)NB * 3 + 4) * 4 + 255) / 256 * 256;
	part_bytes = (uint64_t)(pair_units() + NB + kDirectMaxWg) * (2u << plan.lg) * 4;   // the pair path's own item count, not `units`; + k_pair_direct
}

void launch_plan_items(uint32_t NB, uint32_t n_blk, uint32_t units, const uint32_t *tot, uint32_t *rep, uint32_t *item_start,
                       hipStream_t st);      // lotd_bin.hip

// one chunk of points: dL_dy given feature-major or with any strides (g_sn, g_se)
int pair_chunk(const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t n, const float *x, const float *g,
               int64_t g_sn, int64_t g_se, int32_t min_level, int32_t max_level, uint32_t units, float *dparam, uint32_t out_flags,
               void *rec, uint32_t *offs, uint32_t *plan_buf, float *partial, hipStream_t st, const float *vin) {
	// vin != NULL: second order (d(dL/dx)/dparam for dL_ddLdx = vin, [n, 3])
	PairPlan pl;
	uint64_t ow;
	pair_plan(meta, n, min_level, max_level, pl, ow);
	if (pl.n_pseudo == 0) return 0;
	// levels with a handful of buckets leave the record path (k_pair_direct); `pl` keeps the others
	DirectPlan dp;
	dp.n = 0;
	const uint32_t NB_full = pl.bucket_base[pl.n_pseudo];
	{
		const uint64_t skip = pair_direct_plan(pl, n, dp);
		if (skip) pair_plan(meta, n, min_level, max_level, pl, ow, skip);
	}
	uint32_t nb_max = 0;
	for (uint32_t q = 0; q < pl.n_pseudo; ++q) nb_max = nb_max > pl.nb[q] ? nb_max : pl.nb[q];
	const uint32_t NB = pl.bucket_base[pl.n_pseudo];
	uint32_t *tot = plan_buf, *rep = plan_buf + NB, *item_start = plan_buf + 2 * (size_t)NB;
	uint32_t *gmax = plan_buf + 3 * (size_t)NB + 2;                  // two spare words of the plan region: gmax | ticket
	units = pair_units();
	const size_t bin_lds_max = (size_t)1024 * 4 * 16 + (size_t)(kPMaxNb + 2) * 8;
	static bool attr_set_dev[64] = {};
	int dev_id = 0;
	NR3D_HIP_CHECK(hipGetDevice(&dev_id));
	if (!attr_set_dev[dev_id & 63]) {
#ifdef NR3D_EXPERIMENTS
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_bin<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds_max));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_bin<768>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds_max));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_bin<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds_max));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_bin<768, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds_max));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_accum<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_accum<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
#endif
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_bin<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds_max));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_bin<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds_max));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_direct<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_direct<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_direct<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_direct<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_accum<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_pair_accum<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPLdsMax * 8));
		attr_set_dev[dev_id & 63] = true;
	}
	NR3D_HIP_CHECK(hipMemsetAsync(gmax, 0, 2 * sizeof(uint32_t), st));      // gmax | ticket of k_pair_plan
	const uint32_t bp = pair_bp();
	const size_t bin_lds = (size_t)bp * 4 * 16 + (size_t)(nb_max + 1) * 4;     // stage | hist
	// first order: the Dense levels' records in quad form (NR3D_PAIR_QUAD=0: pair records only)
	const uint32_t quad_on = (!vin && pair_quad_enabled()) ? 1u : 0u;
#define NR3D_PAIR_BIN(BP) if (vin) NR3D_PAIR_BIN_(BP, true); else NR3D_PAIR_BIN_(BP, false)
#define NR3D_PAIR_BIN_(BP, SEC) hipLaunchKernelGGL((k_pair_bin<BP, SEC>), dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n, max_level, \
	meta->interpolation_type, x, g, g_sn, g_se, (u32x4 *)rec, offs, gmax, dp, vin, quad_on)
	{
		prof::Scope ps(NR3D_PROF_LOTD_BIN, st);
#ifdef NR3D_EXPERIMENTS
		if (bp == 512) { NR3D_PAIR_BIN(512); } else if (bp == 768) { NR3D_PAIR_BIN(768); } else
#endif
		{ NR3D_PAIR_BIN(1024); }
	}
#undef NR3D_PAIR_BIN
#undef NR3D_PAIR_BIN_
	// levels that skip the records: straight from (x, dL_dy) into LDS; needs stage A's gmax only, so it runs before stage B
	// and its replicas are summed together with stage B's (one launch less)
	float *dpart = partial + (size_t)(units + NB_full) * (2u << pl.lg);       // behind stage B's partial tables
	const uint32_t nbk_direct = dp.n ? dp.bucket_base[dp.n] : 0u;
	if (dp.n) {
		prof::Scope ps(NR3D_PROF_LOTD_DIRECT, st);
		auto direct = [&](auto kern) {
			hipLaunchKernelGGL(kern, dim3(dp.R, nbk_direct), dim3(kPAccThreads), (size_t)(16u << pl.lg), st, dp, md, n, max_level,
			                   meta->interpolation_type, x, g, g_sn, g_se, gmax, dpart, vin);
		};
		if (vin) { if (pair_fixed()) direct(k_pair_direct<true, true>); else direct(k_pair_direct<false, true>); }
		else     { if (pair_fixed()) direct(k_pair_direct<true, false>); else direct(k_pair_direct<false, false>); }
	}
	if (pl.n_pseudo == 0) {
		if (dp.n)
			hipLaunchKernelGGL(k_pair_direct_reduce, dim3(nbk_direct, (2u << pl.lg) / kPAccThreads), dim3(kPAccThreads), 0, st, dp, md, dpart,
			                   dparam, out_flags);
		NR3D_LAUNCH_CHECK();
		return 0;
	}
	hipLaunchKernelGGL(k_pair_plan, dim3(div_up(NB, 16)), dim3(1024), 0, st, pl, offs, units, tot, rep, item_start, gmax + 1);
#define NR3D_PAIR_ACC(U, F) hipLaunchKernelGGL((k_pair_accum<U, F>), dim3(units + NB), dim3(kPAccThreads), (size_t)(16u << pl.lg), st, pl, md, \
	(const u32x4 *)rec, offs, rep, item_start, gmax, partial, dparam NR3D_DBG_ARG(NR3D_XOPT(PAIR_DEBUG, 0)), out_flags)
	{
		prof::Scope ps(NR3D_PROF_LOTD_ACCUM, st);
#ifdef NR3D_EXPERIMENTS
		if (pair_unroll() == 8) { if (pair_fixed()) NR3D_PAIR_ACC(8, true); else NR3D_PAIR_ACC(8, false); } else
#endif
		{ if (pair_fixed()) NR3D_PAIR_ACC(4, true); else NR3D_PAIR_ACC(4, false); }
	}
#undef NR3D_PAIR_ACC
	// the replicated buckets of the record path and the direct levels' buckets are summed in ONE launch
	hipLaunchKernelGGL(k_pair_reduce, dim3(NB + nbk_direct * kRedRows, div_up((2u << pl.lg) / kPAccThreads, kRedRows)), dim3(kPAccThreads), 0, st, pl, md, rep,
	                   item_start, partial, dparam, out_flags, NB, dp, dpart);
	NR3D_LAUNCH_CHECK();
	return 0;
}

}  // namespace lotd
}  // namespace nr3d
