// nr3d_lib_amd/csrc/lotd_bin.hip -- dL/dparam and d(dL/dx)/dparam for Dense/Hash LoTD levels WITHOUT global atomics.
//
// Why: measured on MI355X (tools/ubench_mem.hip, tools/ubench_lds.hip, profiles/) every global atomic flavour
// (f32/u32/u64/f64/pk_f16, any scope, any table size) saturates at ~21-27 G atomics/s chip-wide, so the
// reference's scatter (2^20 pts x 16 levels x 8 corners x 2 feats = 268 M atomicAdds,
// kernel_lod_hashonly_backward_grid, lotd_hash_only.h:380-470) costs >= 12.8 ms.  LDS ds_add_f64 sustains
// ~1.3-1.5 T/s at random addresses (ds_add_f32 only ~0.2 T/s).  So the scatter is reorganised as
//
//   stage A  k_bin    one workgroup = 512 points x 1 pseudo-level: compute the 2^D corner updates, counting-sort
//                     them by table BUCKET (a bucket = the slice of the level's table whose fp64 accumulators fit
//                     one CU's LDS: 16384/G entries) inside the workgroup and write them as AoS records
//                     {local entry, value[G]} plus the per-bucket start offsets;
//   stage B  k_accum  one workgroup = one bucket (x an optional replica that takes a share of the point blocks):
//                     stream that bucket's records (contiguous runs), accumulate into a 128 KiB fp64 LDS table with
//                     ds_add_f64, then add the slice to dL/dparam once (replicas store fp32 partial tables that
//                     k_reduce_partials sums in a fixed order: there is no global atomic on this path).
//
// Both stages are pure streaming (12 B per corner update written once and read once); the fp64 accumulation makes
// the result independent of the update order up to the final f64->f32 rounding.
#include "lotd_vm.h"
#include "lotd_sorted.h"
#include <stdlib.h>
#include <type_traits>

#ifndef NR3D_BIN_LDS_KB
#define NR3D_BIN_LDS_KB 72     // stage-A record staging per workgroup (2 workgroups per CU)
#define NR3D_BIN_MAX_BP 512
#endif
namespace nr3d {
namespace lotd {

constexpr int kAccThreads = 1024;         // stage-B workgroup
constexpr int kLdsDoubles = 16384;        // 128 KiB of fp64 accumulators
constexpr int kRedSlots = 4;              // accumulator slots per thread in k_reduce_partials
#ifndef NR3D_BALLOT_RANK_BUCKETS
#define NR3D_BALLOT_RANK_BUCKETS 4
#endif
constexpr uint32_t kBallotRankBuckets = NR3D_BALLOT_RANK_BUCKETS;   // stage A ranks through ballots up to this many buckets
                                                                    // (NGP config, backward ms: 0: 0.880, 4: 0.862, 12: 0.899)
constexpr int kMaxPlanLevels = 64;        // pseudo levels handled by the binned path
constexpr uint32_t kMaxBuckets = 8192;    // per pseudo level
constexpr int kBinLdsDyn = NR3D_BIN_LDS_KB * 1024 + (int)(kMaxBuckets + 1) * 4;   // dynamic LDS of a stage-A workgroup: record stage + bucket histogram

#define DISPATCH_DG_BIN(D_, G_, ...)                                                 \
	do {                                                                             \
		const uint32_t _d = (D_), _g = (G_);                                         \
		if (_d == 2 && _g == 2) { constexpr int D = 2, G = 2; __VA_ARGS__; }         \
		else if (_d == 2 && _g == 4) { constexpr int D = 2, G = 4; __VA_ARGS__; }    \
		else if (_d == 2 && _g == 8) { constexpr int D = 2, G = 8; __VA_ARGS__; }    \
		else if (_d == 3 && _g == 2) { constexpr int D = 3, G = 2; __VA_ARGS__; }    \
		else if (_d == 3 && _g == 4) { constexpr int D = 3, G = 4; __VA_ARGS__; }    \
		else if (_d == 3 && _g == 8) { constexpr int D = 3, G = 8; __VA_ARGS__; }    \
		else if (_d == 4 && _g == 2) { constexpr int D = 4, G = 2; __VA_ARGS__; }    \
		else if (_d == 4 && _g == 4) { constexpr int D = 4, G = 4; __VA_ARGS__; }    \
		else return ::nr3d::fail("LoTD::bwd (binned): 4-D pseudo levels of 8 features run as their width-4 regrouping (stage_a_meta)"); \
	} while (0)

// Record classes: NR = capacity in records per (point, pseudo level).  Updates that hit the same table entry from
// several corners of one point are summed in registers first (CP: 2 entries per line, VM: 4 per plane + 2 per line...).
//   Dense/Hash 2^D | CP, CPfast 2D | VecZMatXoY 4 + 2 | NPlaneMul, NPlaneSum D * 2^(D-1) (4-D: 32) | VM 3 * (4 + 2) = 18
__host__ __device__ inline uint32_t rec_count(uint32_t type, uint32_t D, bool forest = false) {
	if (forest) {
		// forest of blocks (3-D): every corner may belong to another block, so updates are kept per corner
		// (8 corners x table entries one corner touches) instead of being summed per distinct entry
		switch (type) {
		case NR3D_LOD_Dense: case NR3D_LOD_Hash: return 8u;
		case NR3D_LOD_CP: case NR3D_LOD_NPlaneMul: return 24u;
		case NR3D_LOD_VectorMatrix: return 48u;
		default: return 0u;                          // CPfast / NPlaneSum / VecZMatXoY have no forest form (lotd_forest.h:265-301)
		}
	}
	switch (type) {
	case NR3D_LOD_Dense: case NR3D_LOD_Hash: return 1u << D;
	case NR3D_LOD_CP: case NR3D_LOD_CPfast: return 2u * D;
	case NR3D_LOD_NPlaneSum: return D >= 3 ? D << (D - 1) : 0u;
	case NR3D_LOD_VecZMatXoY: return D == 3 ? 6u : 0u;
	case NR3D_LOD_NPlaneMul: return D << (D - 1);
	case NR3D_LOD_VectorMatrix: return D == 3 ? 18u : 0u;
	default: return 0u;
	}
}
__host__ __device__ static inline uint32_t rec_class(uint32_t n_rec) {
	return n_rec == 0 ? 0u : n_rec <= 8 ? 8u : n_rec <= 16 ? 16u : n_rec <= 24 ? 24u : n_rec <= 32 ? 32u : 48u;
}

struct BinPlan {
	uint32_t qmap[kMaxPlanLevels];        // pseudo level of the meta behind local index q (levels of ONE record class)
	uint32_t nb[kMaxPlanLevels];          // buckets per pseudo level
	uint32_t bucket_base[kMaxPlanLevels + 1];   // flat bucket index of the level's first bucket ([n_pseudo] = total)
	uint32_t offs_base[kMaxPlanLevels];   // start of this pseudo level's offset table (in uint32 units)
	uint32_t epb_log2;                    // log2(entries per bucket)
	uint32_t n_blk;                       // stage-A workgroups along the points of the current chunk
	uint32_t n_pseudo;                    // pseudo levels in this plan
	uint32_t cap;                         // records per slot = stage-A points per workgroup x NR
	uint32_t n_batches;                   // batched params: a level's entry space is n_batches x level size (else 1)
};

// -------------------------------------------------------------------------------------------------
// [n, E] row-major -> [E, n] (32x32 LDS tiles), so the level-major stage A reads its G columns coalesced
// -------------------------------------------------------------------------------------------------
template <typename ST>
__global__ __launch_bounds__(256) void k_transpose(uint32_t n, uint32_t E, const ST *__restrict__ src, int64_t s_sn,
                                                   int64_t s_se, float *__restrict__ dst) {
	__shared__ float tile[32][33];
	const uint32_t i0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
	const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
	for (int k = 0; k < 32; k += 8) {
		const uint32_t i = i0 + ty + k, e = e0 + tx;
		tile[ty + k][tx] = (i < n && e < E) ? to_f32<ST>(src[(int64_t)i * s_sn + (int64_t)e * s_se]) : 0.0f;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < 32; k += 8) {
		const uint32_t e = e0 + ty + k, i = i0 + tx;
		if (i < n && e < E) dst[(size_t)e * n + i] = tile[tx][ty + k];
	}
}

// The same copy for a CONTIGUOUS row-major source (what autograd hands over) with E <= 120 (round 4): a workgroup takes 128
// consecutive points -- one contiguous 128 E-element piece of memory, read with every lane on consecutive addresses (the 32 x 32 tiles
// above read 128-byte pieces, 72 bytes in the last tile column of configs[3]'s E = 50) -- and writes 512 contiguous bytes per feature.
// configs[3] (E = 50, 2^22 points): 0.51 -> 0.3x ms per dL/dparam pass that starts from a row-major dL_dy.
constexpr uint32_t kTrPts = 128;
template <typename ST>
__global__ __launch_bounds__(256) void k_transpose_rows(uint32_t n, uint32_t E, const ST *__restrict__ src, float *__restrict__ dst) {
	extern __shared__ __attribute__((aligned(16))) float tr_tile[];             // [E][kTrPts + 1]
	const uint32_t i0 = blockIdx.x * kTrPts;
	const uint32_t np = min(kTrPts, n - i0), total = np * E;
	const ST *base = src + (size_t)i0 * E;
	for (uint32_t idx = threadIdx.x; idx < total; idx += 256u) {
		const uint32_t i = idx / E, e = idx - i * E;
		tr_tile[e * (kTrPts + 1u) + i] = to_f32<ST>(base[idx]);
	}
	__syncthreads();
	for (uint32_t o = threadIdx.x; o < E * kTrPts; o += 256u) {
		const uint32_t e = o / kTrPts, i = o % kTrPts;
		if (i < np) dst[(size_t)e * n + i0 + i] = tr_tile[e * (kTrPts + 1u) + i];
	}
}
// [n, E] -> [E, n]: the row kernel when the source is contiguous and E <= 120, the tile kernel otherwise
template <typename ST>
static void launch_transpose(uint32_t n, uint32_t E, const ST *src, int64_t s_sn, int64_t s_se, float *dst, hipStream_t st) {
	if (s_se == 1 && s_sn == (int64_t)E && E <= 120u && E >= 2u)          // 120 x 129 floats = 62 KB: inside the default dynamic-LDS limit
		hipLaunchKernelGGL(k_transpose_rows<ST>, dim3(div_up(n, kTrPts)), dim3(256), (size_t)E * (kTrPts + 1u) * 4u, st, n, E, src, dst);
	else
		hipLaunchKernelGGL(k_transpose<ST>, dim3(div_up(n, 32), div_up(E, 32)), dim3(256), 0, st, n, E, src, s_sn, s_se, dst);
}

// -------------------------------------------------------------------------------------------------
// Stage A: bin the parameter updates of BP points x 1 pseudo level by bucket.
// BP (points = threads per workgroup) is the largest power of two whose record staging area
// ((1 + G) words x BP x NR records) fits in 72 KiB of LDS, so 2 workgroups share a CU.
// -------------------------------------------------------------------------------------------------
template <int G, int NR>
struct BinCfg {
	static constexpr int raw = (NR3D_BIN_LDS_KB * 1024 / 4) / ((1 + G) * NR);
	static constexpr int BP = (raw >= 1024 && NR3D_BIN_MAX_BP >= 1024) ? 1024 : raw >= 512 ? 512 : raw >= 256 ? 256 : raw >= 128 ? 128 : raw >= 64 ? 64 : 32;
	static constexpr uint32_t cap = (uint32_t)BP * NR;          // records per (pseudo level, point block)
};
static uint32_t bin_points(uint32_t G, uint32_t NR) {
	const uint32_t raw = ((uint32_t)NR3D_BIN_LDS_KB * 1024u / 4u) / ((1u + G) * NR);
	return (raw >= 1024 && NR3D_BIN_MAX_BP >= 1024) ? 1024 : raw >= 512 ? 512 : raw >= 256 ? 256 : raw >= 128 ? 128 : raw >= 64 ? 64 : 32;
}

// Product of D factor tables T_d; corner k uses slot s_d(k) of table d.
//   CP:        T_d = line d, slot = bit d of k (NS = 2)
//   NPlaneMul: T_j = plane spanning all dims but D-1-j, slot = the other dims' bits (NS = 2^(D-1))
// d/dT_gd[slot] = sum over the corners with that slot of (grad * w) * prod_{j != gd} T_j[s_j(k)]   (corner_scatter order)
template <int D, int G, int NR, int NS, bool CP, typename TB>
__device__ __forceinline__ uint32_t emit_product(const Lvl &L, const Cell<D> &c, const float (&w)[1 << D], const float (&grad)[G],
                                                 TB grid, uint32_t foff, uint32_t (&ent)[NR],
                                                 float (&val)[NR][G]) {
	constexpr uint32_t C = 1u << D;
	float tv[D][NS][G];
	uint32_t te[D][NS];
	auto slot = [](uint32_t k, int d) -> uint32_t { return CP ? ((k >> d) & 1u) : drop_bit<D>(k, D - 1 - d); };
#pragma unroll
	for (int d = 0; d < D; ++d)
#pragma unroll
		for (uint32_t sl = 0; sl < (uint32_t)NS; ++sl) {
			uint32_t e;
			if (CP) e = entry_line<D>(L, d, c.g[d] + sl);
			else {
				uint32_t p[D];
				corner_pos<D>(c, insert_zero(sl, D - 1 - d), p);
				e = entry_nplane_mul<D>(L, d, p);
			}
			te[d][sl] = e;
#pragma unroll
			for (int f = 0; f < G; ++f) tv[d][sl][f] = grid[e * L.F + foff + f];
		}
#pragma unroll
	for (int gd = 0; gd < D; ++gd)
#pragma unroll
		for (uint32_t sl = 0; sl < (uint32_t)NS; ++sl) {
			float acc[G];
#pragma unroll
			for (int f = 0; f < G; ++f) acc[f] = 0.0f;
#pragma unroll
			for (uint32_t k = 0; k < C; ++k) {
				if (slot(k, gd) != sl) continue;
#pragma unroll
				for (int f = 0; f < G; ++f) {
					float cur = grad[f] * w[k];
#pragma unroll
					for (int j = 0; j < D; ++j)
						if (j != gd) cur *= tv[j][slot(k, j)][f];
					acc[f] += cur;
				}
			}
			ent[gd * NS + sl] = te[gd][sl];
#pragma unroll
			for (int f = 0; f < G; ++f) val[gd * NS + sl][f] = acc[f];
		}
	return (uint32_t)(D * NS);
}

// VM (3-D): per component d a plane over the two dims != d (4 slots) times a line along d (2 slots);
// VecZMatXoY is the d = 2 component alone.  Record index is compile-time: VM d * 6 + {0..3 plane, 4..5 line}.
template <int G, int NR, bool VM, typename TB>
__device__ __forceinline__ uint32_t emit_plane_line(const Lvl &L, const Cell<3> &c, const float (&w)[8], const float (&grad)[G],
                                                    TB grid, uint32_t foff, uint32_t (&ent)[NR],
                                                    float (&val)[NR][G]) {
#pragma unroll
	for (int d = VM ? 0 : 2; d < 3; ++d) {
		const int base = VM ? d * 6 : 0;
		float pv[4][G], lv[2][G];
		uint32_t pe[4], le[2];
#pragma unroll
		for (uint32_t m = 0; m < 4; ++m) {
			uint32_t p[3];
			corner_pos<3>(c, insert_zero(m, d), p);
			if (VM) { uint32_t pl[3], ln[3]; entry_vm(L, p, pl, ln); pe[m] = pl[d]; if (m == 0) le[0] = ln[d]; }
			else { pe[m] = L.res[2] + p[1] + p[0] * L.res[0]; if (m == 0) le[0] = p[2]; }
#pragma unroll
			for (int f = 0; f < G; ++f) pv[m][f] = grid[pe[m] * L.F + foff + f];
		}
		le[1] = le[0] + 1u;
#pragma unroll
		for (uint32_t sl = 0; sl < 2; ++sl)
#pragma unroll
			for (int f = 0; f < G; ++f) lv[sl][f] = grid[le[sl] * L.F + foff + f];
#pragma unroll
		for (uint32_t m = 0; m < 4; ++m) {            // plane slots: the two corners that differ along d
			ent[base + m] = pe[m];
			const uint32_t k0 = insert_zero(m, d), k1 = k0 | (1u << d);
#pragma unroll
			for (int f = 0; f < G; ++f) val[base + m][f] = (grad[f] * w[k0]) * lv[0][f] + (grad[f] * w[k1]) * lv[1][f];
		}
#pragma unroll
		for (uint32_t sl = 0; sl < 2; ++sl) {         // line slots: the four corners with bit d == sl
			ent[base + 4 + sl] = le[sl];
#pragma unroll
			for (int f = 0; f < G; ++f) {
				float acc = 0.0f;
#pragma unroll
				for (uint32_t m = 0; m < 4; ++m) acc += (grad[f] * w[insert_zero(m, d) | (sl << d)]) * pv[m][f];
				val[base + 4 + sl][f] = acc;
			}
		}
	}
	return VM ? 18u : 6u;
}

// The parameter updates of one (point, pseudo level): table entry + G values each.  `w[k]` is the weight of corner k
// (first order: the interpolation weight; second order: the combined d/dx weight), `grad` = dL/dy of the G features,
// `grid` = params + level offset, `foff` = first feature of this pseudo level inside the level's entries.
// Same per-update arithmetic as corner_scatter() (lotd_device.h); contributions to one entry are added in corner order.
// CPfast: y = prod_d lerp(line_d) evaluated per line, not per corner (lotd_encoding.h:653-705, second order :970-1038);
// same arithmetic as k_bwd_dparam's CPfast branch, one record per line entry.
template <int D, int G, int NR, bool SECOND, typename TB>
__device__ __forceinline__ uint32_t emit_cpfast(const Lvl &L, const Cell<D> &c, const float (&grad)[G], const float (&vin)[D],
                                                TB grid, uint32_t foff, uint32_t (&ent)[NR],
                                                float (&val)[NR][G]) {
	float lv[D][2][G];
#pragma unroll
	for (int d = 0; d < D; ++d)
#pragma unroll
		for (uint32_t sl = 0; sl < 2; ++sl) {
			const uint32_t e = entry_line<D>(L, d, c.g[d] + sl);
			ent[2 * d + sl] = e;
#pragma unroll
			for (int f = 0; f < G; ++f) lv[d][sl][f] = grid[e * L.F + foff + f];
		}
#pragma unroll
	for (int ld = 0; ld < D; ++ld)
#pragma unroll
		for (int f = 0; f < G; ++f) {
			if (!SECOND) {
				float gl = grad[f];
#pragma unroll
				for (int d = 0; d < D; ++d)
					if (d != ld) gl *= __fmaf_rn(c.w[d], lv[d][1][f], (1.0f - c.w[d]) * lv[d][0][f]);
				val[2 * ld][f] = gl * (1.0f - c.w[ld]);
				val[2 * ld + 1][f] = gl * c.w[ld];
			} else {
				float acc_l = 0.0f, acc_r = 0.0f;
#pragma unroll
				for (int gd = 0; gd < D; ++gd) {
					float gl = grad[f] * c.sc[gd] * vin[gd] * c.dw[gd];
					const float wl = (ld != gd) ? (1.0f - c.w[ld]) : -1.0f;
					const float wr = (ld != gd) ? c.w[ld] : 1.0f;
#pragma unroll
					for (int d = 0; d < D; ++d) {
						if (d == ld) continue;
						const float nl = (d != gd) ? (1.0f - c.w[d]) : -1.0f;
						const float nr = (d != gd) ? c.w[d] : 1.0f;
						gl *= __fmaf_rn(nr, lv[d][1][f], nl * lv[d][0][f]);
					}
					acc_l = __fmaf_rn(gl, wl, acc_l);
					acc_r = __fmaf_rn(gl, wr, acc_r);
				}
				val[2 * ld][f] = acc_l;
				val[2 * ld + 1][f] = acc_r;
			}
		}
	return 2u * D;
}

// NPlaneSum: sum over the D axis "planes" of a (D-1)-linear interpolation; one record per plane corner, D * 2^(D-1) in all
// (same weights as k_bwd_dparam's NPlaneSum branch, reference lotd_encoding.h:268-351 and its backward).
template <int D, int G, int NR, bool SECOND>
__device__ __forceinline__ uint32_t emit_nplane_sum(const Lvl &L, const Cell<D> &c, const float (&grad)[G], const float (&a)[D],
                                                    uint32_t (&ent)[NR], float (&val)[NR][G]) {
	constexpr uint32_t NC = 1u << (D - 1);
#pragma unroll
	for (uint32_t jd = 0; jd < (uint32_t)D; ++jd)
#pragma unroll
		for (uint32_t k = 0; k < NC; ++k) {
			uint32_t pp[D];
#pragma unroll
			for (int d2 = 0; d2 < D - 1; ++d2) {
				const int d3 = (uint32_t)d2 >= jd ? d2 + 1 : d2;
				pp[d2] = c.g[d3] + ((k >> d2) & 1u);
			}
			float w;
			if (!SECOND) {
				w = 1.0f;
#pragma unroll
				for (int d2 = 0; d2 < D - 1; ++d2) {
					const int d3 = (uint32_t)d2 >= jd ? d2 + 1 : d2;
					w *= ((k >> d2) & 1u) ? c.w[d3] : (1.0f - c.w[d3]);
				}
			} else {
				w = 0.0f;
#pragma unroll
				for (int g2 = 0; g2 < D - 1; ++g2) {
					const int g3 = (uint32_t)g2 >= jd ? g2 + 1 : g2;
					float t = 0.0f;
#pragma unroll
					for (int d = 0; d < D; ++d) if (d == g3) t = a[d];
#pragma unroll
					for (int d2 = 0; d2 < D - 1; ++d2) {
						if (d2 == g2) continue;
						const int d3 = (uint32_t)d2 >= jd ? d2 + 1 : d2;
						t *= ((k >> d2) & 1u) ? c.w[d3] : (1.0f - c.w[d3]);
					}
					w += ((k >> g2) & 1u) ? t : -t;
				}
			}
			ent[jd * NC + k] = entry_nplane_sum<D>(L, jd, pp);
#pragma unroll
			for (int f = 0; f < G; ++f) val[jd * NC + k][f] = grad[f] * w;
		}
	return (uint32_t)D * NC;
}

template <int D, int G, int NR, bool DH, bool SECOND, typename TB>
__device__ __forceinline__ uint32_t emit_updates(const Lvl &L, const Cell<D> &c, const float (&w)[1 << D], const float (&grad)[G],
                                                 const float (&a)[D], const float (&vin)[D],
                                                 TB grid, uint32_t foff, uint32_t (&ent)[NR],
                                                 float (&val)[NR][G]) {
	constexpr uint32_t C = 1u << D;
	if (DH || L.type == NR3D_LOD_Dense || L.type == NR3D_LOD_Hash) {     // DH: the meta has no other level types
		if constexpr (NR >= (int)C) {
#pragma unroll
			for (uint32_t k = 0; k < C; ++k) {
				uint32_t p[D];
				corner_pos<D>(c, k, p);
				ent[k] = (L.type == NR3D_LOD_Dense) ? entry_dense<D>(L, p) : entry_hash<D>(L, p);
#pragma unroll
				for (int f = 0; f < G; ++f) val[k][f] = grad[f] * w[k];
			}
			return C;
		}
	} else if (L.type == NR3D_LOD_CP) {
		if constexpr (!DH && NR >= 2 * D) return emit_product<D, G, NR, 2, true>(L, c, w, grad, grid, foff, ent, val);
	} else if (L.type == NR3D_LOD_NPlaneMul) {
		if constexpr (!DH && NR >= D * (1 << (D - 1)))
			return emit_product<D, G, NR, (1 << (D - 1)), false>(L, c, w, grad, grid, foff, ent, val);
	} else if (L.type == NR3D_LOD_VectorMatrix) {
		if constexpr (!DH && D == 3 && NR >= 18) return emit_plane_line<G, NR, true>(L, c, w, grad, grid, foff, ent, val);
	} else if (L.type == NR3D_LOD_VecZMatXoY) {
		if constexpr (!DH && D == 3 && NR >= 6) return emit_plane_line<G, NR, false>(L, c, w, grad, grid, foff, ent, val);
	} else if (L.type == NR3D_LOD_CPfast) {
		if constexpr (!DH && NR >= 2 * D) return emit_cpfast<D, G, NR, SECOND>(L, c, grad, vin, grid, foff, ent, val);
	} else if (L.type == NR3D_LOD_NPlaneSum) {
		if constexpr (!DH && NR >= D * (1 << (D - 1))) return emit_nplane_sum<D, G, NR, SECOND>(L, c, grad, a, ent, val);
	}
	return 0;
}

// forest (FO): the updates of the 8 corners, each in the tables of the block that OWNS the corner (virtual entry = owner
// block * level size + entry).  A corner nobody owns becomes zero updates of the point's own block.  Product types
// multiply by the other factors read from the owner's tables -- the per-corner arithmetic of corner_scatter()
// (lotd_device.h; reference lotd_forest.h:415-636), one record per (corner, table entry): Dense / Hash 8, CP / NPlaneMul 24,
// VM 48.
template <int G, int NR, typename PT>
__device__ __forceinline__ uint32_t emit_forest(const ForestDev &fo, const Batch &ba, const Lvl &L, const Cell<3> &c,
                                                const float (&w)[8], const float (&grad)[G], const int (&bk)[3], uint32_t bi,
                                                const PT *__restrict__ params, uint32_t foff, uint32_t (&ent)[NR],
                                                float (&val)[NR][G]) {
	constexpr uint32_t E = NR / 8;                     // records per corner of this class
	const uint32_t need = rec_count(L.type, 3, true) / 8u;
	if (need != E) return 0;                           // levels are grouped by record class: a class-NR launch only sees its own types
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) {
		uint32_t p[3], pl[3], owner = bi;
		corner_pos<3>(c, k, p);
		const bool ok = resolve_block(fo, L, bk, p, pl, owner);
		const uint32_t vbase = (ok ? owner : bi) * L.size;
		float wg[G];
#pragma unroll
		for (int f = 0; f < G; ++f) wg[f] = ok ? grad[f] * w[k] : 0.0f;
		if (L.type == NR3D_LOD_Dense || L.type == NR3D_LOD_Hash) {
			const uint32_t e = (L.type == NR3D_LOD_Dense) ? entry_dense<3>(L, pl) : entry_hash<3>(L, pl);
			ent[k * E] = ok ? vbase + e : vbase;
#pragma unroll
			for (int f = 0; f < G; ++f) val[k * E][f] = wg[f];
			continue;
		}
		if constexpr (E >= 2) {
			const auto grid = make_tab(params + ((ba.offsets ? (uint32_t)ba.offsets[ok ? owner : bi] : (ok ? owner : bi) * ba.n_params) + L.off));
			if constexpr (E >= 3) {
				if (L.type == NR3D_LOD_CP || L.type == NR3D_LOD_NPlaneMul) {
					uint32_t idx[3];
#pragma unroll
					for (int d = 0; d < 3; ++d) idx[d] = (L.type == NR3D_LOD_CP) ? entry_line<3>(L, d, pl[d]) : entry_nplane_mul<3>(L, d, pl);
#pragma unroll
					for (int gd = 0; gd < 3; ++gd) {
						ent[k * E + gd] = ok ? vbase + idx[gd] : vbase;
#pragma unroll
						for (int f = 0; f < G; ++f) {
							float cur = wg[f];
							if (ok) {
#pragma unroll
								for (int j = 0; j < 3; ++j) if (j != gd) cur *= grid[idx[j] * L.F + foff + f];
							}
							val[k * E + gd][f] = cur;
						}
					}
					continue;
				}
				if constexpr (E >= 6) {
					if (L.type == NR3D_LOD_VectorMatrix) {
						uint32_t ple[3], lne[3];
						entry_vm(L, pl, ple, lne);
#pragma unroll
						for (int d = 0; d < 3; ++d) {
							ent[k * E + 2 * d] = ok ? vbase + ple[d] : vbase;
							ent[k * E + 2 * d + 1] = ok ? vbase + lne[d] : vbase;
#pragma unroll
							for (int f = 0; f < G; ++f) {
								val[k * E + 2 * d][f] = ok ? wg[f] * grid[lne[d] * L.F + foff + f] : 0.0f;
								val[k * E + 2 * d + 1][f] = ok ? wg[f] * grid[ple[d] * L.F + foff + f] : 0.0f;
							}
						}
						continue;
					}
				}
			}
		}
	}
	return (uint32_t)NR;
}

// SPLIT: threads per point.  1: a thread forms all (up to NR) records of its point.  3 (3-D VM levels, class 24): thread
// (component d, point) forms the six records of component d -- the one-thread form holds 24 record slots in ~130 registers and
// its 72 KB stage allows two 256-thread workgroups per CU: 2 waves per SIMD against ~1 us gathers (profiles/
// r03k_c4_counters_before_cp16.txt: VALU 30 %, LDS 17 %, L2 requests 55 % of their ceilings).  Threads are component-major
// (a wave holds 64 consecutive points of ONE component), so the coherent-lane merge sees what it saw.
// LINES (round 4, VM levels with SPLIT == 3): the two LINE updates of a component are not records -- a VM level's three line
// tables are sum_d R_d entries (configs[3]: 288 / 576 / 1152), G fp64 accumulators each fit LDS many times over -- they are added
// to a per-workgroup LDS table with ds_add_f64, exactly as k_cp_direct does for CP lines; only the component's four PLANE updates
// are sorted and written out: 12 instead of 18 records per (point, pseudo level).  For the table to be worth flushing the
// workgroup must see many points, so the grid is (replicas, pseudo levels) and a workgroup walks the point blocks r, r + R, ...
// (the record slots stay indexed by block: stage B does not change); its line table goes to `lines_out` as fp32 and
// k_vm_lines_reduce adds the replicas in order.
constexpr uint32_t kLinesRecPerPoint = 12;
template <int D, int G, bool SECOND, int NR, bool DH, bool FO, typename PT, int SPLIT = 1, bool LINES = false>
__device__ __forceinline__ void bin_body(const BinPlan &plan, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n,
                                         int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                         const float *__restrict__ vin_, const float *__restrict__ g,
                                         int64_t g_sn, int64_t g_se, const PT *__restrict__ params,
                                         const Batch &ba, const ForestDev &fo, uint32_t *__restrict__ rec,
                                         uint32_t *__restrict__ offs_g, float *__restrict__ lines_out = nullptr,
                                         uint32_t line_stride = 0) {
	constexpr int BP = BinCfg<G, NR>::BP;                 // points per workgroup
	constexpr int kThr = BP * SPLIT;                      // threads per workgroup
	constexpr int NRT = NR / SPLIT;                       // record slots per thread
	constexpr uint32_t cap = LINES ? (uint32_t)BP * kLinesRecPerPoint : BinCfg<G, NR>::cap;
	constexpr int C = 1 << D;
	static_assert(!LINES || (SPLIT == 3 && D == 3), "line tables in LDS: the three-threads-per-point VM stage A");
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // [LINES: fp64 line table |] stage[(1+G)*cap] | hist[nb + 1]
	__shared__ uint64_t scan_lds[kThr / 64 > 0 ? kThr / 64 : 1];
	const uint32_t ql = blockIdx.y;
	const uint32_t q = plan.qmap[ql];
	const uint32_t nb = plan.nb[ql];
	const uint32_t level = meta_level_of(md, q);
	const Lvl L = load_level(md, level);
	const uint32_t n_line = LINES ? (L.res[0] + L.res[1] + L.res[2]) * (uint32_t)G : 0u;     // fp64 accumulators of the line table
	double *lines = reinterpret_cast<double *>(smem);
	uint32_t *stage = smem + 2u * (size_t)n_line;
	uint32_t *hist = stage + (size_t)(1 + G) * cap;
	const uint32_t comp = SPLIT == 1 ? 0u : threadIdx.x / (uint32_t)BP;      // wave-uniform (BP is a multiple of 64)
	if (LINES) for (uint32_t t = threadIdx.x; t < n_line; t += kThr) lines[t] = 0.0;
	// (!LINES: exactly one trip, and in a form the compiler folds -- `blk < blockIdx.x + 1` kept the loop, and its invariants in
	// 27 more registers: k_bin_vm3 went from 68 to 95 VGPRs and lost its second workgroup per CU)
#pragma unroll 1
	for (uint32_t blk = blockIdx.x, once = 1u; LINES ? blk < plan.n_blk : once != 0u; blk += gridDim.x, once = 0u) {
	const uint32_t i = blk * BP + (SPLIT == 1 ? threadIdx.x : threadIdx.x % (uint32_t)BP);

	for (uint32_t b = threadIdx.x; b <= nb; b += kThr) hist[b] = 0;
	__syncthreads();

	// batched params (one table set per batch entry): the level's entry space becomes [batch entry][entry]
	uint32_t pbase = 0, bi = 0;
	const bool in_batch = (i < n) && batch_base_index(ba, i, pbase, bi);
	const bool active = in_batch && ((int32_t)level <= max_level);
	uint32_t ent[NRT], rank[NRT], cell[D];
	float val[NRT][G];
	uint32_t n_rec = 0;
#pragma unroll
	for (int d = 0; d < D; ++d) cell[d] = 0xFFFFFFFFu;
#pragma unroll
	for (uint32_t r = 0; r < (uint32_t)NRT; ++r)
#pragma unroll
		for (int f = 0; f < G; ++f) val[r][f] = 0.0f;
	if (active) {
		float xp[D];
#pragma unroll
		for (int d = 0; d < D; ++d) xp[d] = x[(size_t)i * D + d];
		Cell<D> c;
		if constexpr (FO) locate_forest(xp, L, smooth != 0, c); else locate<D>(xp, L, smooth != 0, c);
		// x outside [0, 1] (the Python layer clamps, the C ABI takes what it gets): a cell outside the level would index the tables,
		// the bucket histogram and the record stage out of bounds -- such a point emits nothing (unsigned compares)
		bool inside = true;
		if constexpr (!FO) {
#pragma unroll
			for (int d = 0; d < D; ++d) inside = inside && (c.g[d] + 1u < L.res[d]);
			if (!inside) {                                                  // the emitters below read the tables: give them a cell that exists
#pragma unroll
				for (int d = 0; d < D; ++d) c.g[d] = 0u;
			}
		}
		float grad[G], w[C];
		const uint32_t col0 = meta_col_of(md, q);
#pragma unroll
		for (int f = 0; f < G; ++f) grad[f] = g[(int64_t)i * g_sn + (int64_t)(col0 + f) * g_se];
		float a[D], vin[D];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			vin[d] = SECOND ? vin_[(size_t)i * D + d] : 0.0f;
			a[d] = SECOND ? c.sc[d] * vin[d] * c.dw[d] : 0.0f;
		}
#pragma unroll
		for (uint32_t k = 0; k < (uint32_t)C; ++k) {
			if constexpr (SPLIT == 3) { w[k] = 0.0f; continue; }          // the separable VM emitter takes the Cell itself
			if (!SECOND) {
				w[k] = corner_weight<D>(c, k);
			} else {
				float sum = 0.0f;
#pragma unroll
				for (int d = 0; d < D; ++d) {
					const float t = face_weight<D>(c, k, d, a[d]);
					sum += ((k >> d) & 1u) ? t : -t;
				}
				w[k] = sum;
			}
		}
		if constexpr (FO) {
			int bk[3];
#pragma unroll
			for (int d = 0; d < 3; ++d) bk[d] = fo.block_ks[3 * (size_t)bi + d];
			// a cell whose eight corners lie inside the point's own block (all but a ~6/R fraction) is the plain level shifted by one
			// node: the product types then emit the single-block records -- one per DISTINCT table entry (VM 18, CP 6, NPlaneMul 12)
			// instead of one per corner and factor (48 / 24 / 24); same per-update arithmetic, fewer records to sort, write and add
			bool done = false;
			if constexpr (NR >= 24 && G <= 4) {            // (G = 8: the all-types emitter on top of 48 record slots spills several hundred registers)
				bool interior = L.type == NR3D_LOD_VectorMatrix || L.type == NR3D_LOD_CP || L.type == NR3D_LOD_NPlaneMul;
#pragma unroll
				for (int d = 0; d < 3; ++d) interior = interior && c.g[d] >= 1u && c.g[d] + 1u <= L.res[d];
				if (interior && rec_class(rec_count(L.type, 3, true)) == (uint32_t)NR) {
					Cell<D> cs = c;
#pragma unroll
					for (int d = 0; d < 3; ++d) cs.g[d] -= 1u;
					const uint32_t boff = ba.offsets ? (uint32_t)ba.offsets[bi] : bi * ba.n_params;
					n_rec = emit_updates<D, G, NR, false, SECOND>(L, cs, w, grad, a, vin, make_tab(params + (boff + L.off)), meta_cnt_of(md, q) * G, ent, val);
#pragma unroll
					for (uint32_t r = 0; r < (uint32_t)NRT; ++r) if (r < n_rec) ent[r] += bi * L.size;
					done = true;
				}
			}
			if (!done) n_rec = emit_forest<G, NR>(fo, ba, L, c, w, grad, bk, bi, params, meta_cnt_of(md, q) * G, ent, val);
		} else if constexpr (SPLIT == 3) {
			static_assert(D == 3 && NR == 24 && !DH, "three threads per point: 3-D VM levels (record class 24)");
			if (L.type == NR3D_LOD_VectorMatrix) {
				const auto grid = make_tab(params + (pbase + L.off));
				const uint32_t foff = meta_cnt_of(md, q) * G;
				if (comp == 0) n_rec = emit_vm_component<G, 0, NRT, SECOND>(L, c, a, grad, grid, foff, ent, val);
				else if (comp == 1) n_rec = emit_vm_component<G, 1, NRT, SECOND>(L, c, a, grad, grid, foff, ent, val);
				else n_rec = emit_vm_component<G, 2, NRT, SECOND>(L, c, a, grad, grid, foff, ent, val);
			}
		} else {
			n_rec = emit_updates<D, G, NR, DH, SECOND>(L, c, w, grad, a, vin, make_tab(params + (pbase + L.off)), meta_cnt_of(md, q) * G, ent, val);
		}
#pragma unroll
		for (int d = 0; d < D; ++d) cell[d] = c.g[d];
		if (!inside) n_rec = 0;
	}
	// Coherent inputs (samples along a ray: consecutive points sit in the same cell of a coarse level) would emit the
	// same entries over and over and then collide on the same LDS accumulators in stage B.  Lanes that continue the
	// previous lane's cell are summed into the first lane of their run (segmented wave reduction over the record
	// values) and emit nothing.  Wave-uniform decision: only when at least a quarter of the wave can be merged, so
	// spread-out points pay three shuffles and a ballot.  Measured on the full-loop workload (1.67 M ray samples):
	// stage B 2.08 -> 0.90 ms at unchanged stage-A time; a threshold of three quarters gives almost nothing (1.89 ms).
	{
		const uint32_t lane = threadIdx.x & 63;
		// previous lane's values through DPP wave_shr:1 (a VALU move; ds_bpermute would add to the LDS pipe this
		// kernel is bound by)
		auto prev = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); };
		bool same = active && lane > 0;
#pragma unroll
		for (int d = 0; d < D; ++d) same = same && (prev(cell[d]) == cell[d]);
		same = same && (prev(bi) == bi) && (prev((uint32_t)active) != 0u);
		const unsigned long long cont = __ballot(same);               // bit i: lane i continues lane i-1's run
		if (__popcll(cont) >= 16) {
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				// lanes lane+1 .. lane+off all continue this lane's run  <=>  `off` set bits right above bit `lane`
				const unsigned long long need = (off == 64 ? ~0ull : ((1ull << off) - 1ull));
				const bool take = (lane + off < 64) && (((cont >> (lane + 1)) & need) == need);
#pragma unroll
				for (uint32_t r = 0; r < (uint32_t)NRT; ++r)
#pragma unroll
					for (int f = 0; f < G; ++f) {
						const float t = __shfl_down(val[r][f], off, 64);
						if (take && r < n_rec) val[r][f] += t;
					}
			}
			if (same) n_rec = 0;                                         // merged into the head of the run
		}
	}
	if constexpr (LINES) {
		// records 4 and 5 of a component are its line's two entries: into the LDS table (merged runs: the head lane carries the sum)
		if (active && n_rec == 6u) {
#pragma unroll
			for (uint32_t r = 4; r < 6; ++r)
#pragma unroll
				for (int f = 0; f < G; ++f) atomicAdd(&lines[(size_t)ent[r] * G + f], (double)val[r][f]);
			n_rec = 4u;
		}
	}
	if (nb <= kBallotRankBuckets) {
		// A coarse level's table is one to four buckets: every record of the block would hit the same few histogram
		// counters (LDS atomics on one address serialise).  Rank through ballots instead: per distinct bucket in the wave
		// one atomic by its first lane, the other lanes take their position from the ballot mask.
		const uint32_t lane = threadIdx.x & 63;
#pragma unroll
		for (uint32_t r = 0; r < (uint32_t)NRT; ++r) {
			const bool has = active && r < n_rec;
			uint32_t bkt = 0xFFFFFFFFu;
			if (has) {
				if (!FO) ent[r] += bi * L.size;
				bkt = ent[r] >> plan.epb_log2;
			}
			unsigned long long todo = __ballot(has);
			while (todo) {
				const int leader = __ffsll((long long)todo) - 1;
				const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)bkt, leader);
				const bool mine = has && bkt == v;
				const unsigned long long m = __ballot(mine);
				uint32_t first = 0;
				if ((int)lane == leader) first = atomicAdd(&hist[v], (uint32_t)__popcll(m));
				first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
				if (mine) rank[r] = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
				todo &= ~m;
			}
		}
	} else if (active) {
#pragma unroll
		for (uint32_t r = 0; r < (uint32_t)NRT; ++r)
			if (r < n_rec) {
				if (!FO) ent[r] += bi * L.size;          // forest records already carry their owner block
				rank[r] = atomicAdd(&hist[ent[r] >> plan.epb_log2], 1u);
			}
	}
	__syncthreads();

	// exclusive scan of the bucket histogram (in place); hist[nb] = total
	uint64_t carry = 0;
	for (uint32_t base = 0; base <= nb; base += kThr) {
		const uint32_t b = base + threadIdx.x;
		const uint64_t v = (b < nb) ? hist[b] : 0;
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		uint64_t inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint64_t t = __shfl_up(inc, off, 64);
			if (lane >= off) inc += t;
		}
		if (lane == 63 || (kThr < 64 && lane == kThr - 1)) scan_lds[wave] = inc;
		__syncthreads();
		uint64_t wave_off = 0, tot = 0;
#pragma unroll
		for (int k = 0; k < (kThr + 63) / 64; ++k) { const uint64_t t = scan_lds[k]; if (k < wave) wave_off += t; tot += t; }
		if (b <= nb) hist[b] = (uint32_t)(carry + wave_off + inc - v);
		carry += tot;
		__syncthreads();
	}

	// counting-sort the records into the LDS staging area (AoS: {idx, val[0..G-1]} per record)
	if (active) {
		const uint32_t mask = (1u << plan.epb_log2) - 1u;
#pragma unroll
		for (uint32_t r = 0; r < (uint32_t)NRT; ++r) {
			if (r >= n_rec) continue;
			const uint32_t pos = hist[ent[r] >> plan.epb_log2] + rank[r];
			stage[pos * (1 + G)] = ent[r] & mask;
#pragma unroll
			for (int f = 0; f < G; ++f) stage[pos * (1 + G) + 1 + f] = __float_as_uint(val[r][f]);
		}
	}
	__syncthreads();

	// coalesced 16-byte write-out of the filled prefix (records are (1 + G)-word AoS, so a bucket's run is one
	// contiguous span of the slot and stage B over-fetches at most one cache line per run)
	const uint32_t total = hist[nb];
	uint4 *dst = reinterpret_cast<uint4 *>(rec + ((size_t)ql * plan.n_blk + blk) * (size_t)(1 + G) * cap);
	const uint4 *src = reinterpret_cast<const uint4 *>(stage);
	// written once, read once by the next kernel: non-temporal, so x / dL_dy keep their L2 lines
	for (uint32_t v4 = threadIdx.x; v4 < (total * (1 + G) + 3) / 4; v4 += kThr) {   // tail: <= 3 stale words
		const uint4 t = src[v4];
		uint32_t *d = reinterpret_cast<uint32_t *>(dst + v4);
		__builtin_nontemporal_store(t.x, d); __builtin_nontemporal_store(t.y, d + 1);
		__builtin_nontemporal_store(t.z, d + 2); __builtin_nontemporal_store(t.w, d + 3);
	}
	uint32_t *ob = offs_g + plan.offs_base[ql];
	for (uint32_t b = threadIdx.x; b <= nb; b += kThr) ob[(size_t)b * plan.n_blk + blk] = hist[b];
	if (LINES) __syncthreads();                          // the next block's histogram reset must not overtake these reads
	}
	if constexpr (LINES) {
		__syncthreads();
		float *mine = lines_out + ((size_t)ql * gridDim.x + blockIdx.x) * line_stride;
		for (uint32_t t = threadIdx.x; t < n_line; t += kThr) mine[t] = (float)lines[t];
	}
}

// PT: storage type of the tables the product-type levels read their other factors from (DH instantiations read none)
template <int D, int G, bool SECOND, int NR, bool DH, typename PT = float>
__global__ __launch_bounds__((BinCfg<G, NR>::BP)) void k_bin(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n,
                                                           int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                           const float *__restrict__ vin_, const float *__restrict__ g,
                                                           int64_t g_sn, int64_t g_se, const PT *__restrict__ params,
                                                           Batch ba, uint32_t *__restrict__ rec,
                                                           uint32_t *__restrict__ offs_g) {
	bin_body<D, G, SECOND, NR, DH, false>(plan, md, n, max_level, smooth, x, vin_, g, g_sn, g_se, params, ba, ForestDev{}, rec, offs_g);
}

// stage A of 3-D VM levels (record class 24) with three threads per point (bin_body, SPLIT == 3)
template <int G, bool SECOND, typename PT>
__global__ __launch_bounds__((BinCfg<G, 24>::BP * 3)) void k_bin_vm3(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n,
                                                                     int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                                     const float *__restrict__ vin_, const float *__restrict__ g,
                                                                     int64_t g_sn, int64_t g_se, const PT *__restrict__ params,
                                                                     Batch ba, uint32_t *__restrict__ rec,
                                                                     uint32_t *__restrict__ offs_g) {
	bin_body<3, G, SECOND, 24, false, false, PT, 3>(plan, md, n, max_level, smooth, x, vin_, g, g_sn, g_se, params, ba, ForestDev{}, rec, offs_g);
}

// ... and with the line updates accumulated in LDS (bin_body, LINES): grid = (replicas, pseudo levels), each workgroup walks the
// point blocks r, r + R, ...; 12 plane records per (point, pseudo level), lines_out [pseudo level][replica][line entries x G] fp32
template <int G, bool SECOND, typename PT>
__global__ __launch_bounds__((BinCfg<G, 24>::BP * 3), 6) /* 6 waves per SIMD = two workgroups per CU */ void k_bin_vm3l(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n,
                                                                      int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                                      const float *__restrict__ vin_, const float *__restrict__ g,
                                                                      int64_t g_sn, int64_t g_se, const PT *__restrict__ params,
                                                                      Batch ba, uint32_t *__restrict__ rec,
                                                                      uint32_t *__restrict__ offs_g, float *__restrict__ lines_out,
                                                                      uint32_t line_stride) {
	bin_body<3, G, SECOND, 24, false, false, PT, 3, true>(plan, md, n, max_level, smooth, x, vin_, g, g_sn, g_se, params, ba, ForestDev{}, rec,
	                                                      offs_g, lines_out, line_stride);
}
// dL/dparam of the line tables += the replicas' partial tables, replica 0 first (one thread per (line entry, feature))
template <int G>
__global__ __launch_bounds__(256) void k_vm_lines_reduce(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md, uint32_t R, uint32_t line_stride,
                                                         const float *__restrict__ lines_part, float *__restrict__ dparam) {
	const uint32_t ql = blockIdx.y, q = plan.qmap[ql];
	const Lvl L = load_level(md, meta_level_of(md, q));
	const uint32_t n_line = (L.res[0] + L.res[1] + L.res[2]) * (uint32_t)G, t = blockIdx.x * 256u + threadIdx.x;
	if (t >= n_line) return;
	const float *p0 = lines_part + (size_t)ql * R * line_stride + t;
	float sum = 0.0f;
	for (uint32_t r = 0; r < R; ++r) sum += p0[(size_t)r * line_stride];
	dparam[L.off + (size_t)(t / G) * L.F + meta_cnt_of(md, q) * G + (t % G)] += sum;
}

// stage A for a forest of blocks (3-D): same sort, corner owners resolved through the octree; NR = 8 Dense / Hash,
// 24 CP / NPlaneMul, 48 VM
template <int G, bool SECOND, int NR, typename PT = float>
__global__ __launch_bounds__((BinCfg<G, NR>::BP)) void k_bin_forest(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md,
                                                                   uint32_t n, int32_t max_level, uint32_t smooth,
                                                                   const float *__restrict__ x, const float *__restrict__ vin_,
                                                                   const float *__restrict__ g, int64_t g_sn, int64_t g_se,
                                                                   const PT *__restrict__ params, Batch ba, ForestDev fo,
                                                                   uint32_t *__restrict__ rec, uint32_t *__restrict__ offs_g) {
	bin_body<3, G, SECOND, NR, true, true>(plan, md, n, max_level, smooth, x, vin_, g, g_sn, g_se, params, ba, fo, rec, offs_g);
}

// -------------------------------------------------------------------------------------------------
// Stage-B planning on the device: the number of records per bucket is only known after stage A and depends on where the
// points are (uniform points fill hash buckets evenly; samples along rays through a thin shell do not).  A bucket
// with more than one unit of work (1/1024 of all records, ~4 units per CU) is split into replicas over its point
// blocks; empty buckets get no workgroup at all.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bucket_totals(BinPlan plan, const uint32_t *__restrict__ offs_g,
                                                       uint32_t *__restrict__ tot) {
	const uint32_t fb = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (fb >= plan.bucket_base[plan.n_pseudo]) return;
	uint32_t q = 0;
	while (q + 1 < plan.n_pseudo && plan.bucket_base[q + 1] <= fb) ++q;
	const uint32_t *ob0 = offs_g + plan.offs_base[q] + (size_t)(fb - plan.bucket_base[q]) * plan.n_blk;
	const uint32_t *ob1 = ob0 + plan.n_blk;
	uint32_t sum = 0;
	for (uint32_t blk = lane; blk < plan.n_blk; blk += 64) sum += ob1[blk] - ob0[blk];
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
	if (lane == 0) tot[fb] = sum;
}

__global__ __launch_bounds__(1024) void k_plan_items(uint32_t NB, uint32_t n_blk, uint32_t n_units, const uint32_t *__restrict__ tot,
                                                     uint32_t *__restrict__ rep, uint32_t *__restrict__ item_start) {
	__shared__ uint64_t red[16];
	__shared__ uint64_t carry_s;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint64_t part = 0;
	for (uint32_t fb = threadIdx.x; fb < NB; fb += 1024) part += tot[fb];
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
			const uint64_t t = tot[fb];
			r = t == 0 ? 0u : (uint32_t)((t + unit / 2) / unit);
			if (t != 0 && r < 1) r = 1;
			if (r > n_blk) r = n_blk;
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

void launch_plan_items(uint32_t NB, uint32_t n_blk, uint32_t units, const uint32_t *tot, uint32_t *rep, uint32_t *item_start,
                       hipStream_t st) {
	hipLaunchKernelGGL(k_plan_items, dim3(1), dim3(1024), 0, st, NB, n_blk, units, tot, rep, item_start);
}

// where accumulator slot t (= entry el of the bucket, feature f; t = el * G + f) of bucket b lives in dL/dparam;
// nullptr past the end of the level's (batched) entry space
template <int G>
__device__ __forceinline__ float *flush_target(const BinPlan &plan, const Lvl &L, const Batch &ba, uint32_t foff0, uint32_t b,
                                               uint32_t t, float *__restrict__ dparam, uint32_t &slot) {
	const uint32_t epb = 1u << plan.epb_log2;
	const uint32_t el = t / G, f = t - el * G;
	slot = f * epb + el;                                     // feature-major accumulator / partial layout
	const uint32_t ev = b * epb + el;
	if (t >= epb * G || ev >= plan.n_batches * L.size) return nullptr;
	uint32_t entry = ev, pbase = 0;
	if (plan.n_batches > 1 || ba.offsets) {
		const uint32_t bi = ev / L.size;
		entry = ev - bi * L.size;
		pbase = ba.offsets ? (uint32_t)ba.offsets[bi] : bi * ba.n_params;
	}
	return dparam + (pbase + L.off) + ((size_t)entry * L.F + foff0 + f);
}

// -------------------------------------------------------------------------------------------------
// Stage B: one bucket (x replica) -> fp64 LDS accumulation -> slice of dL/dparam
// -------------------------------------------------------------------------------------------------
template <int D, int G>
__global__ __launch_bounds__(kAccThreads) void k_accum(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md,
                                                       const uint32_t *__restrict__ rec,
                                                       const uint32_t *__restrict__ offs_g,
                                                       const uint32_t *__restrict__ rep_g,
                                                       const uint32_t *__restrict__ item_start, Batch ba,
                                                       float *__restrict__ partial, float *__restrict__ dparam) {
	// accumulators are feature-major (acc[f][entry]): the G atomics of a record spread over all LDS banks
	extern __shared__ __attribute__((aligned(16))) double acc[];      // [kLdsDoubles]
	const uint32_t cap = plan.cap;
	// work item -> (flat bucket, replica): items of a bucket are consecutive, item_start is their exclusive prefix
	const uint32_t NB = plan.bucket_base[plan.n_pseudo];
	const cu32_t istart = (cu32_t)item_start, irep = (cu32_t)rep_g;       // uniform reads -> scalar loads
	if (blockIdx.x >= istart[NB]) return;
	uint32_t lo = 0, hi = NB;                                             // last fb with item_start[fb] <= blockIdx.x
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (istart[mid] <= blockIdx.x) lo = mid; else hi = mid; }
	const uint32_t fb = lo, R = irep[fb], r = blockIdx.x - istart[fb];
	uint32_t q = 0;
	while (q + 1 < plan.n_pseudo && plan.bucket_base[q + 1] <= fb) ++q;   // local level of the bucket
	const uint32_t b = fb - plan.bucket_base[q];
	const uint32_t qg = plan.qmap[q];                     // pseudo level of the meta
	const uint32_t level = meta_level_of(md, qg);
	const Lvl L = load_level(md, level);
	const uint32_t foff0 = meta_cnt_of(md, qg) * G;

	for (uint32_t t = threadIdx.x; t < (uint32_t)kLdsDoubles; t += kAccThreads) acc[t] = 0.0;
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr uint32_t n_waves = kAccThreads / 64;
	const uint32_t blk_lo = (uint32_t)(((uint64_t)plan.n_blk * r) / R), blk_hi = (uint32_t)(((uint64_t)plan.n_blk * (r + 1)) / R);
	const uint32_t *ob0 = offs_g + plan.offs_base[q] + (size_t)b * plan.n_blk;
	const uint32_t *ob1 = ob0 + plan.n_blk;
	const uint32_t *rec_q = rec + (size_t)q * plan.n_blk * (size_t)(1 + G) * cap;
	// The point blocks of this replica are split evenly over the waves.  Per step a wave takes up to 64 of its blocks:
	// lane t fetches the run [start, end) of block blk0 + t (coalesced); the runs are then walked one per
	// wave-instruction (lane < run length active, run bounds broadcast with v_readlane), kUnroll runs in flight.
	// No per-record ownership search: a hash level's runs hold 64 +- 8 records, so a run costs one full and at most
	// one sparse pass.  While workgroups are resident the kernel streams at ~5.4 TB/s with the LDS atomic pipe about
	// 70 % busy; deeper load pipelining and a bucket-major record stream (both tried) do not shorten it, the rest is
	// the workgroup tail (one 128-KiB-LDS workgroup per CU).
	const uint32_t epb = 1u << plan.epb_log2;
	const uint32_t per_wave = (blk_hi - blk_lo + n_waves - 1) / n_waves;
	const uint32_t w_lo = min(blk_lo + wave * per_wave, blk_hi), w_hi = min(w_lo + per_wave, blk_hi);
	constexpr int kUnroll = 4;               // runs in flight per wave (measured: 2: 472, 3: 428, 4: 402, 8: 414, 16: 460 us)
	constexpr int W = 1 + G;
	for (uint32_t blk0 = w_lo; blk0 < w_hi; blk0 += 64) {
		const uint32_t mb = blk0 + lane;
		const uint32_t s_l = (mb < w_hi) ? ob0[mb] : 0u;
		const uint32_t e_l = (mb < w_hi) ? ob1[mb] : 0u;
		const uint32_t n_run = min(64u, w_hi - blk0);
		const uint32_t *rec_b = rec_q + (size_t)blk0 * cap * W;
		for (uint32_t j0 = 0; j0 < n_run; j0 += kUnroll) {
			uint32_t idx[kUnroll];
			float val[kUnroll][G];
			uint32_t rs[kUnroll], rn[kUnroll];
			uint32_t longest = 0;
			// Loads are branch-free (lanes past the run re-read its first record and drop it): a load under an `if`
			// makes the compiler drain ALL outstanding loads at the join.
			for (uint32_t off = 0; off == 0 || off < longest; off += 64) {   // > 1 pass: runs longer than 64 records
#pragma unroll
				for (int u = 0; u < kUnroll; ++u) {
					if (off == 0) {
						const uint32_t j = min(j0 + (uint32_t)u, 63u);
						const uint32_t s_j = __builtin_amdgcn_readlane(s_l, j), e_j = __builtin_amdgcn_readlane(e_l, j);
						rs[u] = s_j + j * cap;
						rn[u] = (j0 + (uint32_t)u < n_run) ? e_j - s_j : 0u;
						longest = max(longest, rn[u]);
					}
					const uint32_t t = off + lane;
					const uint32_t *r_p = rec_b + (size_t)(rs[u] + (t < rn[u] ? t : 0u)) * W;
					idx[u] = __builtin_nontemporal_load(r_p);          // read exactly once: do not keep in L2
#pragma unroll
					for (int f = 0; f < G; ++f) val[u][f] = __uint_as_float(__builtin_nontemporal_load(r_p + 1 + f));
				}
#pragma unroll
				for (int u = 0; u < kUnroll; ++u)
					if (off + lane < rn[u]) {
#pragma unroll
						for (int f = 0; f < G; ++f) atomicAdd(&acc[(uint32_t)f * epb + idx[u]], (double)val[u][f]);
					}
			}
		}
	}
	__syncthreads();

	// flush.  The only workgroup of a bucket adds its slice to dL/dparam itself (batches of kFlush read-modify-writes,
	// all loads issued before the first store: the compiler cannot hoist them, dparam may alias itself).  A replica
	// stores its fp32 partial sums instead (plain coalesced stores); k_reduce_partials adds the replicas of a bucket
	// in a fixed order -- no global atomic anywhere, and the result does not depend on scheduling.
	if (R > 1) {
		float *mine = partial + (size_t)blockIdx.x * kLdsDoubles;
		for (uint32_t t = threadIdx.x; t < (uint32_t)kLdsDoubles; t += kAccThreads) mine[t] = (float)acc[t];
		return;
	}
	constexpr int kFlush = 8;
	for (uint32_t tb = threadIdx.x; tb < epb * G; tb += kAccThreads * kFlush) {
		float *p[kFlush];
		float v[kFlush], old[kFlush];
#pragma unroll
		for (int k = 0; k < kFlush; ++k) {
			uint32_t slot;
			p[k] = flush_target<G>(plan, L, ba, foff0, b, tb + (uint32_t)k * kAccThreads, dparam, slot);
			v[k] = p[k] ? (float)acc[slot] : 0.0f;
		}
#pragma unroll
		for (int k = 0; k < kFlush; ++k) old[k] = p[k] ? *p[k] : 0.0f;
#pragma unroll
		for (int k = 0; k < kFlush; ++k) if (p[k]) *p[k] = old[k] + v[k];
	}
}

// dL/dparam slice of a replicated bucket += sum of the replicas' partials, replica 0 first.  NS accumulator slots per
// thread, NR replicas per pass (NS * NR loads in flight, summed in replica order).
template <int G, int NS, int NR>
__device__ __forceinline__ void reduce_rows(const BinPlan &plan, const Lvl &L, const Batch &ba, uint32_t foff0, uint32_t b,
                                            uint32_t R, uint32_t row0, const float *__restrict__ part0,
                                            float *__restrict__ dparam) {
	float *p[NS];
	uint32_t slot[NS];
	float old[NS], sum[NS];
#pragma unroll
	for (int k = 0; k < NS; ++k) {
		p[k] = flush_target<G>(plan, L, ba, foff0, b, (row0 + (uint32_t)k) * kAccThreads + threadIdx.x, dparam, slot[k]);
		if (!p[k]) slot[k] = 0;
		sum[k] = 0.0f;
	}
#pragma unroll
	for (int k = 0; k < NS; ++k) old[k] = p[k] ? *p[k] : 0.0f;
	for (uint32_t r0 = 0; r0 < R; r0 += NR) {
		float v[NS][NR];
#pragma unroll
		for (int k = 0; k < NS; ++k)
#pragma unroll
			for (int j = 0; j < NR; ++j) v[k][j] = (r0 + j < R) ? part0[(size_t)(r0 + j) * kLdsDoubles + slot[k]] : 0.0f;
#pragma unroll
		for (int k = 0; k < NS; ++k)
#pragma unroll
			for (int j = 0; j < NR; ++j) sum[k] += v[k][j];
	}
#pragma unroll
	for (int k = 0; k < NS; ++k) if (p[k]) *p[k] = old[k] + sum[k];
}

// grid (bucket, kLdsDoubles / kAccThreads rows).  The chain rep -> level -> item_start -> partials -> dparam is latency-
// bound: a bucket with few replicas (the hash levels of a level-bucket call: 640 buckets x 2) gives kRedSlots rows to
// a workgroup (61 -> 32 us), a bucket with many (the coarse Dense levels: R up to the number of point blocks) keeps one
// row per workgroup, where the replica loop is the long part and more workgroups are what helps.
template <int D, int G>
__global__ __launch_bounds__(kAccThreads) void k_reduce_partials(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md,
                                                                 const uint32_t *__restrict__ rep_g,
                                                                 const uint32_t *__restrict__ item_start, Batch ba,
                                                                 const float *__restrict__ partial,
                                                                 float *__restrict__ dparam) {
	const uint32_t fb = blockIdx.x;
	const uint32_t R = rep_g[fb];
	if (R <= 1) return;
	const bool wide = R <= 4;
	if (wide && blockIdx.y >= kLdsDoubles / kAccThreads / kRedSlots) return;
	uint32_t q = 0;
	while (q + 1 < plan.n_pseudo && plan.bucket_base[q + 1] <= fb) ++q;
	const uint32_t b = fb - plan.bucket_base[q], qg = plan.qmap[q];
	const Lvl L = load_level(md, meta_level_of(md, qg));
	const uint32_t foff0 = meta_cnt_of(md, qg) * G;
	const float *part0 = partial + (size_t)item_start[fb] * kLdsDoubles;
	if (wide) reduce_rows<G, kRedSlots, 4>(plan, L, ba, foff0, b, R, blockIdx.y * kRedSlots, part0, dparam);
	else reduce_rows<G, 1, 8>(plan, L, ba, foff0, b, R, blockIdx.y, part0, dparam);
}

// -------------------------------------------------------------------------------------------------
// Host side
// -------------------------------------------------------------------------------------------------
// target number of stage-B work items (a bucket holding more than total / units records is split into replicas).
// Measured on the NGP config, 2^20 points (backward ms): 512: 0.904, 768: 0.884, 900: 0.884, 960: 0.880, 1000: 0.897,
// 1024: 0.901, 1280: 0.913, 2048: 0.937.
static uint32_t work_units() {                  // knob of the experiments build only (options.h)
	const int64_t v = NR3D_XOPT(LOTD_ACC_UNITS, 960);
	return (uint32_t)(v < 256 ? 256 : (v > 8192 ? 8192 : v));
}

// Points per pass of the binned path: the record workspace grows with it (1.6 GB per 2^20 points for the 16-level
// NGP config), larger passes amortise the per-bucket zero / flush better (2^22 points: backward 3.74 -> 3.51 ms with
// 2^22 instead of 2^20).  Default 2^22 (288 GB of HBM per GPU); nr3d_lotd_set_dparam_chunk_log2() overrides it.
static int g_chunk_log2 = 0;
void set_dparam_chunk_log2(int lg) { g_chunk_log2 = lg <= 0 ? 0 : (lg < 10 ? 10 : (lg > 24 ? 24 : lg)); }
static uint32_t chunk_points(uint32_t n) {
	const uint32_t chunk = 1u << (g_chunk_log2 ? g_chunk_log2 : 22);
	return n < chunk ? n : chunk;
}

// -------------------------------------------------------------------------------------------------
// CP levels without records (round 3).  A 3-D CP level is three LINE tables, sum_d R_d entries in all (configs[3]: 2304 /
// 4608 / 9216) -- the gradient table of a few of its 2-feature pseudo levels fits LDS as fp64 (<= 144 KiB).  Through the
// record path such a level costs 6 records x 12 B per point and pseudo level written, sorted (every record lands in the ONE
// bucket the table is: the ranks come from ballots) and read back: configs[3]'s fourteen CP pseudo levels were ~2.3 of the
// 8 ms of a dL/dparam pass.  Here a workgroup takes a share of the points and up to four CONSECUTIVE pseudo levels of one
// level (as many as fit LDS; they share the cell, the weights and the entry indices), forms the six updates per point and
// feature pair and adds them to its LDS table with ds_add_f64 (order-independent to fp64 rounding); the workgroups' tables
// are written out as fp32 and summed by k_cp_reduce in replica order.
// The updates in factored form -- a CP value is a product of per-dim line interpolants I_d = (1 - w_d) T_d[0] + w_d T_d[1]:
//     first order   dT_d[s] += g * w_d(s) * prod_{j != d} I_j
//     second order  dT_d[s] += g * ( a_d sgn(s) prod_{j != d} I_j  +  w_d(s) sum_{j != d} a_j (T_j[1] - T_j[0]) I_k ),   a = s w' v
// (the record path sums the same terms corner by corner -- emit_product; other association, agreement to fp32 rounding).
// NR3D_LOTD_CP_DIRECT=0: CP levels take the record path like every other type.
// -------------------------------------------------------------------------------------------------
constexpr int kCpThreads = 1024;
constexpr uint32_t kCpMaxItems = 32, kCpLdsBytes = 144 * 1024, kCpReplicas = 128, kCpMaxPairs = 4;
struct CpPlan {
	uint32_t n_items, R, pts_per_rep;
	uint32_t q[kCpMaxItems];             // first pseudo level of item k
	uint32_t np[kCpMaxItems];            // ... and how many consecutive pseudo levels (feature pairs) it serves
	uint32_t part_off[kCpMaxItems];      // float offset of its R partial tables inside `partial`
};

template <bool SECOND, typename PT>
__global__ __launch_bounds__(kCpThreads) void k_cp_direct(CpPlan cp, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n, uint32_t smooth,
                                                          const float *__restrict__ x, const float *__restrict__ vin_,
                                                          const float *__restrict__ g, int64_t g_sn, int64_t g_se,
                                                          const PT *__restrict__ params, float *__restrict__ partial, uint32_t opt_fix) {
	extern __shared__ __attribute__((aligned(16))) double cp_acc[];            // [entries][2 np]
	unsigned long long *cp_fix = reinterpret_cast<unsigned long long *>(cp_acc);
	__shared__ uint32_t s_bound[3];                                            // max |dL/dy|, max |line entry|, max |dL_ddLdx| as float bits
	const uint32_t r = blockIdx.x, item = blockIdx.y;
	const uint32_t q = cp.q[item], np = cp.np[item], row = 2u * np;
	const Lvl L = load_level(md, meta_level_of(md, q));
	const uint32_t n_acc = L.size * row;
	for (uint32_t t = threadIdx.x; t < n_acc; t += kCpThreads) cp_acc[t] = 0.0;        // all-zero bits: 0.0 and fixed-point 0 alike
	if (threadIdx.x < 3) s_bound[threadIdx.x] = 0u;
	__syncthreads();
	const uint32_t foff = meta_cnt_of(md, q) * 2u, col0 = meta_col_of(md, q);
	const auto grid = make_tab(params + L.off + foff);
	const bool vec = (tab_addr(grid) % (2u * tab_elt(grid))) == 0u && (L.F & 1u) == 0u;
	const bool quad = (tab_addr(grid) % (4u * tab_elt(grid))) == 0u && (L.F & 3u) == 0u;       // 4 features (2 pairs) per request
	const uint32_t p_lo = r * cp.pts_per_rep, p_hi = min(n, p_lo + cp.pts_per_rep);
	// Fixed-point accumulators (round 4): the kernel was bound by the LDS -- ds_add_f64 costs ~60 cycles per wave instruction,
	// ds_add_u64 ~35 (tools/ubench_lds; the pair path's k_pair_accum has used them since round 2).  They need a bound on the
	// updates BEFORE the pass: this workgroup's own maxima -- |dL/dy| over its points and columns, |T| over its line tables
	// (and |dL_ddLdx| for the second order) -- give |update| <= B = gmax pmax^2 (x 7.5 max_d scale_d vmax, second order: |a_d| <=
	// 1.5 scale_d |v_d|, the bracket <= 5 pmax^2), hence a scale 2^s with |update| 2^s < 2^44 and the sum of all this
	// workgroup's updates < 2^62: exact sums, order-independent, resolution B 2^-44 (an fp32 ulp of B is B 2^-24).  Every
	// workgroup has its own scale (its table leaves as fp32).  Non-finite or zero bounds keep the fp64 accumulators.
	{
		uint32_t gb = 0u, pb = 0u, vb = 0u;
		for (uint32_t i = p_lo + threadIdx.x; i < p_hi; i += kCpThreads) {
			for (uint32_t f = 0; f < row; ++f) gb = max(gb, __float_as_uint(g[(int64_t)i * g_sn + (int64_t)(col0 + f) * g_se]) & 0x7FFFFFFFu);
			if (SECOND)
#pragma unroll
				for (int d = 0; d < 3; ++d) vb = max(vb, __float_as_uint(vin_[(size_t)i * 3 + d]) & 0x7FFFFFFFu);
		}
		for (uint32_t t = threadIdx.x; t < n_acc; t += kCpThreads)
			pb = max(pb, __float_as_uint((float)grid[(t / row) * L.F + (t % row)]) & 0x7FFFFFFFu);
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			gb = max(gb, (uint32_t)__shfl_xor((int)gb, off, 64)); pb = max(pb, (uint32_t)__shfl_xor((int)pb, off, 64));
			vb = max(vb, (uint32_t)__shfl_xor((int)vb, off, 64));
		}
		if ((threadIdx.x & 63u) == 0u) { atomicMax(&s_bound[0], gb); atomicMax(&s_bound[1], pb); atomicMax(&s_bound[2], vb); }
	}
	__syncthreads();
	double fscale = 0.0, finv = 0.0;
	bool fix = false;
	{
		const float gmax = __uint_as_float(s_bound[0]), pmax = __uint_as_float(s_bound[1]), vmax = __uint_as_float(s_bound[2]);
		float B = gmax * pmax * pmax;
		if (SECOND) B *= 7.5f * fmaxf((float)L.res[0], fmaxf((float)L.res[1], (float)L.res[2])) * vmax;
		const uint32_t bb = __float_as_uint(B);
		if (opt_fix && bb != 0u && bb < 0x7F000000u && B >= 1e-30f) {
			const int e = (int)(bb >> 23) - 126;                                       // B < 2^e
			uint32_t lg = 0;
			while ((1u << lg) < cp.pts_per_rep) ++lg;                                  // updates per accumulator <= points of the workgroup
			const int sc = min(62 - (int)lg - 1, 44) - e;
			if (sc > -1000 && sc < 1000) {
				fscale = __longlong_as_double((long long)(sc + 1023) << 52);
				finv = __longlong_as_double((long long)(1023 - sc) << 52);
				fix = true;
			}
		}
	}
	auto add = [&](double *dst, float v) {                                             // block-uniform branch
		if (fix) atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)(__double_as_longlong(__fma_rn((double)v, fscale, 0x1.8p52)) - 0x4338000000000000LL));
		else atomicAdd(dst, (double)v);
	};
#pragma unroll 2
	for (uint32_t i = p_lo + threadIdx.x; i < p_hi; i += kCpThreads) {
		float xp[3], a[3], w0[3], w1[3];
		uint32_t e0[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) xp[d] = x[(size_t)i * 3 + d];
		Cell<3> c;
		locate<3>(xp, L, smooth != 0, c);
		// x outside [0, 1]: a cell outside the level would index the line tables and the LDS table out of bounds -- it adds nothing
		if (!(c.g[0] + 1u < L.res[0] && c.g[1] + 1u < L.res[1] && c.g[2] + 1u < L.res[2])) continue;
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			a[d] = SECOND ? c.sc[d] * vin_[(size_t)i * 3 + d] * c.dw[d] : 0.0f;
			w1[d] = c.w[d]; w0[d] = 1.0f - c.w[d];
			e0[d] = entry_line<3>(L, d, c.g[d]);
		}
		// NF features of the item at feature offset fo: six entries read (one request each), 6 NF updates
		auto update = [&](auto nf_tag, uint32_t fo, bool vec_ok) {
			constexpr int NF = decltype(nf_tag)::value;
			float T0[3][NF], T1[3][NF], I[3][NF], gr[NF];
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				ld_feats<NF>(grid, e0[d] * L.F + fo, vec_ok, T0[d]);
				ld_feats<NF>(grid, (e0[d] + 1u) * L.F + fo, vec_ok, T1[d]);
			}
#pragma unroll
			for (int f = 0; f < NF; ++f) gr[f] = g[(int64_t)i * g_sn + (int64_t)(col0 + fo + f) * g_se];
#pragma unroll
			for (int d = 0; d < 3; ++d)
#pragma unroll
				for (int f = 0; f < NF; ++f) I[d][f] = __fmaf_rn(w1[d], T1[d][f], w0[d] * T0[d][f]);
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				const int j = (d + 1) % 3, k = (d + 2) % 3;
				double *dst0 = &cp_acc[(size_t)e0[d] * row + fo], *dst1 = dst0 + row;
#pragma unroll
				for (int f = 0; f < NF; ++f) {
					const float other = I[j][f] * I[k][f];
					float v0, v1;
					if (!SECOND) { v0 = gr[f] * (w0[d] * other); v1 = gr[f] * (w1[d] * other); }
					else {
						const float cross = __fmaf_rn(a[j] * (T1[j][f] - T0[j][f]), I[k][f], a[k] * (T1[k][f] - T0[k][f]) * I[j][f]);
						const float own = a[d] * other;
						v0 = gr[f] * __fmaf_rn(w0[d], cross, -own);
						v1 = gr[f] * __fmaf_rn(w1[d], cross, own);
					}
					add(dst0 + f, v0);
					add(dst1 + f, v1);
				}
			}
		};
#pragma unroll 1
		for (uint32_t fo = 0; fo < row;) {
			if (quad && fo + 4u <= row) { update(std::integral_constant<int, 4>{}, fo, true); fo += 4u; }
			else { update(std::integral_constant<int, 2>{}, fo, vec); fo += 2u; }
		}
	}
	__syncthreads();
	float *mine = partial + cp.part_off[item] + (size_t)r * n_acc;
	for (uint32_t t = threadIdx.x; t < n_acc; t += kCpThreads) mine[t] = fix ? (float)((double)(long long)cp_fix[t] * finv) : (float)cp_acc[t];
}

// dL/dparam of an item's pseudo levels += sum of its replicas' tables (replica 0 first)
__global__ __launch_bounds__(256) void k_cp_reduce(CpPlan cp, const nr3d_lotd_meta_t *__restrict__ md, const float *__restrict__ partial,
                                                   float *__restrict__ dparam) {
	const uint32_t item = blockIdx.y, q = cp.q[item], row = 2u * cp.np[item];
	const Lvl L = load_level(md, meta_level_of(md, q));
	const uint32_t n_acc = L.size * row, t = blockIdx.x * 256u + threadIdx.x;
	if (t >= n_acc) return;
	const float *p0 = partial + cp.part_off[item] + t;
	float sum = 0.0f;
	for (uint32_t r = 0; r < cp.R; ++r) sum += p0[(size_t)r * n_acc];
	float *dst = dparam + L.off + (size_t)(t / row) * L.F + meta_cnt_of(md, q) * 2u + (t % row);
	*dst += sum;
}

static bool cp_direct_enabled() { return opt::on(NR3D_OPT_CP_DIRECT); }
// the CP pseudo levels k_cp_direct serves (mask over the meta's pseudo levels; 0: none): unbatched 3-D metas with 2-feature
// pseudo levels, levels in [min_level, max_level] whose fp64 table fits LDS, partial tables inside `part_floats`
static uint64_t cp_plan(const nr3d_lotd_meta_t *m, uint32_t n, int32_t min_level, int32_t max_level, uint64_t part_floats, CpPlan &cp,
                        uint32_t &max_acc) {
	cp.n_items = 0; cp.R = 1; cp.pts_per_rep = n; max_acc = 0;
	if (!cp_direct_enabled() || m->n_dims_to_encode != 3 || m->n_feat_per_pseudo_lvl != 2 || m->n_pseudo_levels > 64u || n == 0) return 0;
	uint64_t mask = 0, floats_per_replica = 0;
	for (uint32_t q = 0; q < m->n_pseudo_levels;) {
		const uint32_t lv = m->map_levels[q];
		const nr3d_lotd_level_t &L = m->levels[lv];
		uint32_t nq = 1;                                       // the level's pseudo levels are consecutive
		while (q + nq < m->n_pseudo_levels && m->map_levels[q + nq] == lv) ++nq;
		const uint32_t q0 = q;
		q += nq;
		if (L.type != NR3D_LOD_CP || (int32_t)lv < min_level || (int32_t)lv > max_level) continue;
		uint32_t fit = (uint32_t)(kCpLdsBytes / ((uint64_t)L.size * 16u));
		fit = fit > kCpMaxPairs ? kCpMaxPairs : fit;
		if (fit == 0 || cp.n_items + div_up(nq, fit) > kCpMaxItems) continue;
		for (uint32_t k = 0; k < nq; k += fit) {
			const uint32_t np = (nq - k) < fit ? (nq - k) : fit;
			cp.q[cp.n_items] = q0 + k; cp.np[cp.n_items] = np;
			++cp.n_items;
			floats_per_replica += (uint64_t)L.size * 2u * np;
			max_acc = max_acc > L.size * 2u * np ? max_acc : L.size * 2u * np;
		}
		for (uint32_t k = 0; k < nq; ++k) mask |= 1ull << (q0 + k);
	}
	if (!cp.n_items) return 0;
	// replicas: >= 4096 points each, the partial tables inside the workspace, and the launch a whole number of rounds of
	// one workgroup per CU (256) as nearly as the divisors allow
	uint32_t r_max = kCpReplicas;
	const uint32_t by_points = div_up(n, 4096u);
	r_max = r_max > by_points ? by_points : r_max;
	while (r_max > 1 && floats_per_replica * r_max > part_floats) --r_max;
	if (floats_per_replica * r_max > part_floats) { cp.n_items = 0; return 0; }
	uint32_t R = r_max;
	double best = 0.0;
	for (uint32_t cand = r_max; cand >= 1 && cand * 2u > r_max; --cand) {
		const uint32_t blocks = cand * cp.n_items, rounds = div_up(blocks, 256u);
		const double fill = (double)blocks / (256.0 * rounds);
		if (fill > best + 1e-9) { best = fill; R = cand; }
	}
	cp.R = R;
	cp.pts_per_rep = div_up(n, R);
	uint64_t off = 0;
	for (uint32_t k = 0; k < cp.n_items; ++k) {
		cp.part_off[k] = (uint32_t)off;
		off += (uint64_t)m->levels[m->map_levels[cp.q[k]]].size * 2u * cp.np[k] * R;
	}
	return mask;
}

// -------------------------------------------------------------------------------------------------
// Small VM levels without records (round 4).  A VM level whose three planes each split into at most kVmDirectChunks LDS-sized
// pieces -- configs[3]: level 2, [128, 96, 64] x 8 features = four pseudo levels, 4/7 of all VM records -- is accumulated like the CP
// levels: a workgroup owns ONE COMPONENT d of ONE pseudo level (the plane over the two dims != d, or a band of its rows when the plane
// has more than 8192 entries, plus line d) and a share of the points; it walks its points, forms the component's six updates
// (emit_vm_component: the arithmetic of the record path, bit for bit) for every point whose cell lies in its band, and adds them
// to fp64 accumulators in LDS.  No record is written, sorted or read back; the workgroups' tables leave as fp32 and
// k_vm_direct_reduce adds the replicas in a fixed order.  Plane-major (not table-slice-major, the first version of this round:
// 2.08 -> 1.28 ms with its lines fixed, still half its lanes idle in every atomic) means every lane of a wave has work: a point
// is evaluated exactly once per (pseudo level, component), in the band that owns its cell's row -- a band holds one row more than it
// owns, for the corners one row up; that row is added to its owner's in the reduction.
// -------------------------------------------------------------------------------------------------
constexpr uint32_t kVmDirectChunks = 4, kVmDirectLg = 13, kVmDirectMaxItems = 64, kVmDirectThreads = 1024;
constexpr uint32_t kVmDirectMaxLines = 1024;   // entries of ONE line: its fp64 table (16 KiB) sits behind the plane band in LDS
struct VmPlan {
	uint32_t n_items, R, pts_per_rep, stride;     // stride: floats of one (item, replica) partial table = 2 (8192 + kVmDirectMaxLines)
	uint16_t q[kVmDirectMaxItems];          // pseudo level of item k
	uint8_t d[kVmDirectMaxItems];           // ... its component (plane over the dims != d, line d)
	uint16_t row0[kVmDirectMaxItems];       // ... first CELL row (along the plane's first dim) it owns
	uint16_t nrows[kVmDirectMaxItems];      // ... and how many (its LDS band holds nrows + 1 rows of entries)
};

template <bool SECOND, typename PT>
__global__ __launch_bounds__(kVmDirectThreads) void k_vm_direct(VmPlan vp, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n, uint32_t smooth,
                                                                const float *__restrict__ x, const float *__restrict__ vin_,
                                                                const float *__restrict__ g, int64_t g_sn, int64_t g_se,
                                                                const PT *__restrict__ params, float *__restrict__ partial, uint32_t opt_fix) {
	extern __shared__ __attribute__((aligned(16))) double vm_acc[];            // plane band [<= 8192 entries][2] | line d [Rd][2]
	constexpr uint32_t kEnt = 1u << kVmDirectLg;
	const uint32_t r = blockIdx.x, item = blockIdx.y;
	const uint32_t q = vp.q[item], dc = vp.d[item], row0 = vp.row0[item], nrows = vp.nrows[item];
	const Lvl L = load_level(md, meta_level_of(md, q));
	const VmGeom gm = vm_geom(L.res, (int)dc);
	double *ln_acc = vm_acc + 2u * kEnt;
	const uint32_t n_acc = 2u * kEnt + 2u * gm.Rd;
	for (uint32_t t = threadIdx.x; t < n_acc; t += kVmDirectThreads) vm_acc[t] = 0.0;
	__syncthreads();
	const uint32_t foff = meta_cnt_of(md, q) * 2u, col0 = meta_col_of(md, q);
	const auto grid = make_tab(params + L.off);
	const uint32_t band_lo = gm.plane_lo + row0 * gm.Rb;                       // first entry of this workgroup's band
	const uint32_t i_lo = r * vp.pts_per_rep, i_hi = min(n, i_lo + vp.pts_per_rep);
	// optional fixed-point accumulators from the workgroup's own bound, as k_cp_direct: |update| <= gmax pmax (first order),
	// <= gmax pmax 7.5 max R vmax (second order: plane g (wo a_d dl + C_m LI) <= 4 |a| pmax, line g (w PC -+ a PI) <= 5 |a| pmax, |a| <= 1.5 R |v|)
	double fscale = 0.0, finv = 0.0;
	bool fix = false;
	if (opt_fix) {
		__shared__ uint32_t s_bound[3];
		if (threadIdx.x < 3) s_bound[threadIdx.x] = 0u;
		__syncthreads();
		uint32_t gb = 0u, pb = 0u, vb = 0u;
		for (uint32_t i = i_lo + threadIdx.x; i < i_hi; i += kVmDirectThreads) {
#pragma unroll
			for (uint32_t f = 0; f < 2u; ++f) gb = max(gb, __float_as_uint(g[(int64_t)i * g_sn + (int64_t)(col0 + f) * g_se]) & 0x7FFFFFFFu);
			if (SECOND)
#pragma unroll
				for (int d = 0; d < 3; ++d) vb = max(vb, __float_as_uint(vin_[(size_t)i * 3 + d]) & 0x7FFFFFFFu);
		}
		for (uint32_t t = threadIdx.x; t < 2u * L.size; t += kVmDirectThreads)
			pb = max(pb, __float_as_uint((float)grid[(t >> 1) * L.F + foff + (t & 1u)]) & 0x7FFFFFFFu);
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			gb = max(gb, (uint32_t)__shfl_xor((int)gb, off, 64)); pb = max(pb, (uint32_t)__shfl_xor((int)pb, off, 64));
			vb = max(vb, (uint32_t)__shfl_xor((int)vb, off, 64));
		}
		if ((threadIdx.x & 63u) == 0u) { atomicMax(&s_bound[0], gb); atomicMax(&s_bound[1], pb); atomicMax(&s_bound[2], vb); }
		__syncthreads();
		float B = __uint_as_float(s_bound[0]) * __uint_as_float(s_bound[1]);
		if (SECOND) B *= 8.0f * fmaxf((float)L.res[0], fmaxf((float)L.res[1], (float)L.res[2])) * __uint_as_float(s_bound[2]);
		const uint32_t bb = __float_as_uint(B);
		if (bb != 0u && bb < 0x7F000000u && B >= 1e-30f) {
			const int e = (int)(bb >> 23) - 126;
			uint32_t lg = 0;
			while ((1u << lg) < vp.pts_per_rep) ++lg;
			const int sc = min(62 - (int)lg - 3, 44) - e;
			if (sc > -1000 && sc < 1000) {
				fscale = __longlong_as_double((long long)(sc + 1023) << 52);
				finv = __longlong_as_double((long long)(1023 - sc) << 52);
				fix = true;
			}
		}
	}
	auto add = [&](double *dst, float v) {                                             // block-uniform branch
		if (fix) atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)(__double_as_longlong(__fma_rn((double)v, fscale, 0x1.8p52)) - 0x4338000000000000LL));
		else atomicAdd(dst, (double)v);
	};
	for (uint32_t i = i_lo + threadIdx.x; i < i_hi; i += kVmDirectThreads) {
		float xp[3], a[3], grad[2];
#pragma unroll
		for (int d = 0; d < 3; ++d) xp[d] = x[(size_t)i * 3 + d];
		Cell<3> c;
		locate<3>(xp, L, smooth != 0, c);
		// x outside [0, 1] (the Python layer clamps, the C ABI takes what it gets): a cell outside the level would index the tables and
		// the LDS band out of bounds -- such a point adds nothing (unsigned compares: a negative floor is a huge index)
		if (!(c.g[0] + 1u < L.res[0] && c.g[1] + 1u < L.res[1] && c.g[2] + 1u < L.res[2])) continue;
		const uint32_t ca = gm.a == 0 ? c.g[0] : c.g[1];
		if (ca - row0 >= nrows) continue;                                       // another band of this plane owns the point
#pragma unroll
		for (int d = 0; d < 3; ++d) a[d] = SECOND ? c.sc[d] * vin_[(size_t)i * 3 + d] * c.dw[d] : 0.0f;
#pragma unroll
		for (int f = 0; f < 2; ++f) grad[f] = g[(int64_t)i * g_sn + (int64_t)(col0 + f) * g_se];
		uint32_t ent[6];
		float val[6][2];
		if (dc == 0u) emit_vm_component<2, 0, 6, SECOND>(L, c, a, grad, grid, foff, ent, val);            // block-uniform
		else if (dc == 1u) emit_vm_component<2, 1, 6, SECOND>(L, c, a, grad, grid, foff, ent, val);
		else emit_vm_component<2, 2, 6, SECOND>(L, c, a, grad, grid, foff, ent, val);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			double *dst = &vm_acc[(size_t)(ent[k] - band_lo) * 2u];
			add(dst, val[k][0]);
			add(dst + 1, val[k][1]);
		}
#pragma unroll
		for (int k = 4; k < 6; ++k) {
			double *dst = &ln_acc[(size_t)(ent[k] - gm.line_lo) * 2u];
			add(dst, val[k][0]);
			add(dst + 1, val[k][1]);
		}
	}
	__syncthreads();
	float *mine = partial + ((size_t)item * vp.R + r) * vp.stride;
	const unsigned long long *vm_fix = reinterpret_cast<const unsigned long long *>(vm_acc);
	for (uint32_t t = threadIdx.x; t < n_acc; t += kVmDirectThreads) mine[t] = fix ? (float)((double)(long long)vm_fix[t] * finv) : (float)vm_acc[t];
}

// dL/dparam += the replicas' tables, replica 0 first.  A band's extra row belongs to the NEXT band of the same plane: that band's
// workgroups add both (no address is touched by two workgroups); the first band of a component adds line d of all its bands.
__global__ __launch_bounds__(256) void k_vm_direct_reduce(VmPlan vp, const nr3d_lotd_meta_t *__restrict__ md, const float *__restrict__ partial,
                                                          float *__restrict__ dparam) {
	constexpr uint32_t kEnt = 1u << kVmDirectLg;
	const uint32_t item = blockIdx.y, q = vp.q[item], dc = vp.d[item], row0 = vp.row0[item], nrows = vp.nrows[item];
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	const Lvl L = load_level(md, meta_level_of(md, q));
	const VmGeom gm = vm_geom(L.res, (int)dc);
	auto sum_item = [&](uint32_t it, uint32_t at) {
		const float *p0 = partial + (size_t)it * vp.R * vp.stride + at;
		float s = 0.0f;
		for (uint32_t r = 0; r < vp.R; ++r) s += p0[(size_t)r * vp.stride];
		return s;
	};
	if (t < 2u * kEnt) {
		const uint32_t le = t >> 1, lr = le / gm.Rb, col = le - lr * gm.Rb, row = row0 + lr;
		if (lr > nrows || row >= gm.Ra) return;
		const bool has_next = item + 1u < vp.n_items && vp.q[item + 1u] == q && vp.d[item + 1u] == dc;
		if (lr == nrows && has_next) return;                                    // the next band's first row: added there
		float sum = sum_item(item, t);
		if (lr == 0u && row0 > 0u)                                              // ... the previous band's extra row
			sum += sum_item(item - 1u, 2u * ((uint32_t)vp.nrows[item - 1u] * gm.Rb + col) + (t & 1u));
		dparam[L.off + (size_t)(gm.plane_lo + row * gm.Rb + col) * L.F + meta_cnt_of(md, q) * 2u + (t & 1u)] += sum;
	} else if (row0 == 0u && t - 2u * kEnt < 2u * gm.Rd) {
		const uint32_t tl = t - 2u * kEnt;
		float sum = 0.0f;
		for (uint32_t it2 = item; it2 < vp.n_items && vp.q[it2] == q && vp.d[it2] == dc; ++it2) sum += sum_item(it2, t);
		dparam[L.off + (size_t)(gm.line_lo + (tl >> 1)) * L.F + meta_cnt_of(md, q) * 2u + (tl & 1u)] += sum;
	}
}

// the VM pseudo levels k_vm_direct serves (mask; 0: none): unbatched 3-D metas with 2-feature pseudo levels, levels inside
// [min_level, max_level] whose planes split into <= kVmDirectChunks bands each, partial tables inside `part_floats`
static uint64_t vm_direct_plan(const nr3d_lotd_meta_t *m, uint32_t n, int32_t min_level, int32_t max_level, uint64_t part_floats, VmPlan &vp,
                               uint64_t skip = 0) {
	vp.n_items = 0; vp.R = 1; vp.pts_per_rep = n; vp.stride = 2u * ((1u << kVmDirectLg) + kVmDirectMaxLines);
	if (!opt::on(NR3D_OPT_VM_DIRECT) || m->n_dims_to_encode != 3 || m->n_feat_per_pseudo_lvl != 2 || m->n_pseudo_levels > 64u || n == 0) return 0;
	uint64_t mask = 0;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
		const uint32_t lv = m->map_levels[q];
		const nr3d_lotd_level_t &L = m->levels[lv];
		if (L.type != NR3D_LOD_VectorMatrix || (int32_t)lv < min_level || (int32_t)lv > max_level || ((skip >> q) & 1ull)) continue;
		uint32_t chunks[3], rows[3], total = 0;
		bool ok = true;
		for (int d = 0; d < 3 && ok; ++d) {
			const VmGeom gm = vm_geom(L.res, d);
			const uint32_t fit = (1u << kVmDirectLg) / gm.Rb;                  // rows of entries a band can hold
			if (fit < 2u || gm.Rd > kVmDirectMaxLines || gm.Ra < 2u) { ok = false; break; }
			const uint32_t cells = gm.Ra - 1u;                                 // cell rows 0 .. Ra - 2
			chunks[d] = div_up(cells, fit - 1u);
			rows[d] = div_up(cells, chunks[d]);
			ok = chunks[d] <= kVmDirectChunks;
			total += chunks[d];
		}
		if (!ok || vp.n_items + total > kVmDirectMaxItems) continue;
		for (int d = 0; d < 3; ++d) {
			const uint32_t cells = vm_geom(L.res, d).Ra - 1u;
			for (uint32_t c = 0; c < chunks[d]; ++c) {
				const uint32_t r0 = c * rows[d], k = vp.n_items++;
				vp.q[k] = (uint16_t)q; vp.d[k] = (uint8_t)d; vp.row0[k] = (uint16_t)r0;
				vp.nrows[k] = (uint16_t)((cells - r0) < rows[d] ? (cells - r0) : rows[d]);
			}
		}
		mask |= 1ull << q;
	}
	if (!vp.n_items) return 0;
	// one workgroup per CU (144 KiB of LDS each): about two rounds of 256, >= 8192 points per replica
	uint32_t R = (2u * 256u) / vp.n_items;
	const uint32_t by_points = div_up(n, 8192u);
	R = R < 1u ? 1u : R;
	R = R > by_points ? by_points : R;
	while (R > 1u && (uint64_t)R * vp.n_items * vp.stride > part_floats) --R;
	if ((uint64_t)R * vp.n_items * vp.stride > part_floats) { vp.n_items = 0; return 0; }
	vp.R = R;
	vp.pts_per_rep = div_up(n, R);
	return mask;
}

// (the sorted-points path of large VM levels: lotd_sorted.hip, its own translation unit)

// plan for the pseudo levels of record class `cls` (0 levels => n_pseudo == 0)
// only != 0: exactly the pseudo levels of that mask, whatever their class, in blocks of `only_bp` points with `only_nr` records per
// point (the VM levels whose line updates stay in LDS: k_bin_vm3l)
static bool make_plan(const nr3d_lotd_meta_t *m, uint32_t n_chunk, uint32_t n_batches, uint32_t cls, BinPlan &plan,
                      uint64_t &offs_words, int32_t min_level = 0, int32_t max_level = 0x7fffffff, bool forest = false,
                      uint64_t skip = 0, uint64_t only = 0, uint32_t only_bp = 0, uint32_t only_nr = 0) {
	const uint32_t D = m->n_dims_to_encode, G = m->n_feat_per_pseudo_lvl;
	const uint32_t kBinPts = only ? only_bp : bin_points(G, cls);
	if (only) cls = only_nr;
	if (m->n_pseudo_levels > kMaxPlanLevels) return false;
	uint32_t lg = 0;
	while ((1u << (lg + 1)) <= (uint32_t)kLdsDoubles / G) ++lg;
	plan.epb_log2 = lg;
	plan.n_blk = div_up(n_chunk, kBinPts);
	plan.cap = kBinPts * cls;
	plan.n_batches = n_batches ? n_batches : 1u;
	uint32_t nq = 0;
	uint64_t base = 0;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
		const nr3d_lotd_level_t &L = m->levels[m->map_levels[q]];
		if (only) { if (q >= 64u || !((only >> q) & 1ull)) continue; }
		else if (rec_class(rec_count(L.type, D, forest)) != cls) continue;
		if (q < 64u && ((skip >> q) & 1ull)) continue;          // served without records (k_cp_direct) or by another plan
		// levels outside the requested range get no stage-A blocks, offsets or work items (max_level schedules, the
		// level-bucket calls of the data-parallel path)
		if ((int32_t)m->map_levels[q] < min_level || (int32_t)m->map_levels[q] > max_level) continue;
		const uint64_t n_virtual = (uint64_t)plan.n_batches * L.size;
		if (n_virtual > 0xFFFFFFFFull) return false;
		const uint32_t nb = div_up(n_virtual, 1u << lg);
		if (nb > kMaxBuckets) return false;
		plan.qmap[nq] = q;
		plan.nb[nq] = nb;
		plan.bucket_base[nq] = nq ? plan.bucket_base[nq - 1] + plan.nb[nq - 1] : 0u;
		if (base > 0xFFFFFFFFull) return false;
		plan.offs_base[nq] = (uint32_t)base;
		base += (uint64_t)(nb + 1) * plan.n_blk;
		++nq;
	}
	plan.n_pseudo = nq;
	plan.bucket_base[nq] = nq ? plan.bucket_base[nq - 1] + plan.nb[nq - 1] : 0u;
	offs_words = base;
	return true;
}

// every level must have a binned form, and params must not be batched (checked by the caller)
static bool binnable(const nr3d_lotd_meta_t *m, bool forest = false) {
	if (m->n_pseudo_levels > kMaxPlanLevels || (forest && m->n_dims_to_encode != 3)) return false;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q)
		if (rec_count(m->levels[m->map_levels[q]].type, m->n_dims_to_encode, forest) == 0) return false;
	return true;
}

constexpr uint32_t kClasses[5] = {8, 16, 24, 32, 48};

struct BinLayout { uint64_t rec_bytes, offs_bytes, plan_bytes, part_bytes, gt_bytes, total; };

// workspace = max over the record classes (they run one after another) of records + offsets, + the transposed dL/dy
static bool layout(const nr3d_lotd_meta_t *m, uint32_t n_chunk, uint32_t n_batches, BinLayout &l, bool forest = false) {
	l.rec_bytes = l.offs_bytes = l.plan_bytes = l.part_bytes = 0;
	for (uint32_t cls : kClasses) {
		BinPlan plan;
		uint64_t ow;
		if (!make_plan(m, n_chunk, n_batches, cls, plan, ow, 0, 0x7fffffff, forest)) return false;
		const uint64_t rb = (uint64_t)plan.n_pseudo * plan.n_blk * (1 + m->n_feat_per_pseudo_lvl) * plan.cap * 4;
		const uint64_t ob = ((ow * 4 + 255) / 256) * 256;
		l.rec_bytes = rb > l.rec_bytes ? rb : l.rec_bytes;
		l.offs_bytes = ob > l.offs_bytes ? ob : l.offs_bytes;
		const uint64_t pb = (((uint64_t)plan.bucket_base[plan.n_pseudo] * 3 + 4) * 4 + 255) / 256 * 256;   // tot | rep | item_start
		l.plan_bytes = pb > l.plan_bytes ? pb : l.plan_bytes;
		// one fp32 partial table per possible stage-B work item (only replicas write theirs)
		const uint64_t qb = (uint64_t)(work_units() + plan.bucket_base[plan.n_pseudo]) * kLdsDoubles * 4;
		l.part_bytes = qb > l.part_bytes ? qb : l.part_bytes;
	}
	if (!forest && n_batches <= 1 && pair_applies(m)) {       // lotd_pair.hip runs in the same regions
		uint64_t rb, ob, pb, qb;
		pair_layout(m, n_chunk, work_units(), rb, ob, pb, qb);
		l.rec_bytes = rb > l.rec_bytes ? rb : l.rec_bytes;
		l.offs_bytes = ob > l.offs_bytes ? ob : l.offs_bytes;
		l.plan_bytes = pb > l.plan_bytes ? pb : l.plan_bytes;
		l.part_bytes = qb > l.part_bytes ? qb : l.part_bytes;
	}
	l.rec_bytes = ((l.rec_bytes + 255) / 256) * 256;
	l.gt_bytes = (((uint64_t)m->n_encoded_dims * n_chunk * 4 + 255) / 256) * 256;
	l.total = l.rec_bytes + l.offs_bytes + l.plan_bytes + l.part_bytes + l.gt_bytes;
	return true;
}

// A stage-A thread keeps NR record slots of (1 + G) words: with 8-feature pseudo levels the 16-corner classes of 4-D metas (NR = 32)
// and the per-corner classes of a forest (NR = 48) do not fit the register file -- round 4 shipped k_bin<4, 8, ., 32> with 548-660
// and k_bin_forest<8, ., 48, __half> with 459-490 spilled dwords.  Such a meta runs as its width-4 regrouping (every pseudo level
// split into two of four features that write the same columns and table entries: ABI 2's map_col / map_cnt), whose kernels do not
// spill; the wide instantiations are no longer built.  The narrowed meta needs its own device copy: kStageAMetaBytes at the end of
// the workspace.
constexpr uint64_t kStageAMetaBytes = 4096;
static_assert(sizeof(nr3d_lotd_meta_t) <= kStageAMetaBytes, "room for the narrowed meta's device copy");
static bool stage_a_meta(const nr3d_lotd_meta_t *m, bool forest, nr3d_lotd_meta_t &out) {
	if (m->n_feat_per_pseudo_lvl != 8 || !(forest || m->n_dims_to_encode == 4) || 2u * m->n_pseudo_levels > NR3D_LOTD_MAX_PSEUDO) return false;
	out = *m;
	out.n_feat_per_pseudo_lvl = 4;
	out.n_pseudo_levels = 2u * m->n_pseudo_levels;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q)
		for (uint32_t j = 0; j < 2u; ++j) {
			out.map_levels[2 * q + j] = m->map_levels[q];
			out.map_cnt[2 * q + j] = (uint16_t)(2u * m->map_cnt[q] + j);
			out.map_col[2 * q + j] = (uint16_t)(m->map_col[q] + 4u * j);
		}
	return true;
}

// returns 0 when the binned path does not apply (caller falls back to the atomic kernels)
uint64_t dparam_workspace_bytes(const nr3d_lotd_meta_t *m, uint32_t n_points, uint32_t n_batches, bool forest) {
	if (!m || n_points == 0) return 0;
	nr3d_lotd_meta_t narrow;
	const bool narrowed = stage_a_meta(m, forest, narrow);
	if (narrowed) m = &narrow;
	if (!binnable(m, forest)) return 0;
	BinLayout lay;
	if (!layout(m, chunk_points(n_points), n_batches, lay, forest)) return 0;
	return lay.total + (narrowed ? kStageAMetaBytes : 0);
}

// NR3D_OPT_VM_SPLIT = 0: VM levels through the one-thread-per-point stage A (A/B, cross-check)
static bool vm_split_enabled() { return opt::on(NR3D_OPT_VM_SPLIT); }

template <int D, int G, int NR, bool DH>
static int launch_class(bool second, const BinPlan &pl, const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t n,
                        int32_t max_level, const float *xc, const float *vc, const float *gc, int64_t sn, int64_t se,
                        const void *params_, bool p_half, const Batch &ba, uint32_t *rec, uint32_t *offs, uint32_t *plan_buf,
                        float *partial, float *dparam, hipStream_t st, const ForestDev *fo = nullptr) {
	constexpr int BP = BinCfg<G, NR>::BP;
	const float *params = (const float *)params_;        // half tables: only the !DH instantiations read them (k_bin<..., __half>)
	uint32_t nb_max = 0;
	for (uint32_t q = 0; q < pl.n_pseudo; ++q) nb_max = nb_max > pl.nb[q] ? nb_max : pl.nb[q];
	const uint32_t NB = pl.bucket_base[pl.n_pseudo];
	uint32_t *tot = plan_buf, *rep = plan_buf + NB, *item_start = plan_buf + 2 * (size_t)NB;
	const size_t bin_lds = ((size_t)(1 + G) * BinCfg<G, NR>::cap + nb_max + 1) * sizeof(uint32_t);
	static bool attr_set_dev[64] = {};                 // per device: a process may drive several GPUs
	int dev_id = 0;
	NR3D_HIP_CHECK(hipGetDevice(&dev_id));
	bool &attr_set = attr_set_dev[dev_id & 63];
	if (!attr_set) {
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_accum<D, G>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsDoubles * 8));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin<D, G, true, NR, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin<D, G, false, NR, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
		if constexpr (!DH) {
			NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin<D, G, true, NR, false, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
			NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin<D, G, false, NR, false, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
		}
		attr_set = true;
	}
	prof::g_mask & (1u << NR3D_PROF_LOTD_BIN) ? prof::begin(NR3D_PROF_LOTD_BIN, st) : (void)0;
	if (fo) {
		if constexpr (D == 3 && (NR == 8 || NR == 24 || NR == 48)) {
			static bool fattr_dev[64] = {};
			if (!fattr_dev[dev_id & 63]) {
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_forest<G, true, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_forest<G, false, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
				fattr_dev[dev_id & 63] = true;
			}
			auto forest_launch = [&](auto kern, auto *tab) {
				hipLaunchKernelGGL(kern, dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n, max_level,
				                   meta->interpolation_type, xc, vc, gc, sn, se, tab, ba, *fo, rec, offs);
			};
			bool done = false;
			if constexpr (NR != 8) {                       // product types read the owner block's tables; Dense / Hash read none
				if (p_half) {
					static bool hattr_dev[64] = {};
					if (!hattr_dev[dev_id & 63]) {
						NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_forest<G, true, NR, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
						NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_forest<G, false, NR, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
						hattr_dev[dev_id & 63] = true;
					}
					if (second) forest_launch(k_bin_forest<G, true, NR, __half>, (const __half *)params_);
					else forest_launch(k_bin_forest<G, false, NR, __half>, (const __half *)params_);
					done = true;
				}
			}
			if (!done) { if (second) forest_launch(k_bin_forest<G, true, NR>, params); else forest_launch(k_bin_forest<G, false, NR>, params); }
		} else {
			return ::nr3d::fail("LoTD forest: the binned path handles 3-D metas only");
		}
	} else if (D == 3 && NR == 24 && !DH && vm_split_enabled()) {
		// class 24 in 3-D is the VM levels: three threads per point (six records each) instead of one with 18
		if constexpr (D == 3 && NR == 24 && !DH) {
			static bool vattr_dev[64] = {};
			if (!vattr_dev[dev_id & 63]) {
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3<G, true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3<G, false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3<G, true, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3<G, false, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
				vattr_dev[dev_id & 63] = true;
			}
			auto vm_launch = [&](auto kern, auto *tab) {
				hipLaunchKernelGGL(kern, dim3(pl.n_blk, pl.n_pseudo), dim3(BP * 3), bin_lds, st, pl, md, n, max_level,
				                   meta->interpolation_type, xc, vc, gc, sn, se, tab, ba, rec, offs);
			};
			if (p_half) {
				if (second) vm_launch(k_bin_vm3<G, true, __half>, (const __half *)params_); else vm_launch(k_bin_vm3<G, false, __half>, (const __half *)params_);
			} else {
				if (second) vm_launch(k_bin_vm3<G, true, float>, params); else vm_launch(k_bin_vm3<G, false, float>, params);
			}
		}
	} else if (!DH && p_half) {
		if constexpr (!DH) {
			if (second)
				hipLaunchKernelGGL((k_bin<D, G, true, NR, false, __half>), dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n, max_level,
				                   meta->interpolation_type, xc, vc, gc, sn, se, (const __half *)params_, ba, rec, offs);
			else
				hipLaunchKernelGGL((k_bin<D, G, false, NR, false, __half>), dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n, max_level,
				                   meta->interpolation_type, xc, vc, gc, sn, se, (const __half *)params_, ba, rec, offs);
		}
	} else if (second)
		hipLaunchKernelGGL((k_bin<D, G, true, NR, DH>), dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n, max_level,
		                   meta->interpolation_type, xc, vc, gc, sn, se, params, ba, rec, offs);
	else
		hipLaunchKernelGGL((k_bin<D, G, false, NR, DH>), dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n, max_level,
		                   meta->interpolation_type, xc, vc, gc, sn, se, params, ba, rec, offs);
	prof::end(NR3D_PROF_LOTD_BIN, st);
	hipLaunchKernelGGL(k_bucket_totals, dim3(div_up(NB, 4)), dim3(256), 0, st, pl, offs, tot);
	hipLaunchKernelGGL(k_plan_items, dim3(1), dim3(1024), 0, st, NB, pl.n_blk, work_units(), tot, rep, item_start);
	// sum of replicas <= work_units() (rounded shares of the total) + one per non-empty bucket
	{
		prof::Scope ps(NR3D_PROF_LOTD_ACCUM, st);
		hipLaunchKernelGGL((k_accum<D, G>), dim3(work_units() + NB), dim3(kAccThreads), kLdsDoubles * 8, st, pl, md, rec, offs, rep,
		                   item_start, ba, partial, dparam);
	}
	hipLaunchKernelGGL((k_reduce_partials<D, G>), dim3(NB, kLdsDoubles / kAccThreads), dim3(kAccThreads), 0, st, pl, md, rep, item_start, ba, partial, dparam);
	NR3D_LAUNCH_CHECK();
	return 0;
}

}  // namespace lotd
}  // namespace nr3d

// dL_dy [N, E] (any strides, float or half) -> feature-major float [E][N]: what the level-major kernels read (the parameter
// gradient passes make this copy themselves when handed a row-major dL_dy; a caller that runs several of them on one dL_dy --
// d(dL/dx)/dparam and d(dL/dx)/dx of one second-order step -- makes it once and hands it to both)
extern "C" int nr3d_lotd_dLdy_feature_major(uint32_t n_points, uint32_t n_encoded_dims, int grad_dtype, const void *dL_dy,
                                            int64_t g_sn, int64_t g_se, float *out, void *stream) {
	using namespace nr3d;
	using namespace nr3d::lotd;
	NR3D_CHECK(grad_dtype == NR3D_F32 || grad_dtype == NR3D_F16, "dLdy_feature_major: f32 / f16 dL_dy");
	if (n_points == 0 || n_encoded_dims == 0) return 0;
	NR3D_CHECK(dL_dy && out, "dLdy_feature_major: NULL tensor pointer");
	if (grad_dtype == NR3D_F16) launch_transpose<__half>(n_points, n_encoded_dims, (const __half *)dL_dy, g_sn, g_se, out, (hipStream_t)stream);
	else launch_transpose<float>(n_points, n_encoded_dims, (const float *)dL_dy, g_sn, g_se, out, (hipStream_t)stream);
	NR3D_LAUNCH_CHECK();
	return 0;
}

static_assert(sizeof(nr3d_lotd_meta_t) % 4 == 0 && sizeof(nr3d_lotd_meta_t) <= 3584, "the narrowed meta rides in the kernel arguments");
__global__ void k_store_meta(const nr3d_lotd_meta_t m, uint32_t *__restrict__ dst) {
	const uint32_t *src = (const uint32_t *)&m;
	for (uint32_t i = threadIdx.x; i < sizeof(nr3d_lotd_meta_t) / 4; i += blockDim.x) dst[i] = src[i];
}

namespace nr3d {
namespace lotd {

// NR3D_OPT_PAIR_SECOND = 0: d(dL/dx)/dparam of pair-path metas through the 12-byte corner records (A/B, and the cross-check)
static bool pair_second_enabled() { return opt::on(NR3D_OPT_PAIR_SECOND); }

int dparam_binned(bool second, const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, const float *dL_ddLdx,
                  const float *dL_dy, int64_t g_sn, int64_t g_se, const float *x, const float *params, const Batch &batch,
                  uint32_t n_batches, int32_t max_level, float *dparam, void *workspace, uint64_t workspace_bytes,
                  hipStream_t st, bool &handled, const ForestDev *forest, int32_t min_level, bool g_half, bool out_half, bool assign,
                  bool p_half) {
	handled = false;
	BinLayout lay;
	const uint32_t nc = chunk_points(N);
	nr3d_lotd_meta_t narrow;
	const bool narrowed = workspace && stage_a_meta(meta, forest != nullptr, narrow);
	if (narrowed) meta = &narrow;
	if (!workspace || !binnable(meta, forest != nullptr) || !layout(meta, nc, n_batches, lay, forest != nullptr)) return 0;
	if (workspace_bytes < lay.total + (narrowed ? kStageAMetaBytes : 0)) return 0;
	handled = true;
	if (narrowed) {
		// the narrowed meta travels BY VALUE in a launch's kernel arguments (2.6 KB of the 4 KB a launch may carry) and one
		// workgroup writes it to the end of the workspace: no host pointer outlives this call (round-5 advisor: an async copy
		// from the stack-local `narrow` relied on the runtime staging pageable memory before returning, and a stream capture
		// would have kept the dangling pointer)
		k_store_meta<<<1, 256, 0, st>>>(narrow, (uint32_t *)((char *)workspace + lay.total));
		NR3D_HIP_CHECK(hipGetLastError());
		meta_dev = (char *)workspace + lay.total;
	}
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	const uint32_t D = meta->n_dims_to_encode, G = meta->n_feat_per_pseudo_lvl, E = meta->n_encoded_dims;
	uint32_t *rec = (uint32_t *)workspace;
	uint32_t *offs = (uint32_t *)((char *)workspace + lay.rec_bytes);
	uint32_t *plan_buf = (uint32_t *)((char *)workspace + lay.rec_bytes + lay.offs_bytes);
	float *partial = (float *)((char *)workspace + lay.rec_bytes + lay.offs_bytes + lay.plan_bytes);
	float *gt = (float *)((char *)workspace + lay.rec_bytes + lay.offs_bytes + lay.plan_bytes + lay.part_bytes);
	const bool row_major = (g_se == 1 && g_sn == (int64_t)E && E > 1) || g_half;      // half gradients always go through gt
	// an unbatched 3-D Dense/Hash meta with 2-feature pseudo levels: pair records (lotd_pair.hip), first and second order
	const bool use_pair = !forest && n_batches <= 1 && !batch.inds && !batch.offsets && !batch.data_size && pair_applies(meta) &&
	                      !(second && !pair_second_enabled());
	if ((g_half || out_half) && !use_pair)
		return ::nr3d::fail("LoTD::bwd: half gradients are served natively on the pair-record path only (nr3d_lotd_half_params_ok)");
	// uninitialised dparam: the pair path assigns when ONE pass covers every level; otherwise zero-fill and accumulate
	const bool assign_now = assign && use_pair && N <= nc && min_level <= 0 && max_level >= (int32_t)meta->n_levels - 1;
	if (assign && !assign_now)
		NR3D_HIP_CHECK(hipMemsetAsync(dparam, 0, (size_t)(n_batches ? n_batches : 1u) * meta->n_params * (out_half ? 2 : 4), st));

	for (uint32_t p0 = 0; p0 < N; p0 += nc) {
		const uint32_t n = (N - p0) < nc ? (N - p0) : nc;
		const float *xc = x + (size_t)p0 * D;
		const float *vc = dL_ddLdx ? dL_ddLdx + (size_t)p0 * D : nullptr;
		const float *gc = g_half ? reinterpret_cast<const float *>(reinterpret_cast<const __half *>(dL_dy) + (int64_t)p0 * g_sn)
		                         : dL_dy + (int64_t)p0 * g_sn;
		int64_t sn = g_sn, se = g_se;
		Batch ba = batch;                              // this chunk's view of the batch description
		if (ba.inds) ba.inds += p0;
		ba.first_point = p0;
		// the feature-major copy of dL_dy, made when the first path that reads columns asks for it (round 6: a pass whose levels all go over
		// sorted points -- the reference's forest workload -- never does: 0.13 ms)
		bool gt_ready = !row_major;
		auto need_gt = [&]() {
			if (gt_ready) return;
			if (g_half) launch_transpose<__half>(n, E, reinterpret_cast<const __half *>(gc), g_sn, g_se, gt, st);
			else launch_transpose<float>(n, E, gc, g_sn, g_se, gt, st);
			gc = gt; sn = 1; se = (int64_t)n;
			gt_ready = true;
		};
		if (use_pair) {
			need_gt();
			if (int rc = pair_chunk(meta, md, n, xc, gc, sn, se, min_level, max_level, work_units(), dparam,
			                        (out_half ? 1u : 0u) | (assign_now ? 2u : 0u), rec, offs, plan_buf, partial, st,
			                        second ? vc : nullptr))
				return rc;
			continue;
		}
		// CP levels whose table fits LDS skip the records (k_cp_direct): they run first, in the partial-table region the
		// record classes use afterwards
		uint64_t cp_mask = 0;
		if (!forest && n_batches <= 1 && !batch.inds && !batch.offsets && !batch.data_size) {
			CpPlan cp;
			uint32_t max_acc = 0;
			cp_mask = cp_plan(meta, n, min_level, max_level, lay.part_bytes / 4, cp, max_acc);
			if (cp_mask) {
				static bool cp_attr[64] = {};
				int dev_id = 0;
				NR3D_HIP_CHECK(hipGetDevice(&dev_id));
				if (!cp_attr[dev_id & 63]) {
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_cp_direct<false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCpLdsBytes));
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_cp_direct<true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCpLdsBytes));
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_cp_direct<false, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCpLdsBytes));
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_cp_direct<true, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCpLdsBytes));
					cp_attr[dev_id & 63] = true;
				}
				need_gt();
				const size_t lds = (size_t)max_acc * 8;
				auto cp_launch = [&](auto kern, auto *tab) {
					hipLaunchKernelGGL(kern, dim3(cp.R, cp.n_items), dim3(kCpThreads), lds, st, cp, md, n, meta->interpolation_type, xc, vc,
					                   gc, sn, se, tab, partial, opt::on(NR3D_OPT_DIRECT_FIXED) ? 1u : 0u);
				};
				if (p_half) {
					if (second) cp_launch(k_cp_direct<true, __half>, (const __half *)params); else cp_launch(k_cp_direct<false, __half>, (const __half *)params);
				} else {
					if (second) cp_launch(k_cp_direct<true, float>, params); else cp_launch(k_cp_direct<false, float>, params);
				}
				hipLaunchKernelGGL(k_cp_reduce, dim3(div_up(max_acc, 256u), cp.n_items), dim3(256), 0, st, cp, md, partial, dparam);
				NR3D_LAUNCH_CHECK();
			}
		}
		// VM levels over sorted points (single tables, batches, forests): points sorted by (block, coordinate), bands accumulated in LDS, no
		// records.  Asked BEFORE k_vm_direct since round 6: with unit records and fixed-point accumulators it beats k_vm_direct + records on
		// configs[3] too (dL/dparam 4.90 -> 4.46 ms, d(dL/dx)/dparam 5.04 -> 4.71 at 2^22 points), so small tables take it from 2^21 points on
		// (lotd_sorted.hip); its scratch is the record / offsets region, which the classes below use afterwards
		if (!g_half) {
			VsPlan vsp;
			const uint64_t smask = vm_sorted_plan(meta, n, n_batches, forest != nullptr, min_level, max_level, cp_mask, vsp);
			if (smask) {
				VsScratch vss;
				vm_sorted_scratch(vsp, n, E, second, forest != nullptr, vss);
				if (vss.total <= lay.rec_bytes + lay.offs_bytes) {
					if (int rc = vm_sorted_run(second, vsp, meta, md, n, xc, vc, dL_dy + (int64_t)p0 * g_sn, g_sn, g_se, params, p_half, ba, forest, dparam,
					                           (char *)workspace, vss, st))
						return rc;
					cp_mask |= smask;
				}
			}
		}
		// small VM levels skip the records altogether (k_vm_direct), like the CP levels above
		if (!forest && n_batches <= 1 && !batch.inds && !batch.offsets && !batch.data_size) {
			VmPlan vp;
			const uint64_t vmask = vm_direct_plan(meta, n, min_level, max_level, lay.part_bytes / 4, vp, cp_mask);
			if (vmask) {
				static bool vattr[64] = {};
				int dev_id = 0;
				NR3D_HIP_CHECK(hipGetDevice(&dev_id));
				constexpr int kVmLds = ((2 << kVmDirectLg) + 2 * (int)kVmDirectMaxLines) * 8;
				if (!vattr[dev_id & 63]) {
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_vm_direct<false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kVmLds));
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_vm_direct<true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kVmLds));
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_vm_direct<false, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kVmLds));
					NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_vm_direct<true, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kVmLds));
					vattr[dev_id & 63] = true;
				}
				need_gt();
				auto vd_launch = [&](auto kern, auto *tab) {
					hipLaunchKernelGGL(kern, dim3(vp.R, vp.n_items), dim3(kVmDirectThreads), (size_t)vp.stride * 8, st, vp, md, n, meta->interpolation_type, xc, vc,
					                   gc, sn, se, tab, partial, opt::get(NR3D_OPT_DIRECT_FIXED) == 2 ? 1u : 0u);
				};
				// (fixed-point accumulators pay in k_cp_direct, which is LDS bound: 1.29 -> 0.98 ms on configs[3]; in k_vm_direct the
				// bound scan and the conversions cost more than they save -- slice-major 1.28 -> 1.48 ms, plane-major 5.13 -> 5.21 ms per
				// pass: it keeps fp64 unless NR3D_OPT_DIRECT_FIXED is 2)
				{
					prof::Scope ps(NR3D_PROF_LOTD_DIRECT, st);
					if (p_half) {
						if (second) vd_launch(k_vm_direct<true, __half>, (const __half *)params); else vd_launch(k_vm_direct<false, __half>, (const __half *)params);
					} else {
						if (second) vd_launch(k_vm_direct<true, float>, params); else vd_launch(k_vm_direct<false, float>, params);
					}
				}
				hipLaunchKernelGGL(k_vm_direct_reduce, dim3(div_up(vp.stride, 256u), vp.n_items), dim3(256), 0, st, vp, md, partial, dparam);
				NR3D_LAUNCH_CHECK();
				cp_mask |= vmask;
			}
		}
		// VM levels: the line tables' gradients accumulate in LDS inside stage A, only the plane updates travel as records
		// (k_bin_vm3l; 12 instead of 18 records per point and pseudo level), then the usual stage B over those records
		if (!forest && n_batches <= 1 && !batch.inds && !batch.offsets && !batch.data_size && D == 3 && G == 2 &&
		    opt::on(NR3D_OPT_VM_LINES_DIRECT) && vm_split_enabled() && meta->n_pseudo_levels <= 64u) {
			uint64_t vm_mask = 0;
			uint32_t line_stride = 0;
			for (uint32_t q = 0; q < meta->n_pseudo_levels; ++q) {
				const nr3d_lotd_level_t &Lq = meta->levels[meta->map_levels[q]];
				if (Lq.type != NR3D_LOD_VectorMatrix || (int32_t)meta->map_levels[q] < min_level || (int32_t)meta->map_levels[q] > max_level) continue;
				if ((cp_mask >> q) & 1ull) continue;                        // served by k_vm_direct
				const uint32_t nl = (Lq.res[0] + Lq.res[1] + Lq.res[2]) * G;
				if ((uint64_t)nl * 8u > 32u * 1024u) continue;              // fp64 line table <= 32 KiB: two workgroups still share a CU
				vm_mask |= 1ull << q;
				line_stride = nl > line_stride ? nl : line_stride;
			}
			if (vm_mask) {
				constexpr int BPL = BinCfg<2, 24>::BP;
				BinPlan pl;
				uint64_t ow;
				make_plan(meta, n, n_batches, 0, pl, ow, min_level, max_level, false, cp_mask, vm_mask, (uint32_t)BPL, kLinesRecPerPoint);
				// replicas: about four workgroups per CU in all (two are resident), never more than there are point blocks
				uint32_t R = (4u * 256u) / pl.n_pseudo;
				R = R < 1u ? 1u : (R > pl.n_blk ? pl.n_blk : R);
				while (R > 1u && (uint64_t)R * pl.n_pseudo * line_stride * 4u > lay.part_bytes) --R;
				if (pl.n_pseudo && (uint64_t)R * pl.n_pseudo * line_stride * 4u <= lay.part_bytes) {
					uint32_t nb_max = 0;
					for (uint32_t q = 0; q < pl.n_pseudo; ++q) nb_max = nb_max > pl.nb[q] ? nb_max : pl.nb[q];
					const uint32_t NB = pl.bucket_base[pl.n_pseudo];
					uint32_t *tot = plan_buf, *rep = plan_buf + NB, *item_start = plan_buf + 2 * (size_t)NB;
					const size_t bin_lds = (size_t)line_stride * 8 + ((size_t)(1 + G) * BPL * kLinesRecPerPoint + nb_max + 1) * sizeof(uint32_t);
					static bool lattr[64] = {};
					int dev_id = 0;
					NR3D_HIP_CHECK(hipGetDevice(&dev_id));
					if (!lattr[dev_id & 63]) {
						NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3l<2, true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
						NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3l<2, false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
						NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3l<2, true, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
						NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin_vm3l<2, false, __half>, hipFuncAttributeMaxDynamicSharedMemorySize, kBinLdsDyn));
						NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_accum<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsDoubles * 8));
						lattr[dev_id & 63] = true;
					}
					NR3D_CHECK(bin_lds <= (size_t)kBinLdsDyn, "LoTD::bwd: VM line tables do not fit the stage-A LDS budget");
					need_gt();
					auto vm_launch = [&](auto kern, auto *tab) {
						hipLaunchKernelGGL(kern, dim3(R, pl.n_pseudo), dim3(BPL * 3), bin_lds, st, pl, md, n, max_level, meta->interpolation_type,
						                   xc, vc, gc, sn, se, tab, ba, rec, offs, partial, line_stride);
					};
					{
						prof::Scope ps(NR3D_PROF_LOTD_BIN, st);
						if (p_half) {
							if (second) vm_launch(k_bin_vm3l<2, true, __half>, (const __half *)params); else vm_launch(k_bin_vm3l<2, false, __half>, (const __half *)params);
						} else {
							if (second) vm_launch(k_bin_vm3l<2, true, float>, params); else vm_launch(k_bin_vm3l<2, false, float>, params);
						}
					}
					// the line tables first: stage B's replicas reuse the partial-table region afterwards
					hipLaunchKernelGGL(k_vm_lines_reduce<2>, dim3(div_up(line_stride, 256u), pl.n_pseudo), dim3(256), 0, st, pl, md, R, line_stride,
					                   partial, dparam);
					hipLaunchKernelGGL(k_bucket_totals, dim3(div_up(NB, 4)), dim3(256), 0, st, pl, offs, tot);
					hipLaunchKernelGGL(k_plan_items, dim3(1), dim3(1024), 0, st, NB, pl.n_blk, work_units(), tot, rep, item_start);
					{
						prof::Scope ps(NR3D_PROF_LOTD_ACCUM, st);
						hipLaunchKernelGGL((k_accum<3, 2>), dim3(work_units() + NB), dim3(kAccThreads), kLdsDoubles * 8, st, pl, md, rec, offs, rep,
						                   item_start, ba, partial, dparam);
					}
					hipLaunchKernelGGL((k_reduce_partials<3, 2>), dim3(NB, kLdsDoubles / kAccThreads), dim3(kAccThreads), 0, st, pl, md, rep, item_start, ba,
					                   partial, dparam);
					NR3D_LAUNCH_CHECK();
					cp_mask |= vm_mask;                          // the record classes below leave these levels out
				}
			}
		}
		for (uint32_t cls : kClasses) {
			BinPlan pl;
			uint64_t ow;
			make_plan(meta, n, n_batches, cls, pl, ow, min_level, max_level, forest != nullptr, cp_mask);
			if (pl.n_pseudo == 0) continue;
			need_gt();
			int rc = 0;
			if (forest) {                                  // 3-D (binnable); one stage-A kernel per record class
				const uint32_t G = meta->n_feat_per_pseudo_lvl;
#define NR3D_FOREST_CLASS(G_, NR_) rc = launch_class<3, G_, NR_, true>(second, pl, meta, md, n, max_level, xc, vc, gc, sn, se, params, p_half, ba, rec, offs, plan_buf, partial, dparam, st, forest)
#define NR3D_FOREST_G(G_) do { if (cls == 8) NR3D_FOREST_CLASS(G_, 8); else if (cls == 24) NR3D_FOREST_CLASS(G_, 24); \
	else if (cls == 48) NR3D_FOREST_CLASS(G_, 48); } while (0)
				if (G == 2) NR3D_FOREST_G(2); else if (G == 4) NR3D_FOREST_G(4);
				else return ::nr3d::fail("LoTD forest bwd (binned): pseudo levels of 8 features run as their width-4 regrouping (stage_a_meta)");
#undef NR3D_FOREST_G
#undef NR3D_FOREST_CLASS
				if (rc) return rc;
				continue;
			}
			// only the (D, class) pairs some level type can produce are instantiated
			DISPATCH_DG_BIN(D, G, {
				// hash-only metas (every level Dense or Hash) get kernels without the product-type code
				if (meta->c_hash_only) {
					if constexpr (D <= 3) rc = launch_class<D, G, 8, true>(second, pl, meta, md, n, max_level, xc, vc, gc, sn, se, params, p_half, ba, rec, offs, plan_buf, partial, dparam, st, forest);
					else rc = launch_class<D, G, 16, true>(second, pl, meta, md, n, max_level, xc, vc, gc, sn, se, params, p_half, ba, rec, offs, plan_buf, partial, dparam, st);
				} else if (cls == 8) rc = launch_class<D, G, 8, false>(second, pl, meta, md, n, max_level, xc, vc, gc, sn, se, params, p_half, ba, rec, offs, plan_buf, partial, dparam, st);
				else if (cls == 16) {
					if constexpr (D >= 3) rc = launch_class<D, G, 16, false>(second, pl, meta, md, n, max_level, xc, vc, gc, sn, se, params, p_half, ba, rec, offs, plan_buf, partial, dparam, st);
				} else if (cls == 24) {
					if constexpr (D == 3) rc = launch_class<D, G, 24, false>(second, pl, meta, md, n, max_level, xc, vc, gc, sn, se, params, p_half, ba, rec, offs, plan_buf, partial, dparam, st);
				} else {
					if constexpr (D == 4) rc = launch_class<D, G, 32, false>(second, pl, meta, md, n, max_level, xc, vc, gc, sn, se, params, p_half, ba, rec, offs, plan_buf, partial, dparam, st);
				}
			});
			if (rc) return rc;
		}
	}
	return 0;
}

}  // namespace lotd
}  // namespace nr3d
