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
//                     ds_add_f64, then write the slice of dL/dparam once (plain stores; f32 atomics only for the
//                     few replicated small levels).
//
// Both stages are pure streaming (12 B per corner update written once and read once); the fp64 accumulation makes
// the result independent of the update order up to the final f64->f32 rounding.
#include "lotd_device.h"
#include <stdlib.h>

namespace nr3d {
namespace lotd {

constexpr int kAccThreads = 1024;         // stage-B workgroup
constexpr int kLdsDoubles = 16384;        // 128 KiB of fp64 accumulators
constexpr int kMaxPlanLevels = 64;        // pseudo levels handled by the binned path
constexpr uint32_t kMaxBuckets = 8192;    // per pseudo level

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
		else { constexpr int D = 4, G = 8; __VA_ARGS__; }                            \
	} while (0)

struct BinPlan {
	uint32_t nb[kMaxPlanLevels];          // buckets per pseudo level
	uint32_t rep[kMaxPlanLevels];         // replicas per bucket (stage B)
	uint32_t order[kMaxPlanLevels];       // stage-B launch order of the pseudo levels: largest workgroups first
	uint32_t offs_base[kMaxPlanLevels];   // start of this pseudo level's offset table (in uint32 units)
	uint32_t epb_log2;                    // log2(entries per bucket)
	uint32_t n_blk;                       // stage-A workgroups along the points of the current chunk
	uint32_t n_pseudo;
};

// -------------------------------------------------------------------------------------------------
// [n, E] row-major -> [E, n] (32x32 LDS tiles), so the level-major stage A reads its G columns coalesced
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose(uint32_t n, uint32_t E, const float *__restrict__ src, int64_t s_sn,
                                                   int64_t s_se, float *__restrict__ dst) {
	__shared__ float tile[32][33];
	const uint32_t i0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
	const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
	for (int k = 0; k < 32; k += 8) {
		const uint32_t i = i0 + ty + k, e = e0 + tx;
		tile[ty + k][tx] = (i < n && e < E) ? src[(int64_t)i * s_sn + (int64_t)e * s_se] : 0.0f;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < 32; k += 8) {
		const uint32_t e = e0 + ty + k, i = i0 + tx;
		if (i < n && e < E) dst[(size_t)e * n + i] = tile[tx][ty + k];
	}
}

// -------------------------------------------------------------------------------------------------
// Stage A: bin the corner updates of BP points x 1 pseudo level by bucket.
// BP (points = threads per workgroup) is the largest power of two whose record staging area
// ((1 + G) words x BP x 2^D records) fits in 64 KiB of LDS, so 2 workgroups share a CU.
// -------------------------------------------------------------------------------------------------
template <int D, int G>
struct BinCfg {
	static constexpr int raw = 16384 / ((1 + G) * (1 << D));
	static constexpr int BP = raw >= 512 ? 512 : raw >= 256 ? 256 : raw >= 128 ? 128 : raw >= 64 ? 64 : 32;
	static constexpr uint32_t cap = (uint32_t)BP << D;          // records per (pseudo level, point block)
};

template <int D, int G, bool SECOND>
__global__ __launch_bounds__((BinCfg<D, G>::BP)) void k_bin(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md, uint32_t n,
                                                          int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                          const float *__restrict__ vin_, const float *__restrict__ g,
                                                          int64_t g_sn, int64_t g_se, uint32_t *__restrict__ rec,
                                                          uint32_t *__restrict__ offs_g) {
	constexpr int BP = BinCfg<D, G>::BP;
	constexpr uint32_t cap = BinCfg<D, G>::cap;
	constexpr int C = 1 << D;
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];   // stage[(1+G)*cap] | hist[nb + 1]
	__shared__ uint64_t scan_lds[BP / 64 > 0 ? BP / 64 : 1];
	uint32_t *stage = smem;
	uint32_t *hist = smem + (size_t)(1 + G) * cap;
	const uint32_t blk = blockIdx.x, q = blockIdx.y;
	const uint32_t nb = plan.nb[q];
	const uint32_t level = md->map_levels[q];
	const uint32_t i = blk * BP + threadIdx.x;
	const Lvl L = load_level(md, level);

	for (uint32_t b = threadIdx.x; b <= nb; b += BP) hist[b] = 0;
	__syncthreads();

	const bool active = (i < n) && ((int32_t)level <= max_level);
	uint32_t ent[C], rank[C];
	float w[C], grad[G];
	if (active) {
		float xp[D];
#pragma unroll
		for (int d = 0; d < D; ++d) xp[d] = x[(size_t)i * D + d];
		Cell<D> c;
		locate<D>(xp, L, smooth != 0, c);
#pragma unroll
		for (int f = 0; f < G; ++f) grad[f] = g[(int64_t)i * g_sn + (int64_t)(q * G + f) * g_se];
		float a[D];
#pragma unroll
		for (int d = 0; d < D; ++d) a[d] = SECOND ? c.sc[d] * vin_[(size_t)i * D + d] * c.dw[d] : 0.0f;
#pragma unroll
		for (uint32_t k = 0; k < (uint32_t)C; ++k) {
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
			uint32_t p[D];
			corner_pos<D>(c, k, p);
			ent[k] = (L.type == NR3D_LOD_Dense) ? entry_dense<D>(L, p) : entry_hash<D>(L, p);
			rank[k] = atomicAdd(&hist[ent[k] >> plan.epb_log2], 1u);
		}
	}
	__syncthreads();

	// exclusive scan of the bucket histogram (in place); hist[nb] = total
	uint64_t carry = 0;
	for (uint32_t base = 0; base <= nb; base += BP) {
		const uint32_t b = base + threadIdx.x;
		const uint64_t v = (b < nb) ? hist[b] : 0;
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		uint64_t inc = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint64_t t = __shfl_up(inc, off, 64);
			if (lane >= off) inc += t;
		}
		if (lane == 63 || (BP < 64 && lane == BP - 1)) scan_lds[wave] = inc;
		__syncthreads();
		uint64_t wave_off = 0, tot = 0;
#pragma unroll
		for (int k = 0; k < (BP + 63) / 64; ++k) { const uint64_t t = scan_lds[k]; if (k < wave) wave_off += t; tot += t; }
		if (b <= nb) hist[b] = (uint32_t)(carry + wave_off + inc - v);
		carry += tot;
		__syncthreads();
	}

	// counting-sort the records into the LDS staging area (AoS: {idx, val[0..G-1]} per record)
	if (active) {
		const uint32_t mask = (1u << plan.epb_log2) - 1u;
#pragma unroll
		for (uint32_t k = 0; k < (uint32_t)C; ++k) {
			const uint32_t pos = hist[ent[k] >> plan.epb_log2] + rank[k];
			stage[pos * (1 + G)] = ent[k] & mask;
#pragma unroll
			for (int f = 0; f < G; ++f) stage[pos * (1 + G) + 1 + f] = __float_as_uint(grad[f] * w[k]);
		}
	}
	__syncthreads();

	// coalesced 16-byte write-out of the filled prefix (records are (1 + G)-word AoS, so a bucket's run is one
	// contiguous span of the slot and stage B over-fetches at most one cache line per run)
	const uint32_t total = hist[nb];                          // multiple of 2^D >= 4
	uint4 *dst = reinterpret_cast<uint4 *>(rec + ((size_t)q * plan.n_blk + blk) * (size_t)(1 + G) * cap);
	const uint4 *src = reinterpret_cast<const uint4 *>(stage);
	for (uint32_t v4 = threadIdx.x; v4 < total / 4 * (1 + G); v4 += BP) dst[v4] = src[v4];
	uint32_t *ob = offs_g + plan.offs_base[q];
	for (uint32_t b = threadIdx.x; b <= nb; b += BP) ob[(size_t)b * plan.n_blk + blk] = hist[b];
}

// -------------------------------------------------------------------------------------------------
// Stage B: one bucket (x replica) -> fp64 LDS accumulation -> slice of dL/dparam
// -------------------------------------------------------------------------------------------------
template <int D, int G>
__global__ __launch_bounds__(kAccThreads) void k_accum(BinPlan plan, const nr3d_lotd_meta_t *__restrict__ md,
                                                       const uint32_t *__restrict__ rec,
                                                       const uint32_t *__restrict__ offs_g,
                                                       float *__restrict__ dparam) {
	// accumulators are feature-major (acc[f][entry]): the G atomics of a record spread over all LDS banks
	extern __shared__ __attribute__((aligned(16))) double acc[];      // [kLdsDoubles]
	constexpr uint32_t cap = BinCfg<D, G>::cap;
	const uint32_t q = plan.order[blockIdx.y];
	const uint32_t nb = plan.nb[q], R = plan.rep[q];
	if (blockIdx.x >= nb * R) return;
	const uint32_t b = blockIdx.x / R, r = blockIdx.x - b * R;
	const uint32_t level = md->map_levels[q];
	const Lvl L = load_level(md, level);
	const uint32_t foff0 = (uint32_t)md->map_cnt[q] * G;

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
	// one sparse pass.  The kernel is bound by the LDS atomic pipe (SQ_ACTIVE_INST_LDS + SQ_LDS_BANK_CONFLICT ~ 100 %
	// of the busy cycles: 64 random 8-byte addresses over 16 bank pairs cost ~58 cycles per ds_add_f64), not by HBM,
	// so neither deeper load pipelining nor a bucket-major record stream (both tried) shortens it.
	const uint32_t epb = 1u << plan.epb_log2;
	const uint32_t per_wave = (blk_hi - blk_lo + n_waves - 1) / n_waves;
	const uint32_t w_lo = min(blk_lo + wave * per_wave, blk_hi), w_hi = min(w_lo + per_wave, blk_hi);
	constexpr int kUnroll = 8;
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
					idx[u] = r_p[0];
#pragma unroll
					for (int f = 0; f < G; ++f) val[u][f] = __uint_as_float(r_p[1 + f]);
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

	// flush: batches of kFlush read-modify-writes per thread, all loads issued before the first store (the compiler
	// cannot hoist them itself: dparam may alias itself across iterations)
	float *dst = dparam + L.off;
	constexpr int kFlush = 8;
	for (uint32_t tb = threadIdx.x; tb < epb * G; tb += kAccThreads * kFlush) {
		float *p[kFlush];
		float v[kFlush], old[kFlush];
#pragma unroll
		for (int k = 0; k < kFlush; ++k) {
			const uint32_t t = tb + (uint32_t)k * kAccThreads;
			const uint32_t el = t / G, f = t - el * G;
			const uint32_t entry = b * epb + el;
			const bool in = (t < epb * G) && (entry < L.size);
			p[k] = in ? dst + ((size_t)entry * L.F + foff0 + f) : nullptr;
			v[k] = in ? (float)acc[f * epb + el] : 0.0f;
		}
		if (R == 1) {                              // this workgroup is the only writer of the slice
#pragma unroll
			for (int k = 0; k < kFlush; ++k) old[k] = p[k] ? *p[k] : 0.0f;
#pragma unroll
			for (int k = 0; k < kFlush; ++k) if (p[k]) *p[k] = old[k] + v[k];
		} else {
#pragma unroll
			for (int k = 0; k < kFlush; ++k) if (p[k] && v[k] != 0.0f) atomic_add_f32(p[k], v[k]);
		}
	}
}

// -------------------------------------------------------------------------------------------------
// Host side
// -------------------------------------------------------------------------------------------------
static uint32_t chunk_points(uint32_t n) {
	static uint32_t chunk = 0;
	if (!chunk) {
		const char *e = getenv("NR3D_LOTD_BIN_CHUNK_LOG2");
		const int lg = e ? atoi(e) : 20;
		chunk = 1u << (lg < 12 ? 12 : (lg > 24 ? 24 : lg));
	}
	return n < chunk ? n : chunk;
}

static uint32_t bin_points(uint32_t D, uint32_t G) {
	const uint32_t raw = 16384u / ((1u + G) * (1u << D));
	return raw >= 512 ? 512 : raw >= 256 ? 256 : raw >= 128 ? 128 : raw >= 64 ? 64 : 32;
}

static bool make_plan(const nr3d_lotd_meta_t *m, uint32_t n_chunk, BinPlan &plan, uint64_t &offs_words) {
	const uint32_t G = m->n_feat_per_pseudo_lvl;
	const uint32_t kBinPts = bin_points(m->n_dims_to_encode, G);
	if (m->n_pseudo_levels > kMaxPlanLevels || !m->c_hash_only) return false;
	uint32_t lg = 0;
	while ((1u << (lg + 1)) <= (uint32_t)kLdsDoubles / G) ++lg;
	plan.epb_log2 = lg;
	plan.n_blk = div_up(n_chunk, kBinPts);
	plan.n_pseudo = m->n_pseudo_levels;
	uint64_t base = 0;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
		const nr3d_lotd_level_t &L = m->levels[m->map_levels[q]];
		const uint32_t nb = div_up(L.size, 1u << lg);
		if (nb > kMaxBuckets) return false;
		plan.nb[q] = nb;
		// enough stage-B workgroups per level to spread over the chip, never more replicas than point blocks
		uint32_t rep = 1;
		plan.rep[q] = rep;
		if (base > 0xFFFFFFFFull) return false;
		plan.offs_base[q] = (uint32_t)base;
		base += (uint64_t)(nb + 1) * plan.n_blk;
	}
	offs_words = base;
	// A bucket of the largest level is the unsplittable unit of stage-B work (one workgroup, ~n * 2^D / nb_max records
	// when the points are spread out).  Smaller levels are replicated until their workgroups are about that size too
	// (rounded down: a few larger workgroups, launched first, pack better than many that spill into another round),
	// and the levels are launched by decreasing workgroup size.
	uint32_t nb_max = 1;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) nb_max = plan.nb[q] > nb_max ? plan.nb[q] : nb_max;
	const uint32_t unit = nb_max < 64 ? 64 : nb_max;              // at least 64 workgroups per level to cover the chip
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
		uint32_t rep = unit / plan.nb[q];
		rep = rep < 1 ? 1 : rep;
		if (rep > plan.n_blk) rep = plan.n_blk ? plan.n_blk : 1u;
		plan.rep[q] = rep;
		plan.order[q] = q;
	}
	for (uint32_t i = 1; i < m->n_pseudo_levels; ++i) {           // insertion sort by nb * rep ascending (= work descending)
		const uint32_t q = plan.order[i];
		const uint32_t key = plan.nb[q] * plan.rep[q];
		uint32_t j = i;
		while (j > 0 && plan.nb[plan.order[j - 1]] * plan.rep[plan.order[j - 1]] > key) { plan.order[j] = plan.order[j - 1]; --j; }
		plan.order[j] = q;
	}
	return true;
}

struct BinLayout { uint64_t rec_bytes, offs_bytes, gt_bytes, total; };

static BinLayout layout(const nr3d_lotd_meta_t *m, const BinPlan &plan, uint64_t offs_words, uint32_t n_chunk) {
	BinLayout l;
	const uint64_t cap = (uint64_t)bin_points(m->n_dims_to_encode, m->n_feat_per_pseudo_lvl) << m->n_dims_to_encode;
	l.rec_bytes = (uint64_t)m->n_pseudo_levels * plan.n_blk * (1 + m->n_feat_per_pseudo_lvl) * cap * 4;
	l.offs_bytes = ((offs_words * 4 + 255) / 256) * 256;
	l.gt_bytes = (((uint64_t)m->n_encoded_dims * n_chunk * 4 + 255) / 256) * 256;
	l.total = l.rec_bytes + l.offs_bytes + l.gt_bytes;
	return l;
}

// returns 0 when the binned path does not apply (caller falls back to the atomic kernels)
uint64_t dparam_workspace_bytes(const nr3d_lotd_meta_t *m, uint32_t n_points) {
	if (!m || n_points == 0) return 0;
	BinPlan plan;
	uint64_t offs_words;
	const uint32_t nc = chunk_points(n_points);
	if (!make_plan(m, nc, plan, offs_words)) return 0;
	return layout(m, plan, offs_words, nc).total;
}

int dparam_binned(bool second, const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, const float *dL_ddLdx,
                  const float *dL_dy, int64_t g_sn, int64_t g_se, const float *x, int32_t max_level, float *dparam,
                  void *workspace, uint64_t workspace_bytes, hipStream_t st, bool &handled) {
	handled = false;
	BinPlan plan;
	uint64_t offs_words;
	const uint32_t nc = chunk_points(N);
	if (!workspace || !make_plan(meta, nc, plan, offs_words)) return 0;
	const BinLayout lay = layout(meta, plan, offs_words, nc);
	if (workspace_bytes < lay.total) return 0;
	handled = true;
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	const uint32_t D = meta->n_dims_to_encode, G = meta->n_feat_per_pseudo_lvl, E = meta->n_encoded_dims;
	uint32_t *rec = (uint32_t *)workspace;
	uint32_t *offs = (uint32_t *)((char *)workspace + lay.rec_bytes);
	float *gt = (float *)((char *)workspace + lay.rec_bytes + lay.offs_bytes);
	uint32_t nb_max = 0, acc_max = 0;
	for (uint32_t q = 0; q < plan.n_pseudo; ++q) {
		nb_max = nb_max > plan.nb[q] ? nb_max : plan.nb[q];
		const uint32_t a = plan.nb[q] * plan.rep[q];
		acc_max = acc_max > a ? acc_max : a;
	}
	const bool row_major = (g_se == 1 && g_sn == (int64_t)E && E > 1);

	for (uint32_t p0 = 0; p0 < N; p0 += nc) {
		const uint32_t n = (N - p0) < nc ? (N - p0) : nc;
		BinPlan pl = plan;
		if (n != nc) {   // last, shorter chunk: fewer point blocks (offset tables shrink, bases stay valid upper bounds)
			uint64_t ow;
			make_plan(meta, n, pl, ow);
		}
		const float *xc = x + (size_t)p0 * D;
		const float *vc = dL_ddLdx ? dL_ddLdx + (size_t)p0 * D : nullptr;
		const float *gc = dL_dy + (int64_t)p0 * g_sn;
		int64_t sn = g_sn, se = g_se;
		if (row_major) {
			hipLaunchKernelGGL(k_transpose, dim3(div_up(n, 32), div_up(E, 32)), dim3(256), 0, st, n, E, gc, g_sn, g_se, gt);
			gc = gt; sn = 1; se = (int64_t)n;
		}
		DISPATCH_DG_BIN(D, G, {
			constexpr int BP = BinCfg<D, G>::BP;
			const size_t bin_lds = ((size_t)(1 + G) * BinCfg<D, G>::cap + nb_max + 1) * sizeof(uint32_t);
			static bool attr_set = false;
			if (!attr_set) {
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_accum<D, G>, hipFuncAttributeMaxDynamicSharedMemorySize,
				                                   kLdsDoubles * 8));
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin<D, G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + (kMaxBuckets + 1) * 4));
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_bin<D, G, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + (kMaxBuckets + 1) * 4));
				attr_set = true;
			}
			if (second)
				hipLaunchKernelGGL((k_bin<D, G, true>), dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n,
				                   max_level, meta->interpolation_type, xc, vc, gc, sn, se, rec, offs);
			else
				hipLaunchKernelGGL((k_bin<D, G, false>), dim3(pl.n_blk, pl.n_pseudo), dim3(BP), bin_lds, st, pl, md, n,
				                   max_level, meta->interpolation_type, xc, vc, gc, sn, se, rec, offs);
			hipLaunchKernelGGL((k_accum<D, G>), dim3(acc_max, pl.n_pseudo), dim3(kAccThreads), kLdsDoubles * 8, st, pl, md,
			                   rec, offs, dparam);
		});
		NR3D_LAUNCH_CHECK();
	}
	return 0;
}

}  // namespace lotd
}  // namespace nr3d
