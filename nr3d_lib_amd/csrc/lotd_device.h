// nr3d_lib_amd/csrc/lotd_device.h -- device-side building blocks of the LoTD kernels (gfx950).
//
// Semantics follow the reference's device math (csrc/lotd/include/lotd/lotd_cuda.h index/value
// functions :92-492, pos_fract :959-1077; corner loops csrc/lotd/include/lotd/lotd_encoding.h),
// re-organised for CDNA4: one lane owns one (point, pseudo-level) pair, keeps all 2^D corner values
// of a feature group in VGPRs (so y and dy/dx come from ONE set of gathers), and the per-level
// descriptor is fetched with scalar loads from a device copy of the meta.
//
// The library is compiled with -ffp-contract=off; every fused multiply-add below is an explicit
// fmaf().  The one that decides integer results is the cell locator (x*scale+0.5 -> floor), fused
// exactly like the reference's nvcc build.
#pragma once
#include "common.h"

namespace nr3d {
namespace lotd {

constexpr int kBlock = 256;
constexpr uint32_t kPrimes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};

// ---------------------------------------------------------------------------------------------
// Block -> (pseudo level, point chunk) schedule.
//   mode 0  level-major       : q = b / n_chunks
//   mode 1  XCD-affine        : workgroup b runs on XCD (b % 8) (observed dispatch order; only used
//                               for cache affinity, never for correctness).  Each XCD walks the
//                               levels {xcd, xcd+8, ...} one after another, so one 4 MiB hash table
//                               at a time is live in that XCD's private 4 MiB L2.
//   mode 2  chunk-major       : q = b % n_pseudo (all levels of a chunk back to back)
//   mode 3  XCD-affine, cost-balanced (default): the (level, chunk) items laid out level after level are cut into
//                               8 contiguous pieces of equal estimated COST (L2 gather requests per point differ
//                               between levels: 4 for Dense, 6 for Hash with the paired 16-byte gathers, half of
//                               that when the table fits the L1).  An XCD still walks few tables, one at a time.
// ---------------------------------------------------------------------------------------------
constexpr int kSchedSegs = 16;             // max (level, chunk range) segments per XCD in mode 3
struct Sched {
	uint32_t n_chunks, n_pseudo, mode, n_slots;
	uint64_t skip;                         // bit q set: pseudo level q is served by another launch (LDS-staged forward)
	uint32_t seg_cum[8][kSchedSegs + 1];   // blocks of this XCD before segment i
	uint32_t seg_begin[8][kSchedSegs];     // first chunk of segment i
	uint32_t seg_q[8][kSchedSegs];         // pseudo level of segment i (32-bit: a scalar load; 16-bit fields go through the vector L1)
};

__device__ __forceinline__ bool decode_block_raw(const Sched &s, uint32_t b, uint32_t &q, uint32_t &chunk);
__device__ __forceinline__ bool decode_block(const Sched &s, uint32_t b, uint32_t &q, uint32_t &chunk) {
	if (!decode_block_raw(s, b, q, chunk)) return false;
	return s.mode == 3 || q >= 64u || !((s.skip >> q) & 1ull);      // mode 3 leaves skipped levels out of the work line
}
__device__ __forceinline__ bool decode_block_raw(const Sched &s, uint32_t b, uint32_t &q, uint32_t &chunk) {
	if (s.mode == 3) {
		// branch-free: the XCD's row of segment starts is one wide scalar load, the segment is the number of starts <= j
		// (a loop with an early exit compiles to one dependent scalar load + wait per segment: ~1.5k cycles before a wave's
		// first vector instruction)
		const uint32_t xcd = b & 7u, j = b >> 3;
		uint32_t idx = 0;
#pragma unroll
		for (int i = 1; i < kSchedSegs; ++i) idx += (j >= s.seg_cum[xcd][i]) ? 1u : 0u;
		if (j >= s.seg_cum[xcd][kSchedSegs]) return false;
		q = s.seg_q[xcd][idx];
		chunk = s.seg_begin[xcd][idx] + (j - s.seg_cum[xcd][idx]);
		return true;
	} else if (s.mode == 1) {
		const uint32_t xcd = b & 7u, j = b >> 3;
		const uint32_t slot = j / s.n_chunks;
		chunk = j - slot * s.n_chunks;
		q = slot * 8u + xcd;
		return q < s.n_pseudo;
	} else if (s.mode == 2) {
		chunk = b / s.n_pseudo;
		q = b - chunk * s.n_pseudo;
		return true;
	}
	q = b / s.n_chunks;
	chunk = b - q * s.n_chunks;
	return true;
}

struct Batch {
	const int64_t *inds;
	const int64_t *offsets;
	uint32_t data_size;
	uint32_t n_params;
	uint32_t first_point = 0;   // index of point 0 of this launch inside the caller's batch (chunked launches)
};

// returns false when the point is skipped (batch index < 0); base = element offset of the batch entry
__device__ __forceinline__ bool batch_base(const Batch &b, uint32_t i, uint32_t &base) {
	uint32_t bi = 0;
	if (b.inds) {
		const int64_t v = b.inds[i];
		if (v < 0) return false;
		bi = (uint32_t)v;
	} else if (b.data_size) {
		bi = i / b.data_size;
	}
	base = b.offsets ? (uint32_t)b.offsets[bi] : bi * b.n_params;
	return true;
}

// same, also returning the batch entry index
__device__ __forceinline__ bool batch_base_index(const Batch &b, uint32_t i, uint32_t &base, uint32_t &bi) {
	bi = 0;
	if (b.inds) {
		const int64_t v = b.inds[i];
		if (v < 0) return false;
		bi = (uint32_t)v;
	} else if (b.data_size) {
		bi = (b.first_point + i) / b.data_size;
	}
	base = b.offsets ? (uint32_t)b.offsets[bi] : bi * b.n_params;
	return true;
}

struct Lvl {
	uint32_t res[4];
	uint32_t F, type, size, off;
};

// The device copy of the meta is read-only for every kernel and indexed wave-uniformly (the block's level): reading it
// through the constant address space makes these scalar loads (s_load_dwordx8 via the scalar cache, results in SGPRs)
// instead of 64-lane vector loads.  There is no sub-dword scalar load: uint16 tables are read as dwords and shifted.
typedef const __attribute__((address_space(4))) uint32_t *cu32_t;

__device__ __forceinline__ Lvl load_level(const nr3d_lotd_meta_t *__restrict__ md, uint32_t level) {
	Lvl L;
	const cu32_t w = (cu32_t)(&md->levels[level]);     // {res[4], n_feats, type, size, offset}
	static_assert(sizeof(nr3d_lotd_level_t) == 32, "level descriptor layout");
#pragma unroll
	for (int d = 0; d < 4; ++d) L.res[d] = w[d];
	L.F = w[4]; L.type = w[5]; L.size = w[6]; L.off = w[7];
	return L;
}

__device__ __forceinline__ uint32_t meta_u16(const uint16_t *tab, uint32_t i) {
	const cu32_t w = (cu32_t)tab;                      // the tables are 4-byte aligned inside the meta struct
	return (w[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t meta_level_of(const nr3d_lotd_meta_t *__restrict__ md, uint32_t q) { return meta_u16(md->map_levels, q); }
__device__ __forceinline__ uint32_t meta_cnt_of(const nr3d_lotd_meta_t *__restrict__ md, uint32_t q) { return meta_u16(md->map_cnt, q); }
// first output column of pseudo level q (q * G for a plain meta; the original layout's column for a regrouped one)
__device__ __forceinline__ uint32_t meta_col_of(const nr3d_lotd_meta_t *__restrict__ md, uint32_t q) { return meta_u16(md->map_col, q); }

// ---------------------------------------------------------------------------------------------
// Cell locator: g = floor(x*(R-2)+0.5), t = frac, plus interpolation weight and its derivatives.
// ---------------------------------------------------------------------------------------------
template <int D>
struct Cell {
	uint32_t g[D];
	float w[D], dw[D], ddw[D], sc[D];
};

template <int D>
__device__ __forceinline__ void locate(const float *__restrict__ xp, const Lvl &L, bool smooth, Cell<D> &c) {
#pragma unroll
	for (int d = 0; d < D; ++d) {
		const float sc = (float)(L.res[d] - 2u);
		const float v = __fmaf_rn(xp[d], sc, 0.5f);
		const float fl = floorf(v);
		const float t = v - fl;
		c.sc[d] = sc;
		c.g[d] = (uint32_t)fl;
		if (!smooth) {
			c.w[d] = t; c.dw[d] = 1.0f; c.ddw[d] = 0.0f;
		} else {
			c.w[d] = t * t * __fmaf_rn(-2.0f, t, 3.0f);
			c.dw[d] = 6.0f * t * (1.0f - t);
			c.ddw[d] = __fmaf_rn(-12.0f, t, 6.0f);
		}
	}
}

// weight of corner `c` (bit d set => upper cell along d), product taken in dim order from 1
template <int D>
__device__ __forceinline__ float corner_weight(const Cell<D> &c, uint32_t corner) {
	float w = 1.0f;
#pragma unroll
	for (int d = 0; d < D; ++d) w *= ((corner >> d) & 1u) ? c.w[d] : (1.0f - c.w[d]);
	return w;
}

// weight over all dims except `skip`, seeded with `seed`
template <int D>
__device__ __forceinline__ float face_weight(const Cell<D> &c, uint32_t corner, int skip, float seed) {
	float w = seed;
#pragma unroll
	for (int d = 0; d < D; ++d)
		if (d != skip) w *= ((corner >> d) & 1u) ? c.w[d] : (1.0f - c.w[d]);
	return w;
}

template <int D>
__device__ __forceinline__ void corner_pos(const Cell<D> &c, uint32_t corner, uint32_t (&p)[D]) {
#pragma unroll
	for (int d = 0; d < D; ++d) p[d] = c.g[d] + ((corner >> d) & 1u);
}

// ---------------------------------------------------------------------------------------------
// Entry indices (in ENTRIES, i.e. before "* F + feature"); all uint32 wrap-around arithmetic.
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ uint32_t entry_dense(const Lvl &L, const uint32_t (&p)[D]) {
	uint32_t e = p[0];
#pragma unroll
	for (int d = 1; d < D; ++d) e = e * L.res[d] + p[d];   // last dim contiguous
	return e;
}

template <int D>
__device__ __forceinline__ uint32_t entry_hash(const Lvl &L, const uint32_t (&p)[D]) {
	uint32_t h = 0;
#pragma unroll
	for (int d = 0; d < D; ++d) h ^= p[d] * kPrimes[d];
	return ((L.size & (L.size - 1u)) == 0u) ? (h & (L.size - 1u)) : (h % L.size);
}

// CP / VM "line" tables: [ line_0 | line_1 | ... ]
template <int D>
__device__ __forceinline__ uint32_t entry_line(const Lvl &L, int dim, uint32_t pos) {
	uint32_t acc = 0;
#pragma unroll
	for (int d = 0; d < D; ++d) if (d < dim) acc += L.res[d];
	return acc + pos;
}

// plane spanned by all dims but `skip`, later dim contiguous; `stride_out` = number of entries
template <int D>
__device__ __forceinline__ uint32_t plane_local(const Lvl &L, int skip, const uint32_t (&p)[D], uint32_t &plane_size) {
	uint32_t e = 0, sz = 1;
	bool first = true;
#pragma unroll
	for (int d = 0; d < D; ++d) {
		if (d == skip) continue;
		e = first ? p[d] : e * L.res[d] + p[d];
		sz *= L.res[d];
		first = false;
	}
	plane_size = sz;
	return e;
}

// VM (D == 3): entries of plane_d and line_d for corner p.  Layout: [x,y,z lines | yz, xz, xy planes]
__device__ __forceinline__ void entry_vm(const Lvl &L, const uint32_t (&p)[3], uint32_t (&pl)[3], uint32_t (&ln)[3]) {
	const uint32_t lines = L.res[0] + L.res[1] + L.res[2];
	ln[0] = p[0]; ln[1] = L.res[0] + p[1]; ln[2] = L.res[0] + L.res[1] + p[2];
	uint32_t acc = lines, sz;
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		pl[d] = acc + plane_local<3>(L, d, p, sz);
		acc += sz;
	}
}

// NPlaneMul plane for reference "jump_dim" j: spans all dims except (D-1-j), NO per-plane base offset
// (reference quirk: the planes alias, lotd_cuda.h:176-206)
template <int D>
__device__ __forceinline__ uint32_t entry_nplane_mul(const Lvl &L, int j, const uint32_t (&p)[D]) {
	uint32_t sz;
	return plane_local<D>(L, D - 1 - j, p, sz);
}

// NPlaneSum sub-plane lookup, reference grid_index_nplane_sub (lotd_cuda.h:145-174), kept as-is
// including its cubic-resolution assumption for the strides.
template <int D>
__device__ __forceinline__ uint32_t entry_nplane_sum(const Lvl &L, uint32_t jump, const uint32_t (&pp)[D]) {
	constexpr int N1 = D - 1;
	uint32_t stride = 1, e = 0;
#pragma unroll
	for (int d2 = 0; d2 < N1; ++d2) {
		const int d3 = (uint32_t)d2 >= jump ? d2 + 1 : d2;
		e += pp[N1 - 1 - d2] * stride;
		stride *= L.res[N1 - d3];
	}
	return jump * stride + e;
}

// ---------------------------------------------------------------------------------------------
// A level's table read as floats, whatever it is stored as.  `TB` in the helpers below is either `const float *` itself or
// HalfTab over __half storage -- the reference's (float, half, float) dispatch (lotd_encoding.h:1501-1504): parameters are
// READ as half and every product / interpolation runs in fp32, so a half table gives exactly what its fp32 copy gives.
// ---------------------------------------------------------------------------------------------
struct HalfTab {
	const __half *p;
	__device__ __forceinline__ float operator[](size_t i) const { return __half2float(p[i]); }
	__device__ __forceinline__ HalfTab operator+(size_t n) const { return HalfTab{p + n}; }
};
__device__ __forceinline__ const float *make_tab(const float *p) { return p; }
__device__ __forceinline__ HalfTab make_tab(const __half *p) { return HalfTab{p}; }
__device__ __forceinline__ uintptr_t tab_addr(const float *g) { return reinterpret_cast<uintptr_t>(g); }
__device__ __forceinline__ uintptr_t tab_addr(HalfTab g) { return reinterpret_cast<uintptr_t>(g.p); }
// bytes per element of the table behind a TB
__device__ __forceinline__ constexpr uint32_t tab_elt(const float *) { return 4u; }
__device__ __forceinline__ constexpr uint32_t tab_elt(HalfTab) { return 2u; }
struct __attribute__((aligned(8))) Pair16 { float x, y, z, w; };   // two neighbouring F=2 entries, 8-byte aligned
struct __attribute__((aligned(4))) HalfPair8 { __half2 a, b; };    // the same over half storage, 4-byte aligned
// 2 / 4 consecutive elements from an address aligned to that many elements (tab_ld4_pair: to HALF that many)
__device__ __forceinline__ float2 tab_ld2(const float *g, size_t i) { return *reinterpret_cast<const float2 *>(g + i); }
__device__ __forceinline__ float2 tab_ld2(HalfTab g, size_t i) { return __half22float2(*reinterpret_cast<const __half2 *>(g.p + i)); }
__device__ __forceinline__ float4 tab_ld4(const float *g, size_t i) { return *reinterpret_cast<const float4 *>(g + i); }
__device__ __forceinline__ float4 tab_ld4(HalfTab g, size_t i) {
	struct __attribute__((aligned(8))) H4 { __half2 a, b; };
	const H4 t = *reinterpret_cast<const H4 *>(g.p + i);
	const float2 lo = __half22float2(t.a), hi = __half22float2(t.b);
	return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 tab_ld4_pair(const float *g, size_t i) {
	const Pair16 t = *reinterpret_cast<const Pair16 *>(g + i);
	return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float4 tab_ld4_pair(HalfTab g, size_t i) {
	const HalfPair8 t = *reinterpret_cast<const HalfPair8 *>(g.p + i);
	const float2 lo = __half22float2(t.a), hi = __half22float2(t.b);
	return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// ---------------------------------------------------------------------------------------------
// Value of NF consecutive features at one corner, for the N-linear level types.
// ---------------------------------------------------------------------------------------------
template <int D, int NF, typename TB>
__device__ __forceinline__ void corner_value(const Lvl &L, TB grid, uint32_t foff,
                                             const uint32_t (&p)[D], float (&v)[NF]) {
	switch (L.type) {
	case NR3D_LOD_Dense: {
		const uint32_t i = entry_dense<D>(L, p) * L.F + foff;
#pragma unroll
		for (int f = 0; f < NF; ++f) v[f] = grid[i + f];
	} break;
	case NR3D_LOD_Hash: {
		const uint32_t i = entry_hash<D>(L, p) * L.F + foff;
#pragma unroll
		for (int f = 0; f < NF; ++f) v[f] = grid[i + f];
	} break;
	case NR3D_LOD_VectorMatrix:
		if constexpr (D == 3) {
			uint32_t pl[3], ln[3];
			entry_vm(L, p, pl, ln);
#pragma unroll
			for (int f = 0; f < NF; ++f) v[f] = 0.0f;
#pragma unroll
			for (int d = 0; d < 3; ++d)
#pragma unroll
				for (int f = 0; f < NF; ++f)
					v[f] = __fmaf_rn(grid[pl[d] * L.F + foff + f], grid[ln[d] * L.F + foff + f], v[f]);
		}
		break;
	case NR3D_LOD_VecZMatXoY:
		if constexpr (D == 3) {
			const uint32_t ln = p[2] * L.F + foff;
			const uint32_t pl = (L.res[2] + p[1] + p[0] * L.res[0]) * L.F + foff;   // stride R_x, as the reference
#pragma unroll
			for (int f = 0; f < NF; ++f) v[f] = grid[pl + f] * grid[ln + f];
		}
		break;
	case NR3D_LOD_CP: {
#pragma unroll
		for (int f = 0; f < NF; ++f) v[f] = grid[entry_line<D>(L, 0, p[0]) * L.F + foff + f];
#pragma unroll
		for (int d = 1; d < D; ++d)
#pragma unroll
			for (int f = 0; f < NF; ++f) v[f] *= grid[entry_line<D>(L, d, p[d]) * L.F + foff + f];
	} break;
	case NR3D_LOD_NPlaneMul: {
#pragma unroll
		for (int f = 0; f < NF; ++f) v[f] = grid[entry_nplane_mul<D>(L, 0, p) * L.F + foff + f];
#pragma unroll
		for (int j = 1; j < D; ++j)
#pragma unroll
			for (int f = 0; f < NF; ++f) v[f] *= grid[entry_nplane_mul<D>(L, j, p) * L.F + foff + f];
	} break;
	default:
#pragma unroll
		for (int f = 0; f < NF; ++f) v[f] = 0.0f;
	}
}

// bits of corner k without bit `skip`, packed
template <int D>
__device__ __forceinline__ constexpr uint32_t drop_bit(uint32_t k, int skip) {
	return (k & ((1u << skip) - 1u)) | ((k >> (skip + 1)) << skip);
}
// inverse: m with a zero bit inserted at position `at`
__device__ __forceinline__ constexpr uint32_t insert_zero(uint32_t m, int at) {
	return (m & ((1u << at) - 1u)) | ((m >> at) << (at + 1));
}

// two consecutive features of one table entry (8-byte load when the table base allows it)
template <typename TB>
__device__ __forceinline__ void ld_pair(TB grid, uint32_t idx, bool vec, float (&o)[2]) {
	if (vec) { const float2 t = tab_ld2(grid, idx); o[0] = t.x; o[1] = t.y; }
	else { o[0] = grid[idx]; o[1] = grid[idx + 1]; }
}

// ---------------------------------------------------------------------------------------------
// Values of 2 consecutive features at ALL 2^D corners of a cell.  Product types read each distinct table entry once
// (CP: 2D lines entries instead of D per corner; VM: 12 plane + 6 line entries instead of 6 per corner) and then form
// the corner values with exactly corner_value()'s arithmetic, so the results are bit-identical to the per-corner form.
// ---------------------------------------------------------------------------------------------
// ONLY >= 0: the instantiation serves levels of that one type (the other types' code and registers drop out);
// kOnlyDenseHash: Dense and Hash levels only (hash-only metas)
constexpr int kOnlyDenseHash = -2;
// NF consecutive features of one table entry: one 8-byte (NF = 2) / 16-byte (NF = 4) load when the alignment allows
template <int NF, typename TB>
__device__ __forceinline__ void ld_feats(TB grid, uint32_t idx, bool vec, float (&o)[NF]) {
	if constexpr (NF == 2) { ld_pair(grid, idx, vec, o); }
	else {
		static_assert(NF == 4, "ld_feats: 2 or 4 features");
		if (vec) { const float4 t = tab_ld4(grid, idx); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
		else { o[0] = grid[idx]; o[1] = grid[idx + 1]; o[2] = grid[idx + 2]; o[3] = grid[idx + 3]; }
	}
}

// NF = 4 (round 3): the product-type levels of a wide pseudo level read 16 bytes per table entry instead of two 8-byte
// pieces -- the forward of configs[3]'s CP levels was bound by L2 requests (356 M per launch = 244 G/s, profiles/
// r03g_c4_counters.txt), 24 per (point, 8-feature pseudo level) for 6 distinct 32-byte entries.
template <int D, int ONLY = -1, int NF = 2, typename TB = const float *>
__device__ __forceinline__ void corner_values_pair(const Lvl &L, TB grid, uint32_t foff, bool vec,
                                                   const Cell<D> &c, float (&v)[1 << D][NF]) {
	constexpr uint32_t C = 1u << D;
	if constexpr (ONLY == kOnlyDenseHash) {                // hash-only metas: the values corner_value() loads
#pragma unroll
		for (uint32_t k = 0; k < C; ++k) {
			uint32_t p[D];
			corner_pos<D>(c, k, p);
			const uint32_t i = ((L.type == NR3D_LOD_Dense) ? entry_dense<D>(L, p) : entry_hash<D>(L, p)) * L.F + foff;
			ld_feats<NF>(grid, i, vec, v[k]);                 // one request when the alignment allows
		}
		return;
	}
	if (ONLY == NR3D_LOD_CP || (ONLY < 0 && L.type == NR3D_LOD_CP)) {
		float t[D][2][NF];
#pragma unroll
		for (int d = 0; d < D; ++d)
#pragma unroll
			for (uint32_t sl = 0; sl < 2; ++sl) ld_feats<NF>(grid, entry_line<D>(L, d, c.g[d] + sl) * L.F + foff, vec, t[d][sl]);
#pragma unroll
		for (uint32_t k = 0; k < C; ++k)
#pragma unroll
			for (int f = 0; f < NF; ++f) {
				float r = t[0][k & 1u][f];
#pragma unroll
				for (int d = 1; d < D; ++d) r *= t[d][(k >> d) & 1u][f];
				v[k][f] = r;
			}
		return;
	}
	if constexpr (D <= 3 && ONLY < 0) {
		if (L.type == NR3D_LOD_NPlaneMul) {
			constexpr uint32_t NS = 1u << (D - 1);
			float t[D][NS][NF];
#pragma unroll
			for (int j = 0; j < D; ++j)
#pragma unroll
				for (uint32_t sl = 0; sl < NS; ++sl) {
					uint32_t p[D];
					corner_pos<D>(c, insert_zero(sl, D - 1 - j), p);
					ld_feats<NF>(grid, entry_nplane_mul<D>(L, j, p) * L.F + foff, vec, t[j][sl]);
				}
#pragma unroll
			for (uint32_t k = 0; k < C; ++k)
#pragma unroll
				for (int f = 0; f < NF; ++f) {
					float r = t[0][drop_bit<D>(k, D - 1)][f];
#pragma unroll
					for (int j = 1; j < D; ++j) r *= t[j][drop_bit<D>(k, D - 1 - j)][f];
					v[k][f] = r;
				}
			return;
		}
	}
	if constexpr (D == 3 && (ONLY < 0 || ONLY == NR3D_LOD_VectorMatrix)) {
		if (ONLY == NR3D_LOD_VectorMatrix || L.type == NR3D_LOD_VectorMatrix || L.type == NR3D_LOD_VecZMatXoY) {
			const bool vm = ONLY == NR3D_LOD_VectorMatrix || L.type == NR3D_LOD_VectorMatrix;
#pragma unroll
			for (uint32_t k = 0; k < C; ++k)
#pragma unroll
				for (int f = 0; f < NF; ++f) v[k][f] = 0.0f;
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				if (!vm && d != 2) continue;
				float pv[4][NF], lv[2][NF];
				uint32_t le0 = 0;
#pragma unroll
				for (uint32_t m = 0; m < 4; ++m) {
					uint32_t p[3];
					corner_pos<3>(c, insert_zero(m, d), p);
					uint32_t pe;
					if (vm) { uint32_t pl[3], ln[3]; entry_vm(L, p, pl, ln); pe = pl[d]; if (m == 0) le0 = ln[d]; }
					else { pe = L.res[2] + p[1] + p[0] * L.res[0]; if (m == 0) le0 = p[2]; }
					ld_feats<NF>(grid, pe * L.F + foff, vec, pv[m]);
				}
				ld_feats<NF>(grid, le0 * L.F + foff, vec, lv[0]);
				ld_feats<NF>(grid, (le0 + 1u) * L.F + foff, vec, lv[1]);
#pragma unroll
				for (uint32_t k = 0; k < C; ++k)
#pragma unroll
					for (int f = 0; f < NF; ++f) {
						const float pvv = pv[drop_bit<3>(k, d)][f], lvv = lv[(k >> d) & 1u][f];
						v[k][f] = vm ? __fmaf_rn(pvv, lvv, v[k][f]) : pvv * lvv;
					}
			}
			return;
		}
	}
	if constexpr (ONLY < 0) {
#pragma unroll
		for (uint32_t k = 0; k < C; ++k) {
			uint32_t p[D];
			corner_pos<D>(c, k, p);
			corner_value<D, NF>(L, grid, foff, p, v[k]);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Scatter of (grad[f] * weight) at one corner into the parameter-gradient buffer, per level type.
// Product types multiply by the other factors read from `grid` (lotd_cuda.h:494-829).
// ---------------------------------------------------------------------------------------------
template <int D, int NF, typename TB>
__device__ __forceinline__ void corner_scatter(const Lvl &L, TB grid, float *__restrict__ gg,
                                               uint32_t foff, const uint32_t (&p)[D], const float (&grad)[NF],
                                               float weight) {
	switch (L.type) {
	case NR3D_LOD_Dense: {
		const uint32_t i = entry_dense<D>(L, p) * L.F + foff;
#pragma unroll
		for (int f = 0; f < NF; ++f) atomic_add_f32(gg + i + f, grad[f] * weight);
	} break;
	case NR3D_LOD_Hash: {
		const uint32_t i = entry_hash<D>(L, p) * L.F + foff;
#pragma unroll
		for (int f = 0; f < NF; ++f) atomic_add_f32(gg + i + f, grad[f] * weight);
	} break;
	case NR3D_LOD_VectorMatrix:
		if constexpr (D == 3) {
			uint32_t pl[3], ln[3];
			entry_vm(L, p, pl, ln);
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				const uint32_t ip = pl[d] * L.F + foff, il = ln[d] * L.F + foff;
#pragma unroll
				for (int f = 0; f < NF; ++f) {
					const float wg = grad[f] * weight;
					atomic_add_f32(gg + ip + f, wg * grid[il + f]);
					atomic_add_f32(gg + il + f, wg * grid[ip + f]);
				}
			}
		}
		break;
	case NR3D_LOD_VecZMatXoY:
		if constexpr (D == 3) {
			const uint32_t il = p[2] * L.F + foff;
			const uint32_t ip = (L.res[2] + p[1] + p[0] * L.res[0]) * L.F + foff;
#pragma unroll
			for (int f = 0; f < NF; ++f) {
				const float wg = grad[f] * weight;
				atomic_add_f32(gg + ip + f, wg * grid[il + f]);
				atomic_add_f32(gg + il + f, wg * grid[ip + f]);
			}
		}
		break;
	case NR3D_LOD_CP:
	case NR3D_LOD_NPlaneMul: {
		uint32_t idx[D];
#pragma unroll
		for (int d = 0; d < D; ++d)
			idx[d] = (L.type == NR3D_LOD_CP ? entry_line<D>(L, d, p[d]) : entry_nplane_mul<D>(L, d, p)) * L.F + foff;
#pragma unroll
		for (int gd = 0; gd < D; ++gd)
#pragma unroll
			for (int f = 0; f < NF; ++f) {
				float cur = grad[f] * weight;
#pragma unroll
				for (int k = 0; k < D; ++k) if (k != gd) cur *= grid[idx[k] + f];
				atomic_add_f32(gg + idx[gd] + f, cur);
			}
	} break;
	default: break;
	}
}

// sum_f value(corner)[f] * grad[f] * weight, for the types that have a d(dL/dx)/dx path
// (lotd_cuda.h:831-957)
template <int D, int NF, typename TB>
__device__ __forceinline__ float corner_dot(const Lvl &L, TB grid, uint32_t foff,
                                            const uint32_t (&p)[D], const float (&grad)[NF], float weight) {
	float v[NF];
	corner_value<D, NF>(L, grid, foff, p, v);
	float r = 0.0f;
#pragma unroll
	for (int f = 0; f < NF; ++f) r = __fmaf_rn(v[f] * grad[f], weight, r);
	return r;
}

// ---------------------------------------------------------------------------------------------
// Paired 16-byte gathers of F == 2 Dense / Hash levels (forward kernels)
// ---------------------------------------------------------------------------------------------
// The gather rate is bound by L2->L1 line requests (one per lane and instruction, ~260 G/s chip-wide,
// tools/ubench_mem), not by bytes.  Corner pairs that are neighbours in memory come from ONE 16-byte load:
//   Dense -> the pair along the contiguous last dim (always neighbours, 8-byte aligned 16-byte load);
//   Hash  -> the pair along dim 0 (prime 1): for even x0 (and a power-of-two table)
//            hash(x0 + 1, ..) == hash(x0, ..) ^ 1, the other half of the same aligned 16-byte slot.
// Lanes whose partner lives elsewhere fetch it with a second 8-byte load, all of them under ONE branch so that
// no load has to be waited for before the last one is issued.  F == 2 (8-byte entries) only.
template <int D, bool DENSE, typename TB>
__device__ __forceinline__ void gather_pairs(const Lvl &L, const Cell<D> &c, TB grid,
                                             float (&v)[1 << D][2]) {
	constexpr uint32_t PBIT = DENSE ? (1u << (D - 1)) : 1u;
	uint32_t e1s[1 << (D - 1)];
	bool all_adj = true;
#pragma unroll
	for (uint32_t m = 0; m < (1u << (D - 1)); ++m) {
		const uint32_t k0 = DENSE ? m : (m << 1);          // m with a zero bit inserted at the pair dimension
		const uint32_t k1 = k0 | PBIT;
		uint32_t p0[D], p1[D];
		corner_pos<D>(c, k0, p0);
		corner_pos<D>(c, k1, p1);
		const uint32_t e0 = DENSE ? entry_dense<D>(L, p0) : entry_hash<D>(L, p0);
		const uint32_t e1 = DENSE ? e0 + 1u : entry_hash<D>(L, p1);
		const uint32_t base = DENSE ? e0 : min(e0 & ~1u, L.size - 2u);
		const float4 t = tab_ld4_pair(grid, (size_t)base * 2u);
		const bool hi0 = (e0 != base);
		v[k0][0] = hi0 ? t.z : t.x;
		v[k0][1] = hi0 ? t.w : t.y;
		const uint32_t o1 = e1 - base;
		v[k1][0] = (o1 == 1u) ? t.z : t.x;
		v[k1][1] = (o1 == 1u) ? t.w : t.y;
		all_adj = all_adj && (o1 < 2u);
		e1s[m] = e1;
	}
	if (!DENSE && !all_adj) {
#pragma unroll
		for (uint32_t m = 0; m < (1u << (D - 1)); ++m) {
			const uint32_t k1 = (m << 1) | 1u;
			const float2 t = tab_ld2(grid, (size_t)e1s[m] * 2u);
			v[k1][0] = t.x;
			v[k1][1] = t.y;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Forest of blocks (csrc/forest/forest.h, lotd_forest.h): octree lookup, corner resolver, cell locator
// ---------------------------------------------------------------------------------------------
struct ForestDev {
	const uint8_t *__restrict__ octree;
	const int32_t *__restrict__ exsum;
	const int16_t *__restrict__ block_ks;
	uint32_t level, level_poffset, continuity;
};

// forest.h:25-58: walk the byte octree from the root to `level`; index of the node in the breadth-first hierarchy
__device__ __forceinline__ int32_t identify(const ForestDev &fo, int kx, int ky, int kz) {
	const int maxval = (1 << fo.level) - 1;
	if (kx < 0 || ky < 0 || kz < 0 || kx > maxval || ky > maxval || kz > maxval) return -1;
	int32_t ord = 0;
	for (uint32_t l = 0; l < fo.level; ++l) {
		const uint32_t depth = fo.level - l - 1;
		const uint32_t child = (((uint32_t)kx >> depth) & 1u) << 2 | (((uint32_t)ky >> depth) & 1u) << 1 | (((uint32_t)kz >> depth) & 1u);
		const uint32_t bits = fo.octree[ord];
		if (!(bits & (1u << child))) return -1;
		ord = fo.exsum[ord] + (int32_t)__popc(bits & ((2u << child) - 1u));    // inclusive count of set children
	}
	return ord;
}

// the point's block: parameter offset and integer coordinates
struct Block {
	uint32_t offset;
	int k[3];
};

__device__ __forceinline__ bool load_block(const ForestDev &fo, const Batch &ba, uint32_t i, Block &b) {
	uint32_t bi;
	if (!batch_base_index(ba, i, b.offset, bi)) return false;
#pragma unroll
	for (int d = 0; d < 3; ++d) b.k[d] = fo.block_ks[3 * (size_t)bi + d];
	return true;
}

// continuity fixing (lotd_forest.h:55-88): corner position p in 0..R+1 -> position inside the block that owns it and
// that block's index (`owner` comes in as the point's own block); false when nothing is stored there (continuity
// off / no such block)
__device__ __forceinline__ bool resolve_block(const ForestDev &fo, const Lvl &L, const int (&k)[3], const uint32_t (&p)[3],
                                              uint32_t (&pl)[3], uint32_t &owner) {
	int kk[3];
	bool changed = false;
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		if (p[d] == 0u) { kk[d] = k[d] - 1; pl[d] = L.res[d] - 1u; changed = true; }
		else if (p[d] == L.res[d] + 1u) { kk[d] = k[d] + 1; pl[d] = 0u; changed = true; }
		else { kk[d] = k[d]; pl[d] = p[d] - 1u; }
	}
	if (changed) {
		if (!fo.continuity) return false;
		const int32_t pidx = identify(fo, (int16_t)kk[0], (int16_t)kk[1], (int16_t)kk[2]);
		const int32_t bi = pidx < 0 ? -1 : pidx - (int32_t)fo.level_poffset;
		if (bi < 0) return false;
		owner = (uint32_t)bi;
	}
	return true;
}

// same, giving the owner's parameter offset
__device__ __forceinline__ bool resolve(const ForestDev &fo, const Batch &ba, const Lvl &L, const Block &b,
                                        const uint32_t (&p)[3], uint32_t (&pl)[3], uint32_t &offset) {
	uint32_t owner = 0xFFFFFFFFu;
	if (!resolve_block(fo, L, b.k, p, pl, owner)) return false;
	offset = (owner == 0xFFFFFFFFu) ? b.offset : (ba.offsets ? (uint32_t)ba.offsets[owner] : owner * ba.n_params);
	return true;
}

// cell locator with the forest's scale = R ("NOTE: for forest", lotd_forest.h:228-232)
__device__ __forceinline__ void locate_forest(const float (&xp)[3], const Lvl &L, bool smooth, Cell<3> &c) {
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const float sc = (float)L.res[d];
		const float v = __fmaf_rn(xp[d], sc, 0.5f);
		const float fl = floorf(v);
		const float t = v - fl;
		c.sc[d] = sc;
		c.g[d] = (uint32_t)fl;
		if (!smooth) {
			c.w[d] = t; c.dw[d] = 1.0f; c.ddw[d] = 0.0f;
		} else {
			c.w[d] = t * t * __fmaf_rn(-2.0f, t, 3.0f);
			c.dw[d] = 6.0f * t * (1.0f - t);
			c.ddw[d] = __fmaf_rn(-12.0f, t, 6.0f);
		}
	}
}

// lotd_bin.hip: atomic-free parameter-gradient path (all level types but NPlaneSum/CPfast, no batching)
uint64_t dparam_workspace_bytes(const nr3d_lotd_meta_t *m, uint32_t n_points, uint32_t n_batches, bool forest = false);
void set_dparam_chunk_log2(int lg);
// `forest` != NULL: the points live in the blocks of a forest (n_batches = n_trees); Dense/Hash 3-D metas only
// levels below `min_level` are left out (their part of dparam is not touched)
int dparam_binned(bool second, const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, const float *dL_ddLdx,
                  const float *dL_dy, int64_t g_sn, int64_t g_se, const float *x, const float *params, const Batch &batch,
                  uint32_t n_batches, int32_t max_level, float *dparam, void *workspace, uint64_t workspace_bytes,
                  hipStream_t st, bool &handled, const ForestDev *forest = nullptr, int32_t min_level = 0, bool g_half = false,
                  bool out_half = false, bool assign = false, bool p_half = false);
// p_half: `params` points to __half tables (the product-type levels read their other factors from them)
// g_half: dL_dy is __half; out_half: dparam is __half (pair path only); assign: dparam arrives UNINITIALISED -- the pair
// path writes every element when one pass covers all levels, every other case zero-fills it first

// lotd_pair.hip: pair-record form of the same path for unbatched 3-D Dense/Hash metas with 2-feature pseudo levels
bool pair_applies(const nr3d_lotd_meta_t *m);
uint32_t pair_direct_levels(const nr3d_lotd_meta_t *m, uint32_t n_points);   // pseudo levels k_pair_direct serves (no records)
void pair_layout(const nr3d_lotd_meta_t *m, uint32_t n_chunk, uint32_t units, uint64_t &rec_bytes, uint64_t &offs_bytes,
                 uint64_t &plan_bytes, uint64_t &part_bytes);
int pair_chunk(const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t n, const float *x, const float *g,
               int64_t g_sn, int64_t g_se, int32_t min_level, int32_t max_level, uint32_t units, float *dparam,
               uint32_t out_flags /* bit 0: dparam is __half; bit 1: assign (dparam uninitialised, every element of the
               plan's levels is written) */, void *rec, uint32_t *offs, uint32_t *plan_buf, float *partial, hipStream_t st,
               const float *vin = nullptr /* second order: dL_ddLdx [n, 3] */);

}  // namespace lotd
}  // namespace nr3d
