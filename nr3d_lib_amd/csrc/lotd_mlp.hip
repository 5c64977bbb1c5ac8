// nr3d_lib_amd/csrc/lotd_mlp.hip -- LoTD encode + fused decoder FORWARD in one kernel (gfx950), C-ABI entry point
// nr3d_lotd_mlp_forward (round 6).
//
// Where it is used: the no-grad density query of the ray driver (nr3d_lib/graphics/nerf/nerf_ray_query.py:105-127: query_density on
// EVERY marched sample, only to decide which samples are still visible).  As two kernels the encoder wrote 128 B of features per
// sample (883 MB for the 6.9 M marched samples of a 262 144-ray pass) which the decoder read straight back, to produce 4 B.  Here a
// wave owns 64 samples, walks the pseudo levels with the two-lane gather of lotd.hip's k_fwd_pairlane (same cell locator, same lerp
// tree, same association: the features are bit-identical to that kernel's), parks the features in a per-wave LDS tile in sample-major
// order and runs the decoder's layers (mlp_device.h: the SAME dense / dense_x3 code as k_mlp_fwd, so bit-identical to it on those
// features) on them.  Only the first `out_cols` columns of the decoder's output leave (sigma: 4 B per sample).
//
// Applies to: 3-D metas of Dense / Hash levels with 2-feature pseudo levels and <= 32 encoded dims (one input tile), decoders with one
// input tile, hidden width <= 64, one output tile (nr3d_lotd_mlp_forward_ok).  Everything else keeps the two kernels.
#include "lotd_device.h"
#include "mlp_device.h"
#include <stdlib.h>

namespace nr3d {
namespace lotdmlp {

using namespace ::nr3d::lotd;
using ::nr3d::mlp::f16v;
using ::nr3d::mlp::f4v;

#ifndef NR3D_FM_SUB
#define NR3D_FM_SUB 2
#endif
#ifndef NR3D_FM_DEPTH
#define NR3D_FM_DEPTH 2
#endif
#ifndef NR3D_FM_MINW
#define NR3D_FM_MINW 2
#endif
constexpr int kWaves = 4, kThreads = kWaves * 64;
constexpr int kSub = NR3D_FM_SUB;          // 32-sample groups per wave and round (= the MFMA tiles it then feeds)
constexpr int kDepth = NR3D_FM_DEPTH;      // pseudo levels whose gathers are in flight per wave
constexpr int kTS = 36;                    // floats per sample row of the feature tile (16-byte aligned rows; 36 mod 32 = 4: the 32 rows
                                           // of a group spread over all banks for the 4-byte feature writes)

template <typename PT> __device__ __forceinline__ float2 load_pair(const char *p);
template <> __device__ __forceinline__ float2 load_pair<float>(const char *p) { return *reinterpret_cast<const float2 *>(p); }
template <> __device__ __forceinline__ float2 load_pair<__half>(const char *p) { return __half22float2(*reinterpret_cast<const __half2 *>(p)); }

struct Args {
	const nr3d_lotd_meta_t *md;
	uint32_t N, n_pseudo, smooth;
	int32_t max_level;
	const float *x;
	const void *params;
	const float *packed; uint32_t packed_floats;       // the decoder's forward region (f32 layers, or the x3 planes when X3)
	uint32_t n_layers, out_dim, out_cols;
	int hidden_act, out_act;
	float *out; int64_t out_stride;
};

// the gathers of one pseudo level for the wave's kSub groups: everything the finish step needs, loads in flight on return
struct Issue {
	float2 t[kSub][4];
	float wP[kSub], wA[kSub], wB[kSub];
	bool live;
};

template <typename PT>
__device__ __forceinline__ void issue_level(const Args &a, uint32_t q, const float (&xp)[kSub][3], uint32_t side, Issue &o) {
	const uint32_t level = meta_level_of(a.md, q);
	const uint32_t foff0 = meta_cnt_of(a.md, q) * 2u;
	o.live = (int32_t)level <= a.max_level;
	const Lvl L = load_level(a.md, o.live ? level : 0u);
	const bool dense = L.type == NR3D_LOD_Dense;
	const char *__restrict__ base = reinterpret_cast<const char *>(reinterpret_cast<const PT *>(a.params) + L.off + foff0);      // wave-uniform
	const uint32_t stride = L.F * (uint32_t)sizeof(PT);
	const bool small = L.size < (1u << 24);
	const float sc0 = (float)(L.res[0] - 2u), sc1 = (float)(L.res[1] - 2u), sc2 = (float)(L.res[2] - 2u);
	if (!o.live) return;
#pragma unroll
	for (int u = 0; u < kSub; ++u) {
		// cell locator and corner offsets: the expressions of lotd.hip pl_item (bit for bit)
		const float v0 = __fmaf_rn(xp[u][0], sc0, 0.5f), v1 = __fmaf_rn(xp[u][1], sc1, 0.5f), v2 = __fmaf_rn(xp[u][2], sc2, 0.5f);
		const float f0 = floorf(v0), f1 = floorf(v1), f2 = floorf(v2);
		float t0 = v0 - f0, t1 = v1 - f1, t2 = v2 - f2;
		const uint32_t c0 = (uint32_t)f0, c1 = (uint32_t)f1, c2 = (uint32_t)f2;
		if (a.smooth) {
			t0 = t0 * t0 * __fmaf_rn(-2.0f, t0, 3.0f); t1 = t1 * t1 * __fmaf_rn(-2.0f, t1, 3.0f); t2 = t2 * t2 * __fmaf_rn(-2.0f, t2, 3.0f);
		}
		uint32_t e[4], off[4];
		if (dense) {
			uint32_t e00;
			if (small) e00 = __umul24(__umul24(c0, L.res[1]) + c1, L.res[2]) + c2 + side;
			else e00 = (c0 * L.res[1] + c1) * L.res[2] + c2 + side;
			const uint32_t sx = L.res[1] * L.res[2], sy = L.res[2];
			e[0] = e00; e[1] = e00 + sx; e[2] = e00 + sy; e[3] = e00 + sx + sy;
			o.wP[u] = t2; o.wA[u] = t0; o.wB[u] = t1;
		} else {
			const uint32_t hy0 = c1 * kPrimes[1], hy1 = hy0 + kPrimes[1];
			const uint32_t hz0 = c2 * kPrimes[2], hz1 = hz0 + kPrimes[2];
			const uint32_t xs = c0 + side;
			const uint32_t a0 = xs ^ hy0, a1 = xs ^ hy1;
			e[0] = a0 ^ hz0; e[1] = a1 ^ hz0; e[2] = a0 ^ hz1; e[3] = a1 ^ hz1;
			if ((L.size & (L.size - 1u)) == 0u) {
				const uint32_t mask = L.size - 1u;
#pragma unroll
				for (int m = 0; m < 4; ++m) e[m] &= mask;
			} else {
#pragma unroll
				for (int m = 0; m < 4; ++m) e[m] %= L.size;
			}
			o.wP[u] = t0; o.wA[u] = t1; o.wB[u] = t2;
		}
		if (small) {
#pragma unroll
			for (int m = 0; m < 4; ++m) off[m] = __umul24(e[m], stride);
		} else {
#pragma unroll
			for (int m = 0; m < 4; ++m) off[m] = e[m] * stride;
		}
#pragma unroll
		for (int m = 0; m < 4; ++m) o.t[u][m] = load_pair<PT>(base + off[m]);
	}
}

// the lerp tree of lotd.hip pl_item (DYDX = false): this lane's feature of its point, for each group
__device__ __forceinline__ void finish_level(const Issue &o, uint32_t side, float (&yv)[kSub]) {
#pragma unroll
	for (int u = 0; u < kSub; ++u) {
		yv[u] = 0.0f;
		if (!o.live) continue;
		const float wk = side ? 1.0f - o.wP[u] : o.wP[u];
		float b[4];
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const float keep = side ? o.t[u][m].y : o.t[u][m].x;
			const float send = side ? o.t[u][m].x : o.t[u][m].y;
			const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xf, 0xf, true));
			b[m] = __fmaf_rn(wk, recv - keep, keep);
		}
		const float cA0 = b[1] - b[0], cA1 = b[3] - b[2];
		const float dA0 = __fmaf_rn(o.wA[u], cA0, b[0]), dA1 = __fmaf_rn(o.wA[u], cA1, b[2]);
		yv[u] = __fmaf_rn(o.wB[u], dA1 - dA0, dA0);
	}
}

// W_T: tiles of the hidden layers (1 or 2); X3: the layers run on the bf16 MFMA with three-piece splits (mlp_device.h dense_x3)
template <typename PT, int W_T, bool X3>
__global__ __launch_bounds__(kThreads, NR3D_FM_MINW) void k_fwd_mlp(Args a) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	::nr3d::mlp::stage_weights(a.packed, a.packed_floats, lds);
	const uint32_t lane = threadIdx.x & 63u, side = lane & 1u, pl = lane >> 1;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float *tile = lds + a.packed_floats + wave * (kSub * 32 * kTS);
	// columns [2 n_pseudo, 32) of the tile stay zero (they meet zero weights anyway; NaN bit patterns in fresh LDS would not)
	for (uint32_t e = lane; e < kSub * 32 * kTS; e += 64) tile[e] = 0.0f;
	const uint32_t off_hidden = X3 ? ::nr3d::mlp::layer_x3_floats(1, W_T) : ::nr3d::mlp::layer_floats(1, W_T);
	const uint32_t sz_hidden = X3 ? ::nr3d::mlp::layer_x3_floats(W_T, W_T) : ::nr3d::mlp::layer_floats(W_T, W_T);
	const uint32_t per_round = kSub * 32;
	const uint32_t n_rounds = (a.N + per_round - 1) / per_round;
	for (uint32_t round = blockIdx.x * kWaves + wave; round < n_rounds; round += gridDim.x * kWaves) {
		const uint32_t w0 = round * per_round;
		// ---- x of both groups (lanes beyond N re-read the last point and store nothing)
		float xp[kSub][3];
#pragma unroll
		for (int u = 0; u < kSub; ++u) {
			const uint32_t i = w0 + (uint32_t)u * 32u + pl;
			const float *px = a.x + (size_t)(i < a.N ? i : a.N - 1u) * 3u;
			xp[u][0] = px[0]; xp[u][1] = px[1]; xp[u][2] = px[2];
		}
		// ---- encode: the pseudo levels one after another, the gathers of the next kDepth - 1 levels in flight while one is interpolated
		Issue ring[kDepth];
#pragma unroll
		for (int d = 0; d < kDepth; ++d)
			if ((uint32_t)d < a.n_pseudo) issue_level<PT>(a, (uint32_t)d, xp, side, ring[d]);
#pragma unroll 1
		for (uint32_t q0 = 0; q0 < a.n_pseudo; q0 += kDepth) {
#pragma unroll
			for (int d = 0; d < kDepth; ++d) {
				const uint32_t q = q0 + (uint32_t)d;
				if (q < a.n_pseudo) {
					float yv[kSub];
					finish_level(ring[d], side, yv);
#pragma unroll
					for (int u = 0; u < kSub; ++u) {
						// half tables: the encoder's output is half (lotd_encoding.h:1501-1504) -- the decoder sees the rounded features,
						// as it does behind the two calls
						if constexpr (sizeof(PT) == 2) yv[u] = __half2float(__float2half(yv[u]));
						tile[(u * 32 + pl) * kTS + 2u * q + side] = yv[u];
					}
					if (q + kDepth < a.n_pseudo) issue_level<PT>(a, q + kDepth, xp, side, ring[d]);
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		// ---- decode: each 32-sample group on the register map (sample = lane & 31; features 8 a + 4 (lane >> 5) + b)
		const int s = (int)(lane & 31u), h = (int)(lane >> 5);
#pragma unroll
		for (int u = 0; u < kSub; ++u) {
			f16v xin[1], hcur[W_T], yo[1];
#pragma unroll
			for (int qd = 0; qd < 4; ++qd) {
				const f4v v = *reinterpret_cast<const f4v *>(tile + (u * 32 + s) * kTS + 8 * qd + 4 * h);
#pragma unroll
				for (int b = 0; b < 4; ++b) xin[0][4 * qd + b] = v[b];
			}
			const float *wl = lds;
			if constexpr (X3) ::nr3d::mlp::dense_x3<1, W_T, true>(wl, xin, hcur, a.hidden_act, (int)lane);
			else ::nr3d::mlp::dense<1, W_T, true>(wl, xin, hcur, a.hidden_act, (int)lane);
#pragma unroll 1
			for (uint32_t l = 1; l + 1 < a.n_layers; ++l) {
				f16v hn[W_T];
				if constexpr (X3) ::nr3d::mlp::dense_x3<W_T, W_T, true>(wl + off_hidden + (l - 1) * sz_hidden, hcur, hn, a.hidden_act, (int)lane);
				else ::nr3d::mlp::dense<W_T, W_T, true>(wl + off_hidden + (l - 1) * sz_hidden, hcur, hn, a.hidden_act, (int)lane);
#pragma unroll
				for (int t = 0; t < W_T; ++t) hcur[t] = hn[t];
			}
			if constexpr (X3) ::nr3d::mlp::dense_x3<W_T, 1, true>(wl + off_hidden + (a.n_layers - 2) * sz_hidden, hcur, yo, a.out_act, (int)lane);
			else ::nr3d::mlp::dense<W_T, 1, true>(wl + off_hidden + (a.n_layers - 2) * sz_hidden, hcur, yo, a.out_act, (int)lane);
			// the first out_cols columns of the sample's row: lane (s, h) holds columns 8 a + 4 h + b in register 4 a + b
			const uint32_t row = w0 + (uint32_t)u * 32u + (uint32_t)s;
			if (row < a.N) {
				float *dst = a.out + (int64_t)row * a.out_stride;
#pragma unroll
				for (int j = 0; j < 16; ++j) {
					const uint32_t c = 8u * (uint32_t)(j >> 2) + 4u * (uint32_t)h + (uint32_t)(j & 3);
					if (c < a.out_cols) dst[c] = yo[0][j];
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the tile is rewritten by the next round
	}
}

}  // namespace lotdmlp
}  // namespace nr3d

using namespace nr3d;

static bool meta_ok(const nr3d_lotd_meta_t *m) {
	if (!m || m->n_dims_to_encode != 3 || m->n_feat_per_pseudo_lvl != 2 || m->n_encoded_dims > 32 || m->n_pseudo_levels == 0) return false;
	for (uint32_t l = 0; l < m->n_levels; ++l)
		if (m->levels[l].type != NR3D_LOD_Dense && m->levels[l].type != NR3D_LOD_Hash) return false;
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q)
		if (m->map_col[q] != 2u * q) return false;             // regrouped metas (other column order) keep the two kernels
	return true;
}

extern "C" int nr3d_lotd_mlp_forward_ok(const nr3d_lotd_meta_t *meta, const nr3d_mlp_desc_t *desc) {
	bool x3; const float *region; uint32_t floats, in_t, w_t, out_t;
	if (!meta_ok(meta) || !desc || desc->dims[0] != meta->n_encoded_dims) return 0;
	if (!mlp::forward_region(desc, nullptr, x3, region, floats, in_t, w_t, out_t)) return 0;
	return (in_t == 1 && w_t <= 2 && out_t == 1) ? 1 : 0;
}

extern "C" int nr3d_lotd_mlp_forward(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint64_t n_points, const float *x,
                                     const void *params, int param_dtype, int32_t max_level, const nr3d_mlp_desc_t *desc,
                                     const float *packed, float *out, int64_t out_stride, uint32_t out_cols, void *stream) {
	NR3D_CHECK(nr3d_lotd_mlp_forward_ok(meta, desc), "lotd_mlp_forward: needs a 3-D Dense / Hash meta with 2-feature pseudo levels and <= 32 "
	           "encoded dims, and a decoder on them with hidden width <= 64 and <= 32 outputs");
	NR3D_CHECK(param_dtype == NR3D_F32 || param_dtype == NR3D_F16, "lotd_mlp_forward: params must be float or half");
	NR3D_CHECK(out_cols >= 1 && out_cols <= desc->dims[desc->n_layers], "lotd_mlp_forward: out_cols outside the decoder's output");
	NR3D_CHECK(n_points < (1ull << 32), "lotd_mlp_forward: more than 2^32 - 1 points in one call");
	if (n_points == 0) return 0;
	NR3D_CHECK(meta_dev && x && params && packed && out, "lotd_mlp_forward: NULL pointer");
	bool x3; const float *region; uint32_t packed_floats, in_t, w_t, out_t;
	mlp::forward_region(desc, packed, x3, region, packed_floats, in_t, w_t, out_t);
	lotdmlp::Args a;
	a.md = (const nr3d_lotd_meta_t *)meta_dev;
	a.N = (uint32_t)n_points; a.n_pseudo = meta->n_pseudo_levels; a.smooth = meta->interpolation_type;
	a.max_level = max_level;
	a.x = x; a.params = params;
	a.packed = region; a.packed_floats = packed_floats;
	a.n_layers = desc->n_layers; a.out_dim = desc->dims[desc->n_layers]; a.out_cols = out_cols;
	a.hidden_act = (int)desc->hidden_activation; a.out_act = (int)desc->output_activation;
	a.out = out; a.out_stride = out_stride;
	const uint32_t lds_bytes = (packed_floats + lotdmlp::kWaves * lotdmlp::kSub * 32 * lotdmlp::kTS) * 4u;
	NR3D_CHECK(lds_bytes <= 160u * 1024u, "lotd_mlp_forward: the decoder's weights do not fit LDS next to the feature tiles");
	const uint32_t n_rounds = (uint32_t)((n_points + 63) / 64);
	const uint32_t per_cu = 160u * 1024u / lds_bytes;
	uint32_t blocks = (n_rounds + lotdmlp::kWaves - 1) / lotdmlp::kWaves;
	const uint32_t cap = 256u * (per_cu > 4u ? 4u : (per_cu ? per_cu : 1u));
	if (blocks > cap) blocks = cap;
	hipStream_t st = (hipStream_t)stream;
#define NR3D_FM(PT, WT, X3) do { \
		auto kern = lotdmlp::k_fwd_mlp<PT, WT, X3>; \
		if (lds_bytes > 64u * 1024u) NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
		hipLaunchKernelGGL(kern, dim3(blocks), dim3(lotdmlp::kThreads), lds_bytes, st, a); } while (0)
	if (param_dtype == NR3D_F32) {
		if (w_t == 1) { if (x3) NR3D_FM(float, 1, true); else NR3D_FM(float, 1, false); }
		else          { if (x3) NR3D_FM(float, 2, true); else NR3D_FM(float, 2, false); }
	} else {
		if (w_t == 1) { if (x3) NR3D_FM(__half, 1, true); else NR3D_FM(__half, 1, false); }
		else          { if (x3) NR3D_FM(__half, 2, true); else NR3D_FM(__half, 2, false); }
	}
#undef NR3D_FM
	NR3D_LAUNCH_CHECK();
	return 0;
}
