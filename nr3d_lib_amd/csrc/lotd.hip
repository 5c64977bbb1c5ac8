// nr3d_lib_amd/csrc/lotd.hip -- LoTD encoder kernels + C-ABI entry points (gfx950 / MI355X).
//
// Replaces the reference extension nr3d_lib.bindings._lotd:
//   forward            kernel_lod / kernel_lod_hash_only[_with_dydx]   lotd_encoding.h:113-428, lotd_hash_only.h:15-378
//   dL/dparam          kernel_lod[_hashonly]_backward_grid               lotd_encoding.h:467-711, lotd_hash_only.h:380-470
//   dL/dx              ATen mul+sum                                      lotd_encoding.h:1562-1586
//   d(dL/dx)/dparam    kernel_lod[_hashonly]_backward_input_backward_grid  lotd_encoding.h:764-1041
//   d(dL/dx)/dx        kernel_lod[_hashonly]_backward_input_backward_input lotd_encoding.h:1157-1298
//   d(dL/dx)/d(dL/dy)  ATen mul+sum                                      lotd_encoding.h:1703-1727
//   get_grid_index     kernel_lod_get_grid_index                         lotd_encoding.h:1300-1433
//   meta               LoDMeta::create_meta                              lotd_torch_api.cu:29-230
//
// Design (MI355X-first, not a translation):
//  * one lane = one (point, pseudo-level); the 2^D corner values of a feature group live in VGPRs and
//    feed BOTH y and dy/dx (no second gather sweep for the Jacobian);
//  * outputs are written feature-major ([E, N] / [E, N, D] storage, returned as strided views), so a
//    wave's 64 lanes store 256 / 768 contiguous bytes per feature;
//  * per-level descriptors come from a device copy of the meta through scalar loads (the block's
//    level is wave-uniform) instead of a 3.3 KB by-value kernel argument;
//  * blocks are scheduled XCD-affinely: each of the 8 XCDs walks its own levels one after another,
//    so one <=4 MiB table at a time is hot in that XCD's private L2;
//  * the second-order parameter scatter combines the reference's 2*D*2^(D-1) left/right updates into
//    one update per corner in registers (2^D atomics instead of D*2^D);
//  * every output element is written (zeros for skipped points), so callers allocate with empty().
#include "lotd_device.h"
#include <stdlib.h>
#include <vector>
#include <type_traits>
#include <algorithm>

namespace nr3d {
namespace lotd {

// =============================================================================================
// Forward
// =============================================================================================
template <typename YT> __device__ __forceinline__ void store_nt(YT *p, float v);      // below (two-lane kernel's section)

// PT: storage type of the tables AND of y (float, or __half read as float -- lotd_device.h, HalfTab -- and y rounded to
// half from the fp32 result); dy/dx leaves as float
template <int D, int G, bool DYDX, bool DH, int ONLY = -1, typename PT = float>
__global__ __launch_bounds__(kBlock) void k_fwd(Sched s, const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                                int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                const PT *__restrict__ params, Batch ba, uint32_t vec_ok,
                                                PT *__restrict__ y, int64_t y_sn, int64_t y_se,
                                                float *__restrict__ dydx, int64_t d_sn, int64_t d_se) {
	uint32_t q, chunk;
	if (!decode_block(s, blockIdx.x, q, chunk)) return;
	const uint32_t i = chunk * kBlock + threadIdx.x;
	if (i >= N) return;
	const uint32_t level = meta_level_of(md, q);
	const uint32_t foff0 = meta_cnt_of(md, q) * G;
	const uint32_t out0 = meta_col_of(md, q);

	float out_y[G];
	float out_g[G][D];
#pragma unroll
	for (int f = 0; f < G; ++f) {
		out_y[f] = 0.0f;
#pragma unroll
		for (int d = 0; d < D; ++d) out_g[f][d] = 0.0f;
	}

	uint32_t base = 0;
	const bool active = ((int32_t)level <= max_level) && batch_base(ba, i, base);
	if (active) {
		const Lvl L = load_level(md, level);
		const auto grid = make_tab(params + (base + L.off));
		float xp[D];
#pragma unroll
		for (int d = 0; d < D; ++d) xp[d] = x[(size_t)i * D + d];
		Cell<D> c;
		locate<D>(xp, L, smooth != 0, c);

		if (DH || (ONLY < 0 && (L.type == NR3D_LOD_Dense || L.type == NR3D_LOD_Hash))) {
			// ---- all corners of all G features at once; vector loads when alignment allows ----
			float v[1 << D][G];
			bool paired = false;
			if constexpr (G == 2) paired = vec_ok && L.F == 2 && L.size >= 2;
			if (paired) {
				if constexpr (G == 2) {
					if (L.type == NR3D_LOD_Dense) gather_pairs<D, true>(L, c, grid, v);
					else gather_pairs<D, false>(L, c, grid, v);
				}
			} else
#pragma unroll
			for (uint32_t k = 0; k < (1u << D); ++k) {
				uint32_t p[D];
				corner_pos<D>(c, k, p);
				const uint32_t e = (L.type == NR3D_LOD_Dense) ? entry_dense<D>(L, p) : entry_hash<D>(L, p);
				const auto src = grid + (e * L.F + foff0);
				if (vec_ok) {
					if constexpr (G == 2) {
						const float2 t = tab_ld2(src, 0);
						v[k][0] = t.x; v[k][1] = t.y;
					} else {
#pragma unroll
						for (int f4 = 0; f4 < G; f4 += 4) {
							const float4 t = tab_ld4(src, f4);
							v[k][f4] = t.x; v[k][f4 + 1] = t.y; v[k][f4 + 2] = t.z; v[k][f4 + 3] = t.w;
						}
					}
				} else {
#pragma unroll
					for (int f = 0; f < G; ++f) v[k][f] = src[f];
				}
			}
#pragma unroll
			for (uint32_t k = 0; k < (1u << D); ++k) {
				const float w = corner_weight<D>(c, k);
#pragma unroll
				for (int f = 0; f < G; ++f) out_y[f] = __fmaf_rn(w, v[k][f], out_y[f]);
			}
			if (DYDX) {
#pragma unroll
				for (int gd = 0; gd < D; ++gd)
#pragma unroll
					for (uint32_t k = 0; k < (1u << D); ++k) {
						if ((k >> gd) & 1u) continue;   // k = lower corner along gd
						const float w = face_weight<D>(c, k, gd, c.sc[gd] * c.dw[gd]);
#pragma unroll
						for (int f = 0; f < G; ++f)
							out_g[f][gd] = __fmaf_rn(w, v[k | (1u << gd)][f] - v[k][f], out_g[f][gd]);
					}
			}
		} else if (ONLY < 0 && L.type == NR3D_LOD_NPlaneSum) {
			// sum over the D axis planes of an (D-1)-linear interpolation (reference: lotd_encoding.h:268-351)
			if constexpr (D > 2) {
#pragma unroll 1
				for (uint32_t jd = 0; jd < (uint32_t)D; ++jd) {
					float pv[1 << (D - 1)][G];
#pragma unroll
					for (uint32_t k = 0; k < (1u << (D - 1)); ++k) {
						uint32_t pp[D];
#pragma unroll
						for (int d2 = 0; d2 < D - 1; ++d2) {
							const int d3 = (uint32_t)d2 >= jd ? d2 + 1 : d2;
							pp[d2] = c.g[d3] + ((k >> d2) & 1u);
						}
						const uint32_t e = entry_nplane_sum<D>(L, jd, pp) * L.F + foff0;
#pragma unroll
						for (int f = 0; f < G; ++f) pv[k][f] = grid[e + f];
					}
#pragma unroll
					for (uint32_t k = 0; k < (1u << (D - 1)); ++k) {
						float w = 1.0f;
#pragma unroll
						for (int d2 = 0; d2 < D - 1; ++d2) {
							const int d3 = (uint32_t)d2 >= jd ? d2 + 1 : d2;
							w *= ((k >> d2) & 1u) ? c.w[d3] : (1.0f - c.w[d3]);
						}
#pragma unroll
						for (int f = 0; f < G; ++f) out_y[f] = __fmaf_rn(w, pv[k][f], out_y[f]);
					}
					if (DYDX) {
#pragma unroll
						for (int g2 = 0; g2 < D - 1; ++g2) {
							const int g3 = (uint32_t)g2 >= jd ? g2 + 1 : g2;
#pragma unroll
							for (uint32_t k = 0; k < (1u << (D - 1)); ++k) {
								if ((k >> g2) & 1u) continue;
								float w = c.sc[g3] * c.dw[g3];
#pragma unroll
								for (int d2 = 0; d2 < D - 1; ++d2) {
									if (d2 == g2) continue;
									const int d3 = (uint32_t)d2 >= jd ? d2 + 1 : d2;
									w *= ((k >> d2) & 1u) ? c.w[d3] : (1.0f - c.w[d3]);
								}
#pragma unroll
								for (int f = 0; f < G; ++f) {
									// out_g is indexed with a runtime dim here; unrolled select keeps it in VGPRs
#pragma unroll
									for (int d = 0; d < D; ++d)
										if (d == g3) out_g[f][d] = __fmaf_rn(w, pv[k | (1u << g2)][f] - pv[k][f], out_g[f][d]);
								}
							}
						}
					}
				}
			}
		} else if (ONLY < 0 && L.type == NR3D_LOD_CPfast) {
			// product over dims of 1-D linear interpolations (reference: lotd_encoding.h:353-410)
			float lv[D][2][G], li[D][G];
#pragma unroll
			for (int d = 0; d < D; ++d) {
				const uint32_t e0 = entry_line<D>(L, d, c.g[d]) * L.F + foff0;
				const uint32_t e1 = entry_line<D>(L, d, c.g[d] + 1u) * L.F + foff0;
#pragma unroll
				for (int f = 0; f < G; ++f) {
					lv[d][0][f] = grid[e0 + f];
					lv[d][1][f] = grid[e1 + f];
					li[d][f] = __fmaf_rn(c.w[d], lv[d][1][f], (1.0f - c.w[d]) * lv[d][0][f]);
				}
			}
#pragma unroll
			for (int f = 0; f < G; ++f) {
				float r = 1.0f;
#pragma unroll
				for (int d = 0; d < D; ++d) r *= li[d][f];
				out_y[f] = r;
			}
			if (DYDX) {
#pragma unroll
				for (int gd = 0; gd < D; ++gd)
#pragma unroll
					for (int f = 0; f < G; ++f) {
						float r = (c.sc[gd] * c.dw[gd]) * (lv[gd][1][f] - lv[gd][0][f]);
#pragma unroll
						for (int d = 0; d < D; ++d) if (d != gd) r *= li[d][f];
						out_g[f][gd] = r;
					}
			}
		} else {
			// ---- remaining N-linear types (VM, VecZMatXoY, CP, NPlaneMul): feature pairs to bound VGPRs; four features at a
			// time (one 16-byte request per table entry instead of two 8-byte ones) when the pseudo level is that wide and
			// the level's entries are 16-byte aligned
			auto accumulate = [&](auto nf_tag, int f0) {
				constexpr int NF = decltype(nf_tag)::value;
				float v[1 << D][NF];
				corner_values_pair<D, ONLY, NF>(L, grid, foff0 + f0, vec_ok != 0 && (L.F % NF) == 0u, c, v);
				float yy[NF], gg[NF][D];
#pragma unroll
				for (int f = 0; f < NF; ++f) {
					yy[f] = 0.0f;
#pragma unroll
					for (int d = 0; d < D; ++d) gg[f][d] = 0.0f;
				}
#pragma unroll
				for (uint32_t k = 0; k < (1u << D); ++k) {
					const float w = corner_weight<D>(c, k);
#pragma unroll
					for (int f = 0; f < NF; ++f) yy[f] = __fmaf_rn(w, v[k][f], yy[f]);
				}
				if (DYDX) {
#pragma unroll
					for (int gd = 0; gd < D; ++gd)
#pragma unroll
						for (uint32_t k = 0; k < (1u << D); ++k) {
							if ((k >> gd) & 1u) continue;
							const float w = face_weight<D>(c, k, gd, c.sc[gd] * c.dw[gd]);
#pragma unroll
							for (int f = 0; f < NF; ++f) gg[f][gd] = __fmaf_rn(w, v[k | (1u << gd)][f] - v[k][f], gg[f][gd]);
						}
				}
#pragma unroll
				for (int f = 0; f < G; ++f)
					if (f >= f0 && f < f0 + NF) {
						out_y[f] = yy[f - f0];
#pragma unroll
						for (int d = 0; d < D; ++d) out_g[f][d] = gg[f - f0][d];
					}
			};
			bool quads = false;
			if constexpr (G % 4 == 0) quads = vec_ok != 0 && (L.F & 3u) == 0u && ((base + L.off) & 3u) == 0u;
			if (quads) {
				if constexpr (G % 4 == 0) {
#pragma unroll 1
					for (int f0 = 0; f0 < G; f0 += 4) accumulate(std::integral_constant<int, 4>{}, f0);
				}
			} else {
#pragma unroll 1
				for (int f0 = 0; f0 < G; f0 += 2) accumulate(std::integral_constant<int, 2>{}, f0);
			}
		}
	}

	// outputs are written once and not read by this kernel: non-temporal stores keep the level tables in the L2
#pragma unroll
	for (int f = 0; f < G; ++f) store_nt<PT>(&y[(int64_t)i * y_sn + (int64_t)(out0 + f) * y_se], out_y[f]);
	if (DYDX) {
#pragma unroll
		for (int f = 0; f < G; ++f) {
			float *dst = dydx + (int64_t)i * d_sn + (int64_t)(out0 + f) * d_se;
#pragma unroll
			for (int d = 0; d < D; ++d) __builtin_nontemporal_store(out_g[f][d], &dst[d]);
		}
	}
}

// =============================================================================================
// Forward, two lanes per (point, pseudo level): 3-D Dense / Hash levels with 2-feature pseudo levels.
//
// The gather rate of k_fwd is bound by L2 requests -- one 128-byte line per lane and load instruction
// (profiles/r02a_counters.txt: 84 M TCP->TCC requests for 2^20 points, the L2 channels 88 % busy) -- and a lane can ask
// for at most 16 bytes.  But the texture addresser coalesces the lanes of ONE instruction by line, and the two corners
// of a cell along x sit in the same line of a Hash table 15 times out of 16:
//     hash(x, y, z) = x ^ K(y, z)  =>  the 16 x of an aligned block map onto the 16 entries (128 B) of an aligned block.
// So lanes 2i and 2i + 1 share point i: lane s gathers the 4 corners with x-bit s (Dense: z-bit s, adjacent entries), the
// pair's two 8-byte loads travel in the same instruction and leave the CU as ONE request.  4.25 requests per (point,
// Hash level) instead of 6 (paired 16-byte loads work for even x only), 4 for Dense as before.  The lanes then swap
// features through DPP (quad_perm 1,0,3,2): lane s ends up with all 8 corners of feature s and interpolates it.
// =============================================================================================
constexpr int kPlPts = kBlock / 2;                 // points per block

// one table entry's feature pair as floats; PT = float (8 bytes) or __half (4 bytes: the (float, half, float) type
// combination of the reference, lotd_encoding.h:1501-1504 -- no whole-table conversion pass, half the table bytes)
template <typename PT> __device__ __forceinline__ float2 load_pair(const char *p);
template <> __device__ __forceinline__ float2 load_pair<float>(const char *p) { return *reinterpret_cast<const float2 *>(p); }
template <> __device__ __forceinline__ float2 load_pair<__half>(const char *p) { return __half22float2(*reinterpret_cast<const __half2 *>(p)); }
template <typename YT> __device__ __forceinline__ void store_nt(YT *p, float v);
template <> __device__ __forceinline__ void store_nt<float>(float *p, float v) { __builtin_nontemporal_store(v, p); }
template <> __device__ __forceinline__ void store_nt<__half>(__half *p, float v) {
	// the value is the fp32 result rounded to half (what the fp32 kernel + a cast gives): keep the compiler from fusing the
	// last fma with the conversion (v_fma_mixlo_f16 rounds the exact fma once, i.e. differently on fp32 -> half ties)
	asm volatile("" : "+v"(v));
	__builtin_nontemporal_store(__half_as_ushort(__float2half(v)), reinterpret_cast<unsigned short *>(p));
}

// =============================================================================================
// The kernel (round 3 form).
//
// Measured on the round-2 kernel (tools/exp_fwd_variants.py, profiles/r03a_fwd_experiments.txt): with its gathers REMOVED
// it still took 236 of its 355 us -- 201 VALU instructions per wave at 4 cycles each (160 us of SIMD time) plus 64-bit
// address arithmetic for every access.  This form keeps the gather pattern (4.25 requests per point and level) and cuts
// the instruction stream to about a third (the no-gather time drops to 131 us, the no-gather-no-store time to 106):
//  * N-linear interpolation as a tree of lerps, pair dim first: 7 (sub, fma) pairs give the value; the differences the
//    lerps form anyway ARE the derivative's building blocks (3 + 1 more lerps) -- 25 flops for y and dy/dx instead of
//    ~100 for the sum over 8 corner weights + 12 face weights.  Same polynomial, different association: results agree
//    with the corner-sum form (k_fwd, the oracle) to fp32 rounding (measured 3.8e-7 of the column maximum), far inside
//    the 1e-5 contract;
//  * lanes exchange ONE value per corner pair (the feature the partner interpolates) and lerp "own side first":
//    b = keep + w_keep * (recv - keep) -- no lo / hi selects;
//  * addresses as a uniform 64-bit base (SGPRs) + 32-bit lane offset: no 64-bit vector multiplies for the strides, no
//    64-bit vector adds per gather, the Jacobian leaves as one dwordx3 per lane; 24-bit multiplies (full rate) where the
//    level is small enough;
//  * a wave walks SUB consecutive 32-point groups with ALL their gathers in flight at once; the scalar preamble
//    (schedule decode, level descriptor) and the lane constants are paid once.
// What the experiments say about the rest (2^22 points, us per 2^20): gathers + arithmetic alone 245 (255 G requests/s,
// the rate of the pure-gather micro-benchmark, profiles/r01_ubench_mem.txt), + output stores 285, + x reads 302, all
// three 338: the L2 channels serve the gathers at their ceiling and the 7.3 M write and 1.4 M x-miss requests queue in
// the same channels -- the times add instead of overlapping.  More groups in flight (SUB 1 / 2 / 4 / 8: 358 / 352 / 358
// / 355 us at 2^20), plain instead of non-temporal stores (373), a row pitch that is not a power of two (no change), x
// through the scalar cache (+14 us on the round-2 kernel) do not move it.
// `dbg` exists in a -DNR3D_EXPERIMENTS build only (timing experiments, results wrong by design): bit 0 no stores, bit 1 no
// gathers, bit 2 no x loads; in the production build it is the constant 0 and these branches fold away.
// =============================================================================================
constexpr int kPlSub = 2;                          // 32-point groups per wave
// one work item = pseudo level q x the block's `chunk` of kPlPts * SUB points
template <bool DYDX, typename PT, int SUB>
__device__ __forceinline__ void pl_item(uint32_t q, uint32_t chunk, const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                        int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                        const PT *__restrict__ params, PT *__restrict__ y, int64_t y_sn, int64_t y_se,
                                        float *__restrict__ dydx, int64_t d_sn, int64_t d_se NR3D_DBG_PARAM) {
	NR3D_DBG_DECL
	constexpr uint32_t kGroup = 32;                        // points per wave and sub-step
	constexpr uint32_t kPts = kPlPts * SUB;                // points per block
	const uint32_t lane = threadIdx.x & 63u, side = lane & 1u, pl = lane >> 1;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t w0 = chunk * kPts + wave * kGroup * SUB;      // first point of this wave
	if (w0 >= N) return;                                   // whole wave past the end (uniform)
	const uint32_t level = meta_level_of(md, q);
	const uint32_t foff0 = meta_cnt_of(md, q) * 2u;
	const bool live = (int32_t)level <= max_level;
	const Lvl L = load_level(md, live ? level : 0u);
	const bool dense = L.type == NR3D_LOD_Dense;
	const char *__restrict__ base = reinterpret_cast<const char *>(params + L.off + foff0);      // wave-uniform
	const uint32_t stride = L.F * (uint32_t)sizeof(PT);
	const bool small = L.size < (1u << 24);               // entries and strides fit 24-bit multiplies
	const float sc0 = (float)(L.res[0] - 2u), sc1 = (float)(L.res[1] - 2u), sc2 = (float)(L.res[2] - 2u);
	// lane part of the output addresses (host-checked to fit 32 bits), in bytes
	const uint32_t y_lane = (uint32_t)((int64_t)pl * y_sn + (int64_t)side * y_se) * (uint32_t)sizeof(PT);
	const uint32_t d_lane = DYDX ? (uint32_t)((int64_t)pl * d_sn + (int64_t)side * d_se) * 4u : 0u;

	// ---- phase 1: x of all SUB groups (lanes beyond N re-read the last point and store nothing)
	float xp[SUB][3];
#pragma unroll
	for (int u = 0; u < SUB; ++u) {
		const uint32_t g0 = w0 + (uint32_t)u * kGroup;     // uniform
		const uint32_t gc = g0 < N ? g0 : w0;
		const char *xb = reinterpret_cast<const char *>(x) + (size_t)gc * 12u;
		uint32_t off = pl * 12u;
		if (gc + kGroup > N) { const uint32_t i = gc + pl; off = ((i < N ? i : N - 1u) - gc) * 12u; }
		const float *px = reinterpret_cast<const float *>(xb + off);
		if (dbg & 4u) {
			uint32_t h = (gc + pl) * 2654435761u + 12345u;
#pragma unroll
			for (int d = 0; d < 3; ++d) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16; xp[u][d] = (float)(h >> 8) * (1.0f / 16777216.0f); }
		} else { xp[u][0] = px[0]; xp[u][1] = px[1]; xp[u][2] = px[2]; }
	}
	float yv[SUB], gx[SUB], gy[SUB], gz[SUB];
#pragma unroll
	for (int u = 0; u < SUB; ++u) { yv[u] = 0.0f; gx[u] = 0.0f; gy[u] = 0.0f; gz[u] = 0.0f; }
	if (live) {
		// ---- phase 2: cells, corner offsets, ALL 4 * SUB gathers in flight
		float2 t[SUB][4];
		float wP[SUB], wA[SUB], wB[SUB], dwP[SUB], dwA[SUB], dwB[SUB];
#pragma unroll
		for (int u = 0; u < SUB; ++u) {
			// cell locator (explicit fma: decides the integer cell, must match the oracle bit for bit)
			const float v0 = __fmaf_rn(xp[u][0], sc0, 0.5f), v1 = __fmaf_rn(xp[u][1], sc1, 0.5f), v2 = __fmaf_rn(xp[u][2], sc2, 0.5f);
			const float f0 = floorf(v0), f1 = floorf(v1), f2 = floorf(v2);
			float t0 = v0 - f0, t1 = v1 - f1, t2 = v2 - f2;
			const uint32_t c0 = (uint32_t)f0, c1 = (uint32_t)f1, c2 = (uint32_t)f2;
			float dw0 = sc0, dw1 = sc1, dw2 = sc2;         // scale * w'
			if (smooth) {
				dw0 *= 6.0f * t0 * (1.0f - t0); dw1 *= 6.0f * t1 * (1.0f - t1); dw2 *= 6.0f * t2 * (1.0f - t2);
				t0 = t0 * t0 * __fmaf_rn(-2.0f, t0, 3.0f); t1 = t1 * t1 * __fmaf_rn(-2.0f, t1, 3.0f); t2 = t2 * t2 * __fmaf_rn(-2.0f, t2, 3.0f);
			}
			// entries of this lane's four corners: pair dim P (Dense z, Hash x) takes `side`, bit 0 of m = dim A, bit 1 = dim B
			uint32_t e[4], off[4];
			if (dense) {
				uint32_t e00;
				if (small) e00 = __umul24(__umul24(c0, L.res[1]) + c1, L.res[2]) + c2 + side;
				else e00 = (c0 * L.res[1] + c1) * L.res[2] + c2 + side;
				const uint32_t sx = L.res[1] * L.res[2], sy = L.res[2];      // scalar
				e[0] = e00; e[1] = e00 + sx; e[2] = e00 + sy; e[3] = e00 + sx + sy;
				wP[u] = t2; wA[u] = t0; wB[u] = t1; dwP[u] = dw2; dwA[u] = dw0; dwB[u] = dw1;
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
				wP[u] = t0; wA[u] = t1; wB[u] = t2; dwP[u] = dw0; dwA[u] = dw1; dwB[u] = dw2;
			}
			if (small) {
#pragma unroll
				for (int m = 0; m < 4; ++m) off[m] = __umul24(e[m], stride);
			} else {
#pragma unroll
				for (int m = 0; m < 4; ++m) off[m] = e[m] * stride;
			}
			if ((dbg & 8u) && dense) {
				// timing experiment (round 4, results wrong): what a CELL-MAJOR replica of the Dense levels would issue -- a
				// lane's four corner pairs are 32 contiguous bytes of the cell's 64-byte record, two 16-byte loads
				const uint32_t cb = (off[0] & ~63u) + 32u * side;
				const float4 lo4 = *reinterpret_cast<const float4 *>(base + cb), hi4 = *reinterpret_cast<const float4 *>(base + cb + 16u);
				t[u][0] = make_float2(lo4.x, lo4.y); t[u][1] = make_float2(lo4.z, lo4.w);
				t[u][2] = make_float2(hi4.x, hi4.y); t[u][3] = make_float2(hi4.z, hi4.w);
			} else {
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				if (dbg & 2u) t[u][m] = make_float2(__int_as_float(off[m] | 0x3f000000u), __int_as_float(off[m] ^ 0x3f123456u));
				else t[u][m] = load_pair<PT>(base + off[m]);
			}
			}
		}
		// ---- phase 3: pair-dim lerps (the partner gets the feature it interpolates, this lane keeps its own; the true
		// (v1 - v0) along P is sgn * d), then dims A and B
#pragma unroll
		for (int u = 0; u < SUB; ++u) {
			const float wk = side ? 1.0f - wP[u] : wP[u];
			float b[4], d[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const float keep = side ? t[u][m].y : t[u][m].x;
				const float send = side ? t[u][m].x : t[u][m].y;
				const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xf, 0xf, true));
				d[m] = recv - keep;
				b[m] = __fmaf_rn(wk, d[m], keep);
			}
			const float cA0 = b[1] - b[0], cA1 = b[3] - b[2];
			const float dA0 = __fmaf_rn(wA[u], cA0, b[0]), dA1 = __fmaf_rn(wA[u], cA1, b[2]);
			const float eB = dA1 - dA0;
			yv[u] = __fmaf_rn(wB[u], eB, dA0);
			if (DYDX) {
				const float gB = eB;
				const float gA = __fmaf_rn(wB[u], cA1 - cA0, cA0);
				const float p0 = __fmaf_rn(wA[u], d[1] - d[0], d[0]), p1 = __fmaf_rn(wA[u], d[3] - d[2], d[2]);
				const float gP = __fmaf_rn(wB[u], p1 - p0, p0);
				const float rP = gP * (side ? -dwP[u] : dwP[u]), rA = gA * dwA[u], rB = gB * dwB[u];
				if (dense) { gx[u] = rA; gy[u] = rB; gz[u] = rP; }
				else { gx[u] = rP; gy[u] = rA; gz[u] = rB; }
			}
		}
	}
	// ---- phase 4: stores (uniform base + 32-bit lane offset)
#pragma unroll
	for (int u = 0; u < SUB; ++u) {
		const uint32_t g0 = w0 + (uint32_t)u * kGroup;
		if ((dbg & 1u) && yv[u] + gx[u] + gy[u] + gz[u] != 1234.56789f) continue;
		if (g0 + pl < N) {
			char *yb = reinterpret_cast<char *>(y) + ((int64_t)g0 * y_sn + (int64_t)(q * 2u) * y_se) * (int64_t)sizeof(PT);
			store_nt<PT>(reinterpret_cast<PT *>(yb + y_lane), yv[u]);
			if (DYDX) {
				char *db = reinterpret_cast<char *>(dydx) + ((int64_t)g0 * d_sn + (int64_t)(q * 2u) * d_se) * 4;
				float *dst = reinterpret_cast<float *>(db + d_lane);
				__builtin_nontemporal_store(gx[u], &dst[0]);
				__builtin_nontemporal_store(gy[u], &dst[1]);
				__builtin_nontemporal_store(gz[u], &dst[2]);
			}
		}
	}
}

template <bool DYDX, typename PT, int SUB>
__global__ __launch_bounds__(kBlock) void k_fwd_pairlane(Sched s, const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                                   int32_t max_level, uint32_t smooth, const float *__restrict__ x,
                                                   const PT *__restrict__ params, PT *__restrict__ y, int64_t y_sn,
                                                   int64_t y_se, float *__restrict__ dydx, int64_t d_sn, int64_t d_se NR3D_DBG_PARAM) {
	uint32_t q, chunk;
	if (!decode_block(s, blockIdx.x, q, chunk)) return;
	pl_item<DYDX, PT, SUB>(q, chunk, md, N, max_level, smooth, x, params, y, y_sn, y_se, dydx, d_sn, d_se NR3D_DBG_ARG(dbg));
}

// Tried and dropped (round 3, profiles/r03e_dynamic_handout_experiment.txt): handing the items out dynamically -- resident
// blocks pull batches from per-XCD queue counters and steal from the other XCDs' queues once theirs is empty, which would
// even out the XCDs when ray-coherent samples make the coarse levels cheap.  Agent-scope atomics on one line serialise at
// ~80 ns each on this part (they execute behind the per-XCD L2s): 28 672 grabs = 2.36 ms for a 0.35 ms kernel.  Counters
// served by the XCD's own L2 (workgroup scope) would be fast but are invisible to the stealing XCDs.

// =============================================================================================
// Forward, LDS-staged: Dense levels of 2 features whose whole table fits one CU's LDS (NGP config: level 0 = 32 KiB,
// level 1 = 85 KiB of the 160 KiB).  The workgroup copies the table into LDS once (coalesced) and serves the corner
// gathers of kLdsPts points from it: 4 ds_read2_b64 per point instead of 4 L2 requests, and the forward of the fine
// levels -- bound by L2 requests (profiles/r02a_counters.txt: 84 M TCP->TCC requests, TCC 88 % busy) -- loses
// these levels' share.  Same arithmetic in the same order as k_fwd: bit-identical outputs.
// =============================================================================================
constexpr int kLdsThreads = 1024;
constexpr uint32_t kLdsPts = 4096;                 // points per workgroup
constexpr uint32_t kLdsMaxBytes = 96 * 1024;       // table bytes a level may have to be staged
constexpr uint32_t kLdsGroupBytes = 150 * 1024;    // ... and the tables one launch holds together (160 KiB of LDS per CU)

// NL levels per launch (round 3): the two coarsest NGP levels (32 + 85 KiB) share one workgroup -- x is read once, one
// launch instead of two (2 x 13 us -> <see DESIGN>).  q[l] = pseudo level, its table at float2 offset lds_off[l].
struct LdsLevels { uint32_t q[2]; uint32_t lds_off[2]; };

template <bool DYDX, typename PT, int NL>
__global__ __launch_bounds__(kLdsThreads) void k_fwd_lds(const nr3d_lotd_meta_t *__restrict__ md, uint32_t N, LdsLevels lv,
                                                         uint32_t smooth, const float *__restrict__ x,
                                                         const PT *__restrict__ params, PT *__restrict__ y, int64_t y_sn,
                                                         int64_t y_se, float *__restrict__ dydx, int64_t d_sn, int64_t d_se) {
	extern __shared__ __attribute__((aligned(16))) float2 tab_all[];
	Lvl L[NL];
#pragma unroll
	for (int l = 0; l < NL; ++l) {
		L[l] = load_level(md, meta_level_of(md, lv.q[l]));
		const char *__restrict__ src = reinterpret_cast<const char *>(params + L[l].off);
		float2 *tab = tab_all + lv.lds_off[l];
		for (uint32_t e = threadIdx.x; e < L[l].size; e += kLdsThreads) tab[e] = load_pair<PT>(src + (size_t)e * (2 * sizeof(PT)));
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// outputs: uniform base of the wave's 64 points + 32-bit lane offset (host-checked), as in k_fwd_pairlane
	const uint32_t y_lane = (uint32_t)((int64_t)lane * y_sn) * (uint32_t)sizeof(PT);
	const uint32_t d_lane = DYDX ? (uint32_t)((int64_t)lane * d_sn) * 4u : 0u;
	const uint32_t ye = (uint32_t)(y_se * (int64_t)sizeof(PT)), de = DYDX ? (uint32_t)(d_se * 4) : 0u;
#pragma unroll 1
	for (uint32_t k4 = 0; k4 < kLdsPts / kLdsThreads; ++k4) {
		const uint32_t g0 = blockIdx.x * kLdsPts + k4 * kLdsThreads + wave * 64u;      // uniform
		if (g0 >= N) break;
		const uint32_t i = g0 + lane, ic = i < N ? i : N - 1u;
		const float *px = reinterpret_cast<const float *>(reinterpret_cast<const char *>(x) + (size_t)g0 * 12u + (ic - g0) * 12u);
		const float x0 = px[0], x1 = px[1], x2 = px[2];
#pragma unroll
		for (int l = 0; l < NL; ++l) {
			const float2 *__restrict__ tab = tab_all + lv.lds_off[l];
			const float sc0 = (float)(L[l].res[0] - 2u), sc1 = (float)(L[l].res[1] - 2u), sc2 = (float)(L[l].res[2] - 2u);
			const uint32_t sx = L[l].res[1] * L[l].res[2], sy = L[l].res[2];
			// cell locator (explicit fma: decides the integer cell, must match the oracle bit for bit)
			const float v0 = __fmaf_rn(x0, sc0, 0.5f), v1 = __fmaf_rn(x1, sc1, 0.5f), v2 = __fmaf_rn(x2, sc2, 0.5f);
			const float f0 = floorf(v0), f1 = floorf(v1), f2 = floorf(v2);
			float t0 = v0 - f0, t1 = v1 - f1, t2 = v2 - f2;
			float dw0 = sc0, dw1 = sc1, dw2 = sc2;             // scale * w'
			if (smooth) {
				dw0 *= 6.0f * t0 * (1.0f - t0); dw1 *= 6.0f * t1 * (1.0f - t1); dw2 *= 6.0f * t2 * (1.0f - t2);
				t0 = t0 * t0 * __fmaf_rn(-2.0f, t0, 3.0f); t1 = t1 * t1 * __fmaf_rn(-2.0f, t1, 3.0f); t2 = t2 * t2 * __fmaf_rn(-2.0f, t2, 3.0f);
			}
			const uint32_t e00 = ((uint32_t)f0 * L[l].res[1] + (uint32_t)f1) * L[l].res[2] + (uint32_t)f2;
			const uint32_t e[4] = {e00, e00 + sx, e00 + sy, e00 + sx + sy};
			// the lerp tree of k_fwd_pairlane for a Dense level (pair dim z, then x, then y), both features in one lane: feature
			// 0 with the arithmetic of the pair's side-0 lane (keep the lower corner, weight w), feature 1 with the side-1
			// lane's (keep the upper corner, weight 1 - w, difference negated) -- the two kernels give the same bits
			const float w1 = 1.0f - t2;
			float2 b[4], d[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const float2 lo = tab[e[m]], hi = tab[e[m] + 1u];
				d[m] = make_float2(hi.x - lo.x, lo.y - hi.y);
				b[m] = make_float2(__fmaf_rn(t2, d[m].x, lo.x), __fmaf_rn(w1, d[m].y, hi.y));
			}
			float yv[2], gx[2], gy[2], gz[2];
#pragma unroll
			for (int f = 0; f < 2; ++f) {
				auto c = [f](const float2 &v) { return f ? v.y : v.x; };
				const float cA0 = c(b[1]) - c(b[0]), cA1 = c(b[3]) - c(b[2]);
				const float dA0 = __fmaf_rn(t0, cA0, c(b[0])), dA1 = __fmaf_rn(t0, cA1, c(b[2]));
				const float eB = dA1 - dA0;
				yv[f] = __fmaf_rn(t1, eB, dA0);
				if (DYDX) {
					const float gA = __fmaf_rn(t1, cA1 - cA0, cA0);
					const float p0 = __fmaf_rn(t0, c(d[1]) - c(d[0]), c(d[0])), p1 = __fmaf_rn(t0, c(d[3]) - c(d[2]), c(d[2]));
					const float gP = __fmaf_rn(t1, p1 - p0, p0);
					gx[f] = gA * dw0; gy[f] = eB * dw1; gz[f] = gP * (f ? -dw2 : dw2);
				}
			}
			if (i < N) {
				char *yb = reinterpret_cast<char *>(y) + ((int64_t)g0 * y_sn + (int64_t)(lv.q[l] * 2u) * y_se) * (int64_t)sizeof(PT);
				store_nt<PT>(reinterpret_cast<PT *>(yb + y_lane), yv[0]);
				store_nt<PT>(reinterpret_cast<PT *>(yb + y_lane + ye), yv[1]);
				if (DYDX) {
					char *db = reinterpret_cast<char *>(dydx) + ((int64_t)g0 * d_sn + (int64_t)(lv.q[l] * 2u) * d_se) * 4;
#pragma unroll
					for (int f = 0; f < 2; ++f) {
						float *dst = reinterpret_cast<float *>(db + d_lane + (f ? de : 0u));
						__builtin_nontemporal_store(gx[f], &dst[0]);
						__builtin_nontemporal_store(gy[f], &dst[1]);
						__builtin_nontemporal_store(gz[f], &dst[2]);
					}
				}
			}
		}
	}
}

// =============================================================================================
// Forward, LDS-staged BY SLAB (round 5; an experiment that LOST, kept behind NR3D_OPT_FWD_LDS_STAGE = 2): Dense levels of 2
// features whose table does not fit LDS whole but in a few slabs of x-planes (NGP config: level 2 = 30^3, 216 KB, two slabs;
// level 3 = 42^3, 593 KB, six).  The workgroup owns kLdsPts points as in k_fwd_lds.  It first buckets them by slab in LDS
// (wave-aggregated counters: one ballot per slab), fetches every point's coordinates once, then walks the slabs: copy the slab's
// planes (+ one halo plane) into LDS, serve the bucket's points from it -- whole waves, every point once; the results wait in
// registers and leave through LDS in point order as whole rows.  The arithmetic is k_fwd_lds's: bit-identical outputs
// (tests/test_lotd_gpu.py::test_forward_lds_slabs_bit_identical).
// Measured (tools/exp_fwd_slab.py, configs[1], 2^20 points): the two-lane kernel loses levels 2-3 and 48 us (339 -> 291), the
// two slab launches cost 62 us: +14 us net (first version, results stored through the point index: 73 us -- 8 scattered dword
// stores per point are 1.5-3.7 partial line writes per point and level; one 8-byte load per staging round trip instead of eight
// in flight and the coordinates re-fetched inside the slab loop made no difference).  What it is bound by: a workgroup's 2 + 6
// slab rounds are SERIAL -- stage 108-119 KB, barrier, ~700 points of LDS work, barrier: 4-5 us of latency each with one workgroup
// per CU (140 KB of LDS) and nothing to overlap with.  A (chunk x slab) grid -- one slab per workgroup -- would make it one round, and
// one round was measured too (a build forcing one slab per level): 39.5 us for the two launches, ~20 us per level against the ~24 us
// a level costs in the two-lane kernel.  Bucketing + LDS write-out cost what the L2 requests they replace cost: dropped.
constexpr uint32_t kSlabBytes = 140 * 1024;        // LDS for the slab's planes (the bucket lists take 8 KB + the counters)
constexpr uint32_t kSlabMax = 8;                   // slabs per level at most: every workgroup pays one staging round per slab
struct SlabLevel { uint32_t q, nx, n_slabs; };     // pseudo level, cell planes per slab, slabs

template <bool DYDX, typename PT>
__global__ __launch_bounds__(kLdsThreads) void k_fwd_lds_slab(const nr3d_lotd_meta_t *__restrict__ md, uint32_t N, SlabLevel sl,
                                                              uint32_t smooth, const float *__restrict__ x,
                                                              const PT *__restrict__ params, PT *__restrict__ y, int64_t y_sn,
                                                              int64_t y_se, float *__restrict__ dydx, int64_t d_sn, int64_t d_se) {
	extern __shared__ __attribute__((aligned(16))) float2 slab_tab[];
	__shared__ uint16_t list[kLdsPts];             // local point ids, bucket after bucket
	__shared__ uint32_t cnt[kSlabMax], base[kSlabMax + 1], cur[kSlabMax];
	const Lvl L = load_level(md, meta_level_of(md, sl.q));
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t g_first = blockIdx.x * kLdsPts, n_here = (N - g_first) < kLdsPts ? (N - g_first) : kLdsPts;
	const float sc0 = (float)(L.res[0] - 2u), sc1 = (float)(L.res[1] - 2u), sc2 = (float)(L.res[2] - 2u);
	const uint32_t sx = L.res[1] * L.res[2], sy = L.res[2];
	if (threadIdx.x < kSlabMax) { cnt[threadIdx.x] = 0u; cur[threadIdx.x] = 0u; }
	__syncthreads();
	// ---- the slab of every point (its x cell, by the exact locator; a cell outside the level is clamped into the last / first slab
	// and into the staged planes below: x outside [0, 1] reads what k_fwd_lds would not have either, but never outside LDS) ----
	uint32_t slab_of[kLdsPts / kLdsThreads];
	const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
	for (uint32_t k4 = 0; k4 < kLdsPts / kLdsThreads; ++k4) {
		const uint32_t lp = k4 * kLdsThreads + threadIdx.x;
		uint32_t s = 0xFFFFFFFFu;
		if (lp < n_here) {
			const float f0 = floorf(__fmaf_rn(x[(size_t)(g_first + lp) * 3], sc0, 0.5f));
			const uint32_t c0 = f0 > 0.0f ? (uint32_t)fminf(f0, (float)(L.res[0] - 2u)) : 0u;
			s = c0 / sl.nx;
			s = s < sl.n_slabs ? s : sl.n_slabs - 1u;                // (cannot exceed it for a geometry from slab_geometry; never an unlisted point)
		}
		slab_of[k4] = s;
		for (uint32_t t = 0; t < sl.n_slabs; ++t) {
			const uint64_t m = __ballot(s == t);
			if (m && lane == (uint32_t)(__ffsll((long long)m) - 1)) atomicAdd(&cnt[t], (uint32_t)__popcll(m));
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = 0;
		for (uint32_t t = 0; t < sl.n_slabs; ++t) { base[t] = run; run += cnt[t]; }
		base[sl.n_slabs] = run;
	}
	__syncthreads();
#pragma unroll
	for (uint32_t k4 = 0; k4 < kLdsPts / kLdsThreads; ++k4) {
		const uint32_t lp = k4 * kLdsThreads + threadIdx.x, s = slab_of[k4];
		for (uint32_t t = 0; t < sl.n_slabs; ++t) {
			const uint64_t m = __ballot(s == t);
			if (!m) continue;
			const uint32_t leader = (uint32_t)(__ffsll((long long)m) - 1);
			uint32_t at = 0;
			if (lane == leader) at = atomicAdd(&cur[t], (uint32_t)__popcll(m));
			at = __shfl(at, (int)leader, 64);
			if (s == t) list[base[t] + at + (uint32_t)__popcll(m & below)] = (uint16_t)lp;
		}
	}
	__syncthreads();                                                               // the lists are complete
	// list position p = k4 * 1024 + thread: its point and coordinates, fetched ONCE in front of the slab loop (all loads in flight
	// together; inside the loop they were a dependent global round trip per slab).  A bucket is a contiguous range of positions, so
	// the threads that work on a slab are whole waves.
	constexpr uint32_t kPer = kLdsPts / kLdsThreads;
	uint32_t pt[kPer];
	float xp[kPer][3];
#pragma unroll
	for (uint32_t k4 = 0; k4 < kPer; ++k4) {
		const uint32_t p = k4 * kLdsThreads + threadIdx.x;
		pt[k4] = g_first + (p < n_here ? (uint32_t)list[p] : 0u);
#pragma unroll
		for (int d = 0; d < 3; ++d) xp[k4][d] = x[(size_t)pt[k4] * 3 + d];
	}
	// results stay in registers until every slab is done, then go through LDS (the slab area is free by then) into point order and
	// leave as whole rows: written straight through the point index they were 8 scattered dword stores per point -- 1.5-3.7 partial
	// line writes per (point, level), as dear as the 4.25 reads the staging saves (first version: 73 us for levels 2-3)
	float res_y[kPer][2], res_j[kPer][2][3];
	// ---- slab after slab ----
	for (uint32_t t = 0; t < sl.n_slabs; ++t) {
		const uint32_t p0 = t * sl.nx;                                              // first plane of the slab
		const uint32_t planes = (p0 + sl.nx + 1u) <= L.res[0] ? sl.nx + 1u : L.res[0] - p0;
		if (t) __syncthreads();                                                    // the previous slab's readers are done
		{
			// eight loads in flight per thread (a plain copy loop issues one 8-byte load per round trip: 15 round trips per slab)
			const char *__restrict__ src = reinterpret_cast<const char *>(params + L.off) + (size_t)p0 * sx * (2 * sizeof(PT));
			const uint32_t total = planes * sx;
			for (uint32_t e0 = threadIdx.x; e0 < total; e0 += kLdsThreads * 8u) {
				float2 v[8];
#pragma unroll
				for (uint32_t u = 0; u < 8u; ++u) {
					const uint32_t e = e0 + u * kLdsThreads;
					v[u] = load_pair<PT>(src + (size_t)(e < total ? e : e0) * (2 * sizeof(PT)));
				}
#pragma unroll
				for (uint32_t u = 0; u < 8u; ++u) {
					const uint32_t e = e0 + u * kLdsThreads;
					if (e < total) slab_tab[e] = v[u];
				}
			}
		}
		__syncthreads();
		const uint32_t b0 = base[t], b1 = base[t + 1];
#pragma unroll
		for (uint32_t k4 = 0; k4 < kPer; ++k4) {
			const uint32_t p = k4 * kLdsThreads + threadIdx.x;
			if (p < b0 || p >= b1) continue;
			const uint32_t i = pt[k4];
			const float x0 = xp[k4][0], x1 = xp[k4][1], x2 = xp[k4][2];
			// cell locator (explicit fma: decides the integer cell, must match the oracle bit for bit)
			const float v0 = __fmaf_rn(x0, sc0, 0.5f), v1 = __fmaf_rn(x1, sc1, 0.5f), v2 = __fmaf_rn(x2, sc2, 0.5f);
			const float f0 = floorf(v0), f1 = floorf(v1), f2 = floorf(v2);
			float t0 = v0 - f0, t1 = v1 - f1, t2 = v2 - f2;
			float dw0 = sc0, dw1 = sc1, dw2 = sc2;             // scale * w'
			if (smooth) {
				dw0 *= 6.0f * t0 * (1.0f - t0); dw1 *= 6.0f * t1 * (1.0f - t1); dw2 *= 6.0f * t2 * (1.0f - t2);
				t0 = t0 * t0 * __fmaf_rn(-2.0f, t0, 3.0f); t1 = t1 * t1 * __fmaf_rn(-2.0f, t1, 3.0f); t2 = t2 * t2 * __fmaf_rn(-2.0f, t2, 3.0f);
			}
			// local plane of the cell inside the slab (clamped: memory safety for x outside [0, 1])
			uint32_t c0 = f0 > 0.0f ? (uint32_t)f0 : 0u;
			c0 = c0 >= p0 ? c0 - p0 : 0u;
			c0 = c0 + 2u <= planes ? c0 : planes - 2u;
			uint32_t c1 = f1 > 0.0f ? (uint32_t)f1 : 0u, c2 = f2 > 0.0f ? (uint32_t)f2 : 0u;
			c1 = c1 + 2u <= L.res[1] ? c1 : L.res[1] - 2u;
			c2 = c2 + 2u <= L.res[2] ? c2 : L.res[2] - 2u;
			const uint32_t e00 = (c0 * L.res[1] + c1) * L.res[2] + c2;
			const uint32_t e[4] = {e00, e00 + sx, e00 + sy, e00 + sx + sy};
			// the lerp tree of k_fwd_lds / k_fwd_pairlane for a Dense level (pair dim z, then x, then y), both features in one lane
			const float w1 = 1.0f - t2;
			float2 b[4], d[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const float2 lo = slab_tab[e[m]], hi = slab_tab[e[m] + 1u];
				d[m] = make_float2(hi.x - lo.x, lo.y - hi.y);
				b[m] = make_float2(__fmaf_rn(t2, d[m].x, lo.x), __fmaf_rn(w1, d[m].y, hi.y));
			}
			float yv[2], gx[2], gy[2], gz[2];
#pragma unroll
			for (int f = 0; f < 2; ++f) {
				auto c = [f](const float2 &v) { return f ? v.y : v.x; };
				const float cA0 = c(b[1]) - c(b[0]), cA1 = c(b[3]) - c(b[2]);
				const float dA0 = __fmaf_rn(t0, cA0, c(b[0])), dA1 = __fmaf_rn(t0, cA1, c(b[2]));
				const float eB = dA1 - dA0;
				yv[f] = __fmaf_rn(t1, eB, dA0);
				if (DYDX) {
					const float gA = __fmaf_rn(t1, cA1 - cA0, cA0);
					const float q0 = __fmaf_rn(t0, c(d[1]) - c(d[0]), c(d[0])), q1 = __fmaf_rn(t0, c(d[3]) - c(d[2]), c(d[2]));
					const float gP = __fmaf_rn(t1, q1 - q0, q0);
					gx[f] = gA * dw0; gy[f] = eB * dw1; gz[f] = gP * (f ? -dw2 : dw2);
				}
			}
			res_y[k4][0] = yv[0]; res_y[k4][1] = yv[1];
			if (DYDX) {
#pragma unroll
				for (int f = 0; f < 2; ++f) { res_j[k4][f][0] = gx[f]; res_j[k4][f][1] = gy[f]; res_j[k4][f][2] = gz[f]; }
			}
		}
	}
	// ---- results: registers -> LDS in point order -> whole rows ----
	__syncthreads();
	float *ob = reinterpret_cast<float *>(slab_tab);                               // y [2][kLdsPts] | dy_dx [2][kLdsPts][3]
#pragma unroll
	for (uint32_t k4 = 0; k4 < kPer; ++k4) {
		const uint32_t p = k4 * kLdsThreads + threadIdx.x;
		if (p >= n_here) continue;
		const uint32_t lp = (uint32_t)list[p];
#pragma unroll
		for (int f = 0; f < 2; ++f) {
			ob[f * kLdsPts + lp] = res_y[k4][f];
			if (DYDX) {
#pragma unroll
				for (int c = 0; c < 3; ++c) ob[2u * kLdsPts + ((uint32_t)f * kLdsPts + lp) * 3u + c] = res_j[k4][f][c];
			}
		}
	}
	__syncthreads();
#pragma unroll
	for (int f = 0; f < 2; ++f) {
		PT *yc = y + ((int64_t)g_first * y_sn + (int64_t)(sl.q * 2u + f) * y_se);
		for (uint32_t k = threadIdx.x; k < n_here; k += kLdsThreads) yc[(int64_t)k * y_sn] = from_f32<PT>(ob[f * kLdsPts + k]);
		if (DYDX) {
			float *dc = dydx + ((int64_t)g_first * d_sn + (int64_t)(sl.q * 2u + f) * d_se);
			const float *oj = ob + 2u * kLdsPts + (uint32_t)f * kLdsPts * 3u;
			if (d_sn == 3) for (uint32_t k = threadIdx.x; k < 3u * n_here; k += kLdsThreads) dc[k] = oj[k];       // feature-major: one run
			else for (uint32_t k = threadIdx.x; k < 3u * n_here; k += kLdsThreads) dc[(int64_t)(k / 3u) * d_sn + (k % 3u)] = oj[k];
		}
	}
}

// =============================================================================================
// dL/dparam (SECOND == false) and d(dL/dx)/dparam (SECOND == true)
// =============================================================================================
template <int D, int G, bool SECOND, bool DH, typename PT = float>
__global__ __launch_bounds__(kBlock) void k_bwd_dparam(Sched s, const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                                       int32_t min_level, int32_t max_level, uint32_t smooth,
                                                       const float *__restrict__ dL_ddLdx,
                                                       const float *__restrict__ dL_dy, int64_t g_sn, int64_t g_se,
                                                       const float *__restrict__ x, const PT *__restrict__ params,
                                                       Batch ba, float *__restrict__ dparam) {
	uint32_t q, chunk;
	if (!decode_block(s, blockIdx.x, q, chunk)) return;
	const uint32_t i = chunk * kBlock + threadIdx.x;
	if (i >= N) return;
	const uint32_t level = meta_level_of(md, q);
	if ((int32_t)level > max_level || (int32_t)level < min_level) return;
	uint32_t base = 0;
	if (!batch_base(ba, i, base)) return;
	const uint32_t foff0 = meta_cnt_of(md, q) * G;
	const uint32_t out0 = meta_col_of(md, q);
	const Lvl L = load_level(md, level);
	const auto grid = make_tab(params + (base + L.off));
	float *__restrict__ gg = dparam + (base + L.off);

	float xp[D], vin[D];
#pragma unroll
	for (int d = 0; d < D; ++d) {
		xp[d] = x[(size_t)i * D + d];
		vin[d] = SECOND ? dL_ddLdx[(size_t)i * D + d] : 0.0f;
	}
	Cell<D> c;
	locate<D>(xp, L, smooth != 0, c);

	float grad[G];
#pragma unroll
	for (int f = 0; f < G; ++f) grad[f] = dL_dy[(int64_t)i * g_sn + (int64_t)(out0 + f) * g_se];

	// per-dim seed of the second-order weights: scale_d * v_d * w'_d
	float a[D];
#pragma unroll
	for (int d = 0; d < D; ++d) a[d] = SECOND ? c.sc[d] * vin[d] * c.dw[d] : 0.0f;

	if (DH || L.type == NR3D_LOD_Dense || L.type == NR3D_LOD_Hash || L.type == NR3D_LOD_VectorMatrix ||
	    L.type == NR3D_LOD_VecZMatXoY || L.type == NR3D_LOD_CP || L.type == NR3D_LOD_NPlaneMul) {
		// One update per corner.  First order: W_c.  Second order: the reference issues, for every grad dim
		// d and every face corner, -w at the lower and +w at the upper corner (lotd_encoding.h:733-761);
		// all D contributions that land on corner c are summed here first.  The scatter rule is linear in
		// the weight for every type, so the result is the same sum.
#pragma unroll
		for (uint32_t k = 0; k < (1u << D); ++k) {
			float w;
			if (!SECOND) {
				w = corner_weight<D>(c, k);
			} else {
				w = 0.0f;
#pragma unroll
				for (int d = 0; d < D; ++d) {
					const float t = face_weight<D>(c, k, d, a[d]);
					w += ((k >> d) & 1u) ? t : -t;
				}
			}
			uint32_t p[D];
			corner_pos<D>(c, k, p);
			if (DH) {
				const uint32_t e = (L.type == NR3D_LOD_Dense) ? entry_dense<D>(L, p) : entry_hash<D>(L, p);
				float *dst = gg + (e * L.F + foff0);
#pragma unroll
				for (int f = 0; f < G; ++f) atomic_add_f32(dst + f, grad[f] * w);
			} else {
				corner_scatter<D, G>(L, grid, gg, foff0, p, grad, w);
			}
		}
	} else if (L.type == NR3D_LOD_NPlaneSum) {
		if constexpr (D > 2) {
#pragma unroll 1
			for (uint32_t jd = 0; jd < (uint32_t)D; ++jd)
#pragma unroll
				for (uint32_t k = 0; k < (1u << (D - 1)); ++k) {
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
					float *dst = gg + (entry_nplane_sum<D>(L, jd, pp) * L.F + foff0);
#pragma unroll
					for (int f = 0; f < G; ++f) atomic_add_f32(dst + f, grad[f] * w);
				}
		}
	} else if (L.type == NR3D_LOD_CPfast) {
		float lv[D][2][G];
		uint32_t le[D][2];
#pragma unroll
		for (int d = 0; d < D; ++d) {
			le[d][0] = entry_line<D>(L, d, c.g[d]) * L.F + foff0;
			le[d][1] = entry_line<D>(L, d, c.g[d] + 1u) * L.F + foff0;
#pragma unroll
			for (int f = 0; f < G; ++f) { lv[d][0][f] = grid[le[d][0] + f]; lv[d][1][f] = grid[le[d][1] + f]; }
		}
		if (!SECOND) {
			// reference lotd_encoding.h:653-705
#pragma unroll
			for (int gd = 0; gd < D; ++gd)
#pragma unroll
				for (int f = 0; f < G; ++f) {
					float gl = grad[f];
#pragma unroll
					for (int d = 0; d < D; ++d)
						if (d != gd) gl *= __fmaf_rn(c.w[d], lv[d][1][f], (1.0f - c.w[d]) * lv[d][0][f]);
					atomic_add_f32(gg + le[gd][0] + f, gl * (1.0f - c.w[gd]));
					atomic_add_f32(gg + le[gd][1] + f, gl * c.w[gd]);
				}
		} else {
			// reference lotd_encoding.h:970-1038: for every (line dim ld, grad dim gd) pair
#pragma unroll
			for (int ld = 0; ld < D; ++ld)
#pragma unroll
				for (int f = 0; f < G; ++f) {
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
					atomic_add_f32(gg + le[ld][0] + f, acc_l);
					atomic_add_f32(gg + le[ld][1] + f, acc_r);
				}
		}
	}
}

// =============================================================================================
// d(dL/dx)/dx  -- Hessian-vector product; Dense / Hash / VM / VecZMatXoY only (other types: 0)
// Each lane owns ALL pseudo levels of one point (no cross-lane atomics on dL_dx).
// =============================================================================================
// acc[d] += sum_e vin[e] * H[e][d],  H = d^2 (sum_c W_c s_c) / dx_e dx_d, s_c = the corner's value . grad
template <int D>
__device__ __forceinline__ void hvp_from_sdot(const Cell<D> &c, uint32_t smooth, const float (&vin)[D], const float (&sdot)[1 << D],
                                              float (&acc)[D]) {
#pragma unroll
	for (int d = 0; d < D; ++d) {
		float o = 0.0f;
#pragma unroll
		for (int e = 0; e < D; ++e) {
			if (e == d && !smooth) continue;    // linear: zero diagonal
			const float seed = (e == d) ? (c.sc[d] * vin[d]) * (c.sc[d] * c.ddw[d])
			                            : (c.sc[e] * vin[e] * c.dw[e]) * (c.dw[d] * c.sc[d]);
#pragma unroll
			for (uint32_t k = 0; k < (1u << D); ++k) {
				// differentiated dims contribute the sign of the corner (+upper, -lower),
				// the others their interpolation weight (for e == d the sign appears once)
				float w = seed;
#pragma unroll
				for (int m = 0; m < D; ++m) {
					const bool up = (k >> m) & 1u;
					if (m == d || m == e) w *= up ? 1.0f : -1.0f;
					else w *= up ? c.w[m] : (1.0f - c.w[m]);
				}
				o = __fmaf_rn(w, sdot[k], o);
			}
		}
		acc[d] += o;
	}
}

// d(dL/dx)/dx contribution of ONE pseudo level to one point: acc[d] += sum_e vin[e] * d^2(sum_f grad_f y_f)/dx_e dx_d
// (reference: kernel_lod_backward_input_backward_input, lotd_encoding.h:1157-1298; Dense / Hash / VM / VecZMatXoY only)
template <int D, int G, bool DH, typename TB>
__device__ __forceinline__ void hvp_level(const nr3d_lotd_meta_t *__restrict__ md, uint32_t q, const Lvl &L, const Cell<D> &c,
                                          uint32_t smooth, const float (&vin)[D], uint32_t i, const float *__restrict__ dL_dy,
                                          int64_t g_sn, int64_t g_se, TB grid, bool vec_ok,
                                          float (&acc)[D]) {
#pragma unroll 1
	for (int f0 = 0; f0 < G; f0 += 2) {
		const uint32_t foff = meta_cnt_of(md, q) * G + f0;
		float grad[2];
		const uint32_t col0 = meta_col_of(md, q) + (uint32_t)f0;
		grad[0] = dL_dy[(int64_t)i * g_sn + (int64_t)col0 * g_se];
		grad[1] = dL_dy[(int64_t)i * g_sn + (int64_t)(col0 + 1u) * g_se];
		// s[k] = sum_f value(corner k)[f] * grad[f]
		float sdot[1 << D];
		{
			float v[1 << D][2];
			corner_values_pair<D, DH ? kOnlyDenseHash : -1>(L, grid, foff, vec_ok && (L.F & 1u) == 0u, c, v);
#pragma unroll
			for (uint32_t k = 0; k < (1u << D); ++k)      // same arithmetic as corner_dot(..., weight = 1)
				sdot[k] = __fmaf_rn(v[k][1] * grad[1], 1.0f, __fmaf_rn(v[k][0] * grad[0], 1.0f, 0.0f));
		}
		hvp_from_sdot<D>(c, smooth, vin, sdot, acc);
	}
}
__device__ __forceinline__ bool hvp_type(uint32_t t) {
	return t == NR3D_LOD_Dense || t == NR3D_LOD_Hash || t == NR3D_LOD_VectorMatrix || t == NR3D_LOD_VecZMatXoY;
}

// The same for ALL nq pseudo levels of one level at once (metas with product-type levels, lane-serial kernel; round 3).
// The Hessian-vector product is linear in s_c = value(corner c) . grad, so the dot product is accumulated over every feature
// of the level first and hvp_from_sdot runs once per LEVEL instead of once per feature pair; and the table entries are read
// four features per 16-byte load when the level's entries are aligned (the kernel was L2-request bound on configs[3]:
// 625 M requests = 177 G/s, profiles/r03g_c4_counters.txt).  Same sum, other association than the per-pair form.
template <int D, int ONLY = -1, typename TB>
__device__ __forceinline__ void hvp_level_merged(const nr3d_lotd_meta_t *__restrict__ md, uint32_t q, uint32_t nq, uint32_t G,
                                                 const Lvl &L, const Cell<D> &c, uint32_t smooth, const float (&vin)[D], uint32_t i,
                                                 const float *__restrict__ dL_dy, int64_t g_sn, int64_t g_se,
                                                 TB grid, bool vec_ok, bool aligned4, float (&acc)[D]) {
	const uint32_t f_begin = meta_cnt_of(md, q) * G, n_f = nq * G, col_begin = meta_col_of(md, q);
	float sdot[1 << D];
#pragma unroll
	for (uint32_t k = 0; k < (1u << D); ++k) sdot[k] = 0.0f;
	const bool quads = vec_ok && aligned4 && (L.F & 3u) == 0u && (f_begin & 3u) == 0u && (n_f & 3u) == 0u;
	if (quads) {
#pragma unroll 1
		for (uint32_t f0 = 0; f0 < n_f; f0 += 4) {
			float grad[4], v[1 << D][4];
#pragma unroll
			for (int j = 0; j < 4; ++j) grad[j] = dL_dy[(int64_t)i * g_sn + (int64_t)(col_begin + f0 + j) * g_se];
			corner_values_pair<D, ONLY, 4>(L, grid, f_begin + f0, true, c, v);
#pragma unroll
			for (uint32_t k = 0; k < (1u << D); ++k)
#pragma unroll
				for (int j = 0; j < 4; ++j) sdot[k] = __fmaf_rn(v[k][j], grad[j], sdot[k]);
		}
	} else {
#pragma unroll 1
		for (uint32_t f0 = 0; f0 < n_f; f0 += 2) {
			float grad[2], v[1 << D][2];
#pragma unroll
			for (int j = 0; j < 2; ++j) grad[j] = dL_dy[(int64_t)i * g_sn + (int64_t)(col_begin + f0 + j) * g_se];
			corner_values_pair<D, ONLY, 2>(L, grid, f_begin + f0, vec_ok && (L.F & 1u) == 0u, c, v);
#pragma unroll
			for (uint32_t k = 0; k < (1u << D); ++k)
#pragma unroll
				for (int j = 0; j < 2; ++j) sdot[k] = __fmaf_rn(v[k][j], grad[j], sdot[k]);
		}
	}
	hvp_from_sdot<D>(c, smooth, vin, sdot, acc);
}

// one lane = one point, the pseudo levels one after another (no workspace needed).
// ONLY / level_mask / accumulate (metas with product types, 3-D): the all-types instantiation needs 156 VGPRs (4 waves per
// SIMD against ~1 us gathers: 322 M L2 requests at 155 G/s on configs[3]); the levels are therefore served by one launch per
// level TYPE (kOnlyDenseHash, CP, VM, the rest) through single-type instantiations, each walking the levels of its mask in
// order and continuing the sum the previous launch left in dL_dx -- for metas whose levels are grouped by type (configs[3])
// the association of the sum is the one-launch kernel's.
template <int D, int G, bool DH = false, typename PT = float, int ONLY = -1>
__global__ __launch_bounds__(kBlock) void k_bwd_bwd_dx(const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                                       uint32_t n_pseudo, int32_t max_level, uint32_t smooth,
                                                       const float *__restrict__ dL_ddLdx,
                                                       const float *__restrict__ dL_dy, int64_t g_sn, int64_t g_se,
                                                       const float *__restrict__ x, const PT *__restrict__ params,
                                                       Batch ba, bool vec_ok, float *__restrict__ dL_dx,
                                                       uint32_t level_mask, uint32_t accumulate) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= N) return;
	float acc[D];
#pragma unroll
	for (int d = 0; d < D; ++d) acc[d] = accumulate ? dL_dx[(size_t)i * D + d] : 0.0f;
	uint32_t base = 0;
	const bool ok = batch_base(ba, i, base);
	if (ok) {
		float xp[D], vin[D];
#pragma unroll
		for (int d = 0; d < D; ++d) { xp[d] = x[(size_t)i * D + d]; vin[d] = dL_ddLdx[(size_t)i * D + d]; }
#pragma unroll 1
		for (uint32_t q = 0; q < n_pseudo;) {
			const uint32_t level = meta_level_of(md, q);
			uint32_t nq = 1;
			if constexpr (!DH)                        // all pseudo levels of this level in one go (they are consecutive)
				while (q + nq < n_pseudo && meta_level_of(md, q + nq) == level) ++nq;
			const uint32_t q0 = q;
			q += nq;
			if ((int32_t)level > max_level || !((level_mask >> (level & 31u)) & 1u)) continue;
			const Lvl L = load_level(md, level);
			if (!DH && !hvp_type(L.type)) continue;
			Cell<D> c;
			locate<D>(xp, L, smooth != 0, c);
			const auto grid = make_tab(params + (base + L.off));
			if constexpr (DH)
				hvp_level<D, G, DH>(md, q0, L, c, smooth, vin, i, dL_dy, g_sn, g_se, grid, vec_ok, acc);
			else
				hvp_level_merged<D, ONLY>(md, q0, nq, G, L, c, smooth, vin, i, dL_dy, g_sn, g_se, grid, vec_ok,
				                          (tab_addr(grid) % (4u * tab_elt(grid))) == 0u, acc);
		}
	}
#pragma unroll
	for (int d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = acc[d];
}

// The same with one lane per (point, pseudo level), level-major blocks on the forward's XCD-affine schedule: the lane-serial
// form above keeps 8 gathers of ONE level in flight per lane and walks all tables in every workgroup; here the levels
// of a point run side by side (a wave works on one table) and leave their D floats in partial[q][i][:], which
// k_sum_levels adds up in level order -- the serial kernel's order, so (for 2-feature pseudo levels) its bits.
template <int D, int G, bool DH, typename PT = float>
__global__ __launch_bounds__(kBlock) void k_bwd_bwd_dx_lv(Sched s, const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                                          int32_t max_level, uint32_t smooth, const float *__restrict__ dL_ddLdx,
                                                          const float *__restrict__ dL_dy, int64_t g_sn, int64_t g_se,
                                                          const float *__restrict__ x, const PT *__restrict__ params, Batch ba,
                                                          bool vec_ok, float *__restrict__ partial) {
	uint32_t q, chunk;
	if (!decode_block(s, blockIdx.x, q, chunk)) return;
	const uint32_t i = chunk * kBlock + threadIdx.x;
	if (i >= N) return;
	float acc[D];
#pragma unroll
	for (int d = 0; d < D; ++d) acc[d] = 0.0f;
	const uint32_t level = meta_level_of(md, q);
	uint32_t base = 0;
	if ((int32_t)level <= max_level && batch_base(ba, i, base)) {
		const Lvl L = load_level(md, level);
		if (DH || hvp_type(L.type)) {
			float xp[D], vin[D];
#pragma unroll
			for (int d = 0; d < D; ++d) { xp[d] = x[(size_t)i * D + d]; vin[d] = dL_ddLdx[(size_t)i * D + d]; }
			Cell<D> c;
			locate<D>(xp, L, smooth != 0, c);
			hvp_level<D, G, DH>(md, q, L, c, smooth, vin, i, dL_dy, g_sn, g_se, make_tab(params + (base + L.off)), vec_ok, acc);
		}
	}
	float *dst = partial + ((size_t)q * N + i) * D;
#pragma unroll
	for (int d = 0; d < D; ++d) __builtin_nontemporal_store(acc[d], dst + d);
}

// ... and with the forward's TWO lanes per (point, pseudo level) (k_fwd_pairlane: 3-D Dense / Hash, 2-feature pseudo
// levels, unbatched), in the forward's lean form (round 3; the first pair-lane form formed s_c = v_c . grad per corner and
// ran the 8-corner sum hvp_from_sdot: 423 VALU instructions per wave, 630 us for 2^20 points against the forward's 333).
// The contraction with dL_dy is linear, so lane s keeps FEATURE s: it gathers the side-s corner of the four pairs, swaps one
// value per pair (as the forward does) and builds the Hessian of its feature's interpolant from the lerp tree's own
// differences.  With P the pair dim, A, B the other two, d[m] the (signed) difference along P at the (A, B) corner m and
// b[m] the value lerped along P:
//     G_B = lerp_A(b2, b3) - lerp_A(b0, b1)       M_AB = (b3 - b2) - (b1 - b0)            G_A = lerp_B(b1 - b0, b3 - b2)
//     M_PB = lerp_A(d2, d3) - lerp_A(d0, d1)      M_PA = lerp_B(d1 - d0, d3 - d2)         G_P = lerp_B(lerp_A(d0, d1), lerp_A(d2, d3))
//     H[d][e] = s_d w'_d s_e w'_e M_de  (d != e),     H[d][d] = s_d^2 w''_d G_d  (smoothstep only; 0 for linear)
// out = (H v) * dL_dy[feature s]; the partners add their two features through one more DPP swap and lane 0 stores.  Same
// polynomial as hvp_from_sdot (the oracle's corner sum), other association: agreement to fp32 rounding of the level's values.
// dL_dy [N, 2 P] (any strides) -> g_pairs [P][N][2]: a level's pass of the kernel below reads its two columns as one
// contiguous stream.  Reading them in place costs as much as the table gathers (measured, tools/exp_hvp_variants.py,
// profiles/r03i_hvp_experiments.txt: 673 -> 401 us per 2^20 points without the dL_dy loads): 8 of every row's 128 bytes per
// level, and every level's XCD pulls the whole line across the fabric.
constexpr int kGpPts = 128, kGpLv = 16;            // tile: points x pseudo levels
__global__ __launch_bounds__(kBlock) void k_hvp_pairs(uint32_t N, uint32_t P, const float *__restrict__ dL_dy, int64_t g_sn,
                                                      int64_t g_se, float *__restrict__ g_pairs) {
	__shared__ float2 tile[kGpLv][kGpPts + 1];
	const uint32_t i0 = blockIdx.x * kGpPts, q0 = blockIdx.y * kGpLv;
#pragma unroll
	for (uint32_t k = 0; k < kGpPts * kGpLv / kBlock; ++k) {
		const uint32_t idx = k * kBlock + threadIdx.x, i = idx / kGpLv, qq = idx % kGpLv;
		float2 v = make_float2(0.0f, 0.0f);
		if (i0 + i < N && q0 + qq < P) {
			const float *src = dL_dy + (int64_t)(i0 + i) * g_sn + (int64_t)((q0 + qq) * 2u) * g_se;
			v = make_float2(src[0], src[g_se]);
		}
		tile[qq][i] = v;
	}
	__syncthreads();
#pragma unroll
	for (uint32_t k = 0; k < kGpPts * kGpLv / kBlock; ++k) {
		const uint32_t idx = k * kBlock + threadIdx.x, qq = idx / kGpPts, i = idx % kGpPts;
		if (i0 + i < N && q0 + qq < P)
		{
			typedef float f2v __attribute__((ext_vector_type(2)));
			const float2 v = tile[qq][i];
			f2v o; o.x = v.x; o.y = v.y;
			__builtin_nontemporal_store(o, reinterpret_cast<f2v *>(g_pairs) + (size_t)(q0 + qq) * N + i0 + i);
		}
	}
}

constexpr int kHvpSub = 2;                         // 32-point groups per wave
template <int SUB, typename PT>
__global__ __launch_bounds__(kBlock) void k_bwd_bwd_dx_pl(Sched s, const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                                          int32_t max_level, uint32_t smooth, const float *__restrict__ dL_ddLdx,
                                                          const float *__restrict__ g_pairs, int64_t g_fm,
                                                          const float *__restrict__ x, const PT *__restrict__ params,
                                                          float *__restrict__ partial NR3D_DBG_PARAM) {
	NR3D_DBG_DECL
	// g_fm == 0: g_pairs is the [pseudo level][point][2] copy of k_hvp_pairs; > 0: the caller's dL_dy is feature-major already
	// (element (i, e) at e * g_fm + i, e.g. the copy the dL/dparam pass of the same step uses) and is read in place
	uint32_t q, chunk;
	if (!decode_block(s, blockIdx.x, q, chunk)) return;
	constexpr uint32_t kGroup = 32;
	constexpr uint32_t kPts = kPlPts * SUB;
	const uint32_t lane = threadIdx.x & 63u, side = lane & 1u, pl = lane >> 1;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t w0 = chunk * kPts + wave * kGroup * SUB;
	if (w0 >= N) return;
	const uint32_t level = meta_level_of(md, q);
	const uint32_t foff0 = meta_cnt_of(md, q) * 2u;
	const bool live = (int32_t)level <= max_level;
	const Lvl L = load_level(md, live ? level : 0u);
	const bool dense = L.type == NR3D_LOD_Dense;
	const char *__restrict__ base = reinterpret_cast<const char *>(params + L.off + foff0);
	const uint32_t stride = L.F * (uint32_t)sizeof(PT);
	const bool small = L.size < (1u << 24);
	const float sc0 = (float)(L.res[0] - 2u), sc1 = (float)(L.res[1] - 2u), sc2 = (float)(L.res[2] - 2u);

	// ---- phase 1: x, v = dL_ddLdx and this lane's dL_dy column (lanes beyond N re-read the last point and store nothing)
	float xp[SUB][3], vp[SUB][3], gs[SUB];
#pragma unroll
	for (int u = 0; u < SUB; ++u) {
		const uint32_t g0 = w0 + (uint32_t)u * kGroup;
		const uint32_t gc = g0 < N ? g0 : w0;
		uint32_t pi = pl;
		if (gc + kGroup > N) { const uint32_t i = gc + pl; pi = (i < N ? i : N - 1u) - gc; }
		const float *px = reinterpret_cast<const float *>(reinterpret_cast<const char *>(x) + (size_t)gc * 12u + pi * 12u);
		const float *pv = reinterpret_cast<const float *>(reinterpret_cast<const char *>(dL_ddLdx) + (size_t)gc * 12u + pi * 12u);
		const char *gb = reinterpret_cast<const char *>(g_pairs) +
		                 (g_fm ? ((size_t)(q * 2u + side) * (size_t)g_fm + gc) * 4u       // row of column 2 q + side (lane-dependent: VGPR base)
		                       : ((size_t)q * N + gc) * 8u);                             // [pseudo level][point][2]
		const uint32_t g_lane = g_fm ? pi * 4u : pi * 8u + side * 4u;
		xp[u][0] = px[0]; xp[u][1] = px[1]; xp[u][2] = px[2];
		if (dbg & 8u) { vp[u][0] = xp[u][1]; vp[u][1] = xp[u][2]; vp[u][2] = xp[u][0]; }
		else { vp[u][0] = pv[0]; vp[u][1] = pv[1]; vp[u][2] = pv[2]; }
		gs[u] = (dbg & 4u) ? 1.0f + (float)side : *reinterpret_cast<const float *>(gb + g_lane);
	}
	float ox[SUB], oy[SUB], oz[SUB];
#pragma unroll
	for (int u = 0; u < SUB; ++u) { ox[u] = 0.0f; oy[u] = 0.0f; oz[u] = 0.0f; }
	if (live) {
		// ---- phase 2: cells, corner offsets, all 4 * SUB gathers in flight (the forward's addressing)
		float2 t[SUB][4];
		float wP[SUB], wA[SUB], wB[SUB], s1P[SUB], s1A[SUB], s1B[SUB], s2P[SUB], s2A[SUB], s2B[SUB], vP[SUB], vA[SUB], vB[SUB];
#pragma unroll
		for (int u = 0; u < SUB; ++u) {
			const float v0 = __fmaf_rn(xp[u][0], sc0, 0.5f), v1 = __fmaf_rn(xp[u][1], sc1, 0.5f), v2 = __fmaf_rn(xp[u][2], sc2, 0.5f);
			const float f0 = floorf(v0), f1 = floorf(v1), f2 = floorf(v2);
			float t0 = v0 - f0, t1 = v1 - f1, t2 = v2 - f2;
			const uint32_t c0 = (uint32_t)f0, c1 = (uint32_t)f1, c2 = (uint32_t)f2;
			float a0 = sc0, a1 = sc1, a2 = sc2;            // scale * w'
			float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f;          // scale^2 * w''
			if (smooth) {
				h0 = (sc0 * sc0) * __fmaf_rn(-12.0f, t0, 6.0f); h1 = (sc1 * sc1) * __fmaf_rn(-12.0f, t1, 6.0f); h2 = (sc2 * sc2) * __fmaf_rn(-12.0f, t2, 6.0f);
				a0 *= 6.0f * t0 * (1.0f - t0); a1 *= 6.0f * t1 * (1.0f - t1); a2 *= 6.0f * t2 * (1.0f - t2);
				t0 = t0 * t0 * __fmaf_rn(-2.0f, t0, 3.0f); t1 = t1 * t1 * __fmaf_rn(-2.0f, t1, 3.0f); t2 = t2 * t2 * __fmaf_rn(-2.0f, t2, 3.0f);
			}
			uint32_t e[4], off[4];
			if (dense) {
				uint32_t e00;
				if (small) e00 = __umul24(__umul24(c0, L.res[1]) + c1, L.res[2]) + c2 + side;
				else e00 = (c0 * L.res[1] + c1) * L.res[2] + c2 + side;
				const uint32_t sx = L.res[1] * L.res[2], sy = L.res[2];
				e[0] = e00; e[1] = e00 + sx; e[2] = e00 + sy; e[3] = e00 + sx + sy;
				wP[u] = t2; wA[u] = t0; wB[u] = t1; s1P[u] = a2; s1A[u] = a0; s1B[u] = a1; s2P[u] = h2; s2A[u] = h0; s2B[u] = h1;
				vP[u] = vp[u][2]; vA[u] = vp[u][0]; vB[u] = vp[u][1];
			} else {
				const uint32_t hy0 = c1 * kPrimes[1], hy1 = hy0 + kPrimes[1];
				const uint32_t hz0 = c2 * kPrimes[2], hz1 = hz0 + kPrimes[2];
				const uint32_t xs = c0 + side;
				const uint32_t b0 = xs ^ hy0, b1 = xs ^ hy1;
				e[0] = b0 ^ hz0; e[1] = b1 ^ hz0; e[2] = b0 ^ hz1; e[3] = b1 ^ hz1;
				if ((L.size & (L.size - 1u)) == 0u) {
					const uint32_t mask = L.size - 1u;
#pragma unroll
					for (int m = 0; m < 4; ++m) e[m] &= mask;
				} else {
#pragma unroll
					for (int m = 0; m < 4; ++m) e[m] %= L.size;
				}
				wP[u] = t0; wA[u] = t1; wB[u] = t2; s1P[u] = a0; s1A[u] = a1; s1B[u] = a2; s2P[u] = h0; s2A[u] = h1; s2B[u] = h2;
				vP[u] = vp[u][0]; vA[u] = vp[u][1]; vB[u] = vp[u][2];
			}
			if (small) {
#pragma unroll
				for (int m = 0; m < 4; ++m) off[m] = __umul24(e[m], stride);
			} else {
#pragma unroll
				for (int m = 0; m < 4; ++m) off[m] = e[m] * stride;
			}
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				if (dbg & 2u) t[u][m] = make_float2(__int_as_float(off[m] | 0x3f000000u), __int_as_float(off[m] ^ 0x3f123456u));
				else t[u][m] = load_pair<PT>(base + off[m]);
			}
		}
		// ---- phase 3: the lerp tree's differences -> gradient and mixed second differences -> H v
#pragma unroll
		for (int u = 0; u < SUB; ++u) {
			const float wk = side ? 1.0f - wP[u] : wP[u];
			float b[4], d[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const float keep = side ? t[u][m].y : t[u][m].x;
				const float send = side ? t[u][m].x : t[u][m].y;
				const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xf, 0xf, true));
				d[m] = recv - keep;
				b[m] = __fmaf_rn(wk, d[m], keep);
			}
			const float cA0 = b[1] - b[0], cA1 = b[3] - b[2];
			const float mAB = cA1 - cA0;
			const float q0 = d[1] - d[0], q1 = d[3] - d[2];
			const float p0 = __fmaf_rn(wA[u], q0, d[0]), p1 = __fmaf_rn(wA[u], q1, d[2]);
			const float mPB = p1 - p0;                      // the signed differences along P carry sgn = (side ? -1 : 1)
			const float mPA = __fmaf_rn(wB[u], q1 - q0, q0);
			const float sP = side ? -s1P[u] : s1P[u];
			const float uP = vP[u] * sP, uA = vA[u] * s1A[u], uB = vB[u] * s1B[u];
			float oP = sP * __fmaf_rn(uA, mPA, uB * mPB);
			float oA = s1A[u] * __fmaf_rn(uP, mPA, uB * mAB);
			float oB = s1B[u] * __fmaf_rn(uP, mPB, uA * mAB);
			if (smooth) {
				const float dA0 = __fmaf_rn(wA[u], cA0, b[0]), dA1 = __fmaf_rn(wA[u], cA1, b[2]);
				const float gB = dA1 - dA0;
				const float gA = __fmaf_rn(wB[u], mAB, cA0);
				const float gP = __fmaf_rn(wB[u], mPB, p0);
				oP = __fmaf_rn(vP[u] * (side ? -s2P[u] : s2P[u]), gP, oP);
				oA = __fmaf_rn(vA[u] * s2A[u], gA, oA);
				oB = __fmaf_rn(vB[u] * s2B[u], gB, oB);
			}
			oP *= gs[u]; oA *= gs[u]; oB *= gs[u];
			// feature 0 (even lane) + feature 1 (odd lane); only the even lane's sum is stored
			oP += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(oP), 0xB1, 0xf, 0xf, true));
			oA += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(oA), 0xB1, 0xf, 0xf, true));
			oB += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(oB), 0xB1, 0xf, 0xf, true));
			if (dense) { ox[u] = oA; oy[u] = oB; oz[u] = oP; }
			else { ox[u] = oP; oy[u] = oA; oz[u] = oB; }
		}
	}
	// ---- phase 4: partial[q][i][:] (uniform base + 32-bit lane offset)
#pragma unroll
	for (int u = 0; u < SUB; ++u) {
		const uint32_t g0 = w0 + (uint32_t)u * kGroup;
		if ((dbg & 1u) && ox[u] + oy[u] + oz[u] != 1234.56789f) continue;
		if (side == 0u && g0 + pl < N) {
			float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(partial) + ((size_t)q * N + g0) * 12u + pl * 12u);
			__builtin_nontemporal_store(ox[u], &dst[0]);
			__builtin_nontemporal_store(oy[u], &dst[1]);
			__builtin_nontemporal_store(oz[u], &dst[2]);
		}
	}
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_sum_levels(uint32_t N, uint32_t n_pseudo, const float *__restrict__ partial,
                                                       float *__restrict__ dL_dx) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= N) return;
	float acc[D];
#pragma unroll
	for (int d = 0; d < D; ++d) acc[d] = 0.0f;
	for (uint32_t q = 0; q < n_pseudo; ++q) {
		const float *src = partial + ((size_t)q * N + i) * D;
#pragma unroll
		for (int d = 0; d < D; ++d) acc[d] += __builtin_nontemporal_load(src + d);
	}
#pragma unroll
	for (int d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = acc[d];
}


// =============================================================================================
// Dense contractions with the stored Jacobian
// =============================================================================================
template <int D>
__global__ __launch_bounds__(kBlock) void k_contract_dx(uint32_t N, uint32_t E, const float *__restrict__ dL_dy,
                                                        int64_t g_sn, int64_t g_se, const float *__restrict__ dydx,
                                                        int64_t d_sn, int64_t d_se, float *__restrict__ dL_dx) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= N) return;
	float acc[D];
#pragma unroll
	for (int d = 0; d < D; ++d) acc[d] = 0.0f;
	const float *gy = dL_dy + (int64_t)i * g_sn;
	const float *jj = dydx + (int64_t)i * d_sn;
#pragma unroll 4
	for (uint32_t e = 0; e < E; ++e) {
		const float g = gy[(int64_t)e * g_se];
		const float *j = jj + (int64_t)e * d_se;
#pragma unroll
		for (int d = 0; d < D; ++d) acc[d] = __fmaf_rn(g, j[d], acc[d]);
	}
#pragma unroll
	for (int d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = acc[d];
}

// Same contraction for a row-major dL/dy ([N, E], the layout autograd hands over): a lane reading its own row
// touches one cache line per lane and instruction, so the block first transposes its 256 x 32 tile of dL/dy through
// LDS (coalesced 16-byte reads), then streams the feature-major Jacobian.  Same summation order as above.
template <int D, typename GT>
__global__ __launch_bounds__(kBlock) void k_contract_dx_rowmajor(uint32_t N, uint32_t E, const GT *__restrict__ dL_dy,
                                                                 const float *__restrict__ dydx, int64_t d_sn,
                                                                 int64_t d_se, float *__restrict__ dL_dx,
                                                                 float *__restrict__ dL_dy_T) {
	constexpr int TE = 32;
	__shared__ float tile[TE][kBlock + 1];
	const uint32_t i0 = blockIdx.x * kBlock, i = i0 + threadIdx.x;
	const uint32_t n_here = min((uint32_t)kBlock, N - i0);
	float acc[D];
#pragma unroll
	for (int d = 0; d < D; ++d) acc[d] = 0.0f;
	for (uint32_t e0 = 0; e0 < E; e0 += TE) {
		const uint32_t te = min((uint32_t)TE, E - e0);
		if (te == TE && (E & 3u) == 0u) {
#pragma unroll
			for (uint32_t v = threadIdx.x; v < kBlock * TE / 4; v += kBlock) {   // 8 lanes x float4 per point row
				const uint32_t p = v >> 3, e4 = (v & 7u) * 4u;
				if (p < n_here) {
					float4 t;
					if constexpr (sizeof(GT) == 4) {
						t = *reinterpret_cast<const float4 *>(dL_dy + (size_t)(i0 + p) * E + e0 + e4);
					} else {                            // half gradients ((float, half, float) type combination): 8-byte reads
						const __half2 *h = reinterpret_cast<const __half2 *>(dL_dy + (size_t)(i0 + p) * E + e0 + e4);
						const float2 a = __half22float2(h[0]), b = __half22float2(h[1]);
						t = make_float4(a.x, a.y, b.x, b.y);
					}
					tile[e4][p] = t.x; tile[e4 + 1][p] = t.y; tile[e4 + 2][p] = t.z; tile[e4 + 3][p] = t.w;
				}
			}
		} else {
			for (uint32_t v = threadIdx.x; v < kBlock * te; v += kBlock) {
				const uint32_t p = v / te, e = v - p * te;
				if (p < n_here) tile[e][p] = to_f32<GT>(dL_dy[(size_t)(i0 + p) * E + e0 + e]);
			}
		}
		__syncthreads();
		if (dL_dy_T && i < N) {      // by-product: the feature-major copy the parameter scatter wants (coalesced rows)
#pragma unroll 8
			for (uint32_t e = 0; e < te; ++e) __builtin_nontemporal_store(tile[e][threadIdx.x], &dL_dy_T[(size_t)(e0 + e) * N + i]);
		}
		if (i < N) {
			const float *jj = dydx + (int64_t)i * d_sn + (int64_t)e0 * d_se;
#pragma unroll 8
			for (uint32_t e = 0; e < te; ++e) {
				const float g = tile[e][threadIdx.x];
				const float *j = jj + (int64_t)e * d_se;
#pragma unroll
				for (int d = 0; d < D; ++d) acc[d] = __fmaf_rn(g, __builtin_nontemporal_load(j + d), acc[d]);   // streamed once
			}
		}
		__syncthreads();
	}
	if (i < N) {
#pragma unroll
		for (int d = 0; d < D; ++d) dL_dx[(size_t)i * D + d] = acc[d];
	}
}

template <int D>
__global__ __launch_bounds__(kBlock) void k_contract_ddLdy(uint32_t N, uint32_t E, const float *__restrict__ v,
                                                           const float *__restrict__ dydx, int64_t d_sn, int64_t d_se,
                                                           float *__restrict__ out, int64_t o_sn, int64_t o_se) {
	// 2-D launch: blockIdx.y = encoded dim (keeps feature-major Jacobian reads and feature-major writes coalesced)
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t e = blockIdx.y;
	if (i >= N) return;
	const float *j = dydx + (int64_t)i * d_sn + (int64_t)e * d_se;
	float r = 0.0f;
#pragma unroll
	for (int d = 0; d < D; ++d) r = __fmaf_rn(v[(size_t)i * D + d], j[d], r);
	out[(int64_t)i * o_sn + (int64_t)e * o_se] = r;
}

// =============================================================================================
// Corner parameter indices (Dense / Hash only) -> int64 [N, E, 2^D]
// =============================================================================================
template <int D, int G>
__global__ __launch_bounds__(kBlock) void k_grid_index(Sched s, const nr3d_lotd_meta_t *__restrict__ md, uint32_t N,
                                                       uint32_t E, int32_t max_level, const float *__restrict__ x,
                                                       Batch ba, int64_t *__restrict__ out) {
	uint32_t q, chunk;
	if (!decode_block(s, blockIdx.x, q, chunk)) return;
	const uint32_t i = chunk * kBlock + threadIdx.x;
	if (i >= N) return;
	const uint32_t level = meta_level_of(md, q);
	if ((int32_t)level > max_level) return;
	uint32_t base = 0;
	if (!batch_base(ba, i, base)) return;
	const Lvl L = load_level(md, level);
	const uint32_t foff0 = meta_cnt_of(md, q) * G;
	float xp[D];
#pragma unroll
	for (int d = 0; d < D; ++d) xp[d] = x[(size_t)i * D + d];
	Cell<D> c;
	locate<D>(xp, L, false, c);
	int64_t *dst = out + ((size_t)i * E + (size_t)meta_col_of(md, q)) * (1u << D);
#pragma unroll
	for (uint32_t k = 0; k < (1u << D); ++k) {
		uint32_t p[D];
		corner_pos<D>(c, k, p);
		const uint32_t e = (L.type == NR3D_LOD_Dense) ? entry_dense<D>(L, p) : entry_hash<D>(L, p);
		const uint32_t ind = base + L.off + e * L.F + foff0;
#pragma unroll
		for (int f = 0; f < G; ++f) dst[k + f * (1u << D)] = (int64_t)(uint32_t)(ind + f);
	}
}

// =============================================================================================
// Host side
// =============================================================================================
// knobs of the experiments build (options.h): the schedule mode, and LOTD_SCHED_EXCL = 0 for the cost-balanced work line in the
// two-lane forward like every other kernel (A/B)
static uint32_t sched_mode_default() { const int64_t m = NR3D_XOPT(LOTD_SCHED, 3); return (m < 0 || m > 3) ? 3u : (uint32_t)m; }
static bool sched_exclusive_enabled() { return NR3D_XOPT(LOTD_SCHED_EXCL, 1) != 0; }

// estimated cost of one (point, pseudo level) item, in half L2 requests (see lotd_device.h, mode 3)
static uint32_t level_cost(const nr3d_lotd_meta_t *m, uint32_t q, bool pairlane = false) {
	const nr3d_lotd_level_t &L = m->levels[m->map_levels[q]];
	const uint32_t D = m->n_dims_to_encode, G = m->n_feat_per_pseudo_lvl;
	if (pairlane) return 17;                                      // 4 / 4.25 requests for Dense / Hash alike
	uint32_t req = 1u << D;
	const bool paired = (G == 2 && L.n_feats == 2);
	if (L.type == NR3D_LOD_Dense) req = paired ? req / 2 : req;
	else if (L.type == NR3D_LOD_Hash) req = (paired && (L.size & (L.size - 1u)) == 0u) ? req * 3 / 4 : req;
	else req = 2 * D + 2;                                         // line/plane tables: small, mostly L1/L2 hits
	const uint64_t bytes = (uint64_t)L.size * L.n_feats * 4;
	uint32_t c = 2 * req;
	if (bytes <= 32 * 1024) c = req;                              // L1 resident
	return c + 1;                                                 // + the point read / output writes
}

static Sched make_sched(uint32_t N, const nr3d_lotd_meta_t *m, uint32_t &n_blocks, uint64_t skip = 0,
                        uint32_t pts_per_block = kBlock, bool pairlane = false) {
	Sched s;
	s.skip = skip;
	const uint32_t n_pseudo = m->n_pseudo_levels;
	s.n_chunks = div_up(N, pts_per_block);
	s.n_pseudo = n_pseudo;
	s.mode = sched_mode_default();
	s.n_slots = div_up(n_pseudo, 8);
	auto skipped = [&](uint32_t q) { return q < 64u && ((skip >> q) & 1ull); };
	if (s.mode == 3 && pairlane && sched_exclusive_enabled()) {
		// Two-lane forward (every level costs the same for spread-out points).  The 8 * floor(L / 8) LARGEST tables (ties: the
		// finer level) are "owned": they are paired finest with least fine, and a pair of levels belongs to a pair of XCDs,
		// each of which walks ITS half of the chunks of both levels -- one 4 MiB Hash table at a time in an XCD's L2, loaded by
		// two XCDs instead of eight.  Every other level is split into 8 chunk ranges, one per XCD (small tables: loading one
		// into each L2 costs microseconds).  Every XCD gets the same number of items, so spread-out points balance exactly
		// (2^20 points: 351 -> 331 us against the cost-balanced work line, whose segments end mid-level), and the mix per
		// XCD is even too: samples along rays coalesce in the coarse levels and not in the fine ones (per-level times on the
		// full loop's 6.9 M samples, profiles/r03f_fwd_levels.txt: 58 us for levels 2-7, 77 / 105 / 129 for 8 / 9 / 10, 152
		// for 11-15), so an XCD that owned only coarse levels would run dry early.
		std::vector<uint32_t> live;
		for (uint32_t q = 0; q < n_pseudo; ++q) if (!skipped(q)) live.push_back(q);
		std::vector<uint32_t> by_size(live);
		auto bytes_of = [&](uint32_t q) { const nr3d_lotd_level_t &L = m->levels[m->map_levels[q]]; return (uint64_t)L.size * L.n_feats; };
		std::sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) {
			return bytes_of(a) != bytes_of(b) ? bytes_of(a) > bytes_of(b) : a > b; });
		const size_t n_excl = live.size() / 8 * 8;
		std::sort(by_size.begin(), by_size.begin() + n_excl, [](uint32_t a, uint32_t b) { return a > b; });      // owned: finest first
		uint32_t nseg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		bool ok = !live.empty() && live.size() - n_excl + n_excl / 4 <= (size_t)kSchedSegs;
		auto add = [&](int x, uint32_t q, uint32_t c0, uint32_t c1) {
			if (c1 <= c0) return;
			const uint32_t i = nseg[x]++;
			s.seg_q[x][i] = q; s.seg_begin[x][i] = c0; s.seg_cum[x][i + 1] = s.seg_cum[x][i] + (c1 - c0);
		};
		if (ok) {
			for (int x = 0; x < 8; ++x) s.seg_cum[x][0] = 0;
			const uint32_t half = s.n_chunks / 2;
			for (size_t i = 0; i < n_excl / 2; ++i) {               // level pair i = (i-th finest, i-th least fine) -> XCDs 2p, 2p + 1
				const int p = (int)(i & 3u);
				const uint32_t qa = by_size[i], qb = by_size[n_excl - 1 - i];
				add(2 * p, qa, 0, half); add(2 * p, qb, 0, half);
				add(2 * p + 1, qb, half, s.n_chunks); add(2 * p + 1, qa, half, s.n_chunks);
			}
			for (size_t k = n_excl; k < by_size.size(); ++k)
				for (int x = 0; x < 8; ++x)
					add(x, by_size[k], (uint32_t)((uint64_t)s.n_chunks * x / 8), (uint32_t)((uint64_t)s.n_chunks * (x + 1) / 8));
			uint32_t max_blocks = 0;
			for (int x = 0; x < 8; ++x) {
				for (int i = nseg[x]; i < kSchedSegs; ++i) { s.seg_cum[x][i + 1] = s.seg_cum[x][i]; s.seg_q[x][i] = 0; s.seg_begin[x][i] = 0; }
				max_blocks = s.seg_cum[x][kSchedSegs] > max_blocks ? s.seg_cum[x][kSchedSegs] : max_blocks;
			}
			n_blocks = 8u * max_blocks;
			return s;
		}
	}
	if (s.mode == 3) {
		// work line: level q occupies n_chunks items of cost c_q each; XCD x takes the items that START in
		// [x, x + 1) * total / 8
		uint64_t total = 0;
		for (uint32_t q = 0; q < n_pseudo; ++q) if (!skipped(q)) total += (uint64_t)level_cost(m, q, pairlane) * s.n_chunks;
		uint32_t max_blocks = 0;
		bool ok = total > 0;
		uint64_t pos = 0;                                             // start of level q on the line
		uint32_t nseg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (int x = 0; x < 8; ++x) s.seg_cum[x][0] = 0;
		// Order of the levels on the line: coarsest, finest, second coarsest, second finest, ...  The cost model prices
		// every level alike, which is right for spread-out points; samples along rays coalesce in the coarse levels (several
		// consecutive samples per cell) and not in the fine ones, so with the levels in natural order the XCDs that own the
		// coarse end finish early (full loop, 262 144 rays: 6.94 ms per iteration against 6.52 with each XCD owning one
		// coarse and one fine level, profiles/r03c_full_loop_sched.txt).  Alternating ends gives every XCD a mix.
		std::vector<uint32_t> order;
		{
			std::vector<uint32_t> live;
			for (uint32_t q = 0; q < n_pseudo; ++q) if (!skipped(q)) live.push_back(q);
			for (size_t lo = 0, hi = live.size(); lo < hi;) {
				order.push_back(live[lo++]);
				if (lo < hi) order.push_back(live[--hi]);
			}
		}
		for (size_t oi = 0; oi < order.size() && ok; ++oi) {
			const uint32_t q = order[oi];
			const uint64_t c = level_cost(m, q, pairlane);
			uint32_t ch = 0;
			while (ch < s.n_chunks) {
				const uint64_t start = pos + (uint64_t)ch * c;
				uint32_t x = (uint32_t)((start * 8) / total);
				x = x > 7 ? 7 : x;
				// last chunk of this level that still starts inside XCD x's share
				const uint64_t bound = ((uint64_t)(x + 1) * total + 7) / 8;          // first position owned by x + 1
				uint64_t ch_end = (bound > pos) ? (bound - pos + c - 1) / c : 0;      // chunks with start < bound
				if (x == 7 || ch_end > s.n_chunks) ch_end = s.n_chunks;
				if (ch_end <= ch) ch_end = ch + 1;
				if (nseg[x] >= (uint32_t)kSchedSegs) { ok = false; break; }
				const uint32_t i = nseg[x]++;
				s.seg_q[x][i] = q;
				s.seg_begin[x][i] = ch;
				s.seg_cum[x][i + 1] = s.seg_cum[x][i] + (uint32_t)(ch_end - ch);
				ch = (uint32_t)ch_end;
			}
			pos += c * s.n_chunks;
		}
		if (ok) {
			for (int x = 0; x < 8; ++x) {
				for (int i = nseg[x]; i < kSchedSegs; ++i) { s.seg_cum[x][i + 1] = s.seg_cum[x][i]; s.seg_q[x][i] = 0; s.seg_begin[x][i] = 0; }
				max_blocks = s.seg_cum[x][kSchedSegs] > max_blocks ? s.seg_cum[x][kSchedSegs] : max_blocks;
			}
			n_blocks = 8u * max_blocks;
			return s;
		}
		s.mode = 1;                                                   // too fragmented: plain XCD-affine
	}
	n_blocks = (s.mode == 1) ? 8u * s.n_slots * s.n_chunks : n_pseudo * s.n_chunks;
	return s;
}

static bool pairlane_enabled() { return opt::on(NR3D_OPT_FWD_PAIRLANE); }
static bool lds_stage_enabled() { return opt::on(NR3D_OPT_FWD_LDS_STAGE); }
static uint32_t lds_stage_min_points() { const int64_t v = NR3D_XOPT(LOTD_LDS_MIN_POINTS, 1 << 18); return v < 1 ? 1u : (uint32_t)v; }

static int check_common(const nr3d_lotd_meta_t *m, const void *meta_dev, int x_dtype, int p_dtype, bool half_params_ok = false) {
	NR3D_CHECK(m != nullptr, "LoTD: meta is NULL");
	NR3D_CHECK(meta_dev != nullptr, "LoTD: meta_dev (device copy of the meta) is NULL");
	NR3D_CHECK(x_dtype == NR3D_F32 && (p_dtype == NR3D_F32 || (half_params_ok && p_dtype == NR3D_F16)),
	           "LoTD: kernels compute in f32; convert half inputs/params on the caller side (got x=%d, params=%d)",
	           x_dtype, p_dtype);
	NR3D_CHECK(m->n_dims_to_encode >= 2 && m->n_dims_to_encode <= 4, "LoTD: `n_dims_to_encode` must be 2/3/4");
	const uint32_t G = m->n_feat_per_pseudo_lvl;
	NR3D_CHECK(G == 2 || G == 4 || G == 8, "LoTDEncoding: `n_feat_per_pseudo_lvl` must be one of [2,4,8]");
	return 0;
}

#define DISPATCH_DG(D_, G_, ...)                                                     \
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

// both storage types of one k_fwd instantiation (inside DISPATCH_DG, with a two-kernel `launch` in scope)
#define NR3D_FWD2(D_, DY_, DH_, ONLY_) launch(k_fwd<D_, G, DY_, DH_, ONLY_, float>, k_fwd<D_, G, DY_, DH_, ONLY_, __half>)

#define DISPATCH_D(D_, ...)                                          \
	do {                                                             \
		const uint32_t _d = (D_);                                    \
		if (_d == 2) { constexpr int D = 2; __VA_ARGS__; }           \
		else if (_d == 3) { constexpr int D = 3; __VA_ARGS__; }      \
		else { constexpr int D = 4; __VA_ARGS__; }                   \
	} while (0)

}  // namespace lotd
}  // namespace nr3d

using namespace nr3d;
using namespace nr3d::lotd;

// the two-lanes-per-point forward (+ LDS-staged coarse levels) applies: 3-D Dense/Hash meta, 2-feature pseudo levels,
// unbatched, 32-bit byte offsets inside every level
static bool pairlane_meta_ok(const nr3d_lotd_meta_t *meta) {
	if (!meta->c_hash_only || meta->n_dims_to_encode != 3 || meta->n_feat_per_pseudo_lvl != 2) return false;
	for (uint32_t l = 0; l < meta->n_levels; ++l)
		if ((uint64_t)meta->levels[l].size * meta->levels[l].n_feats * 4 >= (1ull << 32)) return false;
	return true;
}

// which Dense levels the forward serves from LDS for a batch of N points: whole tables (k_fwd_lds) and, from 2^19 points on, tables
// that fit in <= kSlabMax slabs of x-planes (k_fwd_lds_slab); bit q of the result = pseudo level q.  slab[q] = its slab geometry.
static bool slab_geometry(const nr3d_lotd_level_t &L, uint32_t q, SlabLevel &sl) {
	const uint64_t plane_bytes = (uint64_t)L.res[1] * L.res[2] * 8u;
	if (plane_bytes == 0 || L.res[0] < 2) return false;
	const uint64_t fit = kSlabBytes / plane_bytes;
	if (fit < 2) return false;
	sl.q = q; sl.nx = (uint32_t)(fit - 1u);
	const uint32_t cells = (uint32_t)L.res[0] - 1u;
	sl.n_slabs = (cells + sl.nx - 1u) / sl.nx;
	return sl.n_slabs >= 1 && sl.n_slabs <= kSlabMax;
}
// (an experiment that lost, kept selectable: option value 2 -- see the kernel's header)
static bool slab_stage_enabled() { return opt::get(NR3D_OPT_FWD_LDS_STAGE) == 2; }
static uint64_t fwd_lds_levels(const nr3d_lotd_meta_t *meta, uint32_t N, int32_t max_level, uint64_t *slab_mask) {
	uint64_t staged = 0, slabs = 0;
	if (lds_stage_enabled() && N >= lds_stage_min_points() && meta->n_pseudo_levels <= 64)
		for (uint32_t q = 0; q < meta->n_pseudo_levels; ++q) {
			const uint32_t lv = meta->map_levels[q];
			const nr3d_lotd_level_t &L = meta->levels[lv];
			if ((int32_t)lv > max_level || L.type != NR3D_LOD_Dense || L.n_feats != 2) continue;
			if ((uint64_t)L.size * 8 <= kLdsMaxBytes) { staged |= 1ull << q; continue; }
			SlabLevel sl;
			if (slab_stage_enabled() && N >= (1u << 19) && slab_geometry(L, q, sl)) { staged |= 1ull << q; slabs |= 1ull << q; }
		}
	if (slab_mask) *slab_mask = slabs;
	return staged;
}

// PT = float | __half (parameter and y storage type).  Returns 1 when it served the call, 0 when the generic kernels
// have to, < 0 ... never; errors through NR3D_CHECK (positive).  `served` out-param keeps the int status free.
template <typename PT>
static int fwd_fast_path(const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t N, const float *x, const PT *params,
                         int32_t max_level, PT *y, int64_t y_sn, int64_t y_se, float *dy_dx, int64_t d_sn, int64_t d_se,
                         hipStream_t st, bool &served) {
	served = false;
	if (!pairlane_enabled() || !pairlane_meta_ok(meta) || ((uintptr_t)params % (2 * sizeof(PT))) != 0) return 0;
	// the kernels address the outputs as uniform base + 32-bit lane offset: (31 rows + 1 column) of a wave must fit
	auto lane_span_ok = [](int64_t sn, int64_t se, int64_t elt) {
		return sn >= 0 && se >= 0 && sn < (1ll << 40) && se < (1ll << 40) && (63 * sn + se + 3) * elt < (1ll << 32);
	};
	if (!lane_span_ok(y_sn, y_se, (int64_t)sizeof(PT)) || (dy_dx && !lane_span_ok(d_sn, d_se, 4))) return 0;
	served = true;
	uint64_t staged = 0;
	if (lds_stage_enabled() && N >= lds_stage_min_points() && meta->n_pseudo_levels <= 64) {
		static bool attr_set_dev[64] = {};
		int dev_id = 0;
		NR3D_HIP_CHECK(hipGetDevice(&dev_id));
		if (!attr_set_dev[dev_id & 63]) {
			NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_lds<true, PT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsGroupBytes));
			NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_lds<false, PT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsGroupBytes));
			NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_lds<true, PT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsGroupBytes));
			NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_lds<false, PT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsGroupBytes));
			attr_set_dev[dev_id & 63] = true;
		}
		const bool pair_levels = NR3D_XOPT(LOTD_LDS_PAIR, 1) != 0;     // experiments build: 0 = one launch per staged level (A/B)
		LdsLevels grp;
		uint32_t n_grp = 0, grp_bytes = 0;
		auto flush = [&]() {
			if (!n_grp) return;
			prof::Scope ps(NR3D_PROF_LOTD_FWD_LDS, st);
#define NR3D_LDS_LAUNCH(DY, NL) hipLaunchKernelGGL((k_fwd_lds<DY, PT, NL>), dim3(div_up(N, kLdsPts)), dim3(kLdsThreads), grp_bytes, st, md, \
				N, grp, meta->interpolation_type, x, params, y, y_sn, y_se, dy_dx, d_sn, d_se)
			if (n_grp == 1) { if (dy_dx) NR3D_LDS_LAUNCH(true, 1); else NR3D_LDS_LAUNCH(false, 1); }
			else            { if (dy_dx) NR3D_LDS_LAUNCH(true, 2); else NR3D_LDS_LAUNCH(false, 2); }
#undef NR3D_LDS_LAUNCH
			n_grp = 0; grp_bytes = 0;
		};
		uint64_t slab_mask = 0;
		const uint64_t lds_mask = fwd_lds_levels(meta, N, max_level, &slab_mask);
		for (uint32_t q = 0; q < meta->n_pseudo_levels; ++q) {
			if (!((lds_mask >> q) & 1ull) || ((slab_mask >> q) & 1ull)) continue;
			const nr3d_lotd_level_t &L = meta->levels[meta->map_levels[q]];
			staged |= 1ull << q;
			const uint32_t bytes = (L.size * 8u + 15u) & ~15u;     // staged as float pairs whatever the storage type
			if (n_grp == 2 || (n_grp == 1 && (!pair_levels || grp_bytes + bytes > kLdsGroupBytes))) flush();
			grp.q[n_grp] = q; grp.lds_off[n_grp] = grp_bytes / 8u;
			++n_grp; grp_bytes += bytes;
		}
		flush();
		if (slab_mask) {
			static bool sattr_dev[64] = {};
			if (!sattr_dev[dev_id & 63]) {
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_lds_slab<true, PT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSlabBytes));
				NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)k_fwd_lds_slab<false, PT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSlabBytes));
				sattr_dev[dev_id & 63] = true;
			}
			prof::Scope ps(NR3D_PROF_LOTD_FWD_LDS, st);
			for (uint32_t q = 0; q < meta->n_pseudo_levels; ++q) {
				if (!((slab_mask >> q) & 1ull)) continue;
				SlabLevel sl;
				slab_geometry(meta->levels[meta->map_levels[q]], q, sl);
				const size_t lds = (size_t)(sl.nx + 1u) * meta->levels[meta->map_levels[q]].res[1] * meta->levels[meta->map_levels[q]].res[2] * 8u;
				if (dy_dx)
					hipLaunchKernelGGL((k_fwd_lds_slab<true, PT>), dim3(div_up(N, kLdsPts)), dim3(kLdsThreads), lds, st, md, N, sl, meta->interpolation_type,
					                   x, params, y, y_sn, y_se, dy_dx, d_sn, d_se);
				else
					hipLaunchKernelGGL((k_fwd_lds_slab<false, PT>), dim3(div_up(N, kLdsPts)), dim3(kLdsThreads), lds, st, md, N, sl, meta->interpolation_type,
					                   x, params, y, y_sn, y_se, dy_dx, d_sn, d_se);
				staged |= 1ull << q;
			}
		}
	}
	// timing experiments (experiments build only; results wrong by design): FWD_DBG bit 0 no stores, 1 no gathers, 2 no x loads;
	// FWD_ONLY_LEVEL = one pseudo level
	const int64_t dbg = NR3D_XOPT(FWD_DBG, 0), only = NR3D_XOPT(FWD_ONLY_LEVEL, -1);
	(void)dbg;
	if (only >= 0) for (uint32_t q = 0; q < meta->n_pseudo_levels && q < 64u; ++q) if ((int)q != only) staged |= 1ull << q;
	uint32_t n_blocks;
	const Sched s = make_sched(N, meta, n_blocks, staged, (uint32_t)(kPlPts * kPlSub), true);
	if (n_blocks != 0) {
		prof::Scope ps(NR3D_PROF_LOTD_FWD, st);
		if (dy_dx)
			hipLaunchKernelGGL((k_fwd_pairlane<true, PT, kPlSub>), dim3(n_blocks), dim3(kBlock), 0, st, s, md, N, max_level,
			                   meta->interpolation_type, x, params, y, y_sn, y_se, dy_dx, d_sn, d_se NR3D_DBG_ARG(dbg));
		else
			hipLaunchKernelGGL((k_fwd_pairlane<false, PT, kPlSub>), dim3(n_blocks), dim3(kBlock), 0, st, s, md, N, max_level,
			                   meta->interpolation_type, x, params, y, y_sn, y_se, dy_dx, d_sn, d_se NR3D_DBG_ARG(dbg));
	}
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_lotd_half_params_ok(const nr3d_lotd_meta_t *meta, int batched) {
	return (meta && !batched && pairlane_enabled() && pairlane_meta_ok(meta) && pair_applies(meta)) ? 1 : 0;
}

static int fwd_generic(const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t N, const void *x, const void *params,
                       bool p_half, const int64_t *batch_inds, const int64_t *batch_offsets, uint32_t batch_data_size,
                       int32_t max_level, void *y, int64_t y_sn, int64_t y_se, void *dy_dx, int64_t d_sn, int64_t d_se,
                       hipStream_t st);

extern "C" int nr3d_lotd_fwd(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype,
                             int param_dtype, const void *x, const void *params, const int64_t *batch_inds,
                             const int64_t *batch_offsets, uint32_t batch_data_size, int32_t max_level, void *y,
                             int64_t y_sn, int64_t y_se, void *dy_dx, int64_t d_sn, int64_t d_se, void *stream) {
	if (int rc = check_common(meta, meta_dev, x_dtype, param_dtype, true)) return rc;
	if (N == 0) return 0;
	NR3D_CHECK(x && params && y, "LoTD::fwd: NULL tensor pointer");
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	hipStream_t st = (hipStream_t)stream;
	const bool batched = batch_inds || batch_offsets || batch_data_size;
	// half tables (the reference's (float, half, float) combination): y is half as well; the two-lane kernels serve
	// what they can, everything else (and every batched call) goes through the general kernel on the same storage
	const bool p_half = param_dtype == NR3D_F16;
	NR3D_CHECK(!p_half || ((uintptr_t)params % 2) == 0, "LoTD::fwd: misaligned half params");
	if (!batched) {
		bool served = false;
		if (p_half) {
			if (((uintptr_t)params % 4) == 0)
				if (int rc = fwd_fast_path<__half>(meta, md, N, (const float *)x, (const __half *)params, max_level, (__half *)y, y_sn,
				                                   y_se, (float *)dy_dx, d_sn, d_se, st, served))
					return rc;
		} else if (int rc = fwd_fast_path<float>(meta, md, N, (const float *)x, (const float *)params, max_level, (float *)y, y_sn,
		                                         y_se, (float *)dy_dx, d_sn, d_se, st, served))
			return rc;
		if (served) return 0;
	}
	return fwd_generic(meta, md, N, x, params, p_half, batch_inds, batch_offsets, batch_data_size, max_level, y, y_sn, y_se, dy_dx,
	                   d_sn, d_se, st);
}

// the general kernel (every level type, D = 2 / 3 / 4, batched tables): float y and dy/dx from float or half tables
static int fwd_generic(const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t N, const void *x, const void *params,
                       bool p_half, const int64_t *batch_inds, const int64_t *batch_offsets, uint32_t batch_data_size,
                       int32_t max_level, void *y, int64_t y_sn, int64_t y_se, void *dy_dx, int64_t d_sn, int64_t d_se,
                       hipStream_t st) {
	uint32_t n_blocks;
	const Batch ba{batch_inds, batch_offsets, batch_data_size, meta->n_params};
	const uint32_t G = meta->n_feat_per_pseudo_lvl;
	// vector gathers need every corner address G elements aligned: base pointer aligned and no caller-chosen offsets
	const uint32_t elt = p_half ? 2u : 4u;
	const uint32_t vec_ok = (((uintptr_t)params % ((G >= 4 ? 4u : 2u) * elt)) == 0 && batch_offsets == nullptr) ? 1u : 0u;
	const bool dh = meta->c_hash_only != 0;
	prof::Scope ps(NR3D_PROF_LOTD_FWD, st);
	// A meta that mixes Dense / Hash levels with product types: the instantiation that carries every level type needs
	// 114 VGPRs (4 waves per SIMD), the Dense / Hash one 54 (8 waves) -- so the Dense / Hash levels of a mixed meta go
	// through the lean kernel in a launch of their own (same code for those levels, same bits), the rest through the
	// general one.  NR3D_LOTD_FWD_SPLIT=0: one launch.
	// Likewise CP and VM levels (3-D) have instantiations of their own: the product types share one code path otherwise.
	uint64_t grp[4] = {0, 0, 0, 0};                      // pseudo levels served by: 0 general, 1 Dense / Hash, 2 CP, 3 VM
	const bool split = !dh && meta->n_pseudo_levels <= 64u && opt::on(NR3D_OPT_FWD_SPLIT);
	const uint64_t all = meta->n_pseudo_levels >= 64u ? ~0ull : ((1ull << meta->n_pseudo_levels) - 1ull);
	if (split) {
		for (uint32_t q = 0; q < meta->n_pseudo_levels; ++q) {
			const uint32_t t = meta->levels[meta->map_levels[q]].type;
			int k = 0;
			if (t == NR3D_LOD_Dense || t == NR3D_LOD_Hash) k = 1;
			else if (meta->n_dims_to_encode == 3 && t == NR3D_LOD_CP) k = 2;
			else if (meta->n_dims_to_encode == 3 && t == NR3D_LOD_VectorMatrix) k = 3;
			grp[k] |= 1ull << q;
		}
	} else {
		grp[dh ? 1 : 0] = all;
	}
	for (int k = 0; k < 4; ++k) {
		if (!grp[k]) continue;
		const uint64_t skip = (split || meta->n_pseudo_levels <= 64u) ? (all & ~grp[k]) : 0ull;
		const Sched s = make_sched(N, meta, n_blocks, skip);
		if (n_blocks == 0) continue;
		DISPATCH_DG(meta->n_dims_to_encode, G, {
			auto launch = [&](auto kern_f32, auto kern_f16) {
				if (p_half)
					hipLaunchKernelGGL(kern_f16, dim3(n_blocks), dim3(kBlock), 0, st, s, md, N, max_level,
					                   meta->interpolation_type, (const float *)x, (const __half *)params, ba, vec_ok,
					                   (__half *)y, y_sn, y_se, (float *)dy_dx, d_sn, d_se);
				else
					hipLaunchKernelGGL(kern_f32, dim3(n_blocks), dim3(kBlock), 0, st, s, md, N, max_level,
					                   meta->interpolation_type, (const float *)x, (const float *)params, ba, vec_ok,
					                   (float *)y, y_sn, y_se, (float *)dy_dx, d_sn, d_se);
			};
			if (k == 1) { if (dy_dx) NR3D_FWD2(D, true, true, -1); else NR3D_FWD2(D, false, true, -1); }
			else if (k == 0) { if (dy_dx) NR3D_FWD2(D, true, false, -1); else NR3D_FWD2(D, false, false, -1); }
			else if constexpr (D == 3) {
				if (k == 2) { if (dy_dx) NR3D_FWD2(3, true, false, NR3D_LOD_CP); else NR3D_FWD2(3, false, false, NR3D_LOD_CP); }
				else { if (dy_dx) NR3D_FWD2(3, true, false, NR3D_LOD_VectorMatrix); else NR3D_FWD2(3, false, false, NR3D_LOD_VectorMatrix); }
			}
		});
	}
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_lotd_bwd_dx(const nr3d_lotd_meta_t *meta, uint32_t N, int x_dtype, int param_dtype,
                                const void *dL_dy, int64_t g_sn, int64_t g_se, const void *dy_dx, int64_t d_sn,
                                int64_t d_se, void *dL_dx, void *dL_dy_T, void *stream) {
	NR3D_CHECK(meta != nullptr, "LoTD: meta is NULL");
	NR3D_CHECK(x_dtype == NR3D_F32 && (param_dtype == NR3D_F32 || param_dtype == NR3D_F16), "LoTD::bwd_dx: f32 x, f32 / f16 dL_dy");
	if (N == 0) return 0;
	NR3D_CHECK(dL_dy && dy_dx && dL_dx, "LoTDEncoding::bwd: need `dy_dx` to comput `dL_dx`.");
	const uint32_t E = meta->n_encoded_dims;
	const bool g_half = param_dtype == NR3D_F16;          // dL_dy has the params' dtype (lotd_torch_api.cu:430-433)
	const bool row_major = (g_se == 1 && g_sn == (int64_t)E && ((uintptr_t)dL_dy % (g_half ? 8 : 16)) == 0);
	NR3D_CHECK(dL_dy_T == nullptr || row_major, "LoTD::bwd_dx: dL_dy_T needs a contiguous, 16-byte aligned [N, E] dL_dy");
	NR3D_CHECK(!g_half || (row_major && (E & 3u) == 0u), "LoTD::bwd_dx: half dL_dy must be a contiguous [N, E] tensor, E % 4 == 0");
	prof::Scope ps(NR3D_PROF_LOTD_CONTRACT_DX, (hipStream_t)stream);
	DISPATCH_D(meta->n_dims_to_encode, {
		if (row_major && g_half)
			hipLaunchKernelGGL((k_contract_dx_rowmajor<D, __half>), dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, N, E,
			                   (const __half *)dL_dy, (const float *)dy_dx, d_sn, d_se, (float *)dL_dx, (float *)dL_dy_T);
		else if (row_major)
			hipLaunchKernelGGL((k_contract_dx_rowmajor<D, float>), dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, N, E,
			                   (const float *)dL_dy, (const float *)dy_dx, d_sn, d_se, (float *)dL_dx, (float *)dL_dy_T);
		else
			hipLaunchKernelGGL(k_contract_dx<D>, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, N, E,
			                   (const float *)dL_dy, g_sn, g_se, (const float *)dy_dx, d_sn, d_se, (float *)dL_dx);
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}

static int launch_bwd_dparam(bool second, const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N,
                             const void *dL_ddLdx, const void *dL_dy, int64_t g_sn, int64_t g_se, const void *x,
                             const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                             uint32_t batch_data_size, uint32_t n_batches, int32_t max_level, void *dL_dparam,
                             void *workspace, uint64_t workspace_bytes, void *stream, int32_t min_level = 0, bool p_half = false) {
	// p_half: `params` are __half tables (read by the product-type levels only; dL_dparam stays float)
	if (N == 0 || max_level <= -1 || min_level > max_level) return 0;
	NR3D_CHECK(dL_dy && x && params && dL_dparam, "LoTD::bwd: NULL tensor pointer");
	// atomic-free binned path: metas without NPlaneSum/CPfast levels, when the caller supplied the workspace (batched
	// params need n_batches, the number of table sets behind `params`)
	const bool batched = batch_inds || batch_offsets || batch_data_size;
	if (workspace && (!batched || n_batches > 0)) {
		bool handled = false;
		const Batch bb{batch_inds, batch_offsets, batch_data_size, meta->n_params};
		if (int rc = dparam_binned(second, meta, meta_dev, N, (const float *)dL_ddLdx, (const float *)dL_dy, g_sn, g_se,
		                           (const float *)x, (const float *)params, bb, batched ? n_batches : 1u, max_level,
		                           (float *)dL_dparam, workspace, workspace_bytes, (hipStream_t)stream, handled, nullptr,
		                           min_level, false, false, false, p_half))
			return rc;
		if (handled) return 0;
	}
	uint32_t n_blocks;
	const Sched s = make_sched(N, meta, n_blocks);
	const Batch ba{batch_inds, batch_offsets, batch_data_size, meta->n_params};
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	const bool dh = meta->c_hash_only != 0;
	DISPATCH_DG(meta->n_dims_to_encode, meta->n_feat_per_pseudo_lvl, {
		auto launch = [&](auto kern, auto *tab) {
			hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(kBlock), 0, (hipStream_t)stream, s, md, N, min_level, max_level,
			                   meta->interpolation_type, (const float *)dL_ddLdx, (const float *)dL_dy, g_sn, g_se,
			                   (const float *)x, tab, ba, (float *)dL_dparam);
		};
		const float *pf = (const float *)params;
		const __half *ph = (const __half *)params;
		// Dense / Hash levels read no table: the hash-only instantiations serve both storage types
		if (dh || !p_half) {
			if (second) { if (dh) launch(k_bwd_dparam<D, G, true, true>, pf); else launch(k_bwd_dparam<D, G, true, false>, pf); }
			else        { if (dh) launch(k_bwd_dparam<D, G, false, true>, pf); else launch(k_bwd_dparam<D, G, false, false>, pf); }
		} else {
			if (second) launch(k_bwd_dparam<D, G, true, false, __half>, ph); else launch(k_bwd_dparam<D, G, false, false, __half>, ph);
		}
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_lotd_bwd_dparam(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype,
                                    int param_dtype, const void *dL_dy, int64_t g_sn, int64_t g_se, const void *x,
                                    const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                                    uint32_t batch_data_size, uint32_t n_batches, int32_t max_level, void *dL_dparam,
                                    void *workspace, uint64_t workspace_bytes, void *stream) {
	if (int rc = check_common(meta, meta_dev, x_dtype, param_dtype, true)) return rc;
	return launch_bwd_dparam(false, meta, meta_dev, N, nullptr, dL_dy, g_sn, g_se, x, params, batch_inds,
	                         batch_offsets, batch_data_size, n_batches, max_level, dL_dparam, workspace, workspace_bytes,
	                         stream, 0, param_dtype == NR3D_F16);
}

extern "C" int nr3d_lotd_pair_path_ok(const nr3d_lotd_meta_t *meta) { return (meta && pair_applies(meta)) ? 1 : 0; }
extern "C" uint64_t nr3d_lotd_fwd_lds_levels(const nr3d_lotd_meta_t *meta, uint32_t n_points, uint64_t *by_slab) {
	if (by_slab) *by_slab = 0;
	if (!meta || !pairlane_enabled() || !pairlane_meta_ok(meta)) return 0;
	return fwd_lds_levels(meta, n_points, 0x7fffffff, by_slab);
}

extern "C" int nr3d_lotd_pair_direct_levels(const nr3d_lotd_meta_t *meta, uint32_t n_points) {
	return (meta && pair_applies(meta)) ? (int)pair_direct_levels(meta, n_points) : 0;
}

extern "C" int nr3d_lotd_bwd_dparam_typed(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int grad_dtype,
                                          const void *dL_dy, int64_t g_sn, int64_t g_se, const void *x, int32_t max_level,
                                          int out_dtype, int assign, void *dL_dparam, void *workspace, uint64_t workspace_bytes,
                                          void *stream) {
	if (int rc = check_common(meta, meta_dev, NR3D_F32, NR3D_F32)) return rc;
	NR3D_CHECK((grad_dtype == NR3D_F32 || grad_dtype == NR3D_F16) && (out_dtype == NR3D_F32 || out_dtype == NR3D_F16),
	           "LoTD::bwd_dparam_typed: f32 / f16 only");
	if (N == 0 || max_level <= -1) {
		if (assign && dL_dparam)
			NR3D_HIP_CHECK(hipMemsetAsync(dL_dparam, 0, (size_t)meta->n_params * (out_dtype == NR3D_F16 ? 2 : 4), (hipStream_t)stream));
		return 0;
	}
	NR3D_CHECK(dL_dy && x && dL_dparam && workspace, "LoTD::bwd: NULL tensor pointer");
	bool handled = false;
	const Batch bb{nullptr, nullptr, 0u, meta->n_params};
	if (int rc = dparam_binned(false, meta, meta_dev, N, nullptr, (const float *)dL_dy, g_sn, g_se, (const float *)x, nullptr, bb,
	                           1u, max_level, (float *)dL_dparam, workspace, workspace_bytes, (hipStream_t)stream, handled, nullptr,
	                           0, grad_dtype == NR3D_F16, out_dtype == NR3D_F16, assign != 0))
		return rc;
	NR3D_CHECK(handled, "LoTD::bwd_dparam_typed: the pair-record path does not apply to this meta / workspace");
	return 0;
}

extern "C" int nr3d_lotd_bwd_dparam_levels(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype,
                                           int param_dtype, const void *dL_dy, int64_t g_sn, int64_t g_se, const void *x,
                                           const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                                           uint32_t batch_data_size, uint32_t n_batches, int32_t min_level,
                                           int32_t max_level, void *dL_dparam, void *workspace, uint64_t workspace_bytes,
                                           void *stream) {
	if (int rc = check_common(meta, meta_dev, x_dtype, param_dtype, true)) return rc;
	NR3D_CHECK(min_level >= 0, "LoTD::bwd: min_level must be >= 0");
	return launch_bwd_dparam(false, meta, meta_dev, N, nullptr, dL_dy, g_sn, g_se, x, params, batch_inds,
	                         batch_offsets, batch_data_size, n_batches, max_level, dL_dparam, workspace, workspace_bytes,
	                         stream, min_level, param_dtype == NR3D_F16);
}

extern "C" void nr3d_lotd_set_dparam_chunk_log2(int log2_points) { set_dparam_chunk_log2(log2_points); }

extern "C" uint64_t nr3d_lotd_dparam_workspace_bytes(const nr3d_lotd_meta_t *meta, uint32_t n_points, uint32_t n_batches) {
	return dparam_workspace_bytes(meta, n_points, n_batches);
}

extern "C" int nr3d_lotd_bwd_bwd_dparam(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype,
                                        int param_dtype, const void *dL_ddLdx, const void *dL_dy, int64_t g_sn,
                                        int64_t g_se, const void *x, const void *params, const int64_t *batch_inds,
                                        const int64_t *batch_offsets, uint32_t batch_data_size, uint32_t n_batches,
                                        int32_t max_level, void *dL_dparam, void *workspace, uint64_t workspace_bytes,
                                        void *stream) {
	if (int rc = check_common(meta, meta_dev, x_dtype, param_dtype, true)) return rc;
	NR3D_CHECK(N == 0 || dL_ddLdx != nullptr, "LoTD::bwd_bwd_input: dL_ddLdx is NULL");
	return launch_bwd_dparam(true, meta, meta_dev, N, dL_ddLdx, dL_dy, g_sn, g_se, x, params, batch_inds,
	                         batch_offsets, batch_data_size, n_batches, max_level, dL_dparam, workspace, workspace_bytes,
	                         stream, 0, param_dtype == NR3D_F16);
}

extern "C" int nr3d_lotd_bwd_bwd_ddLdy(const nr3d_lotd_meta_t *meta, uint32_t N, int x_dtype, int param_dtype,
                                       const void *dL_ddLdx, const void *dy_dx, int64_t d_sn, int64_t d_se,
                                       void *dL_ddLdy, int64_t o_sn, int64_t o_se, void *stream) {
	NR3D_CHECK(meta != nullptr, "LoTD: meta is NULL");
	NR3D_CHECK(x_dtype == NR3D_F32 && param_dtype == NR3D_F32, "LoTD::bwd_bwd_ddLdy: f32 only");
	if (N == 0) return 0;
	NR3D_CHECK(dL_ddLdx && dy_dx && dL_ddLdy, "LoTDEncoding::bwd_bwd_input: need `dy_dx` to compute `dL_d(dLdy)`.");
	DISPATCH_D(meta->n_dims_to_encode, {
		hipLaunchKernelGGL(k_contract_ddLdy<D>, dim3(div_up(N, kBlock), meta->n_encoded_dims), dim3(kBlock), 0,
		                   (hipStream_t)stream, N, meta->n_encoded_dims, (const float *)dL_ddLdx,
		                   (const float *)dy_dx, d_sn, d_se, (float *)dL_ddLdy, o_sn, o_se);
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}

template <typename PT>
static int launch_bwd_bwd_dx_t(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype, int param_dtype,
                             const void *dL_ddLdx, const void *dL_dy, int64_t g_sn, int64_t g_se, const void *x,
                             const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                             uint32_t batch_data_size, int32_t max_level, void *dL_dx, void *workspace, uint64_t workspace_bytes,
                             void *stream) {
	if (int rc = check_common(meta, meta_dev, x_dtype, param_dtype, true)) return rc;
	if (N == 0) return 0;
	NR3D_CHECK(dL_ddLdx && dL_dy && x && params && dL_dx, "LoTD::bwd_bwd_dx: NULL tensor pointer");
	const Batch ba{batch_inds, batch_offsets, batch_data_size, meta->n_params};
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	// hash-only metas (every level Dense or Hash): instantiations without the product types' code
	const bool dh = meta->c_hash_only != 0;
	const bool vec_ok = (((uintptr_t)params % (2 * sizeof(PT))) == 0 && batch_offsets == nullptr);
	const uint64_t need = nr3d_lotd_bwd_bwd_dx_workspace_bytes(meta, N);
	// NR3D_OPT_HVP_LEVELS = 0: the lane-serial kernel even with a workspace
	if (workspace && need && workspace_bytes >= need && opt::on(NR3D_OPT_HVP_LEVELS)) {
		uint32_t n_blocks;
		const bool batched = batch_inds || batch_offsets || batch_data_size;
		const int64_t hvp_dbg = NR3D_XOPT(HVP_DBG, 0);                 // timing experiments only (experiments build; results wrong by design)
		(void)hvp_dbg;
		const bool pl = !batched && pairlane_enabled() && pairlane_meta_ok(meta) && ((uintptr_t)params % (2 * sizeof(PT))) == 0 &&
		                opt::on(NR3D_OPT_HVP_PAIRLANE);
		const Sched s = pl ? make_sched(N, meta, n_blocks, 0, (uint32_t)(kPlPts * kHvpSub), true) : make_sched(N, meta, n_blocks);
		DISPATCH_DG(meta->n_dims_to_encode, meta->n_feat_per_pseudo_lvl, {
			auto launch = [&](auto kern) {
				hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(kBlock), 0, (hipStream_t)stream, s, md, N, max_level,
				                   meta->interpolation_type, (const float *)dL_ddLdx, (const float *)dL_dy, g_sn, g_se,
				                   (const float *)x, (const PT *)params, ba, vec_ok, (float *)workspace);
			};
			if (pl) {
				// workspace: partial [P][N][3] | g_pairs [P][N][2]; a feature-major dL_dy (g_sn == 1) is read in place
				const float *g_pairs = (const float *)dL_dy;
				const int64_t g_fm = (g_sn == 1 && g_se >= (int64_t)N) ? g_se : 0;
				if (!g_fm) {
					float *gp = (float *)workspace + (size_t)N * meta->n_pseudo_levels * 3u;
					hipLaunchKernelGGL(k_hvp_pairs, dim3(div_up(N, kGpPts), div_up(meta->n_pseudo_levels, kGpLv)), dim3(kBlock), 0,
					                   (hipStream_t)stream, N, meta->n_pseudo_levels, (const float *)dL_dy, g_sn, g_se, gp);
					g_pairs = gp;
				}
				hipLaunchKernelGGL((k_bwd_bwd_dx_pl<kHvpSub, PT>), dim3(n_blocks), dim3(kBlock), 0, (hipStream_t)stream, s, md, N, max_level,
				                   meta->interpolation_type, (const float *)dL_ddLdx, g_pairs, g_fm,
				                   (const float *)x, (const PT *)params, (float *)workspace NR3D_DBG_ARG(hvp_dbg));
			}
			else if (dh) launch(k_bwd_bwd_dx_lv<D, G, true, PT>); else launch(k_bwd_bwd_dx_lv<D, G, false, PT>);
			hipLaunchKernelGGL(k_sum_levels<D>, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, N,
			                   meta->n_pseudo_levels, (const float *)workspace, (float *)dL_dx);
		});
		NR3D_LAUNCH_CHECK();
		return 0;
	}
	// level groups by type (3-D metas with product types; NR3D_LOTD_HVP_SPLIT=0: one launch): 0 Dense / Hash, 2 VM, 3 VecZMatXoY;
	// the other types have no d(dL/dx)/dx term (hvp_type) and get no launch
	uint32_t grp[4] = {0, 0, 0, 0};
	const bool by_type = !dh && meta->n_dims_to_encode == 3 && meta->n_levels <= 32u && opt::on(NR3D_OPT_HVP_SPLIT);
	for (uint32_t l = 0; l < meta->n_levels && l < 32u; ++l) {
		const uint32_t t = meta->levels[l].type;
		const int k = (t == NR3D_LOD_Dense || t == NR3D_LOD_Hash) ? 0 : t == NR3D_LOD_VectorMatrix ? 2 : t == NR3D_LOD_VecZMatXoY ? 3 : -1;
		if (k >= 0) grp[k] |= 1u << l;
	}
	if (!by_type) grp[3] = 0xFFFFFFFFu;
	DISPATCH_DG(meta->n_dims_to_encode, meta->n_feat_per_pseudo_lvl, {
		uint32_t launched = 0;
		auto launch = [&](auto kern, uint32_t mask) {
			hipLaunchKernelGGL(kern, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, md, N,
			                   meta->n_pseudo_levels, max_level, meta->interpolation_type, (const float *)dL_ddLdx,
			                   (const float *)dL_dy, g_sn, g_se, (const float *)x, (const PT *)params, ba, vec_ok, (float *)dL_dx,
			                   mask, launched);
			launched = 1u;
		};
		if (dh) launch(k_bwd_bwd_dx<D, G, true, PT>, 0xFFFFFFFFu);
		else if (!by_type) launch(k_bwd_bwd_dx<D, G, false, PT>, 0xFFFFFFFFu);
		else if constexpr (D == 3) {
			// in the order of each group's first level (levels grouped by type keep the one-launch association)
			uint32_t left[4] = {grp[0], grp[1], grp[2], grp[3]};
			for (int n_done = 0; n_done < 4; ++n_done) {
				int best = -1;
				for (int k = 0; k < 4; ++k)
					if (left[k] && (best < 0 || __builtin_ctz(left[k]) < __builtin_ctz(left[best]))) best = k;
				if (best < 0) break;
				const uint32_t mask = left[best];
				left[best] = 0;
				if (best == 0) launch(k_bwd_bwd_dx<3, G, false, PT, kOnlyDenseHash>, mask);
				else if (best == 2) launch(k_bwd_bwd_dx<3, G, false, PT, NR3D_LOD_VectorMatrix>, mask);
				else launch(k_bwd_bwd_dx<3, G, false, PT>, mask);
			}
			if (!launched) launch(k_bwd_bwd_dx<3, G, false, PT>, 0u);       // no level at all: dL_dx = 0
		}
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}

// float or half tables (read as float: HalfTab); dL_ddLdx, dL_dy, x and the result are float
static int launch_bwd_bwd_dx(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype, int param_dtype,
                             const void *dL_ddLdx, const void *dL_dy, int64_t g_sn, int64_t g_se, const void *x,
                             const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                             uint32_t batch_data_size, int32_t max_level, void *dL_dx, void *workspace, uint64_t workspace_bytes,
                             void *stream) {
	if (param_dtype == NR3D_F16)
		return launch_bwd_bwd_dx_t<__half>(meta, meta_dev, N, x_dtype, param_dtype, dL_ddLdx, dL_dy, g_sn, g_se, x, params, batch_inds,
		                                   batch_offsets, batch_data_size, max_level, dL_dx, workspace, workspace_bytes, stream);
	return launch_bwd_bwd_dx_t<float>(meta, meta_dev, N, x_dtype, param_dtype, dL_ddLdx, dL_dy, g_sn, g_se, x, params, batch_inds,
	                                  batch_offsets, batch_data_size, max_level, dL_dx, workspace, workspace_bytes, stream);
}

extern "C" uint64_t nr3d_lotd_bwd_bwd_dx_workspace_bytes(const nr3d_lotd_meta_t *meta, uint32_t n_points) {
	// Dense / Hash metas only: with product types in the instantiation (114+ VGPRs) one lane per level is a loss --
	// configs[3]: 7.8 ms against 4.0 for the lane-serial kernel
	if (!meta || !meta->c_hash_only || meta->n_pseudo_levels > 64u) return 0;
	// per (point, pseudo level): the D partial sums; + the pair-lane kernel's copy of dL_dy by level (2 floats)
	const uint64_t per = meta->n_dims_to_encode + (pairlane_meta_ok(meta) ? 2u : 0u);
	return (uint64_t)n_points * meta->n_pseudo_levels * per * sizeof(float);
}

extern "C" int nr3d_lotd_bwd_bwd_dx(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype,
                                    int param_dtype, const void *dL_ddLdx, const void *dL_dy, int64_t g_sn, int64_t g_se,
                                    const void *x, const void *params, const int64_t *batch_inds,
                                    const int64_t *batch_offsets, uint32_t batch_data_size, int32_t max_level,
                                    void *dL_dx, void *stream) {
	return launch_bwd_bwd_dx(meta, meta_dev, N, x_dtype, param_dtype, dL_ddLdx, dL_dy, g_sn, g_se, x, params, batch_inds,
	                         batch_offsets, batch_data_size, max_level, dL_dx, nullptr, 0, stream);
}

extern "C" int nr3d_lotd_bwd_bwd_dx_ws(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype,
                                       int param_dtype, const void *dL_ddLdx, const void *dL_dy, int64_t g_sn, int64_t g_se,
                                       const void *x, const void *params, const int64_t *batch_inds,
                                       const int64_t *batch_offsets, uint32_t batch_data_size, int32_t max_level,
                                       void *dL_dx, void *workspace, uint64_t workspace_bytes, void *stream) {
	return launch_bwd_bwd_dx(meta, meta_dev, N, x_dtype, param_dtype, dL_ddLdx, dL_dy, g_sn, g_se, x, params, batch_inds,
	                         batch_offsets, batch_data_size, max_level, dL_dx, workspace, workspace_bytes, stream);
}

extern "C" int nr3d_lotd_grid_index(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t N, int x_dtype,
                                    const void *x, const int64_t *batch_inds, const int64_t *batch_offsets,
                                    uint32_t batch_data_size, int32_t max_level, int64_t *grid_inds, void *stream) {
	if (int rc = check_common(meta, meta_dev, x_dtype, NR3D_F32)) return rc;
	for (uint32_t l = 0; l < meta->n_levels; ++l)
		NR3D_CHECK(meta->levels[l].type == NR3D_LOD_Dense || meta->levels[l].type == NR3D_LOD_Hash,
		           "LoTDEncoding::get_grid_index: Only support Dense/Hash type.");
	if (N == 0 || max_level <= -1) return 0;
	NR3D_CHECK(x && grid_inds, "LoTD::get_grid_index: NULL tensor pointer");
	uint32_t n_blocks;
	const Sched s = make_sched(N, meta, n_blocks);
	const Batch ba{batch_inds, batch_offsets, batch_data_size, meta->n_params};
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	DISPATCH_DG(meta->n_dims_to_encode, meta->n_feat_per_pseudo_lvl, {
		hipLaunchKernelGGL((k_grid_index<D, G>), dim3(n_blocks), dim3(kBlock), 0, (hipStream_t)stream, s, md, N,
		                   meta->n_encoded_dims, max_level, (const float *)x, ba, grid_inds);
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}
