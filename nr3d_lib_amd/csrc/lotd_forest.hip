// nr3d_lib_amd/csrc/lotd_forest.hip -- LoTD over a forest of blocks (gfx950), C-ABI entry points
// nr3d_forest_identify / nr3d_lotd_forest_{fwd, bwd_dparam, bwd_bwd_dx}.
//
// Replaces the forest overloads of nr3d_lib.bindings._lotd (csrc/lotd/src/lotd.cpp:44-60; kernels
// csrc/lotd/include/lotd/lotd_forest.h; octree lookup csrc/forest/forest.h:25-97).
//
// A forest level of resolution R is an (R+2)^3 N-linear grid per block whose outer shell aliases the facing layer
// of the six/eighteen/twenty-six neighbouring blocks: corner index 0 along a dim is the left neighbour's R-1, R+1 the
// right neighbour's 0, 1..R the block's own 0..R-1 (the cell locator uses scale = R).  Only the boundary cells of a
// block ever touch a neighbour, so the octree walk (`identify`, one dependent byte + one int per octree level) runs for
// a 6/R fraction of the corners; everything else is the plain per-corner arithmetic of lotd_device.h.
// One lane = one (point, pseudo level); parameter gradients are fp32 hardware atomics or, for
// Dense/Hash metas with a workspace, the atomic-free binned path of lotd_bin.hip (k_bin_forest: blocks = batch entries).
#include "common.h"
#include "lotd_device.h"

namespace nr3d {
namespace lotd {

__device__ __forceinline__ bool forest_type(uint32_t t) {
	return t == NR3D_LOD_Dense || t == NR3D_LOD_VectorMatrix || t == NR3D_LOD_NPlaneMul || t == NR3D_LOD_CP || t == NR3D_LOD_Hash;
}

__global__ __launch_bounds__(256) void k_forest_identify(ForestDev fo, uint64_t n, const int16_t *__restrict__ ks,
                                                         int32_t *__restrict__ out) {
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int32_t pidx = identify(fo, ks[3 * i], ks[3 * i + 1], ks[3 * i + 2]);
	out[i] = pidx < 0 ? -1 : pidx - (int32_t)fo.level_poffset;
}

// =============================================================================================
// forward (+ dy/dx): kernel_lod_forest, lotd_forest.h:158-333; forest_fwd_n_linear :33-156
// =============================================================================================
// PT: storage type of the tables (float, or __half read as float -- lotd_device.h, HalfTab); y and dy/dx are float
template <int G, bool DYDX, typename PT>
__global__ __launch_bounds__(kBlock) void k_forest_fwd(const nr3d_lotd_meta_t *__restrict__ md, ForestDev fo, uint32_t N,
                                                       int32_t max_level, uint32_t smooth,
                                                       const float *__restrict__ x, const PT *__restrict__ params,
                                                       Batch ba, uint32_t pair_ok, float *__restrict__ y, int64_t y_sn, int64_t y_se,
                                                       float *__restrict__ dydx, int64_t d_sn, int64_t d_se) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= N) return;
	const uint32_t q = blockIdx.y;
	const uint32_t level = meta_level_of(md, q);
	const uint32_t foff0 = meta_cnt_of(md, q) * G, out0 = q * G;
	float out_y[G], out_g[G][3];
#pragma unroll
	for (int f = 0; f < G; ++f) {
		out_y[f] = 0.0f;
#pragma unroll
		for (int d = 0; d < 3; ++d) out_g[f][d] = 0.0f;
	}
	Block b;
	if ((int32_t)level <= max_level && load_block(fo, ba, i, b)) {
		const Lvl L = load_level(md, level);
		if (forest_type(L.type)) {
			float xp[3];
#pragma unroll
			for (int d = 0; d < 3; ++d) xp[d] = x[(size_t)i * 3 + d];
			Cell<3> c;
			locate_forest(xp, L, smooth != 0, c);
			float v[8][G];
			// a cell whose 8 corners all lie inside the point's own block (all but a 6/R fraction) is the plain level
			// shifted by one node: same paired 16-byte gathers as the single-block kernel
			// ... and for the product types (round 4) the single-block FACTORED gather: every distinct table entry of the cell
			// once (VM 18, CP 6, NPlaneMul 12) instead of once per corner and factor (48 / 24 / 24) -- the reference's own forest
			// workload (Dense x2 + VM x7, unit_test_forest.py) went 7.4 -> 3.x ms
			bool fast = false;
			const bool dh = L.type == NR3D_LOD_Dense || L.type == NR3D_LOD_Hash;
			if constexpr (G == 2) {
				fast = dh ? (pair_ok && L.F == 2 && L.size >= 2) : true;
#pragma unroll
				for (int d = 0; d < 3; ++d) fast = fast && c.g[d] >= 1u && c.g[d] + 1u <= L.res[d];
			}
			if (fast) {
				if constexpr (G == 2) {
					Cell<3> cs = c;
#pragma unroll
					for (int d = 0; d < 3; ++d) cs.g[d] -= 1u;
					const auto grid = make_tab(params + (b.offset + L.off));
					if (L.type == NR3D_LOD_Dense) gather_pairs<3, true>(L, cs, grid, v);
					else if (L.type == NR3D_LOD_Hash) gather_pairs<3, false>(L, cs, grid, v);
					else corner_values_pair<3, -1, 2>(L, grid, foff0, pair_ok != 0u && (L.F & 1u) == 0u, cs, v);
				}
			} else {
#pragma unroll
				for (uint32_t k = 0; k < 8; ++k) {
					uint32_t p[3], pl[3], off;
					corner_pos<3>(c, k, p);
					if (resolve(fo, ba, L, b, p, pl, off)) {
						corner_value<3, G>(L, make_tab(params + (off + L.off)), foff0, pl, v[k]);
					} else {
#pragma unroll
						for (int f = 0; f < G; ++f) v[k][f] = 0.0f;
					}
				}
			}
#pragma unroll
			for (uint32_t k = 0; k < 8; ++k) {
				const float w = corner_weight<3>(c, k);
#pragma unroll
				for (int f = 0; f < G; ++f) out_y[f] = __fmaf_rn(w, v[k][f], out_y[f]);
			}
			if (DYDX) {
#pragma unroll
				for (int gd = 0; gd < 3; ++gd)
#pragma unroll
					for (uint32_t k = 0; k < 8; ++k) {
						if ((k >> gd) & 1u) continue;
						const float w = face_weight<3>(c, k, gd, c.sc[gd] * c.dw[gd]);
#pragma unroll
						for (int f = 0; f < G; ++f) out_g[f][gd] = __fmaf_rn(w, v[k | (1u << gd)][f] - v[k][f], out_g[f][gd]);
					}
			}
		}
	}
#pragma unroll
	for (int f = 0; f < G; ++f) __builtin_nontemporal_store(out_y[f], &y[(int64_t)i * y_sn + (int64_t)(out0 + f) * y_se]);
	if (DYDX) {
#pragma unroll
		for (int f = 0; f < G; ++f) {
			float *dst = dydx + (int64_t)i * d_sn + (int64_t)(out0 + f) * d_se;
#pragma unroll
			for (int d = 0; d < 3; ++d) __builtin_nontemporal_store(out_g[f][d], &dst[d]);
		}
	}
}

// =============================================================================================
// dL/dparam (SECOND == false; kernel_lod_forest_backward_grid :414-542) and d(dL/dx)/dparam
// (SECOND == true; kernel_lod_forest_backward_input_backward_grid :636-773).  As in lotd.hip, the D signed
// face contributions that land on one corner are summed before the scatter (the scatter is linear in the weight).
// =============================================================================================
template <int G, bool SECOND, typename PT>
__global__ __launch_bounds__(kBlock) void k_forest_bwd_dparam(const nr3d_lotd_meta_t *__restrict__ md, ForestDev fo,
                                                              uint32_t N, uint32_t E, int32_t max_level, uint32_t smooth,
                                                              const float *__restrict__ dL_ddLdx,
                                                              const float *__restrict__ dL_dy, const float *__restrict__ x,
                                                              const PT *__restrict__ params, Batch ba,
                                                              float *__restrict__ dparam) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= N) return;
	const uint32_t q = blockIdx.y;
	const uint32_t level = meta_level_of(md, q);
	if ((int32_t)level > max_level) return;
	Block b;
	if (!load_block(fo, ba, i, b)) return;
	const Lvl L = load_level(md, level);
	if (!forest_type(L.type)) return;
	const uint32_t foff0 = meta_cnt_of(md, q) * G, out0 = q * G;
	float xp[3], a[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) xp[d] = x[(size_t)i * 3 + d];
	Cell<3> c;
	locate_forest(xp, L, smooth != 0, c);
#pragma unroll
	for (int d = 0; d < 3; ++d) a[d] = SECOND ? c.sc[d] * dL_ddLdx[(size_t)i * 3 + d] * c.dw[d] : 0.0f;
	float grad[G];
#pragma unroll
	for (int f = 0; f < G; ++f) grad[f] = dL_dy[(size_t)i * E + out0 + f];
#pragma unroll 1
	for (uint32_t k = 0; k < 8; ++k) {
		float w;
		if (!SECOND) {
			w = corner_weight<3>(c, k);
		} else {
			w = 0.0f;
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				const float t = face_weight<3>(c, k, d, a[d]);
				w += ((k >> d) & 1u) ? t : -t;
			}
		}
		uint32_t p[3], pl[3], off;
		corner_pos<3>(c, k, p);
		if (!resolve(fo, ba, L, b, p, pl, off)) continue;
		corner_scatter<3, G>(L, make_tab(params + (off + L.off)), dparam + (off + L.off), foff0, pl, grad, w);
	}
}

// =============================================================================================
// d(dL/dx)/dx (kernel_lod_forest_backward_input_backward_input :929-1065): Dense / VectorMatrix / Hash.
// One lane owns all pseudo levels of its point (no atomics on dL_dx); arithmetic as k_bwd_bwd_dx in lotd.hip.
// =============================================================================================
template <int G, typename PT>
__global__ __launch_bounds__(kBlock) void k_forest_bwd_bwd_dx(const nr3d_lotd_meta_t *__restrict__ md, ForestDev fo,
                                                              uint32_t N, uint32_t E, uint32_t n_pseudo, int32_t max_level,
                                                              uint32_t smooth, const float *__restrict__ dL_ddLdx,
                                                              const float *__restrict__ dL_dy, const float *__restrict__ x,
                                                              const PT *__restrict__ params, Batch ba,
                                                              float *__restrict__ dL_dx) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= N) return;
	float acc[3] = {0.0f, 0.0f, 0.0f};
	Block b;
	if (load_block(fo, ba, i, b)) {
		float xp[3], vin[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) { xp[d] = x[(size_t)i * 3 + d]; vin[d] = dL_ddLdx[(size_t)i * 3 + d]; }
#pragma unroll 1
		for (uint32_t q = 0; q < n_pseudo; ++q) {
			const uint32_t level = meta_level_of(md, q);
			if ((int32_t)level > max_level) continue;
			const Lvl L = load_level(md, level);
			if (!(L.type == NR3D_LOD_Dense || L.type == NR3D_LOD_Hash || L.type == NR3D_LOD_VectorMatrix)) continue;
			Cell<3> c;
			locate_forest(xp, L, smooth != 0, c);
#pragma unroll 1
			for (int f0 = 0; f0 < G; f0 += 2) {
				const uint32_t foff = meta_cnt_of(md, q) * G + f0;
				const float grad[2] = {dL_dy[(size_t)i * E + q * G + f0], dL_dy[(size_t)i * E + q * G + f0 + 1]};
				float sdot[8];
#pragma unroll
				for (uint32_t k = 0; k < 8; ++k) {
					uint32_t p[3], pl[3], off;
					corner_pos<3>(c, k, p);
					sdot[k] = resolve(fo, ba, L, b, p, pl, off) ? corner_dot<3, 2>(L, make_tab(params + (off + L.off)), foff, pl, grad, 1.0f) : 0.0f;
				}
#pragma unroll
				for (int d = 0; d < 3; ++d) {
					float o = 0.0f;
#pragma unroll
					for (int e = 0; e < 3; ++e) {
						if (e == d && !smooth) continue;
						const float seed = (e == d) ? (c.sc[d] * vin[d]) * (c.sc[d] * c.ddw[d])
						                            : (c.sc[e] * vin[e] * c.dw[e]) * (c.dw[d] * c.sc[d]);
#pragma unroll
						for (uint32_t k = 0; k < 8; ++k) {
							float w = seed;
#pragma unroll
							for (int m = 0; m < 3; ++m) {
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
		}
	}
#pragma unroll
	for (int d = 0; d < 3; ++d) dL_dx[(size_t)i * 3 + d] = acc[d];
}

static int check_forest(const nr3d_lotd_meta_t *m, const void *meta_dev, const nr3d_forest_meta_t *fo) {
	NR3D_CHECK(m && meta_dev, "LoTD forest: meta / meta_dev is NULL");
	NR3D_CHECK(fo && fo->exsum && fo->block_ks && (fo->octree || fo->level == 0),
	           "LoTD forest: forest meta or one of its arrays is NULL");   // a single-block forest has no octree bytes
	NR3D_CHECK(m->n_dims_to_encode == 3, "LoTDEncoding::fwd: lotd-forest only supports `n_dims_to_encode`==3");
	const uint32_t G = m->n_feat_per_pseudo_lvl;
	NR3D_CHECK(G == 2 || G == 4 || G == 8, "LoTDEncoding: `n_feat_per_pseudo_lvl` must be one of [2,4,8]");
	for (uint32_t l = 0; l < m->n_levels; ++l) {
		const uint32_t t = m->levels[l].type;
		NR3D_CHECK(t == NR3D_LOD_Dense || t == NR3D_LOD_VectorMatrix || t == NR3D_LOD_NPlaneMul || t == NR3D_LOD_CP ||
		           t == NR3D_LOD_Hash, "LoTD forest: level %u has type %u; forest levels are Dense / VectorMatrix / NPlaneMul / CP / Hash", l, t);
	}
	NR3D_CHECK((uint64_t)m->n_params * fo->n_trees <= 0xFFFFFFFFull, "LoTD forest: n_trees * n_params exceeds 32-bit parameter offsets");
	return 0;
}

static ForestDev dev_of(const nr3d_forest_meta_t *fo) {
	return ForestDev{fo->octree, fo->exsum, fo->block_ks, fo->level, fo->level_poffset, fo->continuity_enabled ? 1u : 0u};
}

#define DISPATCH_G(G_, ...)                                   \
	do {                                                      \
		const uint32_t _g = (G_);                             \
		if (_g == 2) { constexpr int G = 2; __VA_ARGS__; }    \
		else if (_g == 4) { constexpr int G = 4; __VA_ARGS__; } \
		else { constexpr int G = 8; __VA_ARGS__; }            \
	} while (0)

}  // namespace lotd
}  // namespace nr3d

using namespace nr3d;
using namespace nr3d::lotd;

extern "C" int nr3d_forest_identify(const nr3d_forest_meta_t *forest, uint64_t n, const int16_t *ks, int32_t *block_inds,
                                    void *stream) {
	NR3D_CHECK(forest && forest->exsum && (forest->octree || forest->level == 0), "forest_identify: forest meta or one of its arrays is NULL");
	if (n == 0) return 0;
	NR3D_CHECK(ks && block_inds, "forest_identify: NULL tensor pointer");
	hipLaunchKernelGGL(k_forest_identify, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, dev_of(forest), n, ks, block_inds);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_lotd_forest_fwd(const nr3d_lotd_meta_t *meta, const void *meta_dev, const nr3d_forest_meta_t *forest,
                                    uint32_t N, const float *x, const void *params, int param_dtype, const int64_t *block_inds,
                                    const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level, float *y,
                                    int64_t y_sn, int64_t y_se, float *dy_dx, int64_t d_sn, int64_t d_se, void *stream) {
	if (int rc = check_forest(meta, meta_dev, forest)) return rc;
	NR3D_CHECK(param_dtype == NR3D_F32 || param_dtype == NR3D_F16, "LoTD forest: params must be f32 or f16");
	if (N == 0) return 0;
	NR3D_CHECK(x && params && y, "LoTD forest::fwd: NULL tensor pointer");
	const bool p_half = param_dtype == NR3D_F16;
	const Batch ba{block_inds, block_offsets, batch_data_size, meta->n_params};
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	const dim3 grid(div_up(N, kBlock), meta->n_pseudo_levels);
	// paired gathers: every F == 2 entry 8-byte aligned -> even block and level offsets, no caller-chosen offsets
	const uint32_t pair_ok = ((uintptr_t)params % (p_half ? 4 : 8) == 0 && block_offsets == nullptr && meta->n_params % 2 == 0) ? 1u : 0u;
	DISPATCH_G(meta->n_feat_per_pseudo_lvl, {
		auto launch = [&](auto kern, auto *tab) {
			hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, (hipStream_t)stream, md, dev_of(forest), N, max_level,
			                   meta->interpolation_type, x, tab, ba, pair_ok, y, y_sn, y_se, dy_dx, d_sn, d_se);
		};
		if (p_half) { if (dy_dx) launch(k_forest_fwd<G, true, __half>, (const __half *)params); else launch(k_forest_fwd<G, false, __half>, (const __half *)params); }
		else        { if (dy_dx) launch(k_forest_fwd<G, true, float>, (const float *)params); else launch(k_forest_fwd<G, false, float>, (const float *)params); }
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" uint64_t nr3d_lotd_forest_dparam_workspace_bytes(const nr3d_lotd_meta_t *meta, uint32_t n_points, uint32_t n_trees) {
	return dparam_workspace_bytes(meta, n_points, n_trees, true);
}

extern "C" int nr3d_lotd_forest_bwd_dparam(const nr3d_lotd_meta_t *meta, const void *meta_dev, const nr3d_forest_meta_t *forest,
                                           uint32_t N, const float *dL_ddLdx, const float *dL_dy, const float *x,
                                           const void *params, int param_dtype, const int64_t *block_inds,
                                           const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level,
                                           float *dL_dparam, void *workspace, uint64_t workspace_bytes, void *stream) {
	if (int rc = check_forest(meta, meta_dev, forest)) return rc;
	NR3D_CHECK(param_dtype == NR3D_F32 || param_dtype == NR3D_F16, "LoTD forest: params must be f32 or f16");
	const bool p_half = param_dtype == NR3D_F16;
	if (N == 0 || max_level < 0) return 0;
	NR3D_CHECK(dL_dy && x && params && dL_dparam, "LoTD forest::bwd: NULL tensor pointer");
	const Batch ba{block_inds, block_offsets, batch_data_size, meta->n_params};
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	if (workspace) {
		const ForestDev fo = dev_of(forest);
		bool handled = false;
		const int64_t E = meta->n_encoded_dims;
		if (int rc = dparam_binned(dL_ddLdx != nullptr, meta, meta_dev, N, dL_ddLdx, dL_dy, E, 1, x, (const float *)params, ba,
		                           forest->n_trees, max_level, dL_dparam, workspace, workspace_bytes, (hipStream_t)stream, handled, &fo,
		                           0, false, false, false, p_half)) return rc;
		if (handled) return 0;
	}
	const dim3 grid(div_up(N, kBlock), meta->n_pseudo_levels);
	DISPATCH_G(meta->n_feat_per_pseudo_lvl, {
		auto launch = [&](auto kern, auto *tab) {
			hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, (hipStream_t)stream, md, dev_of(forest), N, meta->n_encoded_dims, max_level,
			                   meta->interpolation_type, dL_ddLdx, dL_dy, x, tab, ba, dL_dparam);
		};
		if (p_half) { if (dL_ddLdx) launch(k_forest_bwd_dparam<G, true, __half>, (const __half *)params); else launch(k_forest_bwd_dparam<G, false, __half>, (const __half *)params); }
		else        { if (dL_ddLdx) launch(k_forest_bwd_dparam<G, true, float>, (const float *)params); else launch(k_forest_bwd_dparam<G, false, float>, (const float *)params); }
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_lotd_forest_bwd_bwd_dx(const nr3d_lotd_meta_t *meta, const void *meta_dev, const nr3d_forest_meta_t *forest,
                                           uint32_t N, const float *dL_ddLdx, const float *dL_dy, const float *x,
                                           const void *params, int param_dtype, const int64_t *block_inds,
                                           const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level, float *dL_dx,
                                           void *stream) {
	if (int rc = check_forest(meta, meta_dev, forest)) return rc;
	NR3D_CHECK(param_dtype == NR3D_F32 || param_dtype == NR3D_F16, "LoTD forest: params must be f32 or f16");
	if (N == 0) return 0;
	NR3D_CHECK(dL_ddLdx && dL_dy && x && params && dL_dx, "LoTD forest::bwd_bwd_input: NULL tensor pointer");
	const Batch ba{block_inds, block_offsets, batch_data_size, meta->n_params};
	const auto md = (const nr3d_lotd_meta_t *)meta_dev;
	DISPATCH_G(meta->n_feat_per_pseudo_lvl, {
		auto launch = [&](auto kern, auto *tab) {
			hipLaunchKernelGGL(kern, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, md, dev_of(forest), N,
			                   meta->n_encoded_dims, meta->n_pseudo_levels, max_level, meta->interpolation_type, dL_ddLdx, dL_dy, x, tab,
			                   ba, dL_dx);
		};
		if (param_dtype == NR3D_F16) launch(k_forest_bwd_bwd_dx<G, __half>, (const __half *)params);
		else launch(k_forest_bwd_bwd_dx<G, float>, (const float *)params);
	});
	NR3D_LAUNCH_CHECK();
	return 0;
}
