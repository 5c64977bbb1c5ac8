/* oracle/lotd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT (see lotd_oracle.h).
 *
 * Plain-C fp32 restatement of the reference LoTD CUDA kernels.  Each function cites the reference
 * file:line it follows (paths relative to the reference checkout).  The loops are deliberately
 * the reference's loops (one "thread" = one loop iteration, corner order c = 0..2^D-1, features
 * innermost), NOT the factored/tiled structure used by the HIP kernels, so that agreement between
 * the two is meaningful.
 *
 * Floating-point contract: the reference is compiled by nvcc with its default --fmad=true, so every
 * a*b+c in device code is a single-rounding FMA.  This file is compiled with -ffp-contract=off and
 * spells the contractions that decide integer results (cell selection in pos_fract) as explicit
 * fmaf(); value-only accumulations use fmaf() where the reference expression is a*b+c.
 *
 * Parity pinning status: see oracle/README.md.
 */
#include "lotd_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * Meta  (csrc/lotd/src/lotd_torch_api.cu:29-230)
 * ---------------------------------------------------------------------------------------------- */
static int all_div(const int32_t *v, uint32_t n, int32_t d) {
	for (uint32_t i = 0; i < n; ++i) if ((v[i] - (v[i] / d) * d) != 0) return 0;
	return 1;
}

int orc_lotd_create_meta(int32_t n_input_dim, uint32_t n_levels, const int32_t *res_multidim,
                         const int32_t *n_feats, const int32_t *types, uint32_t hashmap_size,
                         int use_smooth_step, orc_lotd_meta_t *m, char *errbuf, int errlen) {
#define FAIL(msg) do { snprintf(errbuf, errlen, "%s", msg); return 1; } while (0)
	memset(m, 0, sizeof(*m));
	m->interpolation_type = use_smooth_step ? 1u : 0u;
	if (!(n_input_dim == 2 || n_input_dim == 3 || n_input_dim == 4))
		FAIL("LoTDEncoding: `n_input_dim` must be 2/3/4.");                      /* :44-46 */
	const uint32_t D = (uint32_t)n_input_dim;
	m->n_dims_to_encode = D;
	m->n_levels = n_levels;
	if (n_levels > ORC_MAX_LEVELS) FAIL("LoTDEncoding:` num_level` exceeds maximum level"); /* :51 */

	if (all_div(n_feats, n_levels, 8)) m->n_feat_per_pseudo_lvl = 8;           /* :63-71 */
	else if (all_div(n_feats, n_levels, 4)) m->n_feat_per_pseudo_lvl = 4;
	else if (all_div(n_feats, n_levels, 2)) m->n_feat_per_pseudo_lvl = 2;
	else FAIL("LoTDEncoding: the greatest common divisor of `lod_n_feats` must be at least 2");

	const uint32_t max_params = 0xFFFFFFFFu / 2;                                 /* :77 */
	uint32_t acc = 0;
	float acc_f = 0.0f;
	for (uint32_t l = 0; l < n_levels; ++l) {
		const uint32_t nf = (uint32_t)n_feats[l];
		const uint32_t tp = (uint32_t)types[l];
		m->level_n_feats[l] = nf;
		m->level_types[l] = tp;
		m->n_pseudo_levels += nf / m->n_feat_per_pseudo_lvl;
		m->n_encoded_dims += nf;
		uint32_t res[ORC_MAX_DIMS];
		for (uint32_t d = 0; d < D; ++d) {
			res[d] = (uint32_t)res_multidim[l * D + d];
			if (res[d] <= 2) FAIL("LoTDEncoding: only support grid resolutions >= 3"); /* :100 */
			m->level_res[l][d] = res[d];
		}
		uint32_t size = 0;
		float size_f = 0.f;
		switch (tp) {
		case ORC_Dense:                                                          /* :119-126 */
			size = 1; size_f = 1.0f;
			for (uint32_t d = 0; d < D; ++d) { size *= res[d]; size_f *= (float)res[d]; }
			break;
		case ORC_NPlaneMul:
		case ORC_NPlaneSum:                                                      /* :128-146 */
			for (uint32_t ld = 0; ld < D; ++ld) {
				uint32_t ps = 1; float psf = 1.0f;
				for (uint32_t d = 0; d < D - 1; ++d) {
					uint32_t rd = d >= ld ? d + 1 : d;
					ps *= res[rd]; psf *= (float)res[rd];
				}
				size += ps; size_f += psf;
			}
			break;
		case ORC_VectorMatrix:                                                   /* :148-165 */
			if (D != 3) FAIL("LoTDEncoding: VectorMatrix mode only support 3D encoding.");
			for (uint32_t ld = 0; ld < D; ++ld) {
				uint32_t ps = 1;
				for (uint32_t d = 0; d < D - 1; ++d) {
					uint32_t rd = d >= ld ? d + 1 : d;
					ps *= res[rd];
				}
				size += ps + res[ld];
				size_f += (float)(ps + res[ld]);
			}
			break;
		case ORC_VecZMatXoY:                                                     /* :167-173 */
			if (D != 3) FAIL("LoTDEncoding: VecZMatXoY mode only support 3D encoding.");
			size = res[0] * res[1] + res[2];
			size_f = (float)res[0] * (float)res[1] + (float)res[2];
			break;
		case ORC_CPfast:
		case ORC_CP:                                                             /* :175-183 */
			for (uint32_t d = 0; d < D; ++d) { size += res[d]; size_f += (float)res[d]; }
			break;
		case ORC_Hash:                                                           /* :185-191 */
			if (!hashmap_size) FAIL("LoTDEncoding: Hash mode need `hashmap_size`");
			size = hashmap_size; size_f = (float)hashmap_size;
			break;
		default:
			FAIL("LoTDEncoding: Invalid lod type");
		}
		acc_f += size_f * (float)nf;
		if (acc_f > (float)max_params) FAIL("LoTDEncoding: param size too large."); /* :197-200 */
		m->level_sizes[l] = size;
		m->level_n_params[l] = size * nf;
		m->level_offsets[l] = acc;
		acc += size * nf;
	}
	m->level_offsets[n_levels] = acc;
	m->n_params = acc;
	uint32_t q = 0;                                                              /* :213-224 */
	for (uint32_t l = 0; l < n_levels; ++l) {
		uint32_t n = (uint32_t)n_feats[l] / m->n_feat_per_pseudo_lvl;
		for (uint32_t j = 0; j < n; ++j) { m->map_levels[q + j] = l; m->map_cnt[q + j] = j; }
		q += n;
	}
	if (m->n_encoded_dims > 1024)                                                /* :227 */
		FAIL("LoTDEncoding: total number of features too large. Shoule be <= 1024.");
	return 0;
#undef FAIL
}

/* ------------------------------------------------------------------------------------------------
 * Index functions  (csrc/lotd/include/lotd/lotd_cuda.h:92-296)
 * ---------------------------------------------------------------------------------------------- */
static uint32_t idx_dense(uint32_t D, uint32_t feat, const uint32_t *res, uint32_t F, const uint32_t *p) {
	uint32_t stride = 1, index = 0;                                              /* :92-118 */
	for (uint32_t d = 0; d < D; ++d) {
		index += p[D - 1 - d] * stride;
		stride *= res[D - 1 - d];
	}
	return index * F + feat;
}

static uint32_t idx_hash(uint32_t D, uint32_t feat, uint32_t size, uint32_t F, const uint32_t *p) {
	static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
	                                   2097192037u, 1434869437u, 2165219737u}; /* :120-143 */
	uint32_t h = 0;
	for (uint32_t i = 0; i < D; ++i) h ^= p[i] * primes[i];
	return (h % size) * F + feat;
}

static uint32_t idx_nplane_sub(uint32_t D, uint32_t feat, const uint32_t *res, uint32_t F,
                               const uint32_t *pos_plane, uint32_t jump_dim) {  /* :145-174 */
	const uint32_t N_1 = D - 1;
	uint32_t stride = 1, index = 0;
	for (uint32_t d2 = 0; d2 < N_1; ++d2) {
		uint32_t d3 = d2 >= jump_dim ? d2 + 1 : d2;
		index += pos_plane[N_1 - 1 - d2] * stride;
		stride *= res[N_1 - d3];
	}
	return (jump_dim * stride + index) * F + feat;
}

static uint32_t idx_nplane(uint32_t D, uint32_t feat, const uint32_t *res, uint32_t F,
                           const uint32_t *p, uint32_t jump_dim) {              /* :176-206 */
	const uint32_t N_1 = D - 1;
	uint32_t stride = 1, index = 0;
	for (uint32_t d2 = 0; d2 < N_1; ++d2) {
		uint32_t d3 = d2 >= jump_dim ? d2 + 1 : d2;
		index += p[N_1 - d3] * stride;
		stride *= res[N_1 - d3];
	}
	return index * F + feat;
}

static uint32_t idx_cp_line(uint32_t feat, const uint32_t *res, uint32_t F, uint32_t pos_line,
                            uint32_t line_dim) {                                /* :208-222 */
	uint32_t acc = 0;
	for (uint32_t d = 0; d < line_dim; ++d) acc += res[d];
	return (acc + pos_line) * F + feat;
}

static void idx_cp(uint32_t D, uint32_t feat, const uint32_t *res, uint32_t F, const uint32_t *p,
                   uint32_t *index_line) {                                      /* :224-239 */
	uint32_t acc = 0;
	for (uint32_t ld = 0; ld < D; ++ld) {
		index_line[ld] = (acc + p[ld]) * F + feat;
		acc += res[ld];
	}
}

static void idx_vm(uint32_t D, uint32_t feat, const uint32_t *res, uint32_t F, const uint32_t *p,
                   uint32_t *index_plane, uint32_t *index_line) {               /* :241-277 */
	const uint32_t N_1 = D - 1;
	uint32_t acc_line = 0;
	for (uint32_t ld = 0; ld < D; ++ld) {
		index_line[ld] = (acc_line + p[ld]) * F + feat;
		acc_line += res[ld];
	}
	uint32_t acc_plane = 0;
	for (uint32_t ld = 0; ld < D; ++ld) {
		uint32_t stride = 1, index = 0;
		const uint32_t rev_jump = N_1 - ld;
		for (uint32_t d2 = 0; d2 < N_1; ++d2) {
			uint32_t d3 = d2 >= rev_jump ? d2 + 1 : d2;
			index += p[N_1 - d3] * stride;
			stride *= res[N_1 - d3];
		}
		index_plane[ld] = (acc_line + acc_plane + index) * F + feat;
		acc_plane += stride;
	}
}

static void idx_vm_xoy(uint32_t feat, const uint32_t *res, uint32_t F, const uint32_t *p,
                       uint32_t *index_plane, uint32_t *index_line) {           /* :279-296 */
	*index_line = p[2] * F + feat;
	*index_plane = (res[2] + p[1] + p[0] * res[0]) * F + feat;
}

/* ------------------------------------------------------------------------------------------------
 * Per-(point, level) context
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	uint32_t D, F, type, size;
	uint32_t res[ORC_MAX_DIMS];
	float scale[ORC_MAX_DIMS];
	uint32_t pg[ORC_MAX_DIMS];                 /* pos_grid */
	float pos[ORC_MAX_DIMS], dpos[ORC_MAX_DIMS], ddpos[ORC_MAX_DIMS];
	const float *grid;                         /* already offset to batch + level */
	uint32_t base;                             /* batch_offset + level_offsets[level] */
	int smooth;
	/* forest mode only (lotd_forest.h): the block of the point and how to find its neighbours' parameters */
	const orc_forest_t *forest;
	int16_t bk[3];
	uint32_t block_offset, block_n_params;
	const int64_t *block_offsets;
} ctx_t;

/* Set by the orc_lotd_forest_* wrappers around the shared bodies below; NULL = plain LoTD. */
static const orc_forest_t *g_forest = NULL;

/* identify  (csrc/forest/forest.h:25-58, modified from kaolin): index of the octree node at integer
 * coordinates k on `level` in the breadth-first point hierarchy, or -1 */
int32_t orc_forest_identify(const orc_forest_t *fo, const int16_t *k) {
	const int maxval = (1 << fo->level) - 1;
	if (k[0] < 0 || k[1] < 0 || k[2] < 0 || k[0] > maxval || k[1] > maxval || k[2] > maxval) return -1;
	int ord = 0;
	for (uint32_t l = 0; l < fo->level; ++l) {
		const uint32_t depth = fo->level - l - 1;
		const uint32_t mask = 1u << depth;
		const uint32_t child = ((mask & (uint32_t)k[0]) << 2 | (mask & (uint32_t)k[1]) << 1 | (mask & (uint32_t)k[2])) >> depth;
		const uint8_t bits = fo->octree[ord];
		if (!(bits & (1u << child))) return -1;
		const uint32_t cnt = (uint32_t)__builtin_popcount(bits & ((2u << child) - 1u));   /* inclusive */
		ord = fo->exsum[ord] + (int)cnt;
		if (depth == 0) return ord;
	}
	return ord;
}

/* the "continuity fixing" of every forest_*_n_linear (lotd_forest.h:55-88 and its twins): corner index 0 is
 * the left neighbour's res-1, res+1 the right neighbour's 0, 1..res the block's own 0..res-1.
 * -> 0 when the corner has no parameters (continuity off / neighbour missing); else the corner's position inside
 * its block and the offset of that block's parameters relative to the point's own block. */
static int forest_resolve(const ctx_t *c, const uint32_t *lp, uint32_t *lpl, int64_t *delta) {
	int16_t bl[3];
	int changed = 0;
	for (uint32_t d = 0; d < c->D; ++d) {
		if (lp[d] == 0) { bl[d] = (int16_t)(c->bk[d] - 1); lpl[d] = c->res[d] - 1; changed = 1; }
		else if (lp[d] == c->res[d] + 1) { bl[d] = (int16_t)(c->bk[d] + 1); lpl[d] = 0; changed = 1; }
		else { bl[d] = c->bk[d]; lpl[d] = lp[d] - 1; }
	}
	*delta = 0;
	if (changed) {
		if (!c->forest->continuity_enabled) return 0;
		const int32_t pidx = orc_forest_identify(c->forest, bl);
		const int32_t bi = pidx == -1 ? -1 : pidx - (int32_t)c->forest->level_poffset;
		if (bi < 0) return 0;
		const uint32_t off = c->block_offsets ? (uint32_t)c->block_offsets[bi] : (uint32_t)bi * c->block_n_params;
		*delta = (int64_t)off - (int64_t)c->block_offset;
	}
	return 1;
}

/* pos_fract, all three overloads  (lotd_cuda.h:959-1077); scale = res-2 (lotd_encoding.h:185-191) */
static void pos_fract(ctx_t *c, const float *x) {
	for (uint32_t d = 0; d < c->D; ++d) {
		float val = fmaf(x[d], c->scale[d], 0.5f);      /* nvcc: positions*scale+0.5f -> FMA */
		float fl = floorf(val);
		c->pg[d] = (uint32_t)fl;
		val -= (float)c->pg[d];
		if (!c->smooth) {
			c->pos[d] = val; c->dpos[d] = 1.0f; c->ddpos[d] = 0.0f;
		} else {
			c->pos[d] = val * val * fmaf(-2.0f, val, 3.0f);      /* val*val*(3-2*val) */
			c->dpos[d] = 6.0f * val * (1.0f - val);
			c->ddpos[d] = fmaf(-12.0f, val, 6.0f);               /* 6-12*val */
		}
	}
}

/* Returns 0 if this (point, level) is skipped.  (lotd_encoding.h:161-201 and same preamble in every kernel) */
static int setup_ctx(ctx_t *c, const orc_lotd_meta_t *m, uint32_t level, uint32_t i, const float *x,
                     const float *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                     uint32_t batch_data_size, int32_t max_level) {
	if ((int64_t)level > (int64_t)max_level) return 0;
	uint32_t batch_ind = 0;
	if (batch_inds) {
		if (batch_inds[i] < 0) return 0;
		batch_ind = (uint32_t)batch_inds[i];
	} else if (batch_data_size) {
		batch_ind = i / batch_data_size;
	}
	const uint32_t batch_offset = batch_offsets ? (uint32_t)batch_offsets[batch_ind]
	                                            : batch_ind * m->level_offsets[m->n_levels];
	c->base = batch_offset + m->level_offsets[level];
	c->grid = params ? params + c->base : NULL;
	c->forest = g_forest;
	if (g_forest) {                 /* lotd_forest.h:213-216 */
		c->block_offset = batch_offset;
		c->block_n_params = m->level_offsets[m->n_levels];
		c->block_offsets = batch_offsets;
		for (int d = 0; d < 3; ++d) c->bk[d] = g_forest->block_ks[3 * (size_t)batch_ind + d];
	}
	c->D = m->n_dims_to_encode;
	c->size = m->level_sizes[level];
	c->F = m->level_n_feats[level];
	c->type = m->level_types[level];
	c->smooth = (m->interpolation_type == 1);
	for (uint32_t d = 0; d < c->D; ++d) {
		c->res[d] = m->level_res[level][d];
		c->scale[d] = g_forest ? (float)c->res[d]          /* "NOTE: for forest", lotd_forest.h:231 */
		                       : (float)(c->res[d] - 2);
	}
	pos_fract(c, x + (size_t)i * c->D);
	return 1;
}

/* ------------------------------------------------------------------------------------------------
 * grid_val_*_impl  (lotd_cuda.h:298-492): value of `nf` consecutive features at a corner
 * ---------------------------------------------------------------------------------------------- */
static void grid_val(const ctx_t *c, const uint32_t *lp, uint32_t feat_off, uint32_t nf, float *val) {
	const float *g = c->grid;
	const uint32_t D = c->D, F = c->F;
	uint32_t lpl[ORC_MAX_DIMS];
	if (c->forest) {
		int64_t delta;
		if (!forest_resolve(c, lp, lpl, &delta)) { for (uint32_t f = 0; f < nf; ++f) val[f] = 0.f; return; }
		g += delta; lp = lpl;
	}
	switch (c->type) {
	case ORC_Dense: {
		uint32_t idx = idx_dense(D, feat_off, c->res, F, lp);
		for (uint32_t f = 0; f < nf; ++f) val[f] = g[idx + f];
	} break;
	case ORC_Hash: {
		uint32_t idx = idx_hash(D, feat_off, c->size, F, lp);
		for (uint32_t f = 0; f < nf; ++f) val[f] = g[idx + f];
	} break;
	case ORC_VectorMatrix: {                                                     /* :334-361 */
		uint32_t ip[ORC_MAX_DIMS], il[ORC_MAX_DIMS];
		for (uint32_t f = 0; f < nf; ++f) val[f] = 0.f;
		idx_vm(D, feat_off, c->res, F, lp, ip, il);
		for (uint32_t ld = 0; ld < D; ++ld)
			for (uint32_t f = 0; f < nf; ++f) val[f] = fmaf(g[ip[ld] + f], g[il[ld] + f], val[f]);
	} break;
	case ORC_VecZMatXoY: {                                                       /* :396-414 */
		uint32_t ip, il;
		idx_vm_xoy(feat_off, c->res, F, lp, &ip, &il);
		for (uint32_t f = 0; f < nf; ++f) val[f] = g[ip + f] * g[il + f];
	} break;
	case ORC_NPlaneMul: {                                                        /* :443-468 */
		uint32_t ip[ORC_MAX_DIMS];
		for (uint32_t j = 0; j < D; ++j) ip[j] = idx_nplane(D, feat_off, c->res, F, lp, j);
		for (uint32_t f = 0; f < nf; ++f) {
			float r = g[ip[0] + f];
			for (uint32_t j = 1; j < D; ++j) r *= g[ip[j] + f];
			val[f] = r;
		}
	} break;
	case ORC_CP: {                                                               /* :470-492 */
		uint32_t il[ORC_MAX_DIMS];
		idx_cp(D, feat_off, c->res, F, lp, il);
		for (uint32_t f = 0; f < nf; ++f) {
			float r = g[il[0] + f];
			for (uint32_t j = 1; j < D; ++j) r *= g[il[j] + f];
			val[f] = r;
		}
	} break;
	default:
		for (uint32_t f = 0; f < nf; ++f) val[f] = 0.f;
	}
}

/* add_grid_gridient_*_impl  (lotd_cuda.h:494-829); fp32 "atomicAdd" == plain += here.
 * `gd` (double accumulator) is used instead of `g` when non-NULL (tolerance twin). */
typedef struct { float *g; double *gd; } acc_t;
static inline void acc_add(acc_t a, uint32_t idx, float v) {
	if (a.gd) a.gd[idx] += (double)v; else a.g[idx] += v;
}

static void add_grad(const ctx_t *c, const uint32_t *lp, uint32_t feat_off, uint32_t nf,
                     const float *grad, float weight, acc_t a) {
	const float *g = c->grid;
	const uint32_t D = c->D, F = c->F;
	uint32_t lpl[ORC_MAX_DIMS];
	if (c->forest) {
		int64_t delta;
		if (!forest_resolve(c, lp, lpl, &delta)) return;
		g += delta; lp = lpl;
		if (a.g) a.g += delta;
		if (a.gd) a.gd += delta;
	}
	switch (c->type) {
	case ORC_Dense: {
		uint32_t idx = idx_dense(D, feat_off, c->res, F, lp);
		for (uint32_t f = 0; f < nf; ++f) acc_add(a, idx + f, grad[f] * weight);
	} break;
	case ORC_Hash: {
		uint32_t idx = idx_hash(D, feat_off, c->size, F, lp);
		for (uint32_t f = 0; f < nf; ++f) acc_add(a, idx + f, grad[f] * weight);
	} break;
	case ORC_VectorMatrix: {                                                     /* :580-635 */
		float wg[8];
		uint32_t ip[ORC_MAX_DIMS], il[ORC_MAX_DIMS];
		for (uint32_t f = 0; f < nf; ++f) wg[f] = grad[f] * weight;
		idx_vm(D, feat_off, c->res, F, lp, ip, il);
		for (uint32_t ld = 0; ld < D; ++ld)
			for (uint32_t f = 0; f < nf; ++f) {
				acc_add(a, ip[ld] + f, wg[f] * g[il[ld] + f]);
				acc_add(a, il[ld] + f, wg[f] * g[ip[ld] + f]);
			}
	} break;
	case ORC_VecZMatXoY: {                                                       /* :654-704 */
		float wg[8];
		uint32_t ip, il;
		for (uint32_t f = 0; f < nf; ++f) wg[f] = grad[f] * weight;
		idx_vm_xoy(feat_off, c->res, F, lp, &ip, &il);
		for (uint32_t f = 0; f < nf; ++f) {
			acc_add(a, ip + f, wg[f] * g[il + f]);
			acc_add(a, il + f, wg[f] * g[ip + f]);
		}
	} break;
	case ORC_NPlaneMul:                                                          /* :706-768 */
	case ORC_CP: {                                                               /* :770-829 */
		float wg[8];
		uint32_t ii[ORC_MAX_DIMS];
		if (c->type == ORC_CP) idx_cp(D, feat_off, c->res, F, lp, ii);
		else for (uint32_t j = 0; j < D; ++j) ii[j] = idx_nplane(D, feat_off, c->res, F, lp, j);
		for (uint32_t f = 0; f < nf; ++f) wg[f] = grad[f] * weight;
		for (uint32_t gdim = 0; gdim < D; ++gdim) {
			for (uint32_t f = 0; f < nf; ++f) {
				float cur = wg[f];
				for (uint32_t ng = 0; ng < D - 1; ++ng) {
					uint32_t dim = ng >= gdim ? ng + 1 : ng;
					cur *= g[ii[dim] + f];
				}
				acc_add(a, ii[gdim] + f, cur);
			}
		}
	} break;
	default: break;
	}
}

/* calc_dLdx_dim_*_impl  (lotd_cuda.h:831-957) */
static float calc_dLdx(const ctx_t *c, const uint32_t *lp, uint32_t feat_off, uint32_t nf,
                       const float *grad, float weight) {
	const float *g = c->grid;
	const uint32_t D = c->D, F = c->F;
	float r = 0.f;
	uint32_t lpl[ORC_MAX_DIMS];
	if (c->forest) {
		int64_t delta;
		if (!forest_resolve(c, lp, lpl, &delta)) return 0.f;
		g += delta; lp = lpl;
	}
	switch (c->type) {
	case ORC_Dense: {
		uint32_t idx = idx_dense(D, feat_off, c->res, F, lp);
		for (uint32_t f = 0; f < nf; ++f) r = fmaf(g[idx + f] * grad[f], weight, r);
	} break;
	case ORC_Hash: {
		uint32_t idx = idx_hash(D, feat_off, c->size, F, lp);
		for (uint32_t f = 0; f < nf; ++f) r = fmaf(g[idx + f] * grad[f], weight, r);
	} break;
	case ORC_VectorMatrix: {
		float wg[8];
		uint32_t ip[ORC_MAX_DIMS], il[ORC_MAX_DIMS];
		for (uint32_t f = 0; f < nf; ++f) wg[f] = grad[f] * weight;
		idx_vm(D, feat_off, c->res, F, lp, ip, il);
		for (uint32_t ld = 0; ld < D; ++ld)
			for (uint32_t f = 0; f < nf; ++f) r = fmaf(g[ip[ld] + f] * g[il[ld] + f], wg[f], r);
	} break;
	case ORC_VecZMatXoY: {
		uint32_t ip, il;
		idx_vm_xoy(feat_off, c->res, F, lp, &ip, &il);
		for (uint32_t f = 0; f < nf; ++f) r = fmaf(g[ip + f] * g[il + f] * grad[f], weight, r);
	} break;
	default: break;
	}
	return r;
}

/* ------------------------------------------------------------------------------------------------
 * Forward  (generic kernel_lod, lotd_encoding.h:113-428, fwd_n_linear :31-111; the hash-only
 * kernels lotd_hash_only.h:15-378 + linear_interpolate.cuh:9-150 compute the same sums)
 * ---------------------------------------------------------------------------------------------- */
void orc_lotd_fwd(const orc_lotd_meta_t *m, uint32_t N, const float *x, const float *params,
                  const int64_t *batch_inds, const int64_t *batch_offsets, uint32_t batch_data_size,
                  int32_t max_level, float *y, float *dy_dx) {
	const uint32_t D = m->n_dims_to_encode, E = m->n_encoded_dims, G = m->n_feat_per_pseudo_lvl;
	memset(y, 0, sizeof(float) * (size_t)N * E);
	if (dy_dx) memset(dy_dx, 0, sizeof(float) * (size_t)N * E * D);
	if (max_level <= -1) return;                                   /* lotd_torch_api.cu:294-297 */
#pragma omp parallel for schedule(static)
	for (int64_t ii = 0; ii < (int64_t)N; ++ii) {
		const uint32_t i = (uint32_t)ii;
		for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
			const uint32_t level = m->map_levels[q];
			const uint32_t feat_off = m->map_cnt[q] * G;
			const uint32_t out_off = q * G;
			ctx_t c;
			if (!setup_ctx(&c, m, level, i, x, params, batch_inds, batch_offsets, batch_data_size, max_level))
				continue; /* outputs stay zero (set_zero, :142-164) */
			float result[8] = {0};
			float grads[8][ORC_MAX_DIMS];
			memset(grads, 0, sizeof(grads));
			float val[8], vl[8], vr[8];
			uint32_t lp[ORC_MAX_DIMS];

			if (c.type == ORC_NPlaneSum) {                                       /* :268-351 */
				if (D > 2) {
					for (uint32_t jd = 0; jd < D; ++jd)
						for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
							float w = 1.0f;
							uint32_t pp[ORC_MAX_DIMS];
							for (uint32_t d2 = 0; d2 < D - 1; ++d2) {
								uint32_t d3 = d2 >= jd ? d2 + 1 : d2;
								if ((idx & (1u << d2)) == 0) { w *= 1.0f - c.pos[d3]; pp[d2] = c.pg[d3]; }
								else { w *= c.pos[d3]; pp[d2] = c.pg[d3] + 1; }
							}
							uint32_t index = idx_nplane_sub(D, feat_off, c.res, c.F, pp, jd);
							for (uint32_t f = 0; f < G; ++f) result[f] = fmaf(w, c.grid[index + f], result[f]);
						}
					if (dy_dx)
						for (uint32_t jd = 0; jd < D; ++jd)
							for (uint32_t g2 = 0; g2 < D - 1; ++g2) {
								const uint32_t g3 = g2 >= jd ? g2 + 1 : g2;
								for (uint32_t idx = 0; idx < (1u << (D - 2)); ++idx) {
									float w = c.scale[g3] * c.dpos[g3];
									uint32_t pp[ORC_MAX_DIMS];
									for (uint32_t ng = 0; ng + 2 < D; ++ng) {
										const uint32_t d2 = ng >= g2 ? ng + 1 : ng;
										const uint32_t d3 = d2 >= jd ? d2 + 1 : d2;
										if ((idx & (1u << ng)) == 0) { w *= 1.0f - c.pos[d3]; pp[d2] = c.pg[d3]; }
										else { w *= c.pos[d3]; pp[d2] = c.pg[d3] + 1; }
									}
									pp[g2] = c.pg[g3];
									uint32_t il = idx_nplane_sub(D, feat_off, c.res, c.F, pp, jd);
									pp[g2] = c.pg[g3] + 1;
									uint32_t ir = idx_nplane_sub(D, feat_off, c.res, c.F, pp, jd);
									for (uint32_t f = 0; f < G; ++f)
										grads[f][g3] = fmaf(w, c.grid[ir + f] - c.grid[il + f], grads[f][g3]);
								}
							}
				}
			} else if (c.type == ORC_CPfast) {                                   /* :353-410 */
				float r_[8];
				for (uint32_t f = 0; f < G; ++f) r_[f] = 1.0f;
				for (uint32_t ld = 0; ld < D; ++ld) {
					uint32_t il = idx_cp_line(feat_off, c.res, c.F, c.pg[ld], ld);
					uint32_t ir = idx_cp_line(feat_off, c.res, c.F, c.pg[ld] + 1, ld);
					float w = c.pos[ld];
					for (uint32_t f = 0; f < G; ++f)
						r_[f] *= fmaf(w, c.grid[ir + f], (1.0f - w) * c.grid[il + f]);
				}
				for (uint32_t f = 0; f < G; ++f) result[f] = r_[f];
				if (dy_dx)
					for (uint32_t gd = 0; gd < D; ++gd) {
						float w = c.scale[gd] * c.dpos[gd];
						uint32_t il = idx_cp_line(feat_off, c.res, c.F, c.pg[gd], gd);
						uint32_t ir = idx_cp_line(feat_off, c.res, c.F, c.pg[gd] + 1, gd);
						for (uint32_t f = 0; f < G; ++f) grads[f][gd] = w * (c.grid[ir + f] - c.grid[il + f]);
						for (uint32_t ng = 0; ng < D - 1; ++ng) {
							const uint32_t dim = ng >= gd ? ng + 1 : ng;
							uint32_t jl = idx_cp_line(feat_off, c.res, c.F, c.pg[dim], dim);
							uint32_t jr = idx_cp_line(feat_off, c.res, c.F, c.pg[dim] + 1, dim);
							float nw = c.pos[dim];
							for (uint32_t f = 0; f < G; ++f)
								grads[f][gd] *= fmaf(nw, c.grid[jr + f], (1.0f - nw) * c.grid[jl + f]);
						}
					}
			} else {                                                             /* fwd_n_linear :31-111 */
				for (uint32_t idx = 0; idx < (1u << D); ++idx) {
					float w = 1.0f;
					for (uint32_t d = 0; d < D; ++d) {
						if ((idx & (1u << d)) == 0) { w *= 1.0f - c.pos[d]; lp[d] = c.pg[d]; }
						else { w *= c.pos[d]; lp[d] = c.pg[d] + 1; }
					}
					grid_val(&c, lp, feat_off, G, val);
					for (uint32_t f = 0; f < G; ++f) result[f] = fmaf(w, val[f], result[f]);
				}
				if (dy_dx)
					for (uint32_t gd = 0; gd < D; ++gd)
						for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
							float w = c.scale[gd] * c.dpos[gd];
							for (uint32_t ng = 0; ng < D - 1; ++ng) {
								const uint32_t dim = ng >= gd ? ng + 1 : ng;
								if ((idx & (1u << ng)) == 0) { w *= 1.0f - c.pos[dim]; lp[dim] = c.pg[dim]; }
								else { w *= c.pos[dim]; lp[dim] = c.pg[dim] + 1; }
							}
							lp[gd] = c.pg[gd];
							grid_val(&c, lp, feat_off, G, vl);
							lp[gd] = c.pg[gd] + 1;
							grid_val(&c, lp, feat_off, G, vr);
							for (uint32_t f = 0; f < G; ++f)
								grads[f][gd] = fmaf(w, vr[f] - vl[f], grads[f][gd]);
						}
			}
			for (uint32_t f = 0; f < G; ++f) y[(size_t)i * E + out_off + f] = result[f];
			if (dy_dx)
				for (uint32_t f = 0; f < G; ++f)
					for (uint32_t d = 0; d < D; ++d)
						dy_dx[((size_t)i * E + out_off + f) * D + d] = grads[f][d];
		}
	}
}

/* ------------------------------------------------------------------------------------------------
 * dL/dparam  (kernel_lod_backward_grid, lotd_encoding.h:467-711; bwd_n_linear :430-465;
 *             hash-only twin lotd_hash_only.h:380-470)
 * One reference thread = (point i, pseudo level q, feature pair).  Deterministic order here:
 * pseudo levels in parallel (disjoint slices only when levels differ -> we parallelise over LEVELS),
 * points ascending inside.
 * ---------------------------------------------------------------------------------------------- */
static void bwd_dparam_level(const orc_lotd_meta_t *m, uint32_t level, uint32_t i_begin, uint32_t N, const float *dL_ddLdx,
                             const float *dL_dy, const float *x, const float *params,
                             const int64_t *batch_inds, const int64_t *batch_offsets,
                             uint32_t batch_data_size, int32_t max_level, acc_t acc_all) {
	const uint32_t D = m->n_dims_to_encode, E = m->n_encoded_dims, G = m->n_feat_per_pseudo_lvl;
	const uint32_t NFT = 2; /* N_FEAT_PER_THREAD = min(2, G), :1589 */
	for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
		if (m->map_levels[q] != level) continue;
		for (uint32_t i = i_begin; i < N; ++i) {
			ctx_t c;
			if (!setup_ctx(&c, m, level, i, x, params, batch_inds, batch_offsets, batch_data_size, max_level))
				continue;
			acc_t a;
			a.g = acc_all.g ? acc_all.g + c.base : NULL;
			a.gd = acc_all.gd ? acc_all.gd + c.base : NULL;
			for (uint32_t feature = 0; feature < G; feature += NFT) {
				const uint32_t feat_off = m->map_cnt[q] * G + feature;
				const uint32_t out_off = q * G + feature;
				const float *grad = dL_dy + (size_t)i * E + out_off;
				const float *gin = dL_ddLdx ? dL_ddLdx + (size_t)i * D : NULL;
				uint32_t lp[ORC_MAX_DIMS];

				if (!gin) { /* ---------------- first order ---------------- */
					if (c.type == ORC_NPlaneSum) {                               /* :597-651 */
						if (D > 2)
							for (uint32_t jd = 0; jd < D; ++jd)
								for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
									float w = 1.0f;
									uint32_t pp[ORC_MAX_DIMS];
									for (uint32_t d2 = 0; d2 < D - 1; ++d2) {
										uint32_t d3 = d2 >= jd ? d2 + 1 : d2;
										if ((idx & (1u << d2)) == 0) { w *= 1.0f - c.pos[d3]; pp[d2] = c.pg[d3]; }
										else { w *= c.pos[d3]; pp[d2] = c.pg[d3] + 1; }
									}
									uint32_t index = idx_nplane_sub(D, feat_off, c.res, c.F, pp, jd);
									for (uint32_t f = 0; f < NFT; ++f) acc_add(a, index + f, grad[f] * w);
								}
					} else if (c.type == ORC_CPfast) {                           /* :653-705 */
						for (uint32_t gd = 0; gd < D; ++gd) {
							float gl[2] = {grad[0], grad[1]};
							for (uint32_t ng = 0; ng < D - 1; ++ng) {
								const uint32_t dim = ng >= gd ? ng + 1 : ng;
								uint32_t jl = idx_cp_line(feat_off, c.res, c.F, c.pg[dim], dim);
								uint32_t jr = idx_cp_line(feat_off, c.res, c.F, c.pg[dim] + 1, dim);
								float nw = c.pos[dim];
								for (uint32_t f = 0; f < NFT; ++f)
									gl[f] *= fmaf(nw, c.grid[jr + f], (1.0f - nw) * c.grid[jl + f]);
							}
							uint32_t il = idx_cp_line(feat_off, c.res, c.F, c.pg[gd], gd);
							uint32_t ir = idx_cp_line(feat_off, c.res, c.F, c.pg[gd] + 1, gd);
							for (uint32_t f = 0; f < NFT; ++f) acc_add(a, il + f, gl[f] * (1.0f - c.pos[gd]));
							for (uint32_t f = 0; f < NFT; ++f) acc_add(a, ir + f, gl[f] * c.pos[gd]);
						}
					} else {                                                     /* bwd_n_linear */
						for (uint32_t idx = 0; idx < (1u << D); ++idx) {
							float w = 1.0f;
							for (uint32_t d = 0; d < D; ++d) {
								if ((idx & (1u << d)) == 0) { w *= 1.0f - c.pos[d]; lp[d] = c.pg[d]; }
								else { w *= c.pos[d]; lp[d] = c.pg[d] + 1; }
							}
							add_grad(&c, lp, feat_off, NFT, grad, w, a);
						}
					}
				} else { /* ---------------- second order: d(dL/dx)/dparam ----------------
				          * kernel_lod_backward_input_backward_grid, lotd_encoding.h:764-1041,
				          * bwd_input_bwd_grid_n_linear :713-762 */
					if (c.type == ORC_NPlaneSum) {                               /* :903-968 */
						if (D > 2)
							for (uint32_t jd = 0; jd < D; ++jd)
								for (uint32_t g2 = 0; g2 < D - 1; ++g2) {
									const uint32_t g3 = g2 >= jd ? g2 + 1 : g2;
									const float grad_in = c.scale[g3] * 1.0f * gin[g3] * c.dpos[g3];
									for (uint32_t idx = 0; idx < (1u << (D - 2)); ++idx) {
										float w = grad_in;
										uint32_t pp[ORC_MAX_DIMS];
										for (uint32_t ng = 0; ng + 2 < D; ++ng) {
											const uint32_t d2 = ng >= g2 ? ng + 1 : ng;
											const uint32_t d3 = d2 >= jd ? d2 + 1 : d2;
											if ((idx & (1u << ng)) == 0) { w *= 1.0f - c.pos[d3]; pp[d2] = c.pg[d3]; }
											else { w *= c.pos[d3]; pp[d2] = c.pg[d3] + 1; }
										}
										pp[g2] = c.pg[g3];
										uint32_t il = idx_nplane_sub(D, feat_off, c.res, c.F, pp, jd);
										for (uint32_t f = 0; f < NFT; ++f) acc_add(a, il + f, grad[f] * -w);
										pp[g2] = c.pg[g3] + 1;
										uint32_t ir = idx_nplane_sub(D, feat_off, c.res, c.F, pp, jd);
										for (uint32_t f = 0; f < NFT; ++f) acc_add(a, ir + f, grad[f] * w);
									}
								}
					} else if (c.type == ORC_CPfast) {                           /* :970-1038 */
						for (uint32_t ld = 0; ld < D; ++ld)
							for (uint32_t gd = 0; gd < D; ++gd) {
								float gl[2];
								for (uint32_t f = 0; f < NFT; ++f)
									gl[f] = grad[f] * c.scale[gd] * gin[gd] * c.dpos[gd];
								float w_l = -1.0f, w_r = 1.0f;
								if (ld != gd) { w_l = 1.0f - c.pos[ld]; w_r = c.pos[ld]; }
								for (uint32_t ol = 0; ol < D - 1; ++ol) {
									const uint32_t dim = ol >= ld ? ol + 1 : ol;
									uint32_t jl = idx_cp_line(feat_off, c.res, c.F, c.pg[dim], dim);
									uint32_t jr = idx_cp_line(feat_off, c.res, c.F, c.pg[dim] + 1, dim);
									float nl = -1.0f, nr = 1.0f;
									if (dim != gd) { nl = 1.0f - c.pos[dim]; nr = c.pos[dim]; }
									/* reference loops f < N_FEAT_PER_PSEUDO_LVL over an N_FEAT_PER_THREAD array
									 * (:1030) -- out of bounds when gcd > 2; restated with the array's real extent */
									for (uint32_t f = 0; f < NFT; ++f)
										gl[f] *= fmaf(nr, c.grid[jr + f], nl * c.grid[jl + f]);
								}
								uint32_t il = idx_cp_line(feat_off, c.res, c.F, c.pg[ld], ld);
								uint32_t ir = idx_cp_line(feat_off, c.res, c.F, c.pg[ld] + 1, ld);
								for (uint32_t f = 0; f < NFT; ++f) acc_add(a, il + f, gl[f] * w_l);
								for (uint32_t f = 0; f < NFT; ++f) acc_add(a, ir + f, gl[f] * w_r);
							}
					} else {
						for (uint32_t gd = 0; gd < D; ++gd) {
							const float grad_in = c.scale[gd] * gin[gd] * c.dpos[gd];
							for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
								float w = grad_in;
								for (uint32_t ng = 0; ng < D - 1; ++ng) {
									const uint32_t dim = ng >= gd ? ng + 1 : ng;
									if ((idx & (1u << ng)) == 0) { w *= 1.0f - c.pos[dim]; lp[dim] = c.pg[dim]; }
									else { w *= c.pos[dim]; lp[dim] = c.pg[dim] + 1; }
								}
								lp[gd] = c.pg[gd];
								add_grad(&c, lp, feat_off, NFT, grad, -w, a);
								lp[gd] = c.pg[gd] + 1;
								add_grad(&c, lp, feat_off, NFT, grad, w, a);
							}
						}
					}
				}
			}
		}
	}
}

int orc_set_num_threads(int n) {
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
	return omp_get_max_threads();
#else
	(void)n;
	return 1;
#endif
}

/* Slabs of points per level for the fp64-accumulated mode (the sums are exact to ~1e-16 relative, so splitting the
 * points over S private accumulators and adding those in slab order does not change what the checker sees): without
 * them only n_levels host threads ever work on dL/dparam.  ORC_DPARAM_SLABS overrides (1 = one slab). */
static int dparam_slabs(uint32_t N, uint32_t n_levels) {
	const char *e = getenv("ORC_DPARAM_SLABS");
	int s = 1;
	if (e) s = atoi(e);
	else if (N >= (1u << 15)) {
#ifdef _OPENMP
		s = omp_get_max_threads() / (int)(n_levels ? n_levels : 1u);
#endif
	}
	return s < 1 ? 1 : (s > 16 ? 16 : s);
}

static void bwd_dparam_all(const orc_lotd_meta_t *m, uint32_t N, const float *dL_ddLdx, const float *dL_dy,
                           const float *x, const float *params, const int64_t *batch_inds,
                           const int64_t *batch_offsets, uint32_t batch_data_size, int32_t max_level,
                           int accum_double, float *grad, uint64_t numel) {
	if (max_level <= -1) return;
	double *gd = NULL;
	if (accum_double == 1) gd = (double *)calloc(numel, sizeof(double));
	acc_t a; a.g = gd ? NULL : grad; a.gd = gd;
	/* accum_double == 2: float accumulators AND slabs (the CPU baseline's mode: every host thread works, the float sums
	 * are associated per slab instead of over all points in order) */
	const int S = (gd || accum_double == 2) ? dparam_slabs(N, m->n_levels) : 1;
	if (S > 1) {
		/* private accumulators of slabs 1..S-1: kept between calls and left all-zero by the reduction below */
		static void *priv = NULL;
		static uint64_t priv_cap = 0;
		const uint64_t esz = gd ? 8 : 4, need = (uint64_t)(S - 1) * numel * esz;
		if (need > priv_cap) { free(priv); priv = calloc(need, 1); priv_cap = priv ? need : 0; }
		if (priv) {
			double *pd = (double *)priv;
			float *pf = (float *)priv;
			const int64_t tasks = (int64_t)m->n_levels * S;
#pragma omp parallel for schedule(dynamic, 1)
			for (int64_t t = 0; t < tasks; ++t) {
				const uint32_t l = (uint32_t)(t / S), sl = (uint32_t)(t % S);
				acc_t as;
				as.gd = gd ? (sl ? pd + (uint64_t)(sl - 1) * numel : gd) : NULL;
				as.g = gd ? NULL : (sl ? pf + (uint64_t)(sl - 1) * numel : grad);
				const uint32_t i0 = (uint32_t)((uint64_t)N * sl / S), i1 = (uint32_t)((uint64_t)N * (sl + 1) / S);
				bwd_dparam_level(m, l, i0, i1, dL_ddLdx, dL_dy, x, params, batch_inds, batch_offsets, batch_data_size,
				                 max_level, as);
			}
#pragma omp parallel for schedule(static)
			for (int64_t k = 0; k < (int64_t)numel; ++k) {
				if (gd) {
					double acc = gd[k];
					for (int sl = 1; sl < S; ++sl) { acc += pd[(uint64_t)(sl - 1) * numel + k]; pd[(uint64_t)(sl - 1) * numel + k] = 0.0; }
					grad[k] += (float)acc;
				} else {
					float acc = grad[k];
					for (int sl = 1; sl < S; ++sl) { acc += pf[(uint64_t)(sl - 1) * numel + k]; pf[(uint64_t)(sl - 1) * numel + k] = 0.0f; }
					grad[k] = acc;
				}
			}
			free(gd);
			return;
		}
	}
	/* different levels write disjoint slices of every batch copy -> safe to run levels in parallel */
#pragma omp parallel for schedule(dynamic, 1)
	for (int32_t l = 0; l < (int32_t)m->n_levels; ++l)
		bwd_dparam_level(m, (uint32_t)l, 0, N, dL_ddLdx, dL_dy, x, params, batch_inds, batch_offsets,
		                 batch_data_size, max_level, a);
	if (gd) {
		for (uint64_t k = 0; k < numel; ++k) grad[k] += (float)gd[k];
		free(gd);
	}
}

void orc_lotd_bwd_dparam(const orc_lotd_meta_t *m, uint32_t N, const float *dL_dy, const float *x,
                         const float *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                         uint32_t batch_data_size, int32_t max_level, int accum_double, float *grad,
                         uint64_t numel) {
	bwd_dparam_all(m, N, NULL, dL_dy, x, params, batch_inds, batch_offsets, batch_data_size, max_level,
	               accum_double, grad, numel);
}

void orc_lotd_bwd_bwd_dparam(const orc_lotd_meta_t *m, uint32_t N, const float *dL_ddLdx,
                             const float *dL_dy, const float *x, const float *params,
                             const int64_t *batch_inds, const int64_t *batch_offsets,
                             uint32_t batch_data_size, int32_t max_level, int accum_double,
                             float *grad, uint64_t numel) {
	bwd_dparam_all(m, N, dL_ddLdx, dL_dy, x, params, batch_inds, batch_offsets, batch_data_size,
	               max_level, accum_double, grad, numel);
}

/* dL/dx = sum_j dL_dy[i,j] * dy_dx[i,j,:]   (ATen mul + sum, lotd_encoding.h:1562-1586) */
void orc_lotd_bwd_dx(const orc_lotd_meta_t *m, uint32_t N, const float *dL_dy, const float *dy_dx,
                     float *dL_dx) {
	const uint32_t D = m->n_dims_to_encode, E = m->n_encoded_dims;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)N; ++i)
		for (uint32_t d = 0; d < D; ++d) {
			float s = 0.f;
			for (uint32_t j = 0; j < E; ++j) s += dL_dy[(size_t)i * E + j] * dy_dx[((size_t)i * E + j) * D + d];
			dL_dx[(size_t)i * D + d] = s;
		}
}

/* dL/d(dL/dy) = sum_d dL_ddLdx[i,d] * dy_dx[i,j,d]   (lotd_encoding.h:1703-1727) */
void orc_lotd_bwd_bwd_ddLdy(const orc_lotd_meta_t *m, uint32_t N, const float *dL_ddLdx,
                            const float *dy_dx, float *dL_ddLdy) {
	const uint32_t D = m->n_dims_to_encode, E = m->n_encoded_dims;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t)N; ++i)
		for (uint32_t j = 0; j < E; ++j) {
			float s = 0.f;
			for (uint32_t d = 0; d < D; ++d) s += dL_ddLdx[(size_t)i * D + d] * dy_dx[((size_t)i * E + j) * D + d];
			dL_ddLdy[(size_t)i * E + j] = s;
		}
}

/* d(dL/dx)/dx  (kernel_lod_backward_input_backward_input, lotd_encoding.h:1157-1298;
 * bwd_input_bwd_input_n_linear :1043-1155).  Only Dense / VM / VecZMatXoY / Hash contribute. */
void orc_lotd_bwd_bwd_dx(const orc_lotd_meta_t *m, uint32_t N, const float *dL_ddLdx,
                         const float *dL_dy, const float *x, const float *params,
                         const int64_t *batch_inds, const int64_t *batch_offsets,
                         uint32_t batch_data_size, int32_t max_level, float *dL_dx) {
	const uint32_t D = m->n_dims_to_encode, E = m->n_encoded_dims, G = m->n_feat_per_pseudo_lvl;
	const uint32_t NFT = 2;
	if (max_level <= -1) return;
#pragma omp parallel for schedule(static)
	for (int64_t ii = 0; ii < (int64_t)N; ++ii) {
		const uint32_t i = (uint32_t)ii;
		for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
			const uint32_t level = m->map_levels[q];
			ctx_t c;
			if (!setup_ctx(&c, m, level, i, x, params, batch_inds, batch_offsets, batch_data_size, max_level))
				continue;
			if (!(c.type == ORC_Dense || c.type == ORC_Hash || c.type == ORC_VectorMatrix ||
			      c.type == ORC_VecZMatXoY)) continue;
			for (uint32_t feature = 0; feature < G; feature += NFT) {
				const uint32_t feat_off = m->map_cnt[q] * G + feature;
				const uint32_t out_off = q * G + feature;
				const float *grad = dL_dy + (size_t)i * E + out_off;
				const float *gin = dL_ddLdx + (size_t)i * D;
				float diag[ORC_MAX_DIMS], other[ORC_MAX_DIMS];
				if (c.smooth)
					for (uint32_t gd = 0; gd < D; ++gd)
						diag[gd] = (c.scale[gd] * gin[gd]) * (c.scale[gd] * c.ddpos[gd]);
				for (uint32_t gd = 0; gd < D; ++gd) other[gd] = c.scale[gd] * gin[gd] * c.dpos[gd];
				uint32_t lp[ORC_MAX_DIMS];
				for (uint32_t gd = 0; gd < D; ++gd) {
					float out = 0.f;
					for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
						if (c.smooth) {
							float w = diag[gd];
							for (uint32_t ng = 0; ng < D - 1; ++ng) {
								const uint32_t dim = ng >= gd ? ng + 1 : ng;
								if ((idx & (1u << ng)) == 0) { w *= 1.0f - c.pos[dim]; lp[dim] = c.pg[dim]; }
								else { w *= c.pos[dim]; lp[dim] = c.pg[dim] + 1; }
							}
							lp[gd] = c.pg[gd];
							out += calc_dLdx(&c, lp, feat_off, NFT, grad, -w);
							lp[gd] = c.pg[gd] + 1;
							out += calc_dLdx(&c, lp, feat_off, NFT, grad, w);
						}
						for (uint32_t og = 0; og < D - 1; ++og) {
							const uint32_t rog = og >= gd ? og + 1 : og;
							float w = other[rog] * (c.dpos[gd] * c.scale[gd]);
							for (uint32_t ng = 0; ng < D - 1; ++ng) {
								const uint32_t dim = ng >= rog ? ng + 1 : ng;
								if ((idx & (1u << ng)) == 0) {
									if (dim != gd) w *= 1.0f - c.pos[dim]; else w *= -1.0f;
									lp[dim] = c.pg[dim];
								} else {
									if (dim != gd) w *= c.pos[dim];
									lp[dim] = c.pg[dim] + 1;
								}
							}
							lp[rog] = c.pg[rog];
							out += calc_dLdx(&c, lp, feat_off, NFT, grad, -w);
							lp[rog] = c.pg[rog] + 1;
							out += calc_dLdx(&c, lp, feat_off, NFT, grad, w);
						}
					}
					dL_dx[(size_t)i * D + gd] += out;      /* atomicAdd across threads, :1288-1291 */
				}
			}
		}
	}
}

/* lod_get_grid_index  (kernel lotd_encoding.h:1300-1433; host lotd_torch_api.cu:771-843) */
int orc_lotd_grid_index(const orc_lotd_meta_t *m, uint32_t N, const float *x,
                        const int64_t *batch_inds, const int64_t *batch_offsets,
                        uint32_t batch_data_size, int32_t max_level, int64_t *grid_inds) {
	const uint32_t D = m->n_dims_to_encode, E = m->n_encoded_dims, G = m->n_feat_per_pseudo_lvl;
	const uint32_t C = 1u << D;
	for (uint32_t l = 0; l < m->n_levels; ++l)
		if (!(m->level_types[l] == ORC_Dense || m->level_types[l] == ORC_Hash)) return 1;
	if (max_level <= -1) return 0;
	for (uint32_t i = 0; i < N; ++i)
		for (uint32_t q = 0; q < m->n_pseudo_levels; ++q) {
			const uint32_t level = m->map_levels[q];
			const uint32_t feat_off = m->map_cnt[q] * G;
			const uint32_t out_off = q * G;
			ctx_t c;
			if (!setup_ctx(&c, m, level, i, x, NULL, batch_inds, batch_offsets, batch_data_size, max_level))
				continue;
			int64_t *out = grid_inds + ((size_t)i * E + out_off) * C;
			for (uint32_t idx = 0; idx < C; ++idx) {
				uint32_t lp[ORC_MAX_DIMS];
				for (uint32_t d = 0; d < D; ++d) lp[d] = c.pg[d] + (((idx >> d) & 1u) ? 1u : 0u);
				uint32_t ind = (c.type == ORC_Dense) ? idx_dense(D, feat_off, c.res, c.F, lp)
				                                     : idx_hash(D, feat_off, c.size, c.F, lp);
				for (uint32_t f = 0; f < G; ++f) out[idx + f * C] = (int64_t)(uint32_t)(c.base + ind + f);
			}
		}
	return 0;
}

/* ------------------------------------------------------------------------------------------------
 * LoTD forest  (csrc/lotd/include/lotd/lotd_forest.h): the same N-linear bodies with scale = res, the corner
 * remap of forest_resolve and per-point block_inds / block_offsets in place of the batch arguments.
 * The reference's forest kernels switch over Dense / VectorMatrix / NPlaneMul / CP / Hash only (:263-311);
 * other level types return 1 here.
 * ---------------------------------------------------------------------------------------------- */
static int forest_types_ok(const orc_lotd_meta_t *m) {
	if (m->n_dims_to_encode != 3) return 0;
	for (uint32_t l = 0; l < m->n_levels; ++l) {
		const uint32_t t = m->level_types[l];
		if (!(t == ORC_Dense || t == ORC_VectorMatrix || t == ORC_NPlaneMul || t == ORC_CP || t == ORC_Hash)) return 0;
	}
	return 1;
}

int orc_lotd_forest_fwd(const orc_lotd_meta_t *m, const orc_forest_t *fo, uint32_t N, const float *x,
                        const float *params, const int64_t *block_inds, const int64_t *block_offsets,
                        uint32_t batch_data_size, int32_t max_level, float *y, float *dy_dx) {
	if (!forest_types_ok(m)) return 1;
	g_forest = fo;
	orc_lotd_fwd(m, N, x, params, block_inds, block_offsets, batch_data_size, max_level, y, dy_dx);
	g_forest = NULL;
	return 0;
}

int orc_lotd_forest_bwd_dparam(const orc_lotd_meta_t *m, const orc_forest_t *fo, uint32_t N, const float *dL_ddLdx,
                               const float *dL_dy, const float *x, const float *params, const int64_t *block_inds,
                               const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level,
                               int accum_double, float *grad, uint64_t numel) {
	if (!forest_types_ok(m)) return 1;
	if (max_level <= -1) return 0;
	g_forest = fo;
	/* a corner may belong to a neighbouring block, but always to the SAME level: levels stay disjoint, so the
	 * per-level parallel loop of bwd_dparam_all remains race free */
	bwd_dparam_all(m, N, dL_ddLdx, dL_dy, x, params, block_inds, block_offsets, batch_data_size, max_level,
	               accum_double, grad, numel);
	g_forest = NULL;
	return 0;
}

int orc_lotd_forest_bwd_bwd_dx(const orc_lotd_meta_t *m, const orc_forest_t *fo, uint32_t N, const float *dL_ddLdx,
                               const float *dL_dy, const float *x, const float *params, const int64_t *block_inds,
                               const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level,
                               float *dL_dx) {
	if (!forest_types_ok(m)) return 1;
	g_forest = fo;          /* kernel_lod_forest_backward_input_backward_input: Dense / VM / Hash only (:1029-1057) */
	orc_lotd_bwd_bwd_dx(m, N, dL_ddLdx, dL_dy, x, params, block_inds, block_offsets, batch_data_size, max_level, dL_dx);
	g_forest = NULL;
	return 0;
}
