/* oracle/lotd_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, fp32) of the reference LoTD encoder math, used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.  Nothing under
 * nr3d_lib_amd/ may import, link or call this.
 *
 * Struct mirrors the reference's device-side meta `LoDMetaRef`
 * (csrc/lotd/include/lotd/lotd_cuda.h:29-76).
 */
#ifndef LOTD_ORACLE_H
#define LOTD_ORACLE_H
#include <stdint.h>

#define ORC_MAX_LEVELS 32
#define ORC_MAX_DIMS 4
#define ORC_MAX_PSEUDO (ORC_MAX_LEVELS * 8)

/* csrc/lotd/include/lotd/lotd_types.h:16-25 */
enum {
	ORC_Dense = 0, ORC_VectorMatrix = 1, ORC_VecZMatXoY = 2, ORC_CP = 3,
	ORC_CPfast = 4, ORC_NPlaneMul = 5, ORC_NPlaneSum = 6, ORC_Hash = 7
};

typedef struct {
	uint32_t level_res[ORC_MAX_LEVELS][ORC_MAX_DIMS];
	uint32_t level_n_feats[ORC_MAX_LEVELS];
	uint32_t level_types[ORC_MAX_LEVELS];
	uint32_t level_n_params[ORC_MAX_LEVELS];
	uint32_t level_offsets[ORC_MAX_LEVELS + 1];
	uint32_t level_sizes[ORC_MAX_LEVELS];
	uint32_t map_levels[ORC_MAX_PSEUDO];
	uint32_t map_cnt[ORC_MAX_PSEUDO];
	uint32_t n_levels;
	uint32_t n_pseudo_levels;
	uint32_t n_feat_per_pseudo_lvl;
	uint32_t n_dims_to_encode;
	uint32_t n_encoded_dims;
	uint32_t n_params;
	uint32_t interpolation_type; /* 0 linear, 1 smoothstep */
	uint32_t reserved;
} orc_lotd_meta_t;

/* csrc/forest/forest.h:60-97 (ForestMetaRef); block_ks is int16 [n_trees, 3] */
typedef struct {
	const uint8_t *octree;
	const int32_t *exsum;
	const int16_t *block_ks;
	uint32_t n_trees, level, level_poffset;
	int32_t continuity_enabled;
} orc_forest_t;

/* returns 0 on success, nonzero + message in errbuf otherwise */
int orc_lotd_create_meta(int32_t n_input_dim, uint32_t n_levels, const int32_t *res_multidim /*[L,D]*/,
                         const int32_t *n_feats, const int32_t *types, uint32_t hashmap_size,
                         int use_smooth_step, orc_lotd_meta_t *out, char *errbuf, int errlen);

void orc_lotd_fwd(const orc_lotd_meta_t *m, uint32_t N, const float *x, const float *params,
                  const int64_t *batch_inds, const int64_t *batch_offsets, uint32_t batch_data_size,
                  int32_t max_level, float *y /*[N,E]*/, float *dy_dx /*[N,E,D] or NULL*/);

void orc_lotd_bwd_dparam(const orc_lotd_meta_t *m, uint32_t N, const float *dL_dy, const float *x,
                         const float *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                         uint32_t batch_data_size, int32_t max_level, int accum_double,
                         float *grad /*[numel params], zero-initialised by caller*/, uint64_t numel);

/* host threads of the OpenMP loops in this library (<= 0: leave as is); returns the count in effect */
int orc_set_num_threads(int n);

void orc_lotd_bwd_dx(const orc_lotd_meta_t *m, uint32_t N, const float *dL_dy, const float *dy_dx,
                     float *dL_dx /*[N,D]*/);

void orc_lotd_bwd_bwd_ddLdy(const orc_lotd_meta_t *m, uint32_t N, const float *dL_ddLdx,
                            const float *dy_dx, float *dL_ddLdy /*[N,E]*/);

void orc_lotd_bwd_bwd_dparam(const orc_lotd_meta_t *m, uint32_t N, const float *dL_ddLdx,
                             const float *dL_dy, const float *x, const float *params,
                             const int64_t *batch_inds, const int64_t *batch_offsets,
                             uint32_t batch_data_size, int32_t max_level, int accum_double,
                             float *grad, uint64_t numel);

void orc_lotd_bwd_bwd_dx(const orc_lotd_meta_t *m, uint32_t N, const float *dL_ddLdx,
                         const float *dL_dy, const float *x, const float *params,
                         const int64_t *batch_inds, const int64_t *batch_offsets,
                         uint32_t batch_data_size, int32_t max_level, float *dL_dx /*[N,D] zero-init*/);

int orc_lotd_grid_index(const orc_lotd_meta_t *m, uint32_t N, const float *x,
                        const int64_t *batch_inds, const int64_t *batch_offsets,
                        uint32_t batch_data_size, int32_t max_level,
                        int64_t *grid_inds /*[N,E,2^D] zero-init*/);

int32_t orc_forest_identify(const orc_forest_t *fo, const int16_t *k /*[3]*/);
int orc_lotd_forest_fwd(const orc_lotd_meta_t *m, const orc_forest_t *fo, uint32_t N, const float *x,
                        const float *params, const int64_t *block_inds, const int64_t *block_offsets,
                        uint32_t batch_data_size, int32_t max_level, float *y, float *dy_dx);
/* dL_ddLdx == NULL: first order; else d(dL/dx)/dparam */
int orc_lotd_forest_bwd_dparam(const orc_lotd_meta_t *m, const orc_forest_t *fo, uint32_t N, const float *dL_ddLdx,
                               const float *dL_dy, const float *x, const float *params, const int64_t *block_inds,
                               const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level,
                               int accum_double, float *grad, uint64_t numel);
int orc_lotd_forest_bwd_bwd_dx(const orc_lotd_meta_t *m, const orc_forest_t *fo, uint32_t N, const float *dL_ddLdx,
                               const float *dL_dy, const float *x, const float *params, const int64_t *block_inds,
                               const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level,
                               float *dL_dx);
#endif
