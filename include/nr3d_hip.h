/* include/nr3d_hip.h -- C ABI of libnr3d_hip.so (MI355X / gfx950 kernels for the nr3d hot path).
 *
 * This is the drop-in boundary: every entry point takes plain device/host pointers, sizes, element
 * strides and a hipStream_t (as void*); no torch / ATen types.  Each group cites the reference
 * interface it replaces (paths relative to the reference checkout).  All functions return 0 on
 * success, nonzero on failure with a message available from nr3d_last_error() (thread-local).
 *
 * Ownership: the CALLER allocates every buffer (inputs and outputs) on the device the stream
 * belongs to; the library holds no state besides the thread-local error string and the option table below.  Buffers documented
 * "zero-init" must be zeroed by the caller; all other outputs are fully written by the kernels
 * (skipped points are written as zeros), so they may be allocated uninitialised.
 *
 * dtype codes (NR3D_*) name the element type behind a void*.
 */
#ifndef NR3D_HIP_H
#define NR3D_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { NR3D_F32 = 0, NR3D_F16 = 1, NR3D_F64 = 2, NR3D_I32 = 3, NR3D_I64 = 4, NR3D_U8 = 5, NR3D_I16 = 6, NR3D_I8 = 7 };

/* Bumped whenever an entry point is added, removed or changes its parameters.  nr3d_lib_amd/_abi.py (generated from this header by
 * tools/gen_abi.py at build time) carries the same number next to every entry point's argument types; the Python loader refuses a
 * library whose nr3d_abi_version() differs, so a vendored nr3d_lib_amd/ needs this header neither at import nor at run time. */
#define NR3D_ABI_VERSION 9

const char *nr3d_last_error(void);
int nr3d_abi_version(void);

/* Optional per-kernel timing with HIP events on the launch stream (what the reference's c_profile flag does for whole
 * calls: csrc/lotd/include/lotd/lotd_hash_only.h:748-760, LoDMeta::c_profile lotd_torch_api.h:102-110).
 * nr3d_prof_enable(mask): bit id set => every launch of that kernel is bracketed by an event pair (off: no cost).
 * nr3d_prof_read: sum of the recorded intervals of `id` in ms and their number (synchronises on them); reset != 0
 * forgets them.  Used by bench.py for `roofline.achieved` of the dominant kernel. */
enum {
	NR3D_PROF_LOTD_FWD = 0,          /* k_fwd: forward of all levels served from L2 */
	NR3D_PROF_LOTD_FWD_LDS = 1,      /* k_fwd_lds: forward of the levels staged in LDS (one interval per level) */
	NR3D_PROF_LOTD_CONTRACT_DX = 2,  /* dL/dx = dL/dy . dy/dx */
	NR3D_PROF_LOTD_BIN = 3,          /* dL/dparam stage A */
	NR3D_PROF_LOTD_ACCUM = 4,        /* dL/dparam stage B */
	NR3D_PROF_MARCH = 5,             /* ray marching, count + emit */
	NR3D_PROF_COMPOSITE_FWD = 6,     /* fused alpha composite */
	NR3D_PROF_COMPOSITE_BWD = 7,
	NR3D_PROF_LOTD_DIRECT = 8,       /* dL/dparam of the levels that skip the records (k_pair_direct) */
	NR3D_PROF_COUNT = 9
};
void nr3d_prof_enable(uint32_t mask);
int nr3d_prof_read(int id, double *total_ms, uint32_t *n_intervals, int reset);

/* Selectable code paths (round 4: ONE table instead of environment switches; no launch reads the environment).
 * Every entry chooses between two implementations of the SAME result -- the parity tests run both and compare them with
 * each other and with the oracle -- so a caller never needs them; they exist for A/B measurement and cross-checks.
 * nr3d_set_option(id, value): value < 0 restores the default; returns nonzero for an unknown id.  nr3d_get_option: current value,
 * -1 if unknown.
 * TEST / MEASUREMENT ONLY -- not part of the drop-in surface.  The table is process-wide on purpose (an option set on the Python main
 * thread has to reach the launches PyTorch's autograd engine issues from ITS device thread, which a thread-local table would not);
 * entries are relaxed atomics, so a concurrent set/launch is not a data race, but two host threads that A/B DIFFERENT values at the
 * same time see each other's choice -- results stay correct (both implementations give the same result), measurements do not.
 * Production callers never set one: every default is the measured-fastest path.
 * Measurement knobs and timing experiments are NOT options: they exist only in a -DNR3D_EXPERIMENTS build. */
enum {
	NR3D_OPT_LOTD_PAIR = 0,          /* 1: pair / quad records for dL/dparam of 3-D Dense/Hash metas (lotd_pair.hip); 0: corner records */
	NR3D_OPT_PAIR_QUAD = 1,          /* 1: Dense levels of the pair path as quad records */
	NR3D_OPT_PAIR_SECOND = 2,        /* 1: d(dL/dx)/dparam of pair-path metas on pair records */
	NR3D_OPT_PAIR_DIRECT = 3,        /* 1: levels with <= 4 buckets skip the records (k_pair_direct) */
	NR3D_OPT_PAIR_FIXED = 4,         /* 1: 64-bit fixed-point LDS accumulators; 0: fp64 */
	NR3D_OPT_FWD_PAIRLANE = 5,       /* 1: two-lane forward / Hessian kernels for 3-D Dense/Hash metas; 0: k_fwd (corner sum) */
	NR3D_OPT_FWD_SPLIT = 6,          /* 1: mixed metas launch per level type */
	NR3D_OPT_FWD_LDS_STAGE = 7,      /* 1: coarse Dense levels whose whole table fits LDS are served from it (k_fwd_lds); 2: also tables that fit in <= 8 slabs of
	                                  * x-planes, from 2^19 points on (k_fwd_lds_slab: round-5 experiment, bit-identical, slower: off); 0: none */
	NR3D_OPT_HVP_LEVELS = 8,         /* 1: d(dL/dx)/dx with one lane per (point, level) when a workspace is given */
	NR3D_OPT_HVP_PAIRLANE = 9,       /* 1: ... through the two-lane gather */
	NR3D_OPT_HVP_SPLIT = 10,         /* 1: lane-serial d(dL/dx)/dx launches per level type */
	NR3D_OPT_VM_SPLIT = 11,          /* 1: stage A of VM levels with three threads per point */
	NR3D_OPT_CP_DIRECT = 12,         /* 1: CP levels' dL/dparam accumulated in LDS without records */
	NR3D_OPT_MARCH_GROUP = 13,       /* 0: lanes per ray chosen from the ray count; 1 | 16 | 32 | 64 forces */
	NR3D_OPT_PACK_SCAN = 14,         /* 1: fused composite on wave prefix products; 0: serial replay */
	NR3D_OPT_VM_LINES_DIRECT = 15,   /* 1: VM line-table gradients accumulated in LDS inside stage A, plane updates as records only (default 0: measured
	                                  * slower -- the fp64 LDS atomics cost stage A what the smaller records save stage B) */
	NR3D_OPT_FWD_CELL_MAJOR = 16,    /* 1: forward reads a cell-major replica of the mid Dense levels when the caller supplies one */
	NR3D_OPT_SORT_WAVE = 17,         /* 1: packed_sort with one wave per pack (bitonic); 0: one lane per pack (heapsort) */
	NR3D_OPT_VM_DIRECT = 18,         /* 1: VM levels whose planes split into <= 4 LDS-sized bands accumulate their dL/dparam in LDS without records (k_vm_direct) */
	NR3D_OPT_DIRECT_FIXED = 19,      /* 1: k_cp_direct and k_vm_sorted accumulate in 64-bit fixed point (scale from the workgroup's own bound on its updates); 2: k_vm_direct
	                                  * too (measured slower there, twice: it is not bound by its LDS atomics); 0: fp64 */
	NR3D_OPT_VM_SORTED = 20,         /* 1: a dL/dparam pass with a VM level of >= 2^20 entries (over its blocks) and >= 2^19 points -- or any VM level and >= 2^21 points -- sorts the POINTS by
	                                  * (block, coordinate) and accumulates every VM level band by band in LDS, without records (lotd_sorted.hip;
	                                  * single tables, batches and forests; a forest's small Dense levels ride along as slices); 2: whenever the geometry allows (tests); 3: as 1, VM levels only; 0: records */
	NR3D_OPT_MLP_X3 = 21,            /* 1: the fp32 fused MLP forward runs on the bf16 MFMA with every value split into three bf16 pieces (six piece products,
	                                  * fp32 accumulation: fp32-grade results at 2.7x the matrix rate of the f32 MFMA); 0: v_mfma_f32_32x32x2_f32.
	                                  * Non-finite inputs: a row holding +-inf (or a magnitude above the bf16 maximum, 3.39e38) comes out as NaN on the
	                                  * x3 route (inf - bf16(inf) = NaN in the split) where the f32 MFMA gives +-inf or NaN (inf * 0); finite rows of the
	                                  * same batch are unaffected on both.  A ReLU pre-activation within ~1 ulp of zero may be masked differently by a
	                                  * forward on one route and a backward recomputation on the other (the gradient of that unit at that sample only). */
	NR3D_OPT_COUNT = 22
};
int nr3d_set_option(int id, int64_t value);
int64_t nr3d_get_option(int id);

/* =================================================================================================
 * LoTD encoder -- replaces nr3d_lib.bindings._lotd
 *   pybind surface   csrc/lotd/src/lotd.cpp:23-110
 *   host API         csrc/lotd/include/lotd/lotd_torch_api.h:79-249, csrc/lotd/src/lotd_torch_api.cu
 *   device meta      csrc/lotd/include/lotd/lotd_cuda.h:29-76 (LoDMetaRef)
 * ============================================================================================== */
#define NR3D_LOTD_MAX_LEVELS 32
#define NR3D_LOTD_MAX_DIMS 4
#define NR3D_LOTD_MAX_PSEUDO 256

/* csrc/lotd/include/lotd/lotd_types.h:16-25 */
enum {
	NR3D_LOD_Dense = 0, NR3D_LOD_VectorMatrix = 1, NR3D_LOD_VecZMatXoY = 2, NR3D_LOD_CP = 3,
	NR3D_LOD_CPfast = 4, NR3D_LOD_NPlaneMul = 5, NR3D_LOD_NPlaneSum = 6, NR3D_LOD_Hash = 7
};

typedef struct nr3d_lotd_level {
	uint32_t res[NR3D_LOTD_MAX_DIMS]; /* level_res_multidim */
	uint32_t n_feats;                 /* level_n_feats  */
	uint32_t type;                    /* level_types    */
	uint32_t size;                    /* level_sizes    (entries, feature width not counted) */
	uint32_t offset;                  /* level_offsets  (elements, inside one batch entry)   */
} nr3d_lotd_level_t;                  /* 32 B: one level per half cache line */

typedef struct nr3d_lotd_meta {
	nr3d_lotd_level_t levels[NR3D_LOTD_MAX_LEVELS];
	uint16_t map_levels[NR3D_LOTD_MAX_PSEUDO];
	uint16_t map_cnt[NR3D_LOTD_MAX_PSEUDO];
	uint32_t n_levels;
	uint32_t n_pseudo_levels;
	uint32_t n_feat_per_pseudo_lvl;
	uint32_t n_dims_to_encode;
	uint32_t n_encoded_dims;
	uint32_t n_params;               /* == level_offsets[n_levels] */
	uint32_t interpolation_type;     /* 0 Linear, 1 Smoothstep (lotd_types.h:78-82) */
	uint32_t c_hash_only;            /* every level is Dense or Hash */
	/* ABI 2 (appended; everything above is unchanged): first output column of pseudo level q.  q * n_feat_per_pseudo_lvl
	 * for the meta nr3d_lotd_meta_create builds; a REGROUPED meta (nr3d_lotd_meta_regroup) lists a subset of the levels with
	 * a wider pseudo level, and its columns are those of the original layout. */
	uint16_t map_col[NR3D_LOTD_MAX_PSEUDO];
} nr3d_lotd_meta_t;

/* LoDMeta::create_meta (lotd_torch_api.cu:29-230).  Host only, no GPU needed.
 * res_multidim is [n_levels, n_input_dim] row-major; types are NR3D_LOD_* codes. */
int nr3d_lotd_meta_create(int32_t n_input_dim, uint32_t n_levels, const int32_t *res_multidim,
                          const int32_t *n_feats, const int32_t *types, uint32_t hashmap_size,
                          int use_smooth_step, nr3d_lotd_meta_t *out);

/* Pseudo levels regrouped by feature width (no reference counterpart: the reference processes every level in pseudo levels
 * of the GLOBAL gcd of the widths, lotd_torch_api.cu:58-75 -- a meta that mixes 2- and 16-feature levels walks the
 * 16-feature one as eight 2-feature pseudo levels, repeating the index work eight times).  *out = *meta with
 * n_feat_per_pseudo_lvl = width (2, 4 or 8) and ONLY the pseudo levels of the levels whose own widest admissible width
 * (largest of 8, 4, 2 dividing n_feats, capped at `max_width`) is `width`; levels, offsets, n_encoded_dims and the output columns (map_col) are
 * those of *meta, so calls with the regrouped metas of all three widths write disjoint columns / table slices of the same
 * tensors and together equal one call with *meta.  out->n_pseudo_levels == 0: no level has that width. */
int nr3d_lotd_meta_regroup(const nr3d_lotd_meta_t *meta, uint32_t width, uint32_t max_width, nr3d_lotd_meta_t *out);

/* Batch addressing shared by all LoTD entry points (lotd_encoding.h:166-178):
 *   batch_inds   int64 [N] or NULL (value < 0 => point skipped)
 *   batch_offsets int64 [B] or NULL (element offset of each batch entry; default b * n_params)
 *   batch_data_size  >0 => b = i / batch_data_size when batch_inds is NULL
 *   max_level    levels > max_level contribute zeros; <= -1 => everything zero (lotd_torch_api.cu:294)
 * `meta_dev` is a device-resident byte copy of *meta (the caller uploads it once per meta/device).
 * Strides are in ELEMENTS.  x is [N, D] contiguous; params is 1-D contiguous.
 * param_dtype: NR3D_F32, or NR3D_F16 -- `params` are __half tables, read as half and used as float (the reference's
 * (float, half, float) dispatch, lotd_encoding.h:1501-1504) by every kernel that reads table entries: every meta, batched
 * tables and (ABI 3) the forest entry points included; a half table gives bit for bit what its fp32 copy gives.  With NR3D_F16 nr3d_lotd_fwd writes y as __half
 * (the fp32 result rounded once); dy_dx, dL_dx and -- except through nr3d_lotd_bwd_dparam_typed -- dL_dparam stay float, and
 * so does dL_dy (the caller widens a half dL_dy; the pair-record path below reads it as it is).
 */

/* lod_fwd (lotd_torch_api.cu:232-395): y[i*y_sn + e*y_se] (params dtype);
 * dy_dx[i*dydx_sn + e*dydx_se + d] (x dtype) or NULL. */
int nr3d_lotd_fwd(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points,
                  int x_dtype, int param_dtype, const void *x, const void *params,
                  const int64_t *batch_inds, const int64_t *batch_offsets, uint32_t batch_data_size,
                  int32_t max_level, void *y, int64_t y_sn, int64_t y_se,
                  void *dy_dx, int64_t dydx_sn, int64_t dydx_se, void *stream);

/* lod_bwd, input-gradient half (lotd_encoding.h:1562-1586 / lotd_hash_only.h:839-863):
 * dL_dx[i, d] = sum_e dL_dy[i, e] * dy_dx[i, e, d];  dL_dx is [N, D] contiguous (x dtype). */
int nr3d_lotd_bwd_dx(const nr3d_lotd_meta_t *meta, uint32_t n_points, int x_dtype, int param_dtype,
                     const void *dL_dy, int64_t dldy_sn, int64_t dldy_se,
                     const void *dy_dx, int64_t dydx_sn, int64_t dydx_se, void *dL_dx,
                     void *dL_dy_T /* optional out, f32 [n_encoded_dims, n_points]: feature-major copy of a contiguous
                                      dL_dy; pass it to nr3d_lotd_bwd_dparam as dL_dy with strides (1, n_points) and
                                      the scatter skips its own transposition */,
                     void *stream);

/* lod_bwd, parameter-gradient half (kernel_lod[_hashonly]_backward_grid, lotd_encoding.h:467-711,
 * lotd_hash_only.h:380-470).  dL_dparam: params dtype, same numel as params, ZERO-INIT by caller.
 * workspace: optional device scratch of >= nr3d_lotd_dparam_workspace_bytes() bytes; when given (and the meta has
 * no 4-D NPlaneMul / NPlaneSum level) the scatter runs atomic-free (binned records + fp64 LDS accumulation); otherwise
 * (NULL / too small / those levels) hardware f32 atomics are used.
 * n_batches: number of table sets behind `params` when batch_inds / batch_offsets / batch_data_size are used (the
 * reference derives it from params.numel(); 0 = unknown -> batched calls take the atomic path); 0 or 1 otherwise. */
uint64_t nr3d_lotd_dparam_workspace_bytes(const nr3d_lotd_meta_t *meta, uint32_t n_points, uint32_t n_batches);
/* Tuning knob: points per pass of the atomic-free scatter = 2^log2_points (10..24; 0 restores the default 2^22 or
 * NR3D_LOTD_BIN_CHUNK_LOG2).  The workspace size follows; query it again after changing this. */
void nr3d_lotd_set_dparam_chunk_log2(int log2_points);
int nr3d_lotd_bwd_dparam(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points,
                         int x_dtype, int param_dtype, const void *dL_dy, int64_t dldy_sn, int64_t dldy_se,
                         const void *x, const void *params, const int64_t *batch_inds,
                         const int64_t *batch_offsets, uint32_t batch_data_size, uint32_t n_batches, int32_t max_level,
                         void *dL_dparam, void *workspace, uint64_t workspace_bytes, void *stream);
/* Same, restricted to levels min_level..max_level (the entries of the other levels in dL_dparam are not touched).  No
 * reference counterpart (it only has max_level): a data-parallel caller computes the gradient in level buckets and
 * starts the all-reduce of a finished bucket -- a contiguous slice of dL_dparam, levels are stored one after another
 * -- while the next bucket is being accumulated (bench.py, nr3d_lib_amd/distributed.py).  Per level the arithmetic is
 * that of nr3d_lotd_bwd_dparam; how a hot table slice is split over workgroups follows the records of the call, so the
 * buckets together equal the one-call result to fp32 rounding of the partial sums (each is an fp64 sum), not bitwise. */
int nr3d_lotd_bwd_dparam_levels(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points,
                                int x_dtype, int param_dtype, const void *dL_dy, int64_t dldy_sn, int64_t dldy_se,
                                const void *x, const void *params, const int64_t *batch_inds,
                                const int64_t *batch_offsets, uint32_t batch_data_size, uint32_t n_batches,
                                int32_t min_level, int32_t max_level, void *dL_dparam, void *workspace,
                                uint64_t workspace_bytes, void *stream);

/* dL_dy [N, E] (element (i, e) at i * g_sn + e * g_se; grad_dtype NR3D_F32 | NR3D_F16) -> out float [E][N], feature-major:
 * the layout the level-major kernels read.  The dL/dparam entry points make this copy themselves when handed a row-major
 * dL_dy; a caller that runs several of them on one dL_dy (d(dL/dx)/dparam and d(dL/dx)/dx of one second-order step) makes it
 * once and passes it with strides (1, N) to nr3d_lotd_bwd_bwd_dparam and nr3d_lotd_bwd_bwd_dx_ws. */
int nr3d_lotd_dLdy_feature_major(uint32_t n_points, uint32_t n_encoded_dims, int grad_dtype, const void *dL_dy,
                                 int64_t g_sn, int64_t g_se, float *out, void *stream);

/* Native half-parameter storage, the reference's (float, half, float) type combination (<input, param, compute>,
 * csrc/lotd/include/lotd/lotd_encoding.h:1501-1504; PARAM_T accumulators lotd_encoding.h:72): x and dy_dx float, params /
 * y / dL_dy / dL_dparam __half, arithmetic in fp32.  nr3d_lotd_half_params_ok: 1 when nr3d_lotd_fwd (param_dtype
 * NR3D_F16: y is __half too), nr3d_lotd_bwd_dx (param_dtype NR3D_F16: dL_dy is __half, contiguous [N, E]) and
 * nr3d_lotd_bwd_dparam_typed serve this meta with half GRADIENTS as well (dL_dy read and dL_dparam written as __half) --
 * unbatched 3-D Dense/Hash metas with 2-feature pseudo levels; for every other meta the half TABLES are still read as they
 * are (param_dtype above) and only dL_dy / dL_dparam pass through float.  Unlike the reference's __half2 atomics the
 * parameter gradient is accumulated exactly and rounded to half once. */
int nr3d_lotd_half_params_ok(const nr3d_lotd_meta_t *meta, int batched);
/* first-order dL/dparam of the pair-record path (nr3d_lotd_pair_path_ok) with explicit dtypes: grad_dtype of dL_dy (any
 * strides; F32 for the feature-major copy that nr3d_lotd_bwd_dx leaves), out_dtype of dL_dparam.  assign == 0:
 * dL_dparam is ZERO-INIT by the caller and accumulated into, like nr3d_lotd_bwd_dparam; assign != 0: dL_dparam arrives
 * UNINITIALISED and is fully defined on return (the flush writes instead of read-modify-writing when one pass covers
 * all levels; the library zero-fills it itself otherwise).  workspace as nr3d_lotd_bwd_dparam. */
int nr3d_lotd_pair_path_ok(const nr3d_lotd_meta_t *meta);
/* pseudo levels of a pair-path meta whose dL/dparam is accumulated straight from (x, dL_dy) in LDS instead of through
 * records (levels with <= 4 buckets; 0 when the pair path does not apply or NR3D_OPT_PAIR_DIRECT is 0).  Informational: which
 * kernel serves which level (bench.py's per-kernel byte model). */
int nr3d_lotd_pair_direct_levels(const nr3d_lotd_meta_t *meta, uint32_t n_points);
/* which pseudo levels (bit q) nr3d_lotd_fwd serves from LDS for a batch of n_points: whole Dense tables and (*by_slab, ABI 5) Dense
 * tables staged slab by slab; 0 when the two-lane forward does not apply.  bench.py prices the forward kernels on the levels each serves. */
uint64_t nr3d_lotd_fwd_lds_levels(const nr3d_lotd_meta_t *meta, uint32_t n_points, uint64_t *by_slab);
int nr3d_lotd_bwd_dparam_typed(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points, int grad_dtype,
                               const void *dL_dy, int64_t g_sn, int64_t g_se, const void *x, int32_t max_level,
                               int out_dtype, int assign, void *dL_dparam, void *workspace, uint64_t workspace_bytes,
                               void *stream);

/* lod_bwd_bwd_input (lotd_torch_api.cu:575-729), three independent outputs:
 * (i)  dL_ddLdy[i, e] = sum_d dL_ddLdx[i, d] * dy_dx[i, e, d]      (lotd_encoding.h:1703-1727) */
int nr3d_lotd_bwd_bwd_ddLdy(const nr3d_lotd_meta_t *meta, uint32_t n_points, int x_dtype, int param_dtype,
                            const void *dL_ddLdx, const void *dy_dx, int64_t dydx_sn, int64_t dydx_se,
                            void *dL_ddLdy, int64_t out_sn, int64_t out_se, void *stream);
/* (ii) d(dL/dx)/dparam (lotd_encoding.h:764-1041, lotd_hash_only.h:472-574); ZERO-INIT output. */
int nr3d_lotd_bwd_bwd_dparam(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points,
                             int x_dtype, int param_dtype, const void *dL_ddLdx,
                             const void *dL_dy, int64_t dldy_sn, int64_t dldy_se, const void *x,
                             const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                             uint32_t batch_data_size, uint32_t n_batches, int32_t max_level, void *dL_dparam,
                             void *workspace, uint64_t workspace_bytes, void *stream);
/* (iii) d(dL/dx)/dx (lotd_encoding.h:1157-1298, lotd_hash_only.h:576-695); dL_dx [N, D] fully written. */
int nr3d_lotd_bwd_bwd_dx(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points,
                         int x_dtype, int param_dtype, const void *dL_ddLdx,
                         const void *dL_dy, int64_t dldy_sn, int64_t dldy_se, const void *x,
                         const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                         uint32_t batch_data_size, int32_t max_level, void *dL_dx, void *stream);
/* the same with a scratch buffer of nr3d_lotd_bwd_bwd_dx_workspace_bytes(meta, n_points) bytes (0: not served -- Dense /
 * Hash metas only): the
 * pseudo levels of a point are then worked on side by side (one lane per (point, pseudo level), the forward's level-major
 * schedule) and summed in level order afterwards, instead of one after another in one lane.  Same values (the same bits
 * for 2-feature pseudo levels).  A NULL / short workspace falls back to nr3d_lotd_bwd_bwd_dx. */
uint64_t nr3d_lotd_bwd_bwd_dx_workspace_bytes(const nr3d_lotd_meta_t *meta, uint32_t n_points);
int nr3d_lotd_bwd_bwd_dx_ws(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points,
                            int x_dtype, int param_dtype, const void *dL_ddLdx,
                            const void *dL_dy, int64_t dldy_sn, int64_t dldy_se, const void *x,
                            const void *params, const int64_t *batch_inds, const int64_t *batch_offsets,
                            uint32_t batch_data_size, int32_t max_level, void *dL_dx, void *workspace,
                            uint64_t workspace_bytes, void *stream);

/* lod_get_grid_index (lotd_torch_api.cu:771-855; kernel lotd_encoding.h:1300-1433):
 * grid_inds int64 [N, n_encoded_dims, 2^D] contiguous, ZERO-INIT.  Dense/Hash levels only. */
int nr3d_lotd_grid_index(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint32_t n_points, int x_dtype,
                         const void *x, const int64_t *batch_inds, const int64_t *batch_offsets,
                         uint32_t batch_data_size, int32_t max_level, int64_t *grid_inds, void *stream);

/* =================================================================================================
 * Forest of blocks -- replaces nr3d_lib.bindings._forest.ForestMeta and the forest overloads of
 * nr3d_lib.bindings._lotd (csrc/forest/forest_cpp_api.h:18-37, csrc/forest/forest.h:25-97,
 * csrc/lotd/src/lotd.cpp:44-60, csrc/lotd/include/lotd/lotd_forest.h).
 * The octree is the breadth-first byte octree of a kaolin SPC: one occupancy byte per non-leaf node (child
 * index = x<<2 | y<<1 | z), exsum = exclusive prefix sum of the bytes' popcounts, node 0 = root; the blocks are
 * the nodes on `level`, block index = node index - level_poffset, block_ks their integer coordinates.
 * All three arrays are DEVICE pointers; the struct itself lives on the host and is passed by value to kernels.
 * ============================================================================================== */
typedef struct nr3d_forest_meta {
	const uint8_t *octree;        /* [n_nodes] */
	const int32_t *exsum;         /* [n_nodes + 1] */
	const int16_t *block_ks;      /* [n_trees, 3] */
	float world_block_size[3];
	float world_origin[3];
	int32_t resolution[3];
	uint32_t n_trees;
	uint32_t level;
	uint32_t level_poffset;
	int32_t continuity_enabled;   /* look corner values up in the neighbouring block across block faces */
} nr3d_forest_meta_t;

/* block index (or -1) of integer block coordinates ks (int16 [n,3]) -- `identify`, forest.h:25-58 /
 * ForestMetaRef::map_block_ind :88-95 (the reference queries through kaolin's unbatched_query on the host side,
 * nr3d_lib/models/spatial/forest.py:244-260). */
int nr3d_forest_identify(const nr3d_forest_meta_t *forest, uint64_t n, const int16_t *ks, int32_t *block_inds,
                         void *stream);

/* lod_fwd(metas=(lod_meta, forest_meta), ...)  (lotd_torch_api.cu:333-361, kernel_lod_forest lotd_forest.h:158-333).
 * x in [0,1]^3 INSIDE the point's block; block_inds int64 [N] (<0: the point is skipped, outputs zero) or NULL with
 * batch_data_size (points per block, blocks in order) or neither (block 0); block_offsets int64 [n_trees] or NULL
 * (block b's parameters start at b * n_params).  f32, D == 3, level types Dense / VectorMatrix / NPlaneMul / CP / Hash
 * (the reference's forest kernels handle no others).  y[i*y_sn + e*y_se]; dy_dx[i*d_sn + e*d_se + d] or NULL
 * (strides in elements, as for nr3d_lotd_fwd; feature-major storage y_sn = 1, y_se = N makes the stores coalesced). */
int nr3d_lotd_forest_fwd(const nr3d_lotd_meta_t *meta, const void *meta_dev, const nr3d_forest_meta_t *forest,
                         uint32_t n_points, const float *x, const void *params, int param_dtype /* NR3D_F32 | NR3D_F16 */,
                         const int64_t *block_inds,
                         const int64_t *block_offsets, uint32_t batch_data_size, int32_t max_level, float *y,
                         int64_t y_sn, int64_t y_se, float *dy_dx, int64_t d_sn, int64_t d_se, void *stream);

/* scratch for the atomic-free forest parameter gradient (per-corner records: a corner may belong to a neighbouring
 * block; Dense / Hash / VecZMatXoY / CP / NPlaneMul / VM levels, lotd_forest.h:415-636); 0 = not applicable (atomics). */
uint64_t nr3d_lotd_forest_dparam_workspace_bytes(const nr3d_lotd_meta_t *meta, uint32_t n_points, uint32_t n_trees);
/* dL/dparam (dL_ddLdx == NULL; kernel_lod_forest_backward_grid :414-542) or d(dL/dx)/dparam
 * (kernel_lod_forest_backward_input_backward_grid :636-773).  dL_dparam [n_trees * n_params] ZERO-INIT by the caller;
 * contributions of corners that lie in a neighbouring block go to that block's parameters.
 * workspace: >= nr3d_lotd_dparam_workspace_bytes(meta, n_points, n_trees) bytes of device scratch, or NULL.  With it,
 * metas whose levels are all Dense / Hash take the atomic-free sort + segmented-sum path of nr3d_lotd_bwd_dparam (the
 * blocks play the role of batch entries; a corner owned by a neighbour is binned into that block's table); otherwise
 * (or when the path does not apply) fp32 hardware atomics. */
int nr3d_lotd_forest_bwd_dparam(const nr3d_lotd_meta_t *meta, const void *meta_dev, const nr3d_forest_meta_t *forest,
                                uint32_t n_points, const float *dL_ddLdx, const float *dL_dy, const float *x,
                                const void *params, int param_dtype, const int64_t *block_inds, const int64_t *block_offsets,
                                uint32_t batch_data_size, int32_t max_level, float *dL_dparam, void *workspace,
                                uint64_t workspace_bytes, void *stream);

/* d(dL/dx)/dx (kernel_lod_forest_backward_input_backward_input :929-1065): Dense / VectorMatrix / Hash levels
 * contribute.  dL_dx [N,3] is overwritten.  (dL/dx and dL/d(dL/dy) are contractions with dy_dx:
 * nr3d_lotd_bwd_dx / nr3d_lotd_bwd_bwd_ddLdy.) */
int nr3d_lotd_forest_bwd_bwd_dx(const nr3d_lotd_meta_t *meta, const void *meta_dev, const nr3d_forest_meta_t *forest,
                                uint32_t n_points, const float *dL_ddLdx, const float *dL_dy, const float *x,
                                const void *params, int param_dtype, const int64_t *block_inds, const int64_t *block_offsets,
                                uint32_t batch_data_size, int32_t max_level, float *dL_dx, void *stream);

/* =================================================================================================
 * occ_grid ray marching -- replaces nr3d_lib.bindings._occ_grid
 *   csrc/occ_grid/include/occ_grid/cpp_api.h:14-66, csrc/occ_grid/src/ray_marching.cu:136-244,
 *   csrc/occ_grid/src/batched_marching.cu:153-287
 * Two-phase: *_count writes num_steps[n_rays] AND the exclusive scan packed_info[n_rays,2]
 * (= [cumsum - num, num], int32) plus the grand total into total_steps[0] (device int32/int64);
 * the caller reads total_steps back (the single host sync), allocates outputs, calls *_emit.
 * total_steps / total / totals of the count entry points may be DEVICE memory (then copy it back) or pinned, device-visible HOST
 * memory (hipHostMalloc): the scan's last store lands there and the readback is a stream synchronisation, no copy launch.
 * Optional sample cache (>= nr3d_ray_marching_cache_bytes(n_rays, max_steps) bytes, or NULL): *_count also stores
 * every sample in it and *_emit(sample_cache, cache_max_steps = that max_steps) becomes a parallel compaction
 * instead of a second march (the reference always marches twice, ray_marching.cu:170-240).
 *   grid_binary: uint8/bool [ (B,) Rx, Ry, Rz ], z contiguous.  roi: f32 [6] or [B,6].
 *   batched != 0: batch_inds int32 [n_rays] or NULL (<0 skips), batch_data_size as for LoTD.
 * ============================================================================================== */
enum { NR3D_CONTRACT_AABB = 0, NR3D_CONTRACT_UN_BOUNDED_TANH = 1, NR3D_CONTRACT_UN_BOUNDED_SPHERE = 2 };

int nr3d_ray_marching_count(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                            const float *t_max, const float *roi, const int32_t grid_res[3],
                            const uint8_t *grid_binary, int contraction_type, float step_size,
                            float max_step_size, float dt_gamma, uint32_t max_steps, int batched,
                            const int32_t *batch_inds, uint32_t batch_data_size,
                            int32_t *packed_info /*[n_rays,2]*/, int64_t *total_steps /*[1]*/,
                            void *scan_tmp /* >= nr3d_scan_tmp_bytes(n_rays) bytes */,
                            void *sample_cache /*or NULL*/, uint64_t sample_cache_bytes, void *stream);

uint64_t nr3d_ray_marching_cache_bytes(uint32_t n_rays, uint32_t max_steps);

int nr3d_ray_marching_emit(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                           const float *t_max, const float *roi, const int32_t grid_res[3],
                           const uint8_t *grid_binary, int contraction_type, float step_size,
                           float max_step_size, float dt_gamma, int batched, const int32_t *batch_inds,
                           uint32_t batch_data_size, const int32_t *packed_info, float *t_starts,
                           float *t_ends, int32_t *ridx, int32_t *bidx /*NULL unless batched*/,
                           int32_t *gidx /*or NULL*/, const void *sample_cache /*or NULL*/,
                           uint32_t cache_max_steps, void *stream);

/* forest_ray_marching (csrc/occ_grid/src/forest_marching.cu:16-303): marching through the occupancy grids of the
 * blocks a ray crosses.  The block segments of every ray (seg_block_inds int32 [S], seg_entries / seg_exits f32 [S],
 * packed per ray by seg_pack_infos int32 [n_rays,2]) come from the caller's octree ray trace.
 * grid_binary uint8/bool [n_trees, Rx, Ry, Rz].  Same two-phase contract as nr3d_ray_marching_count/_emit. */
int nr3d_forest_ray_marching_count(const nr3d_forest_meta_t *forest, uint32_t n_rays, const float *rays_o,
                                   const float *rays_d, const float *t_min, const float *t_max,
                                   const int32_t *seg_block_inds, const float *seg_entries, const float *seg_exits,
                                   const int32_t *seg_pack_infos, const int32_t grid_res[3], const uint8_t *grid_binary,
                                   float step_size, float max_step_size, float dt_gamma, uint32_t max_steps,
                                   int32_t *packed_info /*[n_rays,2]*/, int64_t *total_steps /*[1]*/, void *scan_tmp,
                                   void *stream);
int nr3d_forest_ray_marching_emit(const nr3d_forest_meta_t *forest, uint32_t n_rays, const float *rays_o,
                                  const float *rays_d, const float *t_min, const float *t_max,
                                  const int32_t *seg_block_inds, const float *seg_entries, const float *seg_exits,
                                  const int32_t *seg_pack_infos, const int32_t grid_res[3], const uint8_t *grid_binary,
                                  float step_size, float max_step_size, float dt_gamma, const int32_t *packed_info,
                                  float *t_starts, float *t_ends, int32_t *ridx, int32_t *blidx, int32_t *gidx /*or NULL*/,
                                  void *stream);

/* Scratch bytes needed by the device-wide scans used in two-phase ops (any n). */
uint64_t nr3d_scan_tmp_bytes(uint64_t n);

/* Occupancy-value grid maintenance -- replaces torch_scatter.scatter_max in update_occ_val_grid[_idx]_ /
 * update_batched_occ_val_grid[_idx]_ (nr3d_lib/models/accelerations/occgrid/utils.py:80-125):
 *   grid[v] <- max(ema_decay * grid[v], max_{samples in v} occ_val)  for touched voxels, unchanged otherwise.
 * scatter_max: vmax [n_batches * Rx*Ry*Rz] f32 scratch := -inf, then the per-voxel maximum of occ_val.  Voxel of
 *   sample i: gidx[i] (int64 [n,3]) or, when gidx == NULL, ((pts[i]/2 + 0.5) * res).long().clamp(0, res-1);
 *   batch of sample i: bidx[i] (int64) or i / per_batch (per_batch == 0: single grid).
 * apply_max: applies the decayed maximum in place.  A sharded caller all-reduces(MAX) vmax between the two. */
int nr3d_occ_scatter_max(uint64_t n, const int64_t *gidx, const float *pts, const int64_t *bidx, uint64_t per_batch,
                         const float *occ_val, const int32_t grid_res[3], uint32_t n_batches, float *vmax, void *stream);
int nr3d_occ_apply_max(uint64_t n_voxels, float ema_decay, const float *vmax, float *occ_val_grid, void *stream);

/* =================================================================================================
 * Fused fully-connected decoder -- the MLP right after the encoder: nr3d_lib/models/blocks/mlp.py:27-127 (`MLP` /
 * `FCBlock`: D hidden DenseLayers + an output layer, nr3d_lib/models/layers.py:228-300), in the reference a chain of
 * GEMM + elementwise launches (or tiny-cuda-nn's fused fp16 network behind nr3d_lib/models/tcnn_adapter.py:74-237).
 * fp32 in / fp32 accumulate; the whole network in one kernel, activations in registers.  Forward: on the bf16 MFMA with every value
 * split into three bf16 pieces and the six significant piece products (fp32-grade; NR3D_OPT_MLP_X3, ABI 5) or on the f32 MFMA; backward:
 * f32 MFMA.  The packed buffer holds [f32 layers | the layers' bf16 pieces | transposed f32 layers (with_backward)]: opaque to the caller.
 *   dims[0] = in_features, dims[1..n_layers-1] = hidden widths, dims[n_layers] = out_features; every width 1..128;
 *   weights[l]: f32 [dims[l+1], dims[l]] row-major (torch nn.Linear layout), biases[l]: f32 [dims[l+1]] or NULL.
 * nr3d_mlp_packed_floats: size of the packed-weights buffer, or 0 when the fused kernels do not apply (a single
 *   layer, a width > 128, packed weights beyond the LDS budget) -- the caller then keeps its unfused path.
 * nr3d_mlp_pack: weights/biases (HOST arrays of n_layers DEVICE pointers) -> packed, in MFMA operand order; call it
 *   whenever the parameters changed (once per optimiser step).
 * nr3d_mlp_forward: y[i*y_stride + o] for x[i*x_stride + f*x_feature_stride], x either row-major (x_feature_stride == 1;
 *   rows need not be padded or aligned, 16-byte aligned rows take vector loads) or feature-major (x_stride == 1: the
 *   [E, N] storage the LoTD kernels write, consumed without a transposing copy -- a half-wave then reads 128 contiguous
 *   bytes of one feature).  nr3d_mlp_backward takes x the same way and stores dL_dx in either layout
 *   (gx_stride / gx_feature_stride; feature-major dL_dx is what nr3d_lotd_bwd_dparam reads without its transposition
 *   pass).
 * ============================================================================================== */
#define NR3D_MLP_MAX_LAYERS 8
enum { NR3D_MLP_ACT_NONE = 0, NR3D_MLP_ACT_RELU = 1 };

typedef struct nr3d_mlp_desc {
	uint32_t n_layers;                          /* linear layers: hidden layers + 1 */
	uint32_t dims[NR3D_MLP_MAX_LAYERS + 1];
	uint32_t hidden_activation;                 /* NR3D_MLP_ACT_* after every hidden layer */
	uint32_t output_activation;
} nr3d_mlp_desc_t;

uint64_t nr3d_mlp_packed_floats(const nr3d_mlp_desc_t *desc);
/* extra floats of the packed buffer that nr3d_mlp_backward needs (a 4-float header: the backward reads the forward layers), or 0
 * when the fused backward does not apply (hidden width > 64, more than 2 hidden layers wider than 32 / 3 narrower ones, input or output wider
 * than the hidden layers): the caller then differentiates its unfused path. */
uint64_t nr3d_mlp_backward_packed_floats(const nr3d_mlp_desc_t *desc);
/* packed: nr3d_mlp_packed_floats (+ nr3d_mlp_backward_packed_floats when with_backward != 0) floats */
int nr3d_mlp_pack(const nr3d_mlp_desc_t *desc, const float *const *weights, const float *const *biases, float *packed,
                  int with_backward, void *stream);
/* dL/dx (or NULL), dL/dW_l [dims[l+1], dims[l]] and dL/db_l (dL_db or entries may be NULL) from x and dL/dy; the forward
 * is recomputed in registers, nothing but x has to be kept from the forward pass.  Parameter gradients are ADDED to
 * dL_dW / dL_db (fp32 atomics, one per element and workgroup): zero them for plain gradients. */
int nr3d_mlp_backward(const nr3d_mlp_desc_t *desc, uint64_t n, const float *x, int64_t x_stride, int64_t x_feature_stride,
                      const float *dL_dy, int64_t gy_stride, const float *packed, float *dL_dx, int64_t gx_stride,
                      int64_t gx_feature_stride, float *const *dL_dW, float *const *dL_db, void *stream);
int nr3d_mlp_forward(const nr3d_mlp_desc_t *desc, uint64_t n, const float *x, int64_t x_stride, int64_t x_feature_stride,
                     const float *packed, float *y, int64_t y_stride, void *stream);

/* LoTD encode + this decoder's FORWARD in one kernel (csrc/lotd_mlp.hip; no reference counterpart -- the reference runs lod_fwd and
 * the decoder as separate ops): for the no-grad density query of the ray driver (nr3d_lib/graphics/nerf/nerf_ray_query.py:105-127),
 * which needs ONE number per marched sample.  out[i, c] for c < out_cols = column c of decoder(encode(x[i])) -- the SAME values as
 * nr3d_lotd_forward followed by nr3d_mlp_forward (bit-identical when both take the two-lane forward, i.e. with the fwd_lds_stage option off; to fp32
 * rounding of the interpolation otherwise), without the [N, n_encoded_dims] features ever reaching memory.  x [N, 3] float contiguous
 * (already in the encoder's [0, 1] range), params / param_dtype / max_level / meta_dev as nr3d_lotd_forward, packed = the decoder's
 * packed buffer (nr3d_mlp_pack), out float with row stride out_stride.  nr3d_lotd_mlp_forward_ok: 1 when the pair is inside the
 * kernel's range (3-D meta of Dense / Hash levels, 2-feature pseudo levels, <= 32 encoded dims = the decoder's input width; hidden
 * width <= 64, <= 32 outputs), else 0: the caller keeps the two calls. */
int nr3d_lotd_mlp_forward_ok(const nr3d_lotd_meta_t *meta, const nr3d_mlp_desc_t *desc);
int nr3d_lotd_mlp_forward(const nr3d_lotd_meta_t *meta, const void *meta_dev, uint64_t n_points, const float *x, const void *params,
                          int param_dtype, int32_t max_level, const nr3d_mlp_desc_t *desc, const float *packed, float *out,
                          int64_t out_stride, uint32_t out_cols, void *stream);

/* The same decoder in HALF precision on the f16 MFMA (csrc/mlp_half.hip) -- the contract of the reference's fast decoder,
 * tiny-cuda-nn's FullyFusedMLP behind nr3d_lib/models/tcnn_adapter.py:37-51,74-146 (`use_tcnn_backend`,
 * nr3d_lib/models/blocks/__init__.py:3-15): half weights / biases / inputs / outputs, activations rounded to half between
 * the layers; the dot products of a layer are accumulated in fp32 (tcnn accumulates in half).  x, y, dL_dy, dL_dx: __half,
 * same layout rules as the fp32 entry points (strides in elements; row-major rows 8-byte aligned with widths that are multiples
 * of 4 take the vector / prefetched path).  weights[l] / biases[l]: __half, torch nn.Linear layout.  Sizes are BYTES here.
 * nr3d_mlp_half_backward: dL_dW / dL_db are FP32 buffers the workgroups add their partial sums to (zero them first) -- the
 * caller rounds them to the parameters' dtype; the fused backward applies to the same shapes as nr3d_mlp_backward. */
uint64_t nr3d_mlp_half_packed_bytes(const nr3d_mlp_desc_t *desc);
uint64_t nr3d_mlp_half_backward_packed_bytes(const nr3d_mlp_desc_t *desc);
int nr3d_mlp_half_pack(const nr3d_mlp_desc_t *desc, const void *const *weights, const void *const *biases, void *packed,
                       int with_backward, void *stream);
int nr3d_mlp_half_forward(const nr3d_mlp_desc_t *desc, uint64_t n, const void *x, int64_t x_stride, int64_t x_feature_stride,
                          const void *packed, void *y, int64_t y_stride, void *stream);
int nr3d_mlp_half_backward(const nr3d_mlp_desc_t *desc, uint64_t n, const void *x, int64_t x_stride, int64_t x_feature_stride,
                           const void *dL_dy, int64_t gy_stride, const void *packed, void *dL_dx, int64_t gx_stride,
                           int64_t gx_feature_stride, float *const *dL_dW, float *const *dL_db, void *stream);

/* =================================================================================================
 * pack_ops -- replaces nr3d_lib.bindings._pack_ops  (csrc/pack_ops/pack_ops.h:11-65,
 * csrc/pack_ops/pack_ops.cpp:21-58, kernels csrc/pack_ops/pack_ops_cuda.cu)
 * pack_infos: int64 [P,2] = (begin, length), contiguous.  feats: [S] or [S, feat_dim] contiguous.
 * ============================================================================================== */

/* n_per_pack (int64 [P]) -> pack_infos [P,2] = (exclusive cumsum, n) and total[0]; the host-side
 * `cumsum` + `stack` of every two-phase pack op (e.g. pack_ops_cuda.cu:584-586). */
int nr3d_pack_infos_from_n(uint32_t P, const int64_t *n_per_pack, int64_t *pack_infos, int64_t *total,
                           void *scan_tmp, void *stream);

/* interleave_arange / interleave_linstep (:47-218).  starts/step_sizes may be NULL (then scalars
 * start_s/step_s are used; passed as double and converted to dtype).  nidx may be NULL. */
int nr3d_interleave_linstep(uint32_t P, int dtype, const int64_t *pack_infos, const void *starts,
                            const void *step_sizes, double start_s, double step_s, void *out,
                            int64_t *nidx, void *stream);
/* the step counts of interleave_arange (graphics/pack_ops/pack_ops.py: stop.subtract(start).div(step_size).ceil().long(), four ATen
 * launches) in one: num_steps[i] = ceil((stops[i] - starts[i]) / step) as int64, the quotient in float32 for int32 / int64 / float32
 * tensors (ATen's true division) and in float64 for float64.  step_sizes [P] of the tensors' dtype, or NULL: the scalar step_s. */
int nr3d_arange_num_steps(uint32_t P, int dtype, const void *starts, const void *stops, const void *step_sizes, double step_s,
                          int64_t *num_steps, void *stream);

/* interleave_sample_step_wrt_depth_clamped (:480-604), round 1 then round 2. */
int nr3d_sample_step_count(uint32_t P, const float *nears, const float *fars, uint32_t max_steps,
                           float dt_gamma, float min_step, float max_step, int64_t *n_per_pack, void *stream);
int nr3d_sample_step_emit(uint32_t P, const float *nears, const int64_t *pack_infos, float dt_gamma,
                          float min_step, float max_step, float *t_samples, float *deltas, int64_t *nidx,
                          void *stream);
/* interleave_sample_step_wrt_depth_in_packed_segments (:606-795): emit == 0 -> n_per_pack. */
int nr3d_sample_step_segments(uint32_t P, const float *nears, const float *fars, const float *entries,
                              const float *exits, const int64_t *seg_pack_infos, uint32_t max_steps,
                              float dt_gamma, float min_step, float max_step, int emit, int64_t *n_per_pack,
                              const int64_t *pack_infos, float *t_samples, float *deltas, int64_t *nidx,
                              int64_t *sidx, void *stream);

/* packed_sum (:798-861).  out [P, feat_dim], fully written (empty packs -> 0). */
int nr3d_packed_sum(uint32_t P, uint64_t S, uint32_t feat_dim, int dtype, const void *feats,
                    const int64_t *pack_infos, void *out, void *stream);
/* Rows of `out` outside every pack (packed_scan / packed_diff / packed_binary; the reference allocates at::zeros):
 *   ordered_packs == 0: the kernel writes the rows of the packs only -- pass a zeroed `out`;
 *   ordered_packs != 0: the caller states pack_infos[p+1].begin >= pack_infos[p].begin + pack_infos[p].len for every p
 *   (what every producer of pack_infos emits: a marcher, an interleave_* op, get_pack_infos_from_n); the kernel then
 *   zeroes the rows in front of the first pack, between packs and behind the last pack (up to S) itself, so `out`
 *   may be uninitialised and no fill launch is needed in front of a launch-bound op. */
/* packed_cumsum / packed_cumprod (:864-1095).  reference first-element semantics. */
int nr3d_packed_scan(uint32_t P, uint64_t S, uint32_t feat_dim, int dtype, const void *feats,
                     const int64_t *pack_infos, int is_prod, int exclusive, int reverse, int ordered_packs,
                     void *out, void *stream);
/* packed_diff / packed_backward_diff (:1098-1333).  edge_a = appends|prepends, edge_fill = last|first fill
 * (at most one non-NULL).  backward != 0 selects packed_backward_diff. */
int nr3d_packed_diff(uint32_t P, uint64_t S, uint32_t feat_dim, int dtype, const void *feats,
                     const int64_t *pack_infos, const void *edge_a, const void *edge_fill, int backward,
                     int ordered_packs, void *out, void *stream);
/* packed_{add,sub,mul,div,gt,geq,lt,leq,eq,neq} (:1960-2480).  op = PackBinaryOpType value
 * (pack_ops.h:25-37: Add 0, Subtract 1, Multiply 2, Division 3, Matmul 4, Gt 5 ... Neq 10).
 * Comparisons write uint8 (bool).  Matmul: other [P, out_dim, feat_dim], out [S, out_dim]. */
int nr3d_packed_binary(uint32_t P, uint64_t S, uint32_t feat_dim, uint32_t out_dim, int dtype,
                       const void *feats, const void *other, const int64_t *pack_infos, int op,
                       int ordered_packs, void *out, void *stream);
/* packed_searchsorted[_packed_vals] (:1336-1503).  val_pack_infos NULL -> vals is [P, num_to_search]. */
int nr3d_packed_searchsorted(uint32_t P, int dtype, const void *bins, const void *vals,
                             const int64_t *pack_infos, uint32_t num_to_search,
                             const int64_t *val_pack_infos, int64_t *pidx, void *stream);
/* try_merge_two_packs_sorted_aligned (:1505-1631).  The rows of the packs are fully written (ABI 5: the kernel zeroes its own
 * counting rows); rows of pidx_a / pidx_b outside every pack keep the caller's fill -- 0 as the reference's aligned op, -1 as its
 * merge_two_packs_sorted. */
int nr3d_try_merge_two_packs_sorted_aligned(uint32_t P, int dtype, const void *vals_a,
                                            const int64_t *pack_infos_a, const void *vals_b,
                                            const int64_t *pack_infos_b, const int64_t *pack_infos_merged,
                                            int b_sorted, int64_t *pidx_a, int64_t *pidx_b, void *stream);
/* merge_two_packs_sorted (graphics/pack_ops/pack_ops.py:611-640: the torch.unique / nonzero / index chain in front of the aligned
 * merge): the UNION of two sorted, unique pack-id lists as ALIGNED descriptors.  Union pack k is (a's pack | an empty one, b's pack |
 * an empty one): u [<= Pa + Pb] ids, pack_infos_a_u / pack_infos_b_u [<= Pa + Pb, 2], n_u [<= Pa + Pb] merged lengths; the first
 * total_u[0] = Pa + (packs of b that a does not have) rows are written; total_u may be pinned host memory, like the other totals
 * (n_only [1] is device scratch).
 * Then: nr3d_pack_infos_from_n(n_u) and nr3d_try_merge_two_packs_sorted_aligned on the aligned descriptors.
 * Scratch: only_b [Pb], ob [Pb, 2] (int64), scan_tmp >= nr3d_scan_tmp_bytes(Pb). */
int nr3d_merge_pack_union(uint32_t Pa, const int64_t *nidx_a, const int64_t *pack_infos_a, uint32_t Pb, const int64_t *nidx_b,
                          const int64_t *pack_infos_b, int64_t *only_b, int64_t *ob, void *scan_tmp, int64_t *n_only,
                          int64_t *u, int64_t *pack_infos_a_u, int64_t *pack_infos_b_u, int64_t *n_u, int64_t *total_u, void *stream);
/* packed_invert_cdf (:1633-1733). */
int nr3d_packed_invert_cdf(uint32_t P, const float *bins, const float *cdfs, const int64_t *pack_infos,
                           const float *u_vals, uint32_t num_to_sample, float *samples, int64_t *bin_idx,
                           void *stream);
/* packed_sort_qsort (:2634-2763): sorts vals IN PLACE per pack; ids (int64 arange, or NULL) permuted. */
int nr3d_packed_sort(uint32_t P, uint64_t S, int dtype, void *vals, int64_t *ids, const int64_t *pack_infos,
                     void *stream);
/* packed_alpha_to_vw_forward (:1735-1793, :1850-1907).  Any of weights / num_steps / selector may be
 * NULL; weights and selector are fully written (skipped samples -> 0). */
int nr3d_alpha_to_vw_forward(uint32_t P, uint64_t S, const float *alphas, const int64_t *pack_infos,
                             float early_stop_eps, float alpha_thre, float *weights, int64_t *num_steps,
                             uint8_t *selector, void *stream);
/* packed_alpha_to_vw_backward (:1795-1848, :1910-1958).  grad_alphas fully written. */
int nr3d_alpha_to_vw_backward(uint32_t P, uint64_t S, const float *alphas, const float *weights,
                              const float *grad_weights, const int64_t *pack_infos, float early_stop_eps,
                              float alpha_thre, float *grad_alphas, void *stream);
/* Fused alpha composite of a packed volume buffer.  No single reference kernel: replaces the renderer's op chain
 * nr3d_lib/models/fields/nerf/renderer_mixin.py:298-311 = packed_alpha_to_vw (pack_ops_cuda.cu:1735-1793) ->
 * packed_sum (:798-861) -> packed_div (:1960-2062) -> packed_sum(. * t) -> packed_sum(. * rgb), and its autograd
 * (pack_ops.py:97-116, :261-283, :293-392; kernel :1795-1848) by one launch each way.
 *   alphas, t [S]; rgb [S,3] or NULL; ray_index int64 [P] or NULL: per-ray results go to element ray_index[p]
 *   (rays_inds_hit) of mask / depth [num_rays] / rgb_out [num_rays,3] (caller ZERO-INITs them when ray_index is given).
 *   vw [S] fully written, bit-identical to nr3d_alpha_to_vw_forward.  depth = sum(vw*t) / (mask + 1e-10) when
 *   normalize_depth, else sum(vw*t). */
int nr3d_pack_composite_fwd(uint32_t P, const float *alphas, const float *t, const float *rgb, const int64_t *pack_infos,
                            const int64_t *ray_index, float early_stop_eps, float alpha_thre, int normalize_depth,
                            float *vw, float *mask, float *depth, float *rgb_out, void *stream);
/* backward of the above: g_mask / g_depth [num_rays], g_rgb [num_rays,3], g_vw [S] (each may be NULL = zero);
 * mask, depth = the forward's outputs.  grad_alphas [S] fully written; grad_t [S], grad_rgb [S,3] optional. */
int nr3d_pack_composite_bwd(uint32_t P, const float *alphas, const float *vw, const float *t, const float *rgb,
                            const int64_t *pack_infos, const int64_t *ray_index, float early_stop_eps, float alpha_thre,
                            int normalize_depth, const float *mask, const float *depth, const float *g_mask,
                            const float *g_depth, const float *g_rgb, const float *g_vw, float *grad_alphas,
                            float *grad_t, float *grad_rgb, void *stream);
/* -------------------------------------------------------------------------------------------------
 * Ray-query glue (csrc/ray_glue.hip): the device-side steps between march, density query, pruning and composite that the
 * reference runs as chains of ATen ops with host syncs in between.  No single reference kernel each; cited per entry.
 * ---------------------------------------------------------------------------------------------- */
/* alpha = 1 - exp(-sigma * delta)  (nr3d_lib/graphics/nerf/nerf_utils.py:23-24 applied to sigma * deltas,
 * nerf_ray_query.py:126,178) and its gradient grad_sigma = grad_alpha * delta * exp(-sigma * delta).  [S] each. */
int nr3d_tau_to_alpha_fwd(uint64_t S, const float *sigma, const float *delta, float *alpha, void *stream);
int nr3d_tau_to_alpha_bwd(uint64_t S, const float *sigma, const float *delta, const float *grad_alpha, float *grad_sigma,
                          void *stream);
/* Post-processing of the marcher's outputs (nr3d_lib/graphics/raymarch/occgrid_raymarch.py:87-112: nonzero on the counts,
 * index, .long()): packed_info int32 [n_rays, 2] -> the rays with >= 1 sample, ascending: ridx_hit int64 [n_hit],
 * pack_infos int64 [n_hit, 2]; totals int64 [2] = {samples, n_hit} (the caller reads n_hit back; outputs sized n_rays).
 * scan_tmp >= nr3d_scan_tmp_bytes(n_rays).
 * Range of the three compacting entry points (this one, nr3d_ray_marching_count_finished, nr3d_prune_compact_packs): one scan
 * carries the running sample count (36 bits) and the rank among the non-empty packs (28 bits) -- fewer than 2^28 packs /
 * rays per call (checked: nonzero status above that), fewer than 2^36 samples in all (more than a 288 GB device can hold). */
int nr3d_march_finish_rays(uint32_t n_rays, const int32_t *packed_info, int64_t *ridx_hit, int64_t *pack_infos,
                           int64_t *totals, void *scan_tmp, void *stream);
/* ... and per sample (same lines): ridx64 = (int64) ridx, deltas = t_ends - t_starts, samples [S, 3] =
 * fma(rays_d[ridx], t_starts, rays_o[ridx]) (torch.addcmul).  Any output may be NULL. */
int nr3d_march_finish_samples(uint64_t S, const float *rays_o, const float *rays_d, const int32_t *ridx,
                              const float *t_starts, const float *t_ends, int64_t *ridx64, float *deltas, float *samples,
                              void *stream);
/* nr3d_ray_marching_count AND nr3d_march_finish_rays with ONE scan of the counts: packed_info as nr3d_ray_marching_count,
 * ridx_hit [n_rays] / pack_infos [n_rays, 2] (their first n_hit rows are written) as nr3d_march_finish_rays,
 * totals = {number of samples, n_hit}. */
int nr3d_ray_marching_count_finished(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                                     const float *t_max, const float *roi, const int32_t grid_res[3],
                                     const uint8_t *grid_binary, int type, float step_size, float max_step_size,
                                     float dt_gamma, uint32_t max_steps, int batched, const int32_t *batch_inds,
                                     uint32_t batch_data_size, int32_t *packed_info, int64_t *ridx_hit, int64_t *pack_infos,
                                     int64_t *totals, void *scan_tmp, void *sample_cache, uint64_t sample_cache_bytes,
                                     void *stream);

/* configs[2] in ONE call (round 6): ray_marching (csrc/occ_grid/src/ray_marching.cu:136-244) + the post-processing of
 * occgrid_raymarch.py:87-112 + alpha = 1 - exp(-sigma * delta) (nerf_utils.py:23-24) + the renderer's composite chain
 * (nr3d_lib/models/fields/nerf/renderer_mixin.py:298-311), enqueued back to back with NO device->host wait inside:
 *   count (+ sample cache) -> scan (+ hit rays, totals) -> cached emit (+ ridx64 / deltas / samples) -> composite over ALL rays.
 * Every per-sample buffer (t_starts, t_ends, ridx, gidx?, ridx64?, deltas, samples?, alphas, vw) has `rows` >= n_rays * max_steps
 * rows (the bound; checked) and its first S rows are written -- the same values as nr3d_ray_marching_count_finished +
 * nr3d_ray_marching_emit_finished + nr3d_tau_to_alpha_fwd + nr3d_pack_composite_fwd, vw / mask / depth / rgb_out bit-identical to
 * them.  sigma [sigma_rows] / rgb [sigma_rows, 3] (rgb optional): the caller's per-sample density / colour; rows beyond sigma_rows
 * read as density 0 (the caller compares S with sigma_rows after its readback).  mask / depth [n_rays], rgb_out [n_rays, 3] are
 * FULLY written (rays without samples: zeros -- no fill).  totals (device-visible, normally pinned host memory) = {S, n_hit}: read it
 * after a stream synchronisation, AFTER this call returned (and after nr3d_march_composite_bwd was enqueued, if wanted).
 * Needs the sample cache (nr3d_ray_marching_cache_bytes); single grid only (batched / forest marches keep the two-phase calls). */
int nr3d_march_composite_fwd(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                             const float *roi, const int32_t grid_res[3], const uint8_t *grid_binary, int type, float step_size,
                             float max_step_size, float dt_gamma, uint32_t max_steps, int32_t *packed_info, int64_t *ridx_hit,
                             int64_t *pack_infos, int64_t *totals, void *scan_tmp, void *sample_cache, uint64_t sample_cache_bytes,
                             uint64_t rows, float *t_starts, float *t_ends, int32_t *ridx, int32_t *gidx, int64_t *ridx64,
                             float *deltas, float *samples, const float *sigma, uint64_t sigma_rows, const float *rgb,
                             float early_stop_eps, float alpha_thre, int normalize_depth, float *alphas, float *vw, float *mask,
                             float *depth, float *rgb_out, void *stream);
/* Host-side wait for words a kernel writes into device-visible HOST memory (pinned; the `totals` of the two-phase and one-call ops):
 * spins until none of words[0..n) equals `sentinel` (the caller stores the sentinel before the launch; the kernels store the totals
 * with system scope) or `timeout_us` passed.  Returns 0 when the words arrived, 1 on timeout (NOT an error: the caller falls back to a
 * stream synchronisation).  Why: a stream synchronisation returns when EVERYTHING enqueued has drained and costs ~20 us of wake-up
 * latency; the totals are written by the SECOND of the one-call path's launches, and reading them early lets the host build its views
 * while the composite kernels still run.  Holds no lock, touches no HIP API. */
int nr3d_wait_host_words(const int64_t *words, int n, int64_t sentinel, uint32_t timeout_us);
/* its backward (= nr3d_pack_composite_bwd over all rays of packed_info, no ray_index, no g_vw; the same values): g_mask / g_depth
 * [n_rays], g_rgb [n_rays, 3] (each may be NULL = zero) -> grad_alphas [S rows written], grad_t / grad_rgb optional; grad_sigma
 * (optional, needs sigma and deltas) = grad_alphas * exp(-sigma * delta) * delta as nr3d_tau_to_alpha_bwd. */
int nr3d_march_composite_bwd(uint32_t n_rays, const int32_t *packed_info, const float *alphas, const float *vw, const float *t,
                             const float *rgb, float early_stop_eps, float alpha_thre, int normalize_depth, const float *mask,
                             const float *depth, const float *g_mask, const float *g_depth, const float *g_rgb, float *grad_alphas,
                             float *grad_t, float *grad_rgb, const float *sigma, const float *deltas, uint64_t sigma_rows,
                             float *grad_sigma, void *stream);

/* nr3d_ray_marching_emit from the sample cache of nr3d_ray_marching_count AND nr3d_march_finish_samples in one launch
 * (the cached emit is a per-ray copy: the epilogue rides on it): t_starts / t_ends / ridx (/ bidx / gidx) as
 * nr3d_ray_marching_emit, ridx64 / deltas / samples (each optional) as nr3d_march_finish_samples -- the same values. */
int nr3d_ray_marching_emit_finished(uint32_t n_rays, const float *rays_o, const float *rays_d, int batched,
                                    const int32_t *batch_inds, uint32_t batch_data_size, const int32_t *packed_info,
                                    const void *sample_cache, uint32_t cache_max_steps, float *t_starts, float *t_ends,
                                    int32_t *ridx, int32_t *bidx, int32_t *gidx, int64_t *ridx64, float *deltas,
                                    float *samples, void *stream);
/* Visibility pruning (nr3d_lib/graphics/nerf/nerf_utils.py:64-98 packed_volume_render_compression + the index gathers
 * of nerf_ray_query.py:128-137).  counts int64 [P] = kept samples per pack (nr3d_alpha_to_vw_forward's num_steps):
 * begin_all [P] = new begin of EVERY pack; the packs that keep >= 1 sample, ascending: idx_out [P'] (tag[i] if tag is
 * given, else i) and pack_infos_out int64 [P', 2]; totals int64 [2] = {kept samples, P'}. */
int nr3d_prune_compact_packs(uint32_t P, const int64_t *counts, const int64_t *tag, int64_t *begin_all, int64_t *idx_out,
                             int64_t *pack_infos_out, int64_t *totals, void *scan_tmp, void *stream);
/* ... then the kept samples (selector uint8 [S], pack_infos = the un-pruned packs) move to their compact positions,
 * ascending inside a pack: pidx int64 [S'] = their old indices, and up to four per-sample arrays are gathered in the same
 * pass: f1, f2 float [S], f3 float [S, 3], l1 int64 [S] (each in/out pair optional). */
int nr3d_prune_compact_samples(uint32_t P, const int64_t *pack_infos, const int64_t *begin_all, const uint8_t *selector,
                               const float *f1, const float *f2, const float *f3, const int64_t *l1, int64_t *pidx,
                               float *f1_out, float *f2_out, float *f3_out, int64_t *l1_out, void *stream);
/* mark_pack_boundaries_cuda (:2765-2805): boundaries int32 [num]. */
int nr3d_mark_pack_boundaries(uint64_t num, int dtype, const void *pack_ids, int32_t *boundaries, void *stream);

/* octree_mark_consecutive_segments (pack_ops.cpp:58, pack_ops_cuda.cu:2807-2887): pidx int32 [n] = octree nodes hit by
 * every ray in order (packs = rays), point_hierarchies int16 [n_nodes,3]; mark_start / mark_end uint8 [n] ZERO-INIT:
 * first / last node of every run of face-adjacent nodes. */
int nr3d_octree_mark_consecutive_segments(uint32_t P, const int32_t *pidx, const int64_t *pack_infos,
                                          const int16_t *point_hierarchies, uint8_t *mark_start, uint8_t *mark_end,
                                          void *stream);

/* Spatial order of a batch of sample positions (ABI 5; csrc/ray_glue.hip): x float [n, 3] -> order int32 [n], order[k] = the sample
 * at position k of a Morton curve through a 2^bits_per_dim grid over the batch's own bounding box (ties in input order).  The
 * ray-query driver (graphics/nerf/nerf_ray_query.py) runs the field on the rendered samples in this order: samples of neighbouring
 * rays that share grid cells become consecutive lanes, which the LoTD backward merges before its scatter and the forward's
 * gathers coalesce (the reference has no counterpart: nerf_ray_query.py:140-160 queries in ray order).  bits_per_dim 1..10;
 * tmp: nr3d_spatial_order_tmp_bytes(n) bytes. */
uint64_t nr3d_spatial_order_tmp_bytes(uint32_t n);
int nr3d_spatial_order(uint32_t n, const float *x, uint32_t bits_per_dim, int32_t *order, void *tmp, void *stream);
/* the field's inputs in that order: x_out[k] = x[order[k]], ridx_out[k] = ridx[order[k]] (int64 ray index per sample, optional) and
 * dirs_out[k] = dirs[ridx[order[k]]] (per-RAY float [n_rays, 3], optional: the view_dirs[ridx] gather of nerf_ray_query.py:78) */
int nr3d_order_gather_inputs(uint32_t n, const int32_t *order, const float *x, const int64_t *ridx, const float *dirs, float *x_out,
                             int64_t *ridx_out, float *dirs_out, void *stream);
/* rows of up to two per-sample float arrays (a [n, wa], b [n, wb]; a width of 0 skips one) between the two orders:
 * scatter != 0: out[order[k]] = in[k] (the field's outputs back to the samples' own order); else out[k] = in[order[k]] */
int nr3d_order_move_rows(uint32_t n, const int32_t *order, int scatter, const float *a, uint32_t wa, float *a_out, const float *b,
                         uint32_t wb, float *b_out, void *stream);

/* The library's own point sort (csrc/rsort.hip, ABI 5), exported for its tests -- lotd_sorted.hip orders the points of a large-table
 * dL/dparam pass with it (the reference's default build has no library sort either: pack_ops_cuda.cu:2621-2629 compiles thrust
 * out, :2634-2720 is its own kernel).  Stable LSD radix sort of `batch` (1 or 2) independent arrays of (uint32 key, uint32
 * value) pairs of the same length by key bits [0, bits): kin{0,1} / vin{0,1} -> kout{0,1} / vout{0,1} (the second set ignored
 * for batch 1).  vin NULL: the values are the element indices.  n_dev (optional): element count in DEVICE memory, <= n_max.
 * Inputs are left untouched, outputs must not alias them; tmp: nr3d_sort_pairs_u32_tmp_bytes(n_max, batch) bytes. */
uint64_t nr3d_sort_pairs_u32_tmp_bytes(uint32_t n_max, int batch);
int nr3d_sort_pairs_u32(void *tmp, int batch, const uint32_t *kin0, const uint32_t *vin0, uint32_t *kout0, uint32_t *vout0,
                        const uint32_t *kin1, const uint32_t *vin1, uint32_t *kout1, uint32_t *vout1, uint32_t n_max,
                        const uint32_t *n_dev, int bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NR3D_HIP_H */
