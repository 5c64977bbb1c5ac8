// nr3d_lib_amd/csrc/pack_launch.h -- launchers of pack_ops.hip kernels that another translation unit chains behind its own
// (occ_grid.hip: nr3d_march_composite_fwd / _bwd enqueue march -> scan -> emit -> composite without the host in between).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nr3d {
namespace pk {

// fused alpha = 1 - exp(-sigma * delta) + alpha composite, one wave per ray over ALL rays of packed_info (int32 [n_rays, 2])
int launch_composite_rays_fwd(uint32_t n_rays, const int32_t *packed_info, const float *sigma, const float *delta, uint64_t sigma_rows,
                              const float *ts, const float *rgb, float eps, float thre, int normalize, float *alphas, float *vw,
                              float *mask, float *depth, float *rgb_out, hipStream_t st);
// ... with the marcher's cached emit in front, in the same launch (the wave of a ray copies its cached samples to their packed rows, then
// composites them)
int launch_emit_composite_rays_fwd(uint32_t n_rays, const int32_t *packed_info, const void *cache, uint32_t cache_stride,
                                   const float *rays_o, const float *rays_d, float *t_starts, float *t_ends, int32_t *ridx, int32_t *gidx,
                                   int64_t *ridx64, float *deltas, float *samples, const float *sigma, uint64_t sigma_rows,
                                   const float *rgb, float eps, float thre, int normalize, float *alphas, float *vw, float *mask,
                                   float *depth, float *rgb_out, hipStream_t st);
int launch_composite_rays_bwd(uint32_t n_rays, const int32_t *packed_info, const float *alphas, const float *vw, const float *ts,
                              const float *rgb, float eps, float thre, int normalize, const float *mask, const float *depth,
                              const float *g_mask, const float *g_depth, const float *g_rgb, float *grad_alphas, float *grad_t,
                              float *grad_rgb, const float *sigma, const float *delta, uint64_t sigma_rows, float *grad_sigma,
                              hipStream_t st);

}  // namespace pk
}  // namespace nr3d
