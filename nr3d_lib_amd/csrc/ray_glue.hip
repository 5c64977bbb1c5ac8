// nr3d_lib_amd/csrc/ray_glue.hip -- the small device-side steps BETWEEN the three hot kernels of a ray query
// (march -> density query -> visibility pruning -> full query -> composite), which the reference leaves to chains of
// ATen ops with host syncs in between (nr3d_lib/graphics/raymarch/occgrid_raymarch.py:25-112 post-processing,
// nr3d_lib/graphics/nerf/nerf_ray_query.py:117-137 + nerf_utils.py:23-24,64-98 pruning).  In the full loop on MI355X
// those chains were ~110 launches and 20 % of the GPU time of an iteration (profiles/r02z_full_loop_kernel_stats.txt);
// here each chain is one to three launches:
//   nr3d_tau_to_alpha_fwd / _bwd     alpha = 1 - exp(-sigma * delta) and its gradient: one launch each way
//   nr3d_march_finish_rays           rays that got samples -> (ray index, int64 pack_infos) compacted, + their number
//   nr3d_march_finish_samples        per sample: int64 ray index, delta = t1 - t0, position o + d * t0
//   nr3d_prune_compact_packs         kept-sample counts -> new begin of every pack, packs that keep >= 1 sample
//                                    compacted (index, int64 pack_infos), + both totals
//   nr3d_prune_compact_samples       kept samples of every pack moved to their compact positions: sample index, depth,
//                                    delta, position, ray index in ONE pass
// All integer outputs are bit-exact with the op chains they replace (same order: ascending pack, ascending sample).
#include "common.h"
#include "scan.h"
#include "compact.h"
#include "../../include/nr3d_hip.h"

namespace nr3d {
namespace glue {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_tau_to_alpha_fwd(uint64_t S, const float *__restrict__ sigma,
                                                             const float *__restrict__ delta, float *__restrict__ alpha) {
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i < S) alpha[i] = 1.0f - expf(-(sigma[i] * delta[i]));
}

// d alpha / d sigma = delta * exp(-sigma * delta)
__global__ __launch_bounds__(kBlock) void k_tau_to_alpha_bwd(uint64_t S, const float *__restrict__ sigma,
                                                             const float *__restrict__ delta, const float *__restrict__ g_alpha,
                                                             float *__restrict__ g_sigma) {
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i < S) g_sigma[i] = g_alpha[i] * expf(-(sigma[i] * delta[i])) * delta[i];
}

// ------------------------------------------------------------------------------------------------
// per-sample epilogue of the marcher
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_finish_samples(uint64_t S, const float *__restrict__ rays_o,
                                                           const float *__restrict__ rays_d, const int32_t *__restrict__ ridx,
                                                           const float *__restrict__ t0, const float *__restrict__ t1,
                                                           int64_t *__restrict__ ridx64, float *__restrict__ deltas,
                                                           float *__restrict__ samples) {
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= S) return;
	const int32_t r = ridx[i];
	const float a = t0[i];
	if (ridx64) ridx64[i] = (int64_t)r;
	if (deltas) deltas[i] = t1[i] - a;
	if (samples) {
		const float *o = rays_o + (size_t)r * 3, *d = rays_d + (size_t)r * 3;
		// torch.addcmul(o, d, t) is o + d * t with the product contracted into the add by the ATen kernel
#pragma unroll
		for (int k = 0; k < 3; ++k) samples[i * 3 + k] = __fmaf_rn(d[k], a, o[k]);
	}
}

// ------------------------------------------------------------------------------------------------
// kept samples of every pack -> compact positions; one wave per pack, ranks from ballots (ascending sample order)
// ------------------------------------------------------------------------------------------------
constexpr int kWaves = kBlock / 64;
__global__ __launch_bounds__(kBlock) void k_compact_samples(uint32_t P, const int64_t *__restrict__ pi,
                                                            const int64_t *__restrict__ begin_all,
                                                            const uint8_t *__restrict__ selector, const float *__restrict__ f1,
                                                            const float *__restrict__ f2, const float *__restrict__ f3,
                                                            const int64_t *__restrict__ l1, int64_t *__restrict__ pidx,
                                                            float *__restrict__ f1o, float *__restrict__ f2o,
                                                            float *__restrict__ f3o, int64_t *__restrict__ l1o) {
	const uint32_t p = blockIdx.x * kWaves + (threadIdx.x >> 6);
	if (p >= P) return;
	const uint32_t lane = threadIdx.x & 63u;
	const uint64_t begin = (uint64_t)pi[2 * (size_t)p], len = (uint64_t)pi[2 * (size_t)p + 1];
	uint64_t out = (uint64_t)begin_all[p];
	for (uint64_t base = 0; base < len; base += 64) {
		const uint64_t i = begin + base + lane;
		const bool keep = (base + lane < len) && selector[i] != 0;
		const unsigned long long m = __ballot(keep);
		if (keep) {
			const uint64_t o = out + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
			if (pidx) pidx[o] = (int64_t)i;
			if (f1o) f1o[o] = f1[i];
			if (f2o) f2o[o] = f2[i];
			if (f3o) { f3o[o * 3] = f3[i * 3]; f3o[o * 3 + 1] = f3[i * 3 + 1]; f3o[o * 3 + 2] = f3[i * 3 + 2]; }
			if (l1o) l1o[o] = l1[i];
		}
		out += (uint64_t)__popcll(m);
	}
}

}  // namespace glue
}  // namespace nr3d

using namespace nr3d;

extern "C" int nr3d_tau_to_alpha_fwd(uint64_t S, const float *sigma, const float *delta, float *alpha, void *stream) {
	if (S == 0) return 0;
	NR3D_CHECK(sigma && delta && alpha, "tau_to_alpha: NULL tensor pointer");
	hipLaunchKernelGGL(glue::k_tau_to_alpha_fwd, dim3((uint32_t)div_up(S, (uint64_t)glue::kBlock)), dim3(glue::kBlock), 0,
	                   (hipStream_t)stream, S, sigma, delta, alpha);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_tau_to_alpha_bwd(uint64_t S, const float *sigma, const float *delta, const float *grad_alpha,
                                     float *grad_sigma, void *stream) {
	if (S == 0) return 0;
	NR3D_CHECK(sigma && delta && grad_alpha && grad_sigma, "tau_to_alpha backward: NULL tensor pointer");
	hipLaunchKernelGGL(glue::k_tau_to_alpha_bwd, dim3((uint32_t)div_up(S, (uint64_t)glue::kBlock)), dim3(glue::kBlock), 0,
	                   (hipStream_t)stream, S, sigma, delta, grad_alpha, grad_sigma);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_march_finish_rays(uint32_t n_rays, const int32_t *packed_info, int64_t *ridx_hit, int64_t *pack_infos,
                                      int64_t *totals, void *scan_tmp, void *stream) {
	NR3D_CHECK(totals && (n_rays == 0 || (packed_info && ridx_hit && pack_infos && scan_tmp)), "march_finish_rays: NULL tensor pointer");
	glue::PackWriter<int32_t, 2> w{packed_info + 1, nullptr, nullptr, ridx_hit, pack_infos};
	return glue::compact_packs<int32_t, 2>(n_rays, w, totals, scan_tmp, (hipStream_t)stream);
}

extern "C" int nr3d_march_finish_samples(uint64_t S, const float *rays_o, const float *rays_d, const int32_t *ridx,
                                         const float *t_starts, const float *t_ends, int64_t *ridx64, float *deltas,
                                         float *samples, void *stream) {
	if (S == 0) return 0;
	NR3D_CHECK(ridx && t_starts && (!deltas || t_ends) && (!samples || (rays_o && rays_d)), "march_finish_samples: NULL tensor pointer");
	hipLaunchKernelGGL(glue::k_finish_samples, dim3((uint32_t)div_up(S, (uint64_t)glue::kBlock)), dim3(glue::kBlock), 0,
	                   (hipStream_t)stream, S, rays_o, rays_d, ridx, t_starts, t_ends, ridx64, deltas, samples);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_prune_compact_packs(uint32_t P, const int64_t *counts, const int64_t *tag, int64_t *begin_all,
                                        int64_t *idx_out, int64_t *pack_infos_out, int64_t *totals, void *scan_tmp,
                                        void *stream) {
	NR3D_CHECK(totals && (P == 0 || (counts && begin_all && pack_infos_out && scan_tmp)), "prune_compact_packs: NULL tensor pointer");
	glue::PackWriter<int64_t, 1> w{counts, tag, begin_all, idx_out, pack_infos_out};
	return glue::compact_packs<int64_t, 1>(P, w, totals, scan_tmp, (hipStream_t)stream);
}

extern "C" int nr3d_prune_compact_samples(uint32_t P, const int64_t *pack_infos, const int64_t *begin_all,
                                          const uint8_t *selector, const float *f1, const float *f2, const float *f3,
                                          const int64_t *l1, int64_t *pidx, float *f1_out, float *f2_out, float *f3_out,
                                          int64_t *l1_out, void *stream) {
	if (P == 0) return 0;
	NR3D_CHECK(pack_infos && begin_all && selector, "prune_compact_samples: NULL tensor pointer");
	NR3D_CHECK((!f1_out || f1) && (!f2_out || f2) && (!f3_out || f3) && (!l1_out || l1), "prune_compact_samples: an output without its input");
	hipLaunchKernelGGL(glue::k_compact_samples, dim3(div_up(P, (uint32_t)glue::kWaves)), dim3(glue::kBlock), 0, (hipStream_t)stream,
	                   P, pack_infos, begin_all, selector, f1, f2, f3, l1, pidx, f1_out, f2_out, f3_out, l1_out);
	NR3D_LAUNCH_CHECK();
	return 0;
}
