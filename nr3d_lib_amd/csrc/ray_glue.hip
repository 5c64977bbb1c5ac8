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

// ------------------------------------------------------------------------------------------------
// spatial order of a batch of sample positions (round 5).  The samples a ray query hands to the field are ray-major: the
// samples of one ray are neighbours, the samples of the NEXT ray -- which fall into the same grid cells at every level whose
// cells are wider than the ray spacing -- are a ray's length away.  In the LoTD backward those meet as same-address LDS atomics
// (half of stage B's time in the full loop, profiles/r04t_full_loop_pair_accum_experiment.txt); ordered along a Morton curve
// they are consecutive lanes, which stage A merges into one record before anything is written.
//   key = bit-interleaved cell of the sample in a 2^b grid over the batch's own bounding box (b <= 10), stable radix sort (rsort)
//   perm[k] = the sample at position k of the order, inv[i] = the position of sample i
// ------------------------------------------------------------------------------------------------
namespace nr3d {
namespace glue {

__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t b = __float_as_uint(f); return (b >> 31) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o >> 31) ? (o & 0x7FFFFFFFu) : ~o); }

// bounds[0..2] = min, bounds[3..5] = max of the finite coordinates, as order-preserving integers (caller: 0xFF.. / 0).
// A FEW workgroups stride over the points and each ends in six atomics: agent-scope atomics on one line cost ~80 ns each behind
// the per-XCD L2s -- one set per wave of a point-sized grid took 0.56 ms for 1.7 M points.
__global__ __launch_bounds__(1024) void k_so_bounds(uint32_t n, const float *__restrict__ x, uint32_t *__restrict__ bounds) {
	__shared__ uint32_t red[16][6];
	uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
	for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < n; i += gridDim.x * 1024u) {
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			const float v = x[(size_t)i * 3 + d];
			if (fabsf(v) <= 3.0e38f) { const uint32_t o = f2ord(v); lo[d] = o < lo[d] ? o : lo[d]; hi[d] = o > hi[d] ? o : hi[d]; }
		}
	}
#pragma unroll
	for (int d = 0; d < 3; ++d) {
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
			const uint32_t a = __shfl_xor(lo[d], off, 64), b = __shfl_xor(hi[d], off, 64);
			lo[d] = a < lo[d] ? a : lo[d]; hi[d] = b > hi[d] ? b : hi[d];
		}
		if ((threadIdx.x & 63u) == 0u) { red[threadIdx.x >> 6][d] = lo[d]; red[threadIdx.x >> 6][3 + d] = hi[d]; }
	}
	__syncthreads();
	if (threadIdx.x < 6u) {
		uint32_t v = red[0][threadIdx.x];
		for (uint32_t w = 1; w < 16u; ++w) { const uint32_t t = red[w][threadIdx.x]; v = threadIdx.x < 3u ? (t < v ? t : v) : (t > v ? t : v); }
		if (threadIdx.x < 3u) atomicMin(&bounds[threadIdx.x], v); else atomicMax(&bounds[threadIdx.x], v);
	}
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {       // bit k of v -> bit 3 k (v < 2^10)
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}

__global__ __launch_bounds__(kBlock) void k_so_keys(uint32_t n, const float *__restrict__ x, const uint32_t *__restrict__ bounds, uint32_t b,
                                                    uint32_t *__restrict__ keys) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	const float cells = (float)(1u << b);
	uint32_t key = 0u;
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const float lo = ord2f(bounds[d]), hi = ord2f(bounds[3 + d]);
		const float v = x[(size_t)i * 3 + d];
		const float ext = hi - lo;
		float q = ext > 0.0f ? (v - lo) / ext * cells : 0.0f;
		q = fminf(fmaxf(q, 0.0f), cells - 1.0f);                   // NaN -> 0 (fmaxf returns the number)
		key |= spread3((uint32_t)q) << d;
	}
	keys[i] = key;
}

// the field's inputs in the order: position, ray index and (optional) the ray's view direction of the sample at every position
__global__ __launch_bounds__(kBlock) void k_so_gather_in(uint32_t n, const int32_t *__restrict__ order, const float *__restrict__ x,
                                                         const int64_t *__restrict__ ridx, const float *__restrict__ dirs,
                                                         float *__restrict__ x_s, int64_t *__restrict__ ridx_s, float *__restrict__ dirs_s) {
	const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= n) return;
	const uint32_t i = (uint32_t)order[k];
#pragma unroll
	for (int d = 0; d < 3; ++d) x_s[(size_t)k * 3 + d] = x[(size_t)i * 3 + d];
	if (ridx) {
		const int64_t r = ridx[i];
		if (ridx_s) ridx_s[k] = r;
		if (dirs) {
#pragma unroll
			for (int d = 0; d < 3; ++d) dirs_s[(size_t)k * 3 + d] = dirs[(size_t)r * 3 + d];
		}
	}
}

// rows of up to two per-sample float arrays between the order and the samples' own order:
// SCATTER: out[order[k]] = in[k] (the field's outputs back to ray order); else out[k] = in[order[k]] (their gradients)
template <bool SCATTER>
__global__ __launch_bounds__(kBlock) void k_so_rows(uint32_t n, const int32_t *__restrict__ order, const float *__restrict__ a, uint32_t wa,
                                                    float *__restrict__ a_out, const float *__restrict__ b, uint32_t wb, float *__restrict__ b_out) {
	const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= n) return;
	const uint32_t i = (uint32_t)order[k];
	const size_t src = SCATTER ? k : i, dst = SCATTER ? i : k;
	for (uint32_t c = 0; c < wa; ++c) a_out[dst * wa + c] = a[src * wa + c];
	for (uint32_t c = 0; c < wb; ++c) b_out[dst * wb + c] = b[src * wb + c];
}

static inline size_t so_align(size_t b) { return (b + 255u) / 256u * 256u; }

}  // namespace glue
}  // namespace nr3d

extern "C" uint64_t nr3d_spatial_order_tmp_bytes(uint32_t n) {
	return (uint64_t)(256u + 2u * nr3d::glue::so_align(4ull * n) + nr3d::rsort::tmp_bytes(n, 1));
}

extern "C" int nr3d_spatial_order(uint32_t n, const float *x, uint32_t bits_per_dim, int32_t *order, void *tmp, void *stream) {
	using namespace nr3d;
	if (n == 0) return 0;
	NR3D_CHECK(x && order && tmp, "spatial_order: NULL tensor pointer");
	NR3D_CHECK(n < (1u << 31), "spatial_order: %u points in one call, the limit is 2^31 - 1", n);
	NR3D_CHECK(bits_per_dim >= 1 && bits_per_dim <= 10, "spatial_order: bits_per_dim must be 1..10, got %u", bits_per_dim);
	hipStream_t st = (hipStream_t)stream;
	char *p = (char *)tmp;
	uint32_t *bounds = (uint32_t *)p; p += 256;
	uint32_t *keys = (uint32_t *)p; p += glue::so_align(4ull * n);
	uint32_t *keys_out = (uint32_t *)p; p += glue::so_align(4ull * n);
	NR3D_HIP_CHECK(hipMemsetAsync(bounds, 0xFF, 12, st));
	NR3D_HIP_CHECK(hipMemsetAsync(bounds + 3, 0, 12, st));
	const uint32_t nb = div_up(n, (uint32_t)glue::kBlock), nwg = div_up(n, 1024u);
	hipLaunchKernelGGL(glue::k_so_bounds, dim3(nwg < 128u ? nwg : 128u), dim3(1024), 0, st, n, x, bounds);
	hipLaunchKernelGGL(glue::k_so_keys, dim3(nb), dim3(glue::kBlock), 0, st, n, x, bounds, bits_per_dim, keys);
	NR3D_LAUNCH_CHECK();
	const uint32_t *kin[1] = {keys}, *vin[1] = {nullptr};
	uint32_t *kout[1] = {keys_out}, *vout[1] = {(uint32_t *)order};
	return rsort::sort_pairs(p, 1, kin, vin, kout, vout, n, nullptr, (int)(3u * bits_per_dim), st);
}

extern "C" int nr3d_order_gather_inputs(uint32_t n, const int32_t *order, const float *x, const int64_t *ridx, const float *dirs, float *x_out,
                                        int64_t *ridx_out, float *dirs_out, void *stream) {
	using namespace nr3d;
	if (n == 0) return 0;
	NR3D_CHECK(order && x && x_out && (!dirs || (ridx && dirs_out)), "order_gather_inputs: NULL tensor pointer");
	hipLaunchKernelGGL(glue::k_so_gather_in, dim3(div_up(n, (uint32_t)glue::kBlock)), dim3(glue::kBlock), 0, (hipStream_t)stream, n, order, x,
	                   ridx, dirs, x_out, ridx_out, dirs_out);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_order_move_rows(uint32_t n, const int32_t *order, int scatter, const float *a, uint32_t wa, float *a_out, const float *b,
                                    uint32_t wb, float *b_out, void *stream) {
	using namespace nr3d;
	if (n == 0) return 0;
	NR3D_CHECK(order && (wa == 0 || (a && a_out)) && (wb == 0 || (b && b_out)), "order_move_rows: NULL tensor pointer");
	const dim3 grid(div_up(n, (uint32_t)glue::kBlock)), blk(glue::kBlock);
	if (scatter) hipLaunchKernelGGL(glue::k_so_rows<true>, grid, blk, 0, (hipStream_t)stream, n, order, a, wa, a_out, b, wb, b_out);
	else hipLaunchKernelGGL(glue::k_so_rows<false>, grid, blk, 0, (hipStream_t)stream, n, order, a, wa, a_out, b, wb, b_out);
	NR3D_LAUNCH_CHECK();
	return 0;
}
