// nr3d_lib_amd/csrc/pack_ops.hip -- segmented ("packed") tensor ops for volume rendering (gfx950).
//
// Replaces nr3d_lib.bindings._pack_ops (csrc/pack_ops/pack_ops_cuda.cu, bindings pack_ops.cpp:21-58).
// pack_infos: int64 [P,2] = (begin, length).
//
// The reference runs ONE THREAD PER PACK with a sequential loop through global memory, and a host
// cumsum + .item() per op.  Here the unit of work is ONE WAVE (64 lanes) PER PACK:
//   * elementwise / reduction / scan ops: lanes stride over the pack (coalesced), wave64 shuffles for
//     the reduction / Kogge-Stone scan with a carried prefix between 64-element chunks;
//   * serial recurrences whose ROUNDING decides integer outputs (transmittance T -> compaction
//     selector; step samplers t += dt) keep the reference's exact operation order: a chunk of 64
//     values is loaded coalesced, then a wave-uniform loop walks them through lane broadcasts
//     (v_readlane), every lane tracking the same scalar state and lane j keeping the result of step j.
//     Loads/stores stay coalesced and the sequence is bit-identical to the one-thread version.
#include <type_traits>
#include "common.h"
#include "scan.h"
#include "pack_launch.h"
#include <stdlib.h>

namespace nr3d {
namespace pk {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

struct Pack { uint32_t p, begin, len; int lane; bool valid; };

__device__ __forceinline__ Pack my_pack(uint32_t P, const int64_t *__restrict__ pi) {
	Pack k;
	k.lane = threadIdx.x & 63;
	k.p = blockIdx.x * kWaves + (threadIdx.x >> 6);
	k.valid = k.p < P;
	k.begin = k.valid ? (uint32_t)pi[2 * (size_t)k.p] : 0u;
	k.len = k.valid ? (uint32_t)pi[2 * (size_t)k.p + 1] : 0u;
	return k;
}
static inline dim3 grid_for(uint32_t P) { return dim3(div_up(P, kWaves)); }

// ordered packs (begin[p+1] >= begin[p] + len[p], include/nr3d_hip.h): the wave of pack p zeroes the rows between its pack
// and the next one (the last pack: up to S), the wave of pack 0 the rows in front of it -- `out` needs no fill launch
template <typename T>
__device__ __forceinline__ void zero_rows(T *__restrict__ out, uint64_t r0, uint64_t r1, uint32_t w, int lane) {
	for (uint64_t e = r0 * w + lane; e < r1 * w; e += 64) out[e] = (T)0;
}
template <typename T>
__device__ __forceinline__ void fill_gaps(const Pack &k, uint32_t P, const int64_t *__restrict__ pi, uint64_t S, uint32_t w,
                                          T *__restrict__ out) {
	const uint64_t end = (uint64_t)k.begin + k.len;
	const uint64_t next = (k.p + 1 < P) ? (uint64_t)pi[2 * (size_t)(k.p + 1)] : S;
	if (next > end) zero_rows<T>(out, end, next, w, k.lane);
	if (k.p == 0 && k.begin > 0) zero_rows<T>(out, 0, k.begin, w, k.lane);
}

template <typename T> __device__ __forceinline__ T shfl_up_t(T v, int off) { return __shfl_up(v, off, 64); }
template <typename T> __device__ __forceinline__ T shfl_t(T v, int src) { return __shfl(v, src, 64); }
template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int m) { return __shfl_xor(v, m, 64); }

// L2-served load (agent scope: bypasses this CU's vector L1, which may hold a line another lane's atomic
// or store has since changed)
__device__ __forceinline__ int64_t ld_l2(const int64_t *p) {
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T> __device__ __forceinline__ T muladd(T a, T b, T c) { return a * b + c; }
template <> __device__ __forceinline__ float muladd<float>(float a, float b, float c) { return __fmaf_rn(a, b, c); }
template <> __device__ __forceinline__ double muladd<double>(double a, double b, double c) { return fma(a, b, c); }

// ------------------------------------------------------------------------------------------------
// interleave_arange / interleave_linstep
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_linstep(uint32_t P, const int64_t *__restrict__ pi,
                                                    const T *__restrict__ starts, const T *__restrict__ steps, T start,
                                                    T step, T *__restrict__ out, int64_t *__restrict__ nidx) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	const T s0 = starts ? starts[k.p] : start;
	const T st = steps ? steps[k.p] : step;
	for (uint32_t j = k.lane; j < k.len; j += 64) {
		out[k.begin + j] = muladd<T>((T)j, st, s0);
		if (nidx) nidx[k.begin + j] = (int64_t)k.p;
	}
}

// ------------------------------------------------------------------------------------------------
// depth-proportional step samplers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp_lu(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }

__global__ __launch_bounds__(kBlock) void k_sample_count(uint32_t P, const float *__restrict__ nears,
                                                         const float *__restrict__ fars, uint32_t max_steps, float g,
                                                         float lo, float hi, int64_t *__restrict__ n_per_pack) {
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= P) return;
	float t = nears[p];
	const float far = fars[p];
	uint32_t n = 0;
	while (t <= far && n < max_steps) { t += clamp_lu(t * g, lo, hi); ++n; }
	n_per_pack[p] = (int64_t)n;
}

__global__ __launch_bounds__(kBlock) void k_sample_emit(uint32_t P, const float *__restrict__ nears,
                                                        const int64_t *__restrict__ pi, float g, float lo, float hi,
                                                        float *__restrict__ ts, float *__restrict__ dts,
                                                        int64_t *__restrict__ nidx) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	float t = nears[k.p];                       // wave-uniform state
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		float my_t = 0.f, my_dt = 0.f;
		for (uint32_t j = 0; j < n; ++j) {      // serial recurrence, identical on every lane
			const float dt = clamp_lu(t * g, lo, hi);
			if ((uint32_t)k.lane == j) { my_t = t; my_dt = dt; }
			t += dt;
		}
		if ((uint32_t)k.lane < n) {
			ts[k.begin + base + k.lane] = my_t;
			dts[k.begin + base + k.lane] = my_dt;
			nidx[k.begin + base + k.lane] = (int64_t)k.p;
		}
	}
}

// segments variant: data-dependent segment walk, one lane per pack
template <bool EMIT>
__global__ __launch_bounds__(kBlock) void k_sample_segments(uint32_t P, const float *__restrict__ nears,
                                                            const float *__restrict__ fars,
                                                            const float *__restrict__ entries,
                                                            const float *__restrict__ exits,
                                                            const int64_t *__restrict__ spi, uint32_t max_steps, float g,
                                                            float lo, float hi, int64_t *__restrict__ n_per_pack,
                                                            const int64_t *__restrict__ pi, float *__restrict__ ts,
                                                            float *__restrict__ dts, int64_t *__restrict__ nidx,
                                                            int64_t *__restrict__ sidx) {
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= P) return;
	const float near = nears[p], far = fars[p];
	const uint32_t sb = (uint32_t)spi[2 * (size_t)p], se = sb + (uint32_t)spi[2 * (size_t)p + 1];
	const uint32_t b = EMIT ? (uint32_t)pi[2 * (size_t)p] : 0u;
	const uint32_t lim = EMIT ? (uint32_t)pi[2 * (size_t)p + 1] : max_steps;
	float t = near;
	uint32_t step = 0;
	for (uint32_t i = sb; i < se; ++i) {
		const float ce = entries[i], cx = exits[i];
		if (ce >= far || cx <= near) break;
		do { t += lo; } while (t < ce);
		while (t <= cx && t <= far && step < lim) {
			const float dt = clamp_lu(t * g, lo, hi);
			if (EMIT) { ts[b + step] = t; nidx[b + step] = (int64_t)p; sidx[b + step] = (int64_t)i; dts[b + step] = dt; }
			t += dt;
			++step;
		}
	}
	if (!EMIT) n_per_pack[p] = (int64_t)step;
}

// interleave_arange's step counts, ceil((stop - start) / step) as int64 -- the four ATen launches of the reference wrapper
// (graphics/pack_ops/pack_ops.py: stop.subtract(start).div(step).ceil().long()) in one, with ATen's arithmetic: the difference in
// the tensors' own type, the quotient in float32 (integer and float32 tensors; true division) or float64, IEEE division
template <typename T>
__global__ __launch_bounds__(kBlock) void k_arange_counts(uint32_t P, const T *__restrict__ start, const T *__restrict__ stop,
                                                          const T *__restrict__ step_t, double step_s, int64_t *__restrict__ n) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= P) return;
	const T d = stop[i] - start[i];
	if constexpr (sizeof(T) == 8 && !std::is_integral<T>::value) {
		const double sv = step_t ? (double)step_t[i] : step_s;
		n[i] = (int64_t)ceil((double)d / sv);
	} else {
		const float sv = step_t ? (float)step_t[i] : (float)step_s;
		n[i] = (int64_t)ceilf((float)d / sv);
	}
}

// ------------------------------------------------------------------------------------------------
// packed_sum
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sum(uint32_t P, uint32_t fd, const T *__restrict__ in,
                                                const int64_t *__restrict__ pi, T *__restrict__ out) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	for (uint32_t j = 0; j < fd; ++j) {
		T acc = (T)0;
		for (uint32_t i = k.lane; i < k.len; i += 64) acc += in[(size_t)(k.begin + i) * fd + j];
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) acc += shfl_xor_t<T>(acc, m);
		if (k.lane == 0) out[(size_t)k.p * fd + j] = acc;
	}
}

// ------------------------------------------------------------------------------------------------
// packed_cumsum / packed_cumprod.  mode: 0 sum, 1 prod (reference semantics: exclusive prod == 0),
// 2 prod with the documented exclusive semantics (identity 1 first)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_scan(uint32_t P, uint32_t fd, const T *__restrict__ in,
                                                 const int64_t *__restrict__ pi, int mode, int exclusive, int reverse,
                                                 uint64_t S, int ordered, T *__restrict__ out) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	if (ordered) fill_gaps<T>(k, P, pi, S, fd, out);
	if (k.len == 0) return;
	const bool prod = mode != 0;
	const T ident = prod ? (T)1 : (T)0;
	for (uint32_t j = 0; j < fd; ++j) {
		if (mode == 1 && exclusive) {            // reference quirk: first element stays 0 and poisons the pack
			for (uint32_t i = k.lane; i < k.len; i += 64) out[(size_t)(k.begin + i) * fd + j] = (T)0;
			continue;
		}
		T carry = ident;
		for (uint32_t base = 0; base < k.len; base += 64) {
			const uint32_t pos = base + k.lane;                      // position in scan order
			const bool ok = pos < k.len;
			const uint32_t idx = reverse ? (k.len - 1 - pos) : pos;   // position in memory
			const T v = ok ? in[(size_t)(k.begin + idx) * fd + j] : ident;
			T inc = v;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const T t = shfl_up_t<T>(inc, off);
				if (k.lane >= off) inc = prod ? inc * t : inc + t;
			}
			const T incl = prod ? carry * inc : carry + inc;
			T r;
			if (!exclusive) r = incl;
			else {
				const T prev = shfl_up_t<T>(incl, 1);
				r = (k.lane == 0) ? carry : prev;
			}
			if (ok) out[(size_t)(k.begin + idx) * fd + j] = r;
			carry = shfl_t<T>(incl, 63);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// packed_diff / packed_backward_diff
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_diff(uint32_t P, uint32_t fd, const T *__restrict__ in,
                                                 const int64_t *__restrict__ pi, const T *__restrict__ edge_a,
                                                 const T *__restrict__ edge_fill, int backward, uint64_t S, int ordered,
                                                 T *__restrict__ out) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	if (ordered) fill_gaps<T>(k, P, pi, S, fd, out);
	if (k.len == 0) return;
	const uint32_t total = k.len * fd;
	for (uint32_t e = k.lane; e < total; e += 64) {
		const uint32_t i = e / fd, j = e - i * fd;
		const size_t at = (size_t)(k.begin + i) * fd + j;
		T r;
		if (!backward) {
			if (i + 1 < k.len) r = in[at + fd] - in[at];
			else r = edge_a ? (T)(edge_a[(size_t)k.p * fd + j] - in[at]) : (edge_fill ? edge_fill[(size_t)k.p * fd + j] : (T)0);
		} else {
			if (i > 0) r = in[at] - in[at - fd];
			else r = edge_a ? (T)(in[at] - edge_a[(size_t)k.p * fd + j]) : (edge_fill ? edge_fill[(size_t)k.p * fd + j] : (T)0);
		}
		out[at] = r;
	}
}

// ------------------------------------------------------------------------------------------------
// per-pack broadcast binary ops / comparisons / matmul
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_binary(uint32_t P, uint32_t fd, const T *__restrict__ in,
                                                   const T *__restrict__ other, const int64_t *__restrict__ pi, int op,
                                                   uint64_t S, int ordered, T *__restrict__ out, uint8_t *__restrict__ out_b) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	if (ordered) {
		if (op >= 5) fill_gaps<uint8_t>(k, P, pi, S, fd, out_b);
		else fill_gaps<T>(k, P, pi, S, fd, out);
	}
	const uint32_t total = k.len * fd;
	for (uint32_t e = k.lane; e < total; e += 64) {
		const uint32_t j = e % fd;
		const size_t at = (size_t)k.begin * fd + e;
		const T a = in[at], o = other[(size_t)k.p * fd + j];
		switch (op) {
		case 0: out[at] = a + o; break;
		case 1: out[at] = a - o; break;
		case 2: out[at] = a * o; break;
		case 3: out[at] = a / o; break;
		case 5: out_b[at] = a > o; break;
		case 6: out_b[at] = a >= o; break;
		case 7: out_b[at] = a < o; break;
		case 8: out_b[at] = a <= o; break;
		case 9: out_b[at] = a == o; break;
		default: out_b[at] = a != o; break;
		}
	}
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_matmul(uint32_t P, uint32_t fd, uint32_t od, const T *__restrict__ in,
                                                   const T *__restrict__ other, const int64_t *__restrict__ pi,
                                                   uint64_t S, int ordered, T *__restrict__ out) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	if (ordered) fill_gaps<T>(k, P, pi, S, od, out);
	const T *o = other + (size_t)k.p * od * fd;
	const uint32_t total = k.len * od;
	for (uint32_t e = k.lane; e < total; e += 64) {
		const uint32_t i = e / od, j = e - i * od;
		T r = (T)0;
		for (uint32_t c = 0; c < fd; ++c) r = muladd<T>(in[(size_t)(k.begin + i) * fd + c], o[(size_t)j * fd + c], r);
		out[(size_t)(k.begin + i) * od + j] = r;
	}
}

// ------------------------------------------------------------------------------------------------
// lower-bound searches
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint32_t lower_bound(T val, const T *__restrict__ data, uint32_t length) {
	uint32_t first = 0, count = length;
	while (count > 0) {
		const uint32_t step = count >> 1, it = first + step;
		if (data[it] < val) { first = it + 1; count -= step + 1; } else count = step;
	}
	return first;
}
template <typename T>
__device__ __forceinline__ uint32_t lower_bound_clamped(T val, const T *__restrict__ data, uint32_t length) {
	if (length == 0) return 0;
	return min(lower_bound<T>(val, data, length), length - 1);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_searchsorted(uint32_t P, const T *__restrict__ bins,
                                                         const T *__restrict__ vals, const int64_t *__restrict__ pi,
                                                         uint32_t n_search, const int64_t *__restrict__ vpi,
                                                         int64_t *__restrict__ pidx) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	uint32_t ob = k.p * n_search, n = n_search;
	if (vpi) { ob = (uint32_t)vpi[2 * (size_t)k.p]; n = (uint32_t)vpi[2 * (size_t)k.p + 1]; }
	for (uint32_t i = k.lane; i < n; i += 64)
		pidx[ob + i] = (int64_t)(k.begin + lower_bound_clamped<T>(vals[ob + i], bins + k.begin, k.len));
}

__global__ __launch_bounds__(kBlock) void k_invert_cdf(uint32_t P, const float *__restrict__ bins,
                                                       const float *__restrict__ cdfs, const int64_t *__restrict__ pi,
                                                       const float *__restrict__ u_vals, uint32_t n_sample,
                                                       float *__restrict__ samples, int64_t *__restrict__ bin_idx) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	const float *bn = bins + k.begin, *cd = cdfs + k.begin;
	const uint32_t ob = k.p * n_sample;
	for (uint32_t i = k.lane; i < n_sample; i += 64) {
		const float u = u_vals[ob + i];
		const uint32_t pos = lower_bound_clamped<float>(u, cd, k.len);
		bin_idx[ob + i] = (int64_t)(pos + k.begin);
		float s;
		if (pos == 0) s = bn[0];
		else {
			const float c0 = cd[pos - 1], pmf = cd[pos] - c0;
			s = pmf < 1.0e-5f ? bn[pos - 1] : __fmaf_rn((u - c0) / pmf, bn[pos] - bn[pos - 1], bn[pos - 1]);
		}
		samples[ob + i] = s;
	}
}

// ------------------------------------------------------------------------------------------------
// try_merge_two_packs_sorted_aligned: positions of a's and b's elements in the merged pack.
//   lb_j      = lower_bound of b_j in a
//   pidx_a[i] = out_begin + i + #{j : lb_j <= i}
//   pidx_b[j] = (lb_j == 0 ? out_begin : pidx_a[lb_j - 1] + 1) + (rank of j inside its run of equal
//               consecutive lb values)          -- reference semantics, pack_ops_cuda.cu:1538-1570
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_merge(uint32_t P, const T *__restrict__ va_, const int64_t *__restrict__ pia,
                                                  const T *__restrict__ vb_, const int64_t *__restrict__ pib,
                                                  const int64_t *__restrict__ pim, int64_t *__restrict__ pa_,
                                                  int64_t *__restrict__ pb_) {
	const Pack k = my_pack(P, pia);
	if (!k.valid) return;
	const uint32_t bb = (uint32_t)pib[2 * (size_t)k.p], bl = (uint32_t)pib[2 * (size_t)k.p + 1];
	const int64_t ob = pim[2 * (size_t)k.p];
	const T *va = va_ + k.begin, *vb = vb_ + bb;
	int64_t *pa = pa_ + k.begin, *pb = pb_ + bb;
	// 0. the histogram below counts into the pack's own rows of pa: zero them here, so that the caller can hand over a buffer
	//    filled with -1 -- the value the reference leaves in rows that belong to no pack (round-4 advisor: a zero-filled buffer
	//    gave such rows position 0)
	for (uint32_t i = k.lane; i < k.len; i += 64) pa[i] = 0;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the zeros have reached L2 before any atomic of this wave
	// 1. lower bounds + histogram
	for (uint32_t j = k.lane; j < bl; j += 64) {
		const uint32_t i = lower_bound<T>(vb[j], va, k.len);
		pb[j] = (int64_t)i;
		if (i < k.len) atomicAdd((unsigned long long *)&pa[i], 1ull);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // atomics of this wave have reached L2
	// 2. inclusive scan of (count + 1) over a, offset so that pa[0] = ob + count_0
	int64_t carry = ob - 1;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t i = base + k.lane;
		int64_t inc = (i < k.len) ? ld_l2(&pa[i]) + 1 : 0;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const int64_t t = shfl_up_t<int64_t>(inc, off);
			if (k.lane >= off) inc += t;
		}
		if (i < k.len) pa[i] = carry + inc;
		carry += shfl_t<int64_t>(inc, 63);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // pa stores of this wave have reached L2
	// 3. b positions: run-rank via a max-scan of run starts
	int64_t run_carry = 0;      // start index of the run that is open at the chunk boundary
	int64_t prev_last = -1;     // lb of the last element of the previous chunk
	for (uint32_t base = 0; base < bl; base += 64) {
		const uint32_t j = base + k.lane;
		const bool ok = j < bl;
		const int64_t i = ok ? pb[j] : -2;
		int64_t left = shfl_up_t<int64_t>(i, 1);
		if (k.lane == 0) left = prev_last;
		const bool starts = ok && (i != left);
		int64_t st = starts ? (int64_t)j : -1;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const int64_t t = shfl_up_t<int64_t>(st, off);
			if (k.lane >= off) st = max(st, t);
		}
		if (st < 0) st = run_carry;
		if (ok) {
			const int64_t rank = (int64_t)j - st;
			pb[j] = rank + ((i == 0) ? ob : ld_l2(&pa[i - 1]) + 1);
		}
		const uint32_t last = min(63u, bl - base - 1);
		run_carry = shfl_t<int64_t>(st, (int)last);
		prev_last = shfl_t<int64_t>(i, (int)last);
	}
}

// ------------------------------------------------------------------------------------------------
// packed sort, one LANE per pack: ascending, in place, optional id permutation; heapsort (no auxiliary stack buffer), order
// of equal keys unspecified (the reference's quicksort is unstable as well).  Kept behind NR3D_OPT_SORT_WAVE = 0 (cross-check)
// and as the fallback for packs the wave kernel below cannot hold in registers.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void sift_down(T *v, int64_t *ids, int64_t start, int64_t end) {
	int64_t root = start;
	while (2 * root + 1 <= end) {
		int64_t child = 2 * root + 1, sw = root;
		if (v[sw] < v[child]) sw = child;
		if (child + 1 <= end && v[sw] < v[child + 1]) sw = child + 1;
		if (sw == root) return;
		{ const T t = v[root]; v[root] = v[sw]; v[sw] = t; }
		if (ids) { const int64_t t = ids[root]; ids[root] = ids[sw]; ids[sw] = t; }
		root = sw;
	}
}

// ------------------------------------------------------------------------------------------------
// merge_two_packs_sorted (graphics/pack_ops/pack_ops.py:611-640 of the reference: torch.unique + nonzero + index arithmetic): the
// UNION of two sorted, unique pack-id lists as aligned pack descriptors -- union pack k = (a's pack or an empty one, b's pack or an
// empty one) -- so that the aligned merge kernel serves packs present in a only, in b only or in both.
//   only_b[j] = id_b[j] not in a;   ob[j] = #{j' < j : only_b[j']} (scan);   position of a's pack i = i + ob[lower_bound_b(id_a[i])],
//   of b-only pack j = lower_bound_a(id_b[j]) + ob[j]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_union_flags(uint32_t Pa, const int64_t *__restrict__ ida, uint32_t Pb,
                                                        const int64_t *__restrict__ idb, int64_t *__restrict__ only_b) {
	const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
	if (j >= Pb) return;
	const int64_t id = idb[j];
	const uint32_t i = lower_bound<int64_t>(id, ida, Pa);
	only_b[j] = (i < Pa && ida[i] == id) ? 0 : 1;
}
__global__ __launch_bounds__(kBlock) void k_union_build(uint32_t Pa, const int64_t *__restrict__ ida, const int64_t *__restrict__ pia,
                                                        uint32_t Pb, const int64_t *__restrict__ idb, const int64_t *__restrict__ pib,
                                                        const int64_t *__restrict__ only_b, const int64_t *__restrict__ ob,
                                                        const int64_t *__restrict__ n_only, int64_t *__restrict__ u,
                                                        int64_t *__restrict__ pia_u, int64_t *__restrict__ pib_u, int64_t *__restrict__ n_u,
                                                        int64_t *__restrict__ total_u) {
	const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
	if (t == 0) total_u[0] = (int64_t)Pa + n_only[0];
	if (t < Pa) {
		const int64_t id = ida[t];
		const uint32_t j = lower_bound<int64_t>(id, idb, Pb);
		const bool both = j < Pb && idb[j] == id;
		const size_t pos = (size_t)t + (size_t)(j < Pb ? ob[2 * (size_t)j] : n_only[0]);
		u[pos] = id;
		pia_u[2 * pos] = pia[2 * (size_t)t]; pia_u[2 * pos + 1] = pia[2 * (size_t)t + 1];
		pib_u[2 * pos] = both ? pib[2 * (size_t)j] : 0; pib_u[2 * pos + 1] = both ? pib[2 * (size_t)j + 1] : 0;
		n_u[pos] = pia[2 * (size_t)t + 1] + (both ? pib[2 * (size_t)j + 1] : 0);
	} else if (t < Pa + Pb) {
		const uint32_t j = t - Pa;
		if (!only_b[j]) return;
		const int64_t id = idb[j];
		const size_t pos = (size_t)lower_bound<int64_t>(id, ida, Pa) + (size_t)ob[2 * (size_t)j];
		u[pos] = id;
		pia_u[2 * pos] = 0; pia_u[2 * pos + 1] = 0;
		pib_u[2 * pos] = pib[2 * (size_t)j]; pib_u[2 * pos + 1] = pib[2 * (size_t)j + 1];
		n_u[pos] = pib[2 * (size_t)j + 1];
	}
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_sort(uint32_t P, T *__restrict__ vals, int64_t *__restrict__ ids,
                                                 const int64_t *__restrict__ pi) {
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= P) return;
	const int64_t b = pi[2 * (size_t)p], n = pi[2 * (size_t)p + 1];
	if (n < 2) return;
	T *v = vals + b;
	int64_t *id = ids ? ids + b : nullptr;
	for (int64_t s = (n - 2) / 2; s >= 0; --s) sift_down<T>(v, id, s, n - 1);
	for (int64_t e = n - 1; e > 0; --e) {
		{ const T t = v[e]; v[e] = v[0]; v[0] = t; }
		if (id) { const int64_t t = id[e]; id[e] = id[0]; id[0] = t; }
		sift_down<T>(v, id, 0, e - 1);
	}
}

// ------------------------------------------------------------------------------------------------
// packed sort, one WAVE per pack (round 4; reference kernel_packed_sort_qsort pack_ops_cuda.cu:2634-2763: one thread per pack,
// quicksort with an explicit stack in global memory -- 334 us for 4096 packs x 32..64, 12 ms for 4096 x 320..640 on its GPU).
// A pack of up to 2048 elements lives in the wave's REGISTERS, element i = r * 64 + lane (K = 1 ... 32 registers per lane,
// loads and stores are 256-byte rows), and goes through a bitonic network: compare-exchange partners at distance >= 64 are
// two registers of the same lane (static indices, no data movement), partners at distance < 64 come through a wave shuffle.
// What is compared is (order-preserving unsigned image of the key, position in the pack): all composites are distinct, so the
// network's result is THE stable ascending order -- deterministic, and equal to a stable argsort (the reference's quicksort
// leaves the order of equal keys unspecified).  Floats order by their IEEE total order (-0 before +0, NaNs with the sign bit
// clear last), padding is the all-ones composite.  Packs longer than 2048 (8-byte keys: 1024) fall back to the heapsort above on
// lane 0.
// ------------------------------------------------------------------------------------------------
namespace srt {
struct E32 {                                        // 4-byte keys: one 64-bit composite, key << 32 | position
	uint64_t c;
	static __device__ __forceinline__ bool lt(const E32 &a, const E32 &b) { return a.c < b.c; }
	static __device__ __forceinline__ E32 shfl_xor(const E32 &a, int m) { E32 r; r.c = __shfl_xor(a.c, m, 64); return r; }
	static __device__ __forceinline__ E32 pad() { E32 r; r.c = ~0ull; return r; }
	__device__ __forceinline__ uint32_t pos() const { return (uint32_t)c; }
	static __device__ __forceinline__ E32 load(const void *vals, uint64_t at, uint32_t j, int is_float) {
		const uint32_t b = reinterpret_cast<const uint32_t *>(vals)[at];
		const uint32_t k = is_float ? (b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u)) : (b ^ 0x80000000u);
		E32 r; r.c = ((uint64_t)k << 32) | j; return r;
	}
	__device__ __forceinline__ void store(void *vals, uint64_t at, int is_float) const {
		const uint32_t k = (uint32_t)(c >> 32);
		reinterpret_cast<uint32_t *>(vals)[at] = is_float ? ((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k) : (k ^ 0x80000000u);
	}
};
struct E64 {                                        // 8-byte keys: (key, position) compared lexicographically
	uint64_t k; uint32_t i;
	static __device__ __forceinline__ bool lt(const E64 &a, const E64 &b) { return a.k < b.k || (a.k == b.k && a.i < b.i); }
	static __device__ __forceinline__ E64 shfl_xor(const E64 &a, int m) { E64 r; r.k = __shfl_xor(a.k, m, 64); r.i = __shfl_xor(a.i, m, 64); return r; }
	static __device__ __forceinline__ E64 pad() { E64 r; r.k = ~0ull; r.i = ~0u; return r; }
	__device__ __forceinline__ uint32_t pos() const { return i; }
	static __device__ __forceinline__ E64 load(const void *vals, uint64_t at, uint32_t j, int is_float) {
		const uint64_t b = reinterpret_cast<const uint64_t *>(vals)[at];
		E64 r; r.i = j;
		r.k = is_float ? (b ^ ((uint64_t)((int64_t)b >> 63) | 0x8000000000000000ull)) : (b ^ 0x8000000000000000ull);
		return r;
	}
	__device__ __forceinline__ void store(void *vals, uint64_t at, int is_float) const {
		reinterpret_cast<uint64_t *>(vals)[at] = is_float ? ((k & 0x8000000000000000ull) ? (k ^ 0x8000000000000000ull) : ~k)
		                                                  : (k ^ 0x8000000000000000ull);
	}
};

template <typename E, int K>
__device__ __forceinline__ void bitonic(E (&v)[K], uint32_t lane) {
	constexpr int N = 64 * K;
#pragma unroll
	for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
		for (int j = k >> 1; j >= 1; j >>= 1) {
			if (j >= 64) {                              // partners are registers r and r | jr of the same lane
				const int jr = j >> 6;
#pragma unroll
				for (int r = 0; r < K; ++r) {
					if (r & jr) continue;
					const bool asc = ((r << 6) & k) == 0;   // k >= 128: decided by r alone
					const E a = v[r], b = v[r | jr];
					const bool sw = E::lt(b, a) == asc;
					v[r] = sw ? b : a;
					v[r | jr] = sw ? a : b;
				}
			} else {                                    // partner = the same register of lane ^ j
#pragma unroll
				for (int r = 0; r < K; ++r) {
					const E a = v[r], b = E::shfl_xor(a, j);
					const bool asc = ((((uint32_t)r << 6) | lane) & (uint32_t)k) == 0;
					const bool keep_min = ((lane & (uint32_t)j) == 0) == asc;
					v[r] = (keep_min == E::lt(b, a)) ? b : a;
				}
			}
		}
	}
}

template <typename E, int K>
__device__ __forceinline__ void sort_pack(void *vals, int64_t *ids, uint64_t begin, uint32_t len, uint32_t lane, int is_float) {
	E v[K];
#pragma unroll
	for (int r = 0; r < K; ++r) {
		const uint32_t j = (uint32_t)r * 64u + lane;
		v[r] = j < len ? E::load(vals, begin + j, j, is_float) : E::pad();
	}
	bitonic<E, K>(v, lane);
	int64_t moved[K];
	if (ids) {                                          // ids[begin + j] <- ids[begin + source of j]: every read before any write
#pragma unroll
		for (int r = 0; r < K; ++r) {
			const uint32_t j = (uint32_t)r * 64u + lane;
			moved[r] = j < len ? ids[begin + v[r].pos()] : 0;
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
#pragma unroll
	for (int r = 0; r < K; ++r) {
		const uint32_t j = (uint32_t)r * 64u + lane;
		if (j < len) {
			v[r].store(vals, begin + j, is_float);
			if (ids) ids[begin + j] = moved[r];
		}
	}
}
}  // namespace srt

// longest pack sorted in registers: 2048 elements with 4-byte keys, 1024 with 8-byte keys (the next size would spill)
template <typename E> struct SortMax { static constexpr uint32_t value = sizeof(E) > 8 ? 1024u : 2048u; };
template <typename E, typename T>
__global__ __launch_bounds__(kBlock) void k_sort_wave(uint32_t P, void *__restrict__ vals, int64_t *__restrict__ ids,
                                                      const int64_t *__restrict__ pi, int is_float) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t p = blockIdx.x * kWaves + (threadIdx.x >> 6);
	if (p >= P) return;
	const int64_t b = pi[2 * (size_t)p], n64 = pi[2 * (size_t)p + 1];
	if (n64 < 2) return;
	if (n64 > (int64_t)SortMax<E>::value) {             // rare: longer than the register file holds -> heapsort on one lane
		if (lane == 0) {
			T *v = reinterpret_cast<T *>(vals) + b;
			int64_t *id = ids ? ids + b : nullptr;
			for (int64_t s = (n64 - 2) / 2; s >= 0; --s) sift_down<T>(v, id, s, n64 - 1);
			for (int64_t e = n64 - 1; e > 0; --e) {
				{ const T t = v[e]; v[e] = v[0]; v[0] = t; }
				if (id) { const int64_t t = id[e]; id[e] = id[0]; id[0] = t; }
				sift_down<T>(v, id, 0, e - 1);
			}
		}
		return;
	}
	const uint32_t n = (uint32_t)n64;
	if (n <= 64) srt::sort_pack<E, 1>(vals, ids, (uint64_t)b, n, lane, is_float);
	else if (n <= 128) srt::sort_pack<E, 2>(vals, ids, (uint64_t)b, n, lane, is_float);
	else if (n <= 256) srt::sort_pack<E, 4>(vals, ids, (uint64_t)b, n, lane, is_float);
	else if (n <= 512) srt::sort_pack<E, 8>(vals, ids, (uint64_t)b, n, lane, is_float);
	else if (n <= 1024) srt::sort_pack<E, 16>(vals, ids, (uint64_t)b, n, lane, is_float);
	else if constexpr (SortMax<E>::value > 1024u) srt::sort_pack<E, 32>(vals, ids, (uint64_t)b, n, lane, is_float);
}

// ------------------------------------------------------------------------------------------------
// alpha -> volume-rendering weights (and the compaction selector / counts)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_alpha_fwd(uint32_t P, const float *__restrict__ alphas,
                                                      const int64_t *__restrict__ pi, float eps, float thre,
                                                      float *__restrict__ weights, int64_t *__restrict__ num_steps,
                                                      uint8_t *__restrict__ selector) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	float T = 1.0f;            // wave-uniform
	int cnt = 0;
	bool stopped = false;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const float a_mine = ((uint32_t)k.lane < n) ? alphas[k.begin + base + k.lane] : 0.0f;
		float w_mine = 0.0f;
		bool sel_mine = false;
		if (!stopped) {
			for (uint32_t j = 0; j < n; ++j) {
				if (T < eps) { stopped = true; break; }
				const float a = shfl_t<float>(a_mine, (int)j);
				if (a <= thre) continue;
				if ((uint32_t)k.lane == j) { w_mine = a * T; sel_mine = true; }
				T *= (1.0f - a);
				++cnt;
			}
		}
		if ((uint32_t)k.lane < n) {
			if (weights) weights[k.begin + base + k.lane] = w_mine;
			if (selector) selector[k.begin + base + k.lane] = sel_mine ? 1 : 0;
		}
	}
	if (num_steps && k.lane == 0) num_steps[k.p] = (int64_t)cnt;
}

__global__ __launch_bounds__(kBlock) void k_alpha_bwd(uint32_t P, const float *__restrict__ alphas,
                                                      const float *__restrict__ weights,
                                                      const float *__restrict__ grad_weights,
                                                      const int64_t *__restrict__ pi, float eps, float thre,
                                                      float *__restrict__ grad_alphas) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	// accum = sum_j gw_j * w_j in the reference's serial order (fma chain): the backward divides by
	// max(1 - alpha, 1e-10), so for alpha == 1 a last-bit difference in accum is amplified by 1e10
	float accum = 0.0f;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const bool mine = (uint32_t)k.lane < n;
		const float gw_mine = mine ? grad_weights[k.begin + base + k.lane] : 0.0f;
		const float w_mine = mine ? weights[k.begin + base + k.lane] : 0.0f;
		for (uint32_t j = 0; j < n; ++j)
			accum = __fmaf_rn(shfl_t<float>(gw_mine, (int)j), shfl_t<float>(w_mine, (int)j), accum);
	}
	float T = 1.0f;
	bool stopped = false;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const bool mine = (uint32_t)k.lane < n;
		const float a_mine = mine ? alphas[k.begin + base + k.lane] : 0.0f;
		const float gw_mine = mine ? grad_weights[k.begin + base + k.lane] : 0.0f;
		const float w_mine = mine ? weights[k.begin + base + k.lane] : 0.0f;
		float ga_mine = 0.0f;
		if (!stopped) {
			for (uint32_t j = 0; j < n; ++j) {
				if (T < eps) { stopped = true; break; }
				const float a = shfl_t<float>(a_mine, (int)j);
				if (a < thre) continue;
				if ((uint32_t)k.lane == j) ga_mine = __fmaf_rn(gw_mine, T, -accum) / fmaxf(1.0f - a, 1e-10f);
				accum = __fmaf_rn(-shfl_t<float>(gw_mine, (int)j), shfl_t<float>(w_mine, (int)j), accum);
				T *= (1.0f - a);
			}
		}
		if (mine) grad_alphas[k.begin + base + k.lane] = ga_mine;
	}
}

// ---- lane-per-pack variants: many short packs (>= 1 wave of packs per SIMD).  The recurrences are serial per pack
// either way; with one pack per lane a wave advances 64 packs per instruction instead of broadcasting one pack's
// samples lane by lane.  Same operation order per pack => same bits as the wave-per-pack kernels.
struct __attribute__((packed, aligned(4))) F4u { float v[4]; };     // 4 samples, dword aligned (one dwordx4 access)
struct __attribute__((packed, aligned(1))) B4u { uint8_t v[4]; };

__global__ __launch_bounds__(kBlock) void k_alpha_fwd_lpp(uint32_t P, const float *__restrict__ alphas,
                                                          const int64_t *__restrict__ pi, float eps, float thre,
                                                          float *__restrict__ weights, int64_t *__restrict__ num_steps,
                                                          uint8_t *__restrict__ selector) {
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= P) return;
	const uint32_t begin = (uint32_t)pi[2 * (size_t)p], len = (uint32_t)pi[2 * (size_t)p + 1];
	float T = 1.0f;
	int cnt = 0;
	bool stopped = false;
	auto one = [&](float a, float &w, bool &sel) {
		w = 0.0f; sel = false;
		if (stopped) return;
		if (T < eps) { stopped = true; return; }
		if (!(a <= thre)) { w = a * T; sel = true; T *= (1.0f - a); ++cnt; }
	};
	// 4 samples per memory request (the lanes' packs are in different lines); chunk c + 1 is loaded, branch-free,
	// before chunk c is processed (see k_alpha_bwd_lpp)
	const uint32_t n4 = len / 4;
	if (n4) {
		F4u a_n = *reinterpret_cast<const F4u *>(alphas + begin);
		for (uint32_t c = 0; c < n4; ++c) {
			const F4u a4 = a_n;
			a_n = *reinterpret_cast<const F4u *>(alphas + begin + 4 * min(c + 1, n4 - 1));
			F4u w4; B4u s4;
#pragma unroll
			for (int u = 0; u < 4; ++u) { bool sel; one(a4.v[u], w4.v[u], sel); s4.v[u] = sel ? 1 : 0; }
			if (weights) *reinterpret_cast<F4u *>(weights + begin + 4 * c) = w4;
			if (selector) *reinterpret_cast<B4u *>(selector + begin + 4 * c) = s4;
		}
	}
	for (uint32_t j = 4 * n4; j < len; ++j) {
		float w; bool sel;
		one(alphas[begin + j], w, sel);
		if (weights) weights[begin + j] = w;
		if (selector) selector[begin + j] = sel ? 1 : 0;
	}
	if (num_steps) num_steps[p] = (int64_t)cnt;
}

__global__ __launch_bounds__(kBlock) void k_alpha_bwd_lpp(uint32_t P, const float *__restrict__ alphas,
                                                          const float *__restrict__ weights,
                                                          const float *__restrict__ grad_weights,
                                                          const int64_t *__restrict__ pi, float eps, float thre,
                                                          float *__restrict__ grad_alphas) {
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= P) return;
	const uint32_t begin = (uint32_t)pi[2 * (size_t)p], len = (uint32_t)pi[2 * (size_t)p + 1];
	// Chunks of 4 samples; the loads of chunk c + 1 are issued (branch-free: the last chunk is simply re-read) before
	// chunk c is processed, so the serial recurrences run under the memory latency instead of after it.
	const uint32_t n4 = len / 4;
	auto ld4 = [&](const float *base, uint32_t c) { return *reinterpret_cast<const F4u *>(base + begin + 4 * c); };
	float accum = 0.0f;
	if (n4) {
		F4u g_n = ld4(grad_weights, 0), w_n = ld4(weights, 0);
		for (uint32_t c = 0; c < n4; ++c) {
			const F4u g4 = g_n, w4 = w_n;
			const uint32_t cn = min(c + 1, n4 - 1);
			g_n = ld4(grad_weights, cn); w_n = ld4(weights, cn);
#pragma unroll
			for (int u = 0; u < 4; ++u) accum = __fmaf_rn(g4.v[u], w4.v[u], accum);
		}
	}
	for (uint32_t j = 4 * n4; j < len; ++j) accum = __fmaf_rn(grad_weights[begin + j], weights[begin + j], accum);
	float T = 1.0f;
	bool stopped = false;
	auto one = [&](float a, float gw, float w) -> float {
		if (stopped) return 0.0f;
		if (T < eps) { stopped = true; return 0.0f; }
		if (a < thre) return 0.0f;
		const float ga = __fmaf_rn(gw, T, -accum) / fmaxf(1.0f - a, 1e-10f);
		accum = __fmaf_rn(-gw, w, accum);
		T *= (1.0f - a);
		return ga;
	};
	if (n4) {
		F4u a_n = ld4(alphas, 0), g_n = ld4(grad_weights, 0), w_n = ld4(weights, 0);
		for (uint32_t c = 0; c < n4; ++c) {
			const F4u a4 = a_n, g4 = g_n, w4 = w_n;
			const uint32_t cn = min(c + 1, n4 - 1);
			a_n = ld4(alphas, cn); g_n = ld4(grad_weights, cn); w_n = ld4(weights, cn);
			F4u o4;
#pragma unroll
			for (int u = 0; u < 4; ++u) o4.v[u] = one(a4.v[u], g4.v[u], w4.v[u]);
			*reinterpret_cast<F4u *>(grad_alphas + begin + 4 * c) = o4;
		}
	}
	for (uint32_t j = 4 * n4; j < len; ++j) grad_alphas[begin + j] = one(alphas[begin + j], grad_weights[begin + j], weights[begin + j]);
}

// wave-per-pack for few packs; from 2048 packs on (32 waves of lanes) one lane per pack is faster at every pack
// length measured (4096 packs x 61 samples: 8 / 23 us vs 27 / 42 us forward / backward)
// NR3D_OPT_PACK_SCAN = 0: the fused composite replays the transmittance serially (vw bit-identical to packed_alpha_to_vw)
// instead of the prefix-product kernels.  Knobs of the experiments build: PACK_SCAN_MAX = packs up to which wave-per-pack +
// scan is preferred over one lane per pack (default: always -- 4096 packs x <= 512: 8.5 / 11 us against 38 / 77 us forward /
// backward, 262144 packs: 148 / 289 against 302 / 837), PACK_LPP_MIN.
static inline bool composite_scan() { return opt::on(NR3D_OPT_PACK_SCAN); }
static inline uint32_t composite_scan_max() { return (uint32_t)NR3D_XOPT(PACK_SCAN_MAX, 0xFFFFFFFFll); }
static inline bool lane_per_pack(uint32_t P) { return P >= (uint32_t)NR3D_XOPT(PACK_LPP_MIN, 2048); }

// ------------------------------------------------------------------------------------------------
// Fused alpha composite of a packed volume buffer: the renderer's chain
//     vw = packed_alpha_to_vw(alpha); mask = packed_sum(vw); depth = packed_sum(packed_div(vw, mask + 1e-10) * t)
//     (or packed_sum(vw * t)); rgb = packed_sum(vw[:, None] * rgb)
// (nr3d_lib/models/fields/nerf/renderer_mixin.py:298-311; kernels pack_ops_cuda.cu:1735-1848, :798-861, :1960-2062) as
// ONE pass per ray forward and ONE backward, instead of 5 + 5 launches with [S]-sized temporaries in between.
// The transmittance recurrence keeps the reference's serial order (its rounding decides which samples are cut by
// early_stop_eps), so vw is bit-identical to packed_alpha_to_vw; the per-ray sums are taken as sum(vw * x) and the
// normalised depth as sum(vw * t) / (mask + 1e-10) -- the chain's values up to summation order (<= 1e-5 rel).
// `ray_index` (optional) scatters the per-ray results into [num_rays] outputs (rays_inds_hit) and gathers their grads.
// ------------------------------------------------------------------------------------------------
template <bool RGB>
__device__ __forceinline__ void composite_fwd_serial(const Pack &k, const float *__restrict__ alphas, const float *__restrict__ ts,
                                                     const float *__restrict__ rgb, const int64_t *__restrict__ ray_index, float eps,
                                                     float thre, int normalize, float *__restrict__ vw, float *__restrict__ mask,
                                                     float *__restrict__ depth, float *__restrict__ rgb_out) {
	float T = 1.0f;
	bool stopped = false;
	float s = 0.0f, dt = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const bool mine = (uint32_t)k.lane < n;
		const size_t i = (size_t)k.begin + base + k.lane;
		const float a_mine = mine ? alphas[i] : 0.0f;
		const float t_mine = mine ? ts[i] : 0.0f;
		float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
		if (RGB && mine) { r0 = rgb[3 * i]; r1 = rgb[3 * i + 1]; r2 = rgb[3 * i + 2]; }
		float w_mine = 0.0f;
		if (!stopped) {
			for (uint32_t j = 0; j < n; ++j) {
				if (T < eps) { stopped = true; break; }
				const float a = shfl_t<float>(a_mine, (int)j);
				if (a <= thre) continue;
				if ((uint32_t)k.lane == j) w_mine = a * T;
				T *= (1.0f - a);
			}
		}
		if (mine) vw[i] = w_mine;
		s += w_mine; dt += w_mine * t_mine;
		if (RGB) { c0 += w_mine * r0; c1 += w_mine * r1; c2 += w_mine * r2; }
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		s += shfl_xor_t<float>(s, off); dt += shfl_xor_t<float>(dt, off);
		if (RGB) { c0 += shfl_xor_t<float>(c0, off); c1 += shfl_xor_t<float>(c1, off); c2 += shfl_xor_t<float>(c2, off); }
	}
	if (k.lane == 0) {
		const size_t o = ray_index ? (size_t)ray_index[k.p] : (size_t)k.p;
		mask[o] = s;
		depth[o] = normalize ? dt / (s + 1e-10f) : dt;
		if (RGB) { rgb_out[3 * o] = c0; rgb_out[3 * o + 1] = c1; rgb_out[3 * o + 2] = c2; }
	}
}

template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_fwd(uint32_t P, const float *__restrict__ alphas, const float *__restrict__ ts,
                                                          const float *__restrict__ rgb, const int64_t *__restrict__ pi,
                                                          const int64_t *__restrict__ ray_index, float eps, float thre, int normalize,
                                                          float *__restrict__ vw, float *__restrict__ mask,
                                                          float *__restrict__ depth, float *__restrict__ rgb_out) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	composite_fwd_serial<RGB>(k, alphas, ts, rgb, ray_index, eps, thre, normalize, vw, mask, depth, rgb_out);
}

// dL/dvw of one sample from the per-ray output grads (and the optional direct grad on vw)
struct CompGrad { float gm, cd, dref, g0, g1, g2; };
__device__ __forceinline__ float comp_gw(const CompGrad &c, float t, float r0, float r1, float r2, float gv) {
	float g = __fmaf_rn(c.cd, t - c.dref, c.gm);
	g = __fmaf_rn(c.g0, r0, g); g = __fmaf_rn(c.g1, r1, g); g = __fmaf_rn(c.g2, r2, g);
	return g + gv;
}
__device__ __forceinline__ CompGrad comp_grad(size_t o, int normalize, const float *mask, const float *depth, const float *g_mask,
                                              const float *g_depth, const float *g_rgb) {
	CompGrad c;
	const float gd = g_depth ? g_depth[o] : 0.0f;
	c.gm = g_mask ? g_mask[o] : 0.0f;
	c.cd = normalize ? gd / (mask[o] + 1e-10f) : gd;
	c.dref = normalize ? depth[o] : 0.0f;
	c.g0 = g_rgb ? g_rgb[3 * o] : 0.0f; c.g1 = g_rgb ? g_rgb[3 * o + 1] : 0.0f; c.g2 = g_rgb ? g_rgb[3 * o + 2] : 0.0f;
	return c;
}

template <bool RGB>
__device__ __forceinline__ void composite_bwd_serial(const Pack &k, const float *__restrict__ alphas, const float *__restrict__ vw,
                                                     const float *__restrict__ ts, const float *__restrict__ rgb,
                                                     const int64_t *__restrict__ ray_index, float eps, float thre, int normalize,
                                                     const float *__restrict__ mask, const float *__restrict__ depth,
                                                     const float *__restrict__ g_mask, const float *__restrict__ g_depth,
                                                     const float *__restrict__ g_rgb, const float *__restrict__ g_vw,
                                                     float *__restrict__ grad_alphas, float *__restrict__ grad_t,
                                                     float *__restrict__ grad_rgb) {
	const size_t o = ray_index ? (size_t)ray_index[k.p] : (size_t)k.p;
	const CompGrad cg = comp_grad(o, normalize, mask, depth, g_mask, g_depth, g_rgb);
	// accum = sum_j gw_j * w_j, serial fma chain like packed_alpha_to_vw_backward (the sweep below divides by
	// max(1 - alpha, 1e-10), which amplifies any difference in accum)
	float accum = 0.0f;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const bool mine = (uint32_t)k.lane < n;
		const size_t i = (size_t)k.begin + base + k.lane;
		const float w_mine = mine ? vw[i] : 0.0f;
		float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
		if (RGB && mine) { r0 = rgb[3 * i]; r1 = rgb[3 * i + 1]; r2 = rgb[3 * i + 2]; }
		const float gw_mine = mine ? comp_gw(cg, ts[i], r0, r1, r2, g_vw ? g_vw[i] : 0.0f) : 0.0f;
		for (uint32_t j = 0; j < n; ++j)
			accum = __fmaf_rn(shfl_t<float>(gw_mine, (int)j), shfl_t<float>(w_mine, (int)j), accum);
	}
	float T = 1.0f;
	bool stopped = false;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const bool mine = (uint32_t)k.lane < n;
		const size_t i = (size_t)k.begin + base + k.lane;
		const float a_mine = mine ? alphas[i] : 0.0f;
		const float w_mine = mine ? vw[i] : 0.0f;
		float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
		if (RGB && mine) { r0 = rgb[3 * i]; r1 = rgb[3 * i + 1]; r2 = rgb[3 * i + 2]; }
		const float gw_mine = mine ? comp_gw(cg, ts[i], r0, r1, r2, g_vw ? g_vw[i] : 0.0f) : 0.0f;
		float ga_mine = 0.0f;
		if (!stopped) {
			for (uint32_t j = 0; j < n; ++j) {
				if (T < eps) { stopped = true; break; }
				const float a = shfl_t<float>(a_mine, (int)j);
				if (a < thre) continue;
				if ((uint32_t)k.lane == j) ga_mine = __fmaf_rn(gw_mine, T, -accum) / fmaxf(1.0f - a, 1e-10f);
				accum = __fmaf_rn(-shfl_t<float>(gw_mine, (int)j), shfl_t<float>(w_mine, (int)j), accum);
				T *= (1.0f - a);
			}
		}
		if (mine) {
			grad_alphas[i] = ga_mine;
			if (grad_t) grad_t[i] = cg.cd * w_mine;
			if (RGB && grad_rgb) { grad_rgb[3 * i] = w_mine * cg.g0; grad_rgb[3 * i + 1] = w_mine * cg.g1; grad_rgb[3 * i + 2] = w_mine * cg.g2; }
		}
	}
}

template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_bwd(uint32_t P, const float *__restrict__ alphas, const float *__restrict__ vw,
                                                          const float *__restrict__ ts, const float *__restrict__ rgb,
                                                          const int64_t *__restrict__ pi, const int64_t *__restrict__ ray_index,
                                                          float eps, float thre, int normalize, const float *__restrict__ mask,
                                                          const float *__restrict__ depth, const float *__restrict__ g_mask,
                                                          const float *__restrict__ g_depth, const float *__restrict__ g_rgb,
                                                          const float *__restrict__ g_vw, float *__restrict__ grad_alphas,
                                                          float *__restrict__ grad_t, float *__restrict__ grad_rgb) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	composite_bwd_serial<RGB>(k, alphas, vw, ts, rgb, ray_index, eps, thre, normalize, mask, depth, g_mask, g_depth, g_rgb, g_vw,
	                          grad_alphas, grad_t, grad_rgb);
}

// ------------------------------------------------------------------------------------------------
// Prefix-product forms (wave per pack, few to some thousand packs): the transmittance before sample i is the running
// product of (1 - alpha) over the samples that count, so a 64-sample chunk needs one 6-step wave scan instead of 64
// dependent broadcast-multiply steps.  The products are associated as a tree instead of left to right: T differs from the
// serial recurrence by rounding only (<= ~n * 2^-24 relative, typically 1e-6 at 512 samples; the contract's tolerance is
// 1e-5).  What must NOT depend on rounding is WHICH samples early_stop_eps cuts: a pack where some T lands within
// 2e-4 * eps of the threshold is replayed with the serial kernel body (wave-uniform branch; ~1 pack in 10^4).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_inclusive_mul(float v, int lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const float t = shfl_up_t<float>(v, off); if (lane >= off) v *= t; }
	return v;
}
__device__ __forceinline__ float wave_inclusive_add(float v, int lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const float t = shfl_up_t<float>(v, off); if (lane >= off) v += t; }
	return v;
}

// transmittance of one chunk: `counts` = the sample multiplies T; returns T before my sample, whether my sample lies behind
// the early stop, and flags a pack whose stop decision could hinge on rounding.  Tc / stopped carry across chunks.
struct ChunkT { float Tin; bool cut; };
__device__ __forceinline__ ChunkT chunk_transmittance(float a, bool mine, bool counts, float eps, int lane, float &Tc, bool &stopped,
                                                      bool &ambiguous) {
	const float f = (mine && counts) ? (1.0f - a) : 1.0f;
	const float inc = wave_inclusive_mul(f, lane);
	float exc = shfl_up_t<float>(inc, 1);
	if (lane == 0) exc = 1.0f;
	ChunkT r;
	r.Tin = Tc * exc;
	const unsigned long long below = __ballot(mine && r.Tin < eps);
	const unsigned long long near = __ballot(mine && fabsf(r.Tin - eps) <= eps * 2e-4f);
	const int first = below ? __ffsll((long long)below) - 1 : 64;
	// a near-threshold T only matters up to the first clear crossing
	if (!stopped && (near & ((first >= 63) ? ~0ull : ((2ull << first) - 1ull)))) ambiguous = true;
	r.cut = stopped || lane >= first;
	stopped = stopped || below != 0ull;
	Tc *= shfl_t<float>(inc, 63);
	return r;
}

// body of the prefix-product forward for the pack `k`.  FROM_SIGMA: the opacity is alpha = 1 - exp(-(sigma * delta)) (the expression of
// k_tau_to_alpha_fwd, ray_glue.hip: bit-identical), computed here and stored to `alphas` (an OUTPUT then: not restrict-qualified, the
// serial replay below reads back what this wave's own lanes stored); rows >= sigma_rows read sigma as 0 (the host raises afterwards).
template <bool RGB, bool FROM_SIGMA>
__device__ __forceinline__ void composite_fwd_scan_body(const Pack &k, float *alphas, const float *__restrict__ sigma,
                                                        const float *__restrict__ delta, uint64_t sigma_rows, const float *__restrict__ ts,
                                                        const float *__restrict__ rgb, const int64_t *__restrict__ ray_index, float eps,
                                                        float thre, int normalize, float *__restrict__ vw, float *__restrict__ mask,
                                                        float *__restrict__ depth, float *__restrict__ rgb_out) {
	float Tc = 1.0f, s = 0.0f, dt = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
	bool stopped = false, ambiguous = false;
	auto alpha_at = [&](size_t i) -> float {
		if (!FROM_SIGMA) return alphas[i];
		const float a = 1.0f - expf(-((i < sigma_rows ? sigma[i] : 0.0f) * delta[i]));
		alphas[i] = a;
		return a;
	};
	uint32_t base = 0;
	for (; base < k.len && !ambiguous; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const bool mine = (uint32_t)k.lane < n;
		const size_t i = (size_t)k.begin + base + k.lane;
		const float a = mine ? alpha_at(i) : 0.0f;
		const float t = mine ? ts[i] : 0.0f;
		float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
		if (RGB && mine) { r0 = rgb[3 * i]; r1 = rgb[3 * i + 1]; r2 = rgb[3 * i + 2]; }
		const bool counts = !(a <= thre);
		const ChunkT ct = chunk_transmittance(a, mine, counts, eps, k.lane, Tc, stopped, ambiguous);
		const float w = (mine && counts && !ct.cut) ? a * ct.Tin : 0.0f;
		if (mine) vw[i] = w;
		s += w; dt += w * t;
		if (RGB) { c0 += w * r0; c1 += w * r1; c2 += w * r2; }
	}
	if (ambiguous) {        // wave-uniform
		if (FROM_SIGMA)     // the chunks the loop above did not reach: their opacities first
			for (; base < k.len; base += 64) if ((uint32_t)k.lane < min(64u, k.len - base)) alpha_at((size_t)k.begin + base + k.lane);
		composite_fwd_serial<RGB>(k, alphas, ts, rgb, ray_index, eps, thre, normalize, vw, mask, depth, rgb_out);
		return;
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		s += shfl_xor_t<float>(s, off); dt += shfl_xor_t<float>(dt, off);
		if (RGB) { c0 += shfl_xor_t<float>(c0, off); c1 += shfl_xor_t<float>(c1, off); c2 += shfl_xor_t<float>(c2, off); }
	}
	if (k.lane == 0) {
		const size_t o = ray_index ? (size_t)ray_index[k.p] : (size_t)k.p;
		mask[o] = s;
		depth[o] = normalize ? dt / (s + 1e-10f) : dt;
		if (RGB) { rgb_out[3 * o] = c0; rgb_out[3 * o + 1] = c1; rgb_out[3 * o + 2] = c2; }
	}
}

template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_fwd_scan(uint32_t P, const float *__restrict__ alphas, const float *__restrict__ ts,
                                                               const float *__restrict__ rgb, const int64_t *__restrict__ pi,
                                                               const int64_t *__restrict__ ray_index, float eps, float thre,
                                                               int normalize, float *__restrict__ vw, float *__restrict__ mask,
                                                               float *__restrict__ depth, float *__restrict__ rgb_out) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	composite_fwd_scan_body<RGB, false>(k, const_cast<float *>(alphas), nullptr, nullptr, 0, ts, rgb, ray_index, eps, thre, normalize, vw,
	                                    mask, depth, rgb_out);
}

// ONE WAVE PER RAY over ALL rays of a marcher's packed_info (int32 [n_rays, 2], nr3d_march_composite_fwd): a ray without samples
// writes its zeros itself (no fill launch, no compaction of the hit rays needed before the launch), per-ray results at the ray's own index
__device__ __forceinline__ Pack my_ray(uint32_t n_rays, const int32_t *__restrict__ packed_info) {
	Pack k;
	k.lane = threadIdx.x & 63;
	k.p = blockIdx.x * kWaves + (threadIdx.x >> 6);
	k.valid = k.p < n_rays;
	k.begin = k.valid ? (uint32_t)packed_info[2 * (size_t)k.p] : 0u;
	k.len = k.valid ? (uint32_t)packed_info[2 * (size_t)k.p + 1] : 0u;
	return k;
}

template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_rays_fwd(uint32_t n_rays, const int32_t *__restrict__ packed_info,
                                                               const float *__restrict__ sigma, const float *__restrict__ delta,
                                                               uint64_t sigma_rows, const float *__restrict__ ts,
                                                               const float *__restrict__ rgb, float eps, float thre, int normalize,
                                                               float *alphas, float *__restrict__ vw, float *__restrict__ mask,
                                                               float *__restrict__ depth, float *__restrict__ rgb_out) {
	const Pack k = my_ray(n_rays, packed_info);
	if (!k.valid) return;
	composite_fwd_scan_body<RGB, true>(k, alphas, sigma, delta, sigma_rows, ts, rgb, nullptr, eps, thre, normalize, vw, mask, depth, rgb_out);
}

// the cached emit of the marcher (occ_grid.hip k_emit_cached: the same stores, the same expressions) AND the composite in one launch:
// the wave of ray i copies the ray's cached samples {t0, t1, cell} to their packed rows -- t_starts / t_ends / ridx (/ gidx / ridx64 /
// deltas / samples) -- and composites them straight away.  Row begin + j is written and read back by the same lane (j % 64).
template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_emit_composite_rays_fwd(uint32_t n_rays, const int32_t *__restrict__ packed_info,
                                                                    const uint32_t *__restrict__ cache, uint32_t cache_stride,
                                                                    const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                    float *t_starts, float *__restrict__ t_ends, int32_t *__restrict__ ridx,
                                                                    int32_t *__restrict__ gidx, int64_t *__restrict__ ridx64, float *deltas,
                                                                    float *__restrict__ samples, const float *__restrict__ sigma,
                                                                    uint64_t sigma_rows, const float *__restrict__ rgb, float eps, float thre,
                                                                    int normalize, float *alphas, float *__restrict__ vw,
                                                                    float *__restrict__ mask, float *__restrict__ depth,
                                                                    float *__restrict__ rgb_out) {
	const Pack k = my_ray(n_rays, packed_info);
	if (!k.valid) return;
	const uint32_t *c = cache + (size_t)k.p * cache_stride * 3;
	float o[3] = {0.0f, 0.0f, 0.0f}, d[3] = {0.0f, 0.0f, 0.0f};
	if (samples) {
#pragma unroll
		for (int a = 0; a < 3; ++a) { o[a] = rays_o[3 * (size_t)k.p + a]; d[a] = rays_d[3 * (size_t)k.p + a]; }
	}
	for (uint32_t j = k.lane; j < k.len; j += 64) {
		const float a = __uint_as_float(c[3 * j]), e = __uint_as_float(c[3 * j + 1]);
		const size_t r = (size_t)k.begin + j;
		t_starts[r] = a;
		t_ends[r] = e;
		ridx[r] = (int32_t)k.p;
		if (gidx) gidx[r] = (int32_t)c[3 * j + 2];
		if (ridx64) ridx64[r] = (int64_t)k.p;
		deltas[r] = e - a;
		if (samples) {
#pragma unroll
			for (int q = 0; q < 3; ++q) samples[r * 3 + q] = __fmaf_rn(d[q], a, o[q]);
		}
	}
	composite_fwd_scan_body<RGB, true>(k, alphas, sigma, deltas, sigma_rows, t_starts, rgb, nullptr, eps, thre, normalize, vw, mask, depth,
	                                   rgb_out);
}

template <bool RGB>
__device__ __forceinline__ void composite_bwd_scan_body(const Pack &k, const float *__restrict__ alphas, const float *__restrict__ vw,
                                                        const float *__restrict__ ts, const float *__restrict__ rgb,
                                                        const int64_t *__restrict__ ray_index, float eps, float thre, int normalize,
                                                        const float *__restrict__ mask, const float *__restrict__ depth,
                                                        const float *__restrict__ g_mask, const float *__restrict__ g_depth,
                                                        const float *__restrict__ g_rgb, const float *__restrict__ g_vw,
                                                        float *__restrict__ grad_alphas, float *__restrict__ grad_t,
                                                        float *__restrict__ grad_rgb) {
	const size_t o = ray_index ? (size_t)ray_index[k.p] : (size_t)k.p;
	const CompGrad cg = comp_grad(o, normalize, mask, depth, g_mask, g_depth, g_rgb);
	// total = sum_j gw_j * w_j; the sample's own "still to come" sum is total minus the prefix before it
	float total = 0.0f;
	for (uint32_t base = 0; base < k.len; base += 64) {
		const bool mine = (uint32_t)k.lane < min(64u, k.len - base);
		const size_t i = (size_t)k.begin + base + k.lane;
		if (mine) {
			float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
			if (RGB) { r0 = rgb[3 * i]; r1 = rgb[3 * i + 1]; r2 = rgb[3 * i + 2]; }
			total = __fmaf_rn(comp_gw(cg, ts[i], r0, r1, r2, g_vw ? g_vw[i] : 0.0f), vw[i], total);
		}
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) total += shfl_xor_t<float>(total, off);
	float Tc = 1.0f, qc = 0.0f;
	bool stopped = false, ambiguous = false;
	for (uint32_t base = 0; base < k.len && !ambiguous; base += 64) {
		const uint32_t n = min(64u, k.len - base);
		const bool mine = (uint32_t)k.lane < n;
		const size_t i = (size_t)k.begin + base + k.lane;
		const float a = mine ? alphas[i] : 0.0f;
		const float w = mine ? vw[i] : 0.0f;
		float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
		if (RGB && mine) { r0 = rgb[3 * i]; r1 = rgb[3 * i + 1]; r2 = rgb[3 * i + 2]; }
		const float gw = mine ? comp_gw(cg, ts[i], r0, r1, r2, g_vw ? g_vw[i] : 0.0f) : 0.0f;
		const bool counts = !(a < thre);
		const ChunkT ct = chunk_transmittance(a, mine, counts, eps, k.lane, Tc, stopped, ambiguous);
		const float q = gw * w;
		const float incq = wave_inclusive_add(q, k.lane);
		const float accum = total - (qc + (incq - q));            // sum over the samples from mine on
		qc += shfl_t<float>(incq, 63);
		if (mine) {
			grad_alphas[i] = (counts && !ct.cut) ? __fmaf_rn(gw, ct.Tin, -accum) / fmaxf(1.0f - a, 1e-10f) : 0.0f;
			if (grad_t) grad_t[i] = cg.cd * w;
			if (RGB && grad_rgb) { grad_rgb[3 * i] = w * cg.g0; grad_rgb[3 * i + 1] = w * cg.g1; grad_rgb[3 * i + 2] = w * cg.g2; }
		}
	}
	if (ambiguous)
		composite_bwd_serial<RGB>(k, alphas, vw, ts, rgb, ray_index, eps, thre, normalize, mask, depth, g_mask, g_depth, g_rgb, g_vw,
		                          grad_alphas, grad_t, grad_rgb);
}


template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_bwd_scan(uint32_t P, const float *__restrict__ alphas, const float *__restrict__ vw,
                                                               const float *__restrict__ ts, const float *__restrict__ rgb,
                                                               const int64_t *__restrict__ pi, const int64_t *__restrict__ ray_index,
                                                               float eps, float thre, int normalize, const float *__restrict__ mask,
                                                               const float *__restrict__ depth, const float *__restrict__ g_mask,
                                                               const float *__restrict__ g_depth, const float *__restrict__ g_rgb,
                                                               const float *__restrict__ g_vw, float *__restrict__ grad_alphas,
                                                               float *__restrict__ grad_t, float *__restrict__ grad_rgb) {
	const Pack k = my_pack(P, pi);
	if (!k.valid) return;
	composite_bwd_scan_body<RGB>(k, alphas, vw, ts, rgb, ray_index, eps, thre, normalize, mask, depth, g_mask, g_depth, g_rgb, g_vw,
	                             grad_alphas, grad_t, grad_rgb);
}

// backward over all rays of a marcher's packed_info (see k_composite_rays_fwd); grad_sigma (optional) = grad_alpha * exp(-(sigma *
// delta)) * delta, the expression of k_tau_to_alpha_bwd, from the grad_alphas this wave's own lanes have just stored
template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_rays_bwd(uint32_t n_rays, const int32_t *__restrict__ packed_info,
                                                               const float *__restrict__ alphas, const float *__restrict__ vw,
                                                               const float *__restrict__ ts, const float *__restrict__ rgb, float eps,
                                                               float thre, int normalize, const float *__restrict__ mask,
                                                               const float *__restrict__ depth, const float *__restrict__ g_mask,
                                                               const float *__restrict__ g_depth, const float *__restrict__ g_rgb,
                                                               float *grad_alphas, float *__restrict__ grad_t,
                                                               float *__restrict__ grad_rgb, const float *__restrict__ sigma,
                                                               const float *__restrict__ delta, uint64_t sigma_rows,
                                                               float *__restrict__ grad_sigma) {
	const Pack k = my_ray(n_rays, packed_info);
	if (!k.valid) return;
	composite_bwd_scan_body<RGB>(k, alphas, vw, ts, rgb, nullptr, eps, thre, normalize, mask, depth, g_mask, g_depth, g_rgb, nullptr,
	                             grad_alphas, grad_t, grad_rgb);
	if (grad_sigma)
		for (uint32_t base = 0; base < k.len; base += 64) {
			const size_t i = (size_t)k.begin + base + k.lane;
			if ((uint32_t)k.lane < min(64u, k.len - base))
				grad_sigma[i] = i < sigma_rows ? grad_alphas[i] * expf(-(sigma[i] * delta[i])) * delta[i] : 0.0f;
		}
}

// lane-per-pack forms (many rays): every lane walks its own ray serially, 4 samples per memory request
template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_fwd_lpp(uint32_t P, const float *__restrict__ alphas, const float *__restrict__ ts,
                                                              const float *__restrict__ rgb, const int64_t *__restrict__ pi,
                                                              const int64_t *__restrict__ ray_index, float eps, float thre,
                                                              int normalize, float *__restrict__ vw, float *__restrict__ mask,
                                                              float *__restrict__ depth, float *__restrict__ rgb_out) {
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= P) return;
	const size_t begin = (size_t)pi[2 * (size_t)p];
	const uint32_t len = (uint32_t)pi[2 * (size_t)p + 1];
	float T = 1.0f, s = 0.0f, dt = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
	bool stopped = false;
	auto one = [&](float a, float t, float r0, float r1, float r2) -> float {
		float w = 0.0f;
		if (!stopped) {
			if (T < eps) stopped = true;
			else if (!(a <= thre)) { w = a * T; T *= (1.0f - a); }
		}
		s += w; dt += w * t;
		if (RGB) { c0 += w * r0; c1 += w * r1; c2 += w * r2; }
		return w;
	};
	const uint32_t n4 = len / 4;
	auto ld4 = [&](const float *base, size_t at) { return *reinterpret_cast<const F4u *>(base + at); };
	if (n4) {
		F4u a_n = ld4(alphas, begin), t_n = ld4(ts, begin), q0 = {}, q1 = {}, q2 = {};
		if (RGB) { q0 = ld4(rgb, 3 * begin); q1 = ld4(rgb, 3 * begin + 4); q2 = ld4(rgb, 3 * begin + 8); }
		for (uint32_t c = 0; c < n4; ++c) {
			const F4u a4 = a_n, t4 = t_n, x0 = q0, x1 = q1, x2 = q2;
			const size_t nx = begin + 4 * (size_t)min(c + 1, n4 - 1);
			a_n = ld4(alphas, nx); t_n = ld4(ts, nx);
			if (RGB) { q0 = ld4(rgb, 3 * nx); q1 = ld4(rgb, 3 * nx + 4); q2 = ld4(rgb, 3 * nx + 8); }
			const float cc[12] = {x0.v[0], x0.v[1], x0.v[2], x0.v[3], x1.v[0], x1.v[1], x1.v[2], x1.v[3], x2.v[0], x2.v[1], x2.v[2], x2.v[3]};
			F4u w4;
#pragma unroll
			for (int u = 0; u < 4; ++u) w4.v[u] = one(a4.v[u], t4.v[u], cc[3 * u], cc[3 * u + 1], cc[3 * u + 2]);
			*reinterpret_cast<F4u *>(vw + begin + 4 * (size_t)c) = w4;
		}
	}
	for (uint32_t j = 4 * n4; j < len; ++j) {
		const size_t i = begin + j;
		vw[i] = one(alphas[i], ts[i], RGB ? rgb[3 * i] : 0.0f, RGB ? rgb[3 * i + 1] : 0.0f, RGB ? rgb[3 * i + 2] : 0.0f);
	}
	const size_t o = ray_index ? (size_t)ray_index[p] : (size_t)p;
	mask[o] = s;
	depth[o] = normalize ? dt / (s + 1e-10f) : dt;
	if (RGB) { rgb_out[3 * o] = c0; rgb_out[3 * o + 1] = c1; rgb_out[3 * o + 2] = c2; }
}

template <bool RGB>
__global__ __launch_bounds__(kBlock) void k_composite_bwd_lpp(uint32_t P, const float *__restrict__ alphas, const float *__restrict__ vw,
                                                              const float *__restrict__ ts, const float *__restrict__ rgb,
                                                              const int64_t *__restrict__ pi, const int64_t *__restrict__ ray_index,
                                                              float eps, float thre, int normalize, const float *__restrict__ mask,
                                                              const float *__restrict__ depth, const float *__restrict__ g_mask,
                                                              const float *__restrict__ g_depth, const float *__restrict__ g_rgb,
                                                              const float *__restrict__ g_vw, float *__restrict__ grad_alphas,
                                                              float *__restrict__ grad_t, float *__restrict__ grad_rgb) {
	const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	if (p >= P) return;
	const size_t begin = (size_t)pi[2 * (size_t)p];
	const uint32_t len = (uint32_t)pi[2 * (size_t)p + 1];
	const size_t o = ray_index ? (size_t)ray_index[p] : (size_t)p;
	const CompGrad cg = comp_grad(o, normalize, mask, depth, g_mask, g_depth, g_rgb);
	// Chunks of 4 samples (one 16-byte request per stream and chunk: the lanes' rays sit in different lines); the loads of
	// chunk c + 1 are issued, branch-free (the last chunk is simply re-read), before chunk c is processed, so the serial
	// recurrences run under the memory latency instead of after it (as k_alpha_bwd_lpp).
	struct Chunk { F4u a, w, t, gv, c0, c1, c2; };
	const uint32_t n4 = len / 4;
	auto ld4 = [&](const float *base, size_t at) { return *reinterpret_cast<const F4u *>(base + at); };
	auto load = [&](uint32_t c, bool with_alpha) {
		Chunk k = {};
		const size_t at = begin + 4 * (size_t)c;
		if (with_alpha) k.a = ld4(alphas, at);
		k.w = ld4(vw, at); k.t = ld4(ts, at);
		if (g_vw) k.gv = ld4(g_vw, at);
		if (RGB) { k.c0 = ld4(rgb, 3 * at); k.c1 = ld4(rgb, 3 * at + 4); k.c2 = ld4(rgb, 3 * at + 8); }
		return k;
	};
	auto gw_of = [&](const Chunk &k, int u) {
		const float cc[12] = {k.c0.v[0], k.c0.v[1], k.c0.v[2], k.c0.v[3], k.c1.v[0], k.c1.v[1], k.c1.v[2], k.c1.v[3],
		                      k.c2.v[0], k.c2.v[1], k.c2.v[2], k.c2.v[3]};
		return comp_gw(cg, k.t.v[u], cc[3 * u], cc[3 * u + 1], cc[3 * u + 2], g_vw ? k.gv.v[u] : 0.0f);
	};
	auto gw_at = [&](size_t i) {
		return comp_gw(cg, ts[i], RGB ? rgb[3 * i] : 0.0f, RGB ? rgb[3 * i + 1] : 0.0f, RGB ? rgb[3 * i + 2] : 0.0f, g_vw ? g_vw[i] : 0.0f);
	};
	// pass 1: accum = sum_j gw_j * w_j (serial fma chain, the order of packed_alpha_to_vw_backward)
	float accum = 0.0f;
	if (n4) {
		Chunk nx = load(0, false);
		for (uint32_t c = 0; c < n4; ++c) {
			const Chunk k = nx;
			nx = load(min(c + 1, n4 - 1), false);
#pragma unroll
			for (int u = 0; u < 4; ++u) accum = __fmaf_rn(gw_of(k, u), k.w.v[u], accum);
		}
	}
	for (uint32_t j = 4 * n4; j < len; ++j) accum = __fmaf_rn(gw_at(begin + j), vw[begin + j], accum);
	// pass 2: the transmittance sweep
	float T = 1.0f;
	bool stopped = false;
	auto one = [&](float a, float w, float gw) -> float {
		if (stopped) return 0.0f;
		if (T < eps) { stopped = true; return 0.0f; }
		if (a < thre) return 0.0f;
		const float ga = __fmaf_rn(gw, T, -accum) / fmaxf(1.0f - a, 1e-10f);
		accum = __fmaf_rn(-gw, w, accum);
		T *= (1.0f - a);
		return ga;
	};
	if (n4) {
		Chunk nx = load(0, true);
		for (uint32_t c = 0; c < n4; ++c) {
			const Chunk k = nx;
			nx = load(min(c + 1, n4 - 1), true);
			F4u ga4, gt4, r0 = {}, r1 = {}, r2 = {};
			float gc[12];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				ga4.v[u] = one(k.a.v[u], k.w.v[u], gw_of(k, u));
				gt4.v[u] = cg.cd * k.w.v[u];
				gc[3 * u] = k.w.v[u] * cg.g0; gc[3 * u + 1] = k.w.v[u] * cg.g1; gc[3 * u + 2] = k.w.v[u] * cg.g2;
			}
			const size_t at = begin + 4 * (size_t)c;
			*reinterpret_cast<F4u *>(grad_alphas + at) = ga4;
			if (grad_t) *reinterpret_cast<F4u *>(grad_t + at) = gt4;
			if (RGB && grad_rgb) {
#pragma unroll
				for (int u = 0; u < 4; ++u) { r0.v[u] = gc[u]; r1.v[u] = gc[4 + u]; r2.v[u] = gc[8 + u]; }
				*reinterpret_cast<F4u *>(grad_rgb + 3 * at) = r0;
				*reinterpret_cast<F4u *>(grad_rgb + 3 * at + 4) = r1;
				*reinterpret_cast<F4u *>(grad_rgb + 3 * at + 8) = r2;
			}
		}
	}
	for (uint32_t j = 4 * n4; j < len; ++j) {
		const size_t i = begin + j;
		const float w = vw[i];
		grad_alphas[i] = one(alphas[i], w, gw_at(i));
		if (grad_t) grad_t[i] = cg.cd * w;
		if (RGB && grad_rgb) { grad_rgb[3 * i] = w * cg.g0; grad_rgb[3 * i + 1] = w * cg.g1; grad_rgb[3 * i + 2] = w * cg.g2; }
	}
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_boundaries(uint64_t n, const T *__restrict__ ids, int32_t *__restrict__ b) {
	const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	b[i] = (i == 0) ? 1 : (ids[i - 1] == ids[i] ? 0 : 1);
}

// octree_mark_consecutive_segments (pack_ops_cuda.cu:2807-2841): per ray (pack) of octree nodes hit in order, mark where
// a run of face-adjacent nodes (L1 distance of the integer coordinates <= 1) starts and ends.  One wave per pack.
// The reference indexes the node list without the pack's offset (`point_indices[j]` instead of `[begin + j]`), which
// is only right for the first pack; the pack's own nodes are used here.
__global__ __launch_bounds__(kBlock) void k_mark_consecutive(uint32_t P, const int64_t *__restrict__ pi,
                                                             const int32_t *__restrict__ pidx, const int16_t *__restrict__ pts,
                                                             uint8_t *__restrict__ mark_start, uint8_t *__restrict__ mark_end) {
	const Pack k = my_pack(P, pi);
	if (!k.valid || k.len == 0) return;
	for (uint32_t j = k.lane; j < k.len; j += 64) {
		bool brk = (j == 0);
		if (j > 0) {
			const int16_t *a = pts + 3 * (size_t)pidx[k.begin + j - 1], *b = pts + 3 * (size_t)pidx[k.begin + j];
			brk = (abs((int)b[0] - (int)a[0]) + abs((int)b[1] - (int)a[1]) + abs((int)b[2] - (int)a[2])) > 1;
		}
		if (brk) {
			mark_start[k.begin + j] = 1;
			if (j > 0) mark_end[k.begin + j - 1] = 1;
		}
		if (j == k.len - 1) mark_end[k.begin + j] = 1;
	}
}

}  // namespace pk
}  // namespace nr3d

using namespace nr3d;

#define PK_DISPATCH(dtype, ...)                                                          \
	do {                                                                                 \
		switch (dtype) {                                                                 \
		case NR3D_F32: { using T = float; __VA_ARGS__; } break;                          \
		case NR3D_F64: { using T = double; __VA_ARGS__; } break;                         \
		case NR3D_I32: { using T = int32_t; __VA_ARGS__; } break;                        \
		case NR3D_I64: { using T = int64_t; __VA_ARGS__; } break;                        \
		default: return ::nr3d::fail("pack_ops: unsupported dtype code %d (f32/f64/i32/i64)", (int)(dtype)); \
		}                                                                                \
	} while (0)

extern "C" int nr3d_pack_infos_from_n(uint32_t P, const int64_t *n_per_pack, int64_t *pack_infos, int64_t *total,
                                      void *scan_tmp, void *stream) {
	NR3D_CHECK(total && scan_tmp, "pack_infos_from_n: NULL scratch pointer");
	return scan::pack_infos_from_counts<int64_t, int64_t>(P, n_per_pack, pack_infos, total, scan_tmp, (hipStream_t)stream);
}

extern "C" int nr3d_interleave_linstep(uint32_t P, int dtype, const int64_t *pack_infos, const void *starts,
                                       const void *step_sizes, double start_s, double step_s, void *out, int64_t *nidx,
                                       void *stream) {
	if (P == 0) return 0;
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_linstep<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P,
	                                      pack_infos, (const T *)starts, (const T *)step_sizes, (T)start_s, (T)step_s,
	                                      (T *)out, nidx));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_arange_num_steps(uint32_t P, int dtype, const void *starts, const void *stops, const void *step_sizes, double step_s,
                                     int64_t *num_steps, void *stream) {
	if (P == 0) return 0;
	NR3D_CHECK(starts && stops && num_steps, "arange_num_steps: NULL pointer");
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_arange_counts<T>, dim3(div_up(P, pk::kBlock)), dim3(pk::kBlock), 0, (hipStream_t)stream, P,
	                                      (const T *)starts, (const T *)stops, (const T *)step_sizes, step_s, num_steps));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_sample_step_count(uint32_t P, const float *nears, const float *fars, uint32_t max_steps,
                                      float dt_gamma, float min_step, float max_step, int64_t *n_per_pack, void *stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(pk::k_sample_count, dim3(div_up(P, pk::kBlock)), dim3(pk::kBlock), 0, (hipStream_t)stream, P, nears,
	                   fars, max_steps, dt_gamma, min_step, max_step, n_per_pack);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_sample_step_emit(uint32_t P, const float *nears, const int64_t *pack_infos, float dt_gamma,
                                     float min_step, float max_step, float *t_samples, float *deltas, int64_t *nidx,
                                     void *stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(pk::k_sample_emit, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, nears, pack_infos,
	                   dt_gamma, min_step, max_step, t_samples, deltas, nidx);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_sample_step_segments(uint32_t P, const float *nears, const float *fars, const float *entries,
                                         const float *exits, const int64_t *seg_pack_infos, uint32_t max_steps,
                                         float dt_gamma, float min_step, float max_step, int emit, int64_t *n_per_pack,
                                         const int64_t *pack_infos, float *t_samples, float *deltas, int64_t *nidx,
                                         int64_t *sidx, void *stream) {
	if (P == 0) return 0;
	const dim3 g(div_up(P, pk::kBlock)), b(pk::kBlock);
	if (emit)
		hipLaunchKernelGGL(pk::k_sample_segments<true>, g, b, 0, (hipStream_t)stream, P, nears, fars, entries, exits,
		                   seg_pack_infos, max_steps, dt_gamma, min_step, max_step, n_per_pack, pack_infos, t_samples,
		                   deltas, nidx, sidx);
	else
		hipLaunchKernelGGL(pk::k_sample_segments<false>, g, b, 0, (hipStream_t)stream, P, nears, fars, entries, exits,
		                   seg_pack_infos, max_steps, dt_gamma, min_step, max_step, n_per_pack, pack_infos, t_samples,
		                   deltas, nidx, sidx);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_packed_sum(uint32_t P, uint64_t S, uint32_t fd, int dtype, const void *feats,
                               const int64_t *pack_infos, void *out, void *stream) {
	if (P == 0) return 0;
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_sum<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, fd,
	                                      (const T *)feats, pack_infos, (T *)out));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_packed_scan(uint32_t P, uint64_t S, uint32_t fd, int dtype, const void *feats,
                                const int64_t *pack_infos, int is_prod, int exclusive, int reverse, int ordered_packs,
                                void *out, void *stream) {
	NR3D_CHECK(P > 0 || !ordered_packs || S == 0, "packed_scan: ordered_packs needs at least one pack (no wave to zero `out`)");
	// the gap fill works on the 32-bit pack bounds the kernels use (my_pack): past 2^32 rows it would zero the wrong ranges
	NR3D_CHECK(!ordered_packs || (uint64_t)S < (1ull << 32), "packed_scan: ordered_packs is limited to 2^32 - 1 rows per call, got %llu", (unsigned long long)S);
	if (P == 0) return 0;
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_scan<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, fd,
	                                      (const T *)feats, pack_infos, is_prod, exclusive, reverse, S, ordered_packs, (T *)out));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_packed_diff(uint32_t P, uint64_t S, uint32_t fd, int dtype, const void *feats,
                                const int64_t *pack_infos, const void *edge_a, const void *edge_fill, int backward,
                                int ordered_packs, void *out, void *stream) {
	NR3D_CHECK(!(edge_a && edge_fill), "You should only specify AT MOST one of [appends, prepends, last_fill, first_fill]");
	NR3D_CHECK(P > 0 || !ordered_packs || S == 0, "packed_diff: ordered_packs needs at least one pack (no wave to zero `out`)");
	// the gap fill works on the 32-bit pack bounds the kernels use (my_pack): past 2^32 rows it would zero the wrong ranges
	NR3D_CHECK(!ordered_packs || (uint64_t)S < (1ull << 32), "packed_diff: ordered_packs is limited to 2^32 - 1 rows per call, got %llu", (unsigned long long)S);
	if (P == 0) return 0;
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_diff<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, fd,
	                                      (const T *)feats, pack_infos, (const T *)edge_a, (const T *)edge_fill, backward,
	                                      S, ordered_packs, (T *)out));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_packed_binary(uint32_t P, uint64_t S, uint32_t fd, uint32_t od, int dtype, const void *feats,
                                  const void *other, const int64_t *pack_infos, int op, int ordered_packs, void *out,
                                  void *stream) {
	NR3D_CHECK(op >= 0 && op <= 10, "packed_binary_ops: invalid op %d", op);
	NR3D_CHECK(P > 0 || !ordered_packs || S == 0, "packed_binary_ops: ordered_packs needs at least one pack (no wave to zero `out`)");
	// the gap fill works on the 32-bit pack bounds the kernels use (my_pack): past 2^32 rows it would zero the wrong ranges
	NR3D_CHECK(!ordered_packs || (uint64_t)S < (1ull << 32), "packed_binary_ops: ordered_packs is limited to 2^32 - 1 rows per call, got %llu", (unsigned long long)S);
	if (P == 0) return 0;
	if (op == 4) {
		PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_matmul<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P,
		                                      fd, od, (const T *)feats, (const T *)other, pack_infos, S, ordered_packs, (T *)out));
	} else {
		PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_binary<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P,
		                                      fd, (const T *)feats, (const T *)other, pack_infos, op, S, ordered_packs,
		                                      (T *)out, (uint8_t *)out));
	}
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_packed_searchsorted(uint32_t P, int dtype, const void *bins, const void *vals,
                                        const int64_t *pack_infos, uint32_t num_to_search,
                                        const int64_t *val_pack_infos, int64_t *pidx, void *stream) {
	if (P == 0) return 0;
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_searchsorted<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream,
	                                      P, (const T *)bins, (const T *)vals, pack_infos, num_to_search, val_pack_infos,
	                                      pidx));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_try_merge_two_packs_sorted_aligned(uint32_t P, int dtype, const void *vals_a,
                                                       const int64_t *pack_infos_a, const void *vals_b,
                                                       const int64_t *pack_infos_b, const int64_t *pack_infos_merged,
                                                       int b_sorted, int64_t *pidx_a, int64_t *pidx_b, void *stream) {
	(void)b_sorted;   // the sorted-b shortcut of the reference only narrows the search range; same result
	if (P == 0) return 0;
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_merge<T>, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P,
	                                      (const T *)vals_a, pack_infos_a, (const T *)vals_b, pack_infos_b,
	                                      pack_infos_merged, pidx_a, pidx_b));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_merge_pack_union(uint32_t Pa, const int64_t *nidx_a, const int64_t *pack_infos_a, uint32_t Pb, const int64_t *nidx_b,
                                     const int64_t *pack_infos_b, int64_t *only_b, int64_t *ob, void *scan_tmp, int64_t *n_only,
                                     int64_t *u, int64_t *pack_infos_a_u, int64_t *pack_infos_b_u, int64_t *n_u, int64_t *total_u,
                                     void *stream) {
	NR3D_CHECK(Pa > 0 && Pb > 0, "merge_pack_union: both pack lists must be non-empty");
	NR3D_CHECK(only_b && ob && scan_tmp && n_only && total_u, "merge_pack_union: NULL scratch pointer");
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(pk::k_union_flags, dim3(div_up(Pb, pk::kBlock)), dim3(pk::kBlock), 0, st, Pa, nidx_a, Pb, nidx_b, only_b);
	if (int rc = scan::pack_infos_from_counts<int64_t, int64_t>(Pb, only_b, ob, n_only, scan_tmp, st)) return rc;
	hipLaunchKernelGGL(pk::k_union_build, dim3(div_up((uint64_t)Pa + Pb, pk::kBlock)), dim3(pk::kBlock), 0, st, Pa, nidx_a, pack_infos_a, Pb,
	                   nidx_b, pack_infos_b, only_b, ob, n_only, u, pack_infos_a_u, pack_infos_b_u, n_u, total_u);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_packed_invert_cdf(uint32_t P, const float *bins, const float *cdfs, const int64_t *pack_infos,
                                      const float *u_vals, uint32_t num_to_sample, float *samples, int64_t *bin_idx,
                                      void *stream) {
	if (P == 0) return 0;
	hipLaunchKernelGGL(pk::k_invert_cdf, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, bins, cdfs,
	                   pack_infos, u_vals, num_to_sample, samples, bin_idx);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_packed_sort(uint32_t P, uint64_t S, int dtype, void *vals, int64_t *ids, const int64_t *pack_infos,
                                void *stream) {
	if (P == 0) return 0;
	if (opt::on(NR3D_OPT_SORT_WAVE)) {                  // one wave per pack, bitonic network in registers (default)
		const hipStream_t st = (hipStream_t)stream;
		switch (dtype) {
		case NR3D_F32: hipLaunchKernelGGL((pk::k_sort_wave<pk::srt::E32, float>), pk::grid_for(P), dim3(pk::kBlock), 0, st, P, vals, ids, pack_infos, 1); break;
		case NR3D_I32: hipLaunchKernelGGL((pk::k_sort_wave<pk::srt::E32, int32_t>), pk::grid_for(P), dim3(pk::kBlock), 0, st, P, vals, ids, pack_infos, 0); break;
		case NR3D_F64: hipLaunchKernelGGL((pk::k_sort_wave<pk::srt::E64, double>), pk::grid_for(P), dim3(pk::kBlock), 0, st, P, vals, ids, pack_infos, 1); break;
		case NR3D_I64: hipLaunchKernelGGL((pk::k_sort_wave<pk::srt::E64, int64_t>), pk::grid_for(P), dim3(pk::kBlock), 0, st, P, vals, ids, pack_infos, 0); break;
		default: return ::nr3d::fail("pack_ops: unsupported dtype code %d (f32/f64/i32/i64)", (int)dtype);
		}
		NR3D_LAUNCH_CHECK();
		return 0;
	}
	PK_DISPATCH(dtype, hipLaunchKernelGGL(pk::k_sort<T>, dim3(div_up(P, pk::kBlock)), dim3(pk::kBlock), 0,
	                                      (hipStream_t)stream, P, (T *)vals, ids, pack_infos));
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_alpha_to_vw_forward(uint32_t P, uint64_t S, const float *alphas, const int64_t *pack_infos,
                                        float early_stop_eps, float alpha_thre, float *weights, int64_t *num_steps,
                                        uint8_t *selector, void *stream) {
	if (P == 0) return 0;
	if (pk::lane_per_pack(P))
		hipLaunchKernelGGL(pk::k_alpha_fwd_lpp, dim3(div_up(P, pk::kBlock)), dim3(pk::kBlock), 0, (hipStream_t)stream, P, alphas,
		                   pack_infos, early_stop_eps, alpha_thre, weights, num_steps, selector);
	else
		hipLaunchKernelGGL(pk::k_alpha_fwd, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, alphas, pack_infos,
		                   early_stop_eps, alpha_thre, weights, num_steps, selector);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_alpha_to_vw_backward(uint32_t P, uint64_t S, const float *alphas, const float *weights,
                                         const float *grad_weights, const int64_t *pack_infos, float early_stop_eps,
                                         float alpha_thre, float *grad_alphas, void *stream) {
	if (P == 0) return 0;
	if (pk::lane_per_pack(P))
		hipLaunchKernelGGL(pk::k_alpha_bwd_lpp, dim3(div_up(P, pk::kBlock)), dim3(pk::kBlock), 0, (hipStream_t)stream, P, alphas,
		                   weights, grad_weights, pack_infos, early_stop_eps, alpha_thre, grad_alphas);
	else
		hipLaunchKernelGGL(pk::k_alpha_bwd, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, alphas, weights,
		                   grad_weights, pack_infos, early_stop_eps, alpha_thre, grad_alphas);
	NR3D_LAUNCH_CHECK();
	return 0;
}

namespace nr3d {
namespace pk {
int launch_composite_rays_fwd(uint32_t n_rays, const int32_t *packed_info, const float *sigma, const float *delta, uint64_t sigma_rows,
                              const float *ts, const float *rgb, float eps, float thre, int normalize, float *alphas, float *vw,
                              float *mask, float *depth, float *rgb_out, hipStream_t st) {
	if (n_rays == 0) return 0;
	prof::Scope ps(NR3D_PROF_COMPOSITE_FWD, st);
	if (rgb)
		hipLaunchKernelGGL(k_composite_rays_fwd<true>, grid_for(n_rays), dim3(kBlock), 0, st, n_rays, packed_info, sigma, delta, sigma_rows,
		                   ts, rgb, eps, thre, normalize, alphas, vw, mask, depth, rgb_out);
	else
		hipLaunchKernelGGL(k_composite_rays_fwd<false>, grid_for(n_rays), dim3(kBlock), 0, st, n_rays, packed_info, sigma, delta, sigma_rows,
		                   ts, rgb, eps, thre, normalize, alphas, vw, mask, depth, rgb_out);
	NR3D_LAUNCH_CHECK();
	return 0;
}

int launch_emit_composite_rays_fwd(uint32_t n_rays, const int32_t *packed_info, const void *cache, uint32_t cache_stride,
                                   const float *rays_o, const float *rays_d, float *t_starts, float *t_ends, int32_t *ridx, int32_t *gidx,
                                   int64_t *ridx64, float *deltas, float *samples, const float *sigma, uint64_t sigma_rows,
                                   const float *rgb, float eps, float thre, int normalize, float *alphas, float *vw, float *mask,
                                   float *depth, float *rgb_out, hipStream_t st) {
	if (n_rays == 0) return 0;
	prof::Scope ps(NR3D_PROF_COMPOSITE_FWD, st);
	if (rgb)
		hipLaunchKernelGGL(k_emit_composite_rays_fwd<true>, grid_for(n_rays), dim3(kBlock), 0, st, n_rays, packed_info,
		                   (const uint32_t *)cache, cache_stride, rays_o, rays_d, t_starts, t_ends, ridx, gidx, ridx64, deltas, samples, sigma,
		                   sigma_rows, rgb, eps, thre, normalize, alphas, vw, mask, depth, rgb_out);
	else
		hipLaunchKernelGGL(k_emit_composite_rays_fwd<false>, grid_for(n_rays), dim3(kBlock), 0, st, n_rays, packed_info,
		                   (const uint32_t *)cache, cache_stride, rays_o, rays_d, t_starts, t_ends, ridx, gidx, ridx64, deltas, samples, sigma,
		                   sigma_rows, rgb, eps, thre, normalize, alphas, vw, mask, depth, rgb_out);
	NR3D_LAUNCH_CHECK();
	return 0;
}

int launch_composite_rays_bwd(uint32_t n_rays, const int32_t *packed_info, const float *alphas, const float *vw, const float *ts,
                              const float *rgb, float eps, float thre, int normalize, const float *mask, const float *depth,
                              const float *g_mask, const float *g_depth, const float *g_rgb, float *grad_alphas, float *grad_t,
                              float *grad_rgb, const float *sigma, const float *delta, uint64_t sigma_rows, float *grad_sigma,
                              hipStream_t st) {
	if (n_rays == 0) return 0;
	prof::Scope ps(NR3D_PROF_COMPOSITE_BWD, st);
	if (rgb)
		hipLaunchKernelGGL(k_composite_rays_bwd<true>, grid_for(n_rays), dim3(kBlock), 0, st, n_rays, packed_info, alphas, vw, ts, rgb, eps,
		                   thre, normalize, mask, depth, g_mask, g_depth, g_rgb, grad_alphas, grad_t, grad_rgb, sigma, delta, sigma_rows,
		                   grad_sigma);
	else
		hipLaunchKernelGGL(k_composite_rays_bwd<false>, grid_for(n_rays), dim3(kBlock), 0, st, n_rays, packed_info, alphas, vw, ts, rgb, eps,
		                   thre, normalize, mask, depth, g_mask, g_depth, g_rgb, grad_alphas, grad_t, grad_rgb, sigma, delta, sigma_rows,
		                   grad_sigma);
	NR3D_LAUNCH_CHECK();
	return 0;
}
}  // namespace pk
}  // namespace nr3d

extern "C" int nr3d_pack_composite_fwd(uint32_t P, const float *alphas, const float *t, const float *rgb,
                                       const int64_t *pack_infos, const int64_t *ray_index, float early_stop_eps,
                                       float alpha_thre, int normalize_depth, float *vw, float *mask, float *depth,
                                       float *rgb_out, void *stream) {
	if (P == 0) return 0;
	NR3D_CHECK(alphas && t && pack_infos && vw && mask && depth, "pack_composite_fwd: NULL pointer");
	NR3D_CHECK(!rgb || rgb_out, "pack_composite_fwd: rgb given without rgb_out");
	const bool scan = pk::composite_scan() && P <= pk::composite_scan_max();
	const bool lpp = !scan && pk::lane_per_pack(P);
	const dim3 g = lpp ? dim3(div_up(P, pk::kBlock)) : pk::grid_for(P), b(pk::kBlock);
#define NR3D_COMP_FWD(K) hipLaunchKernelGGL(K, g, b, 0, (hipStream_t)stream, P, alphas, t, rgb, pack_infos, ray_index, \
	early_stop_eps, alpha_thre, normalize_depth, vw, mask, depth, rgb_out)
	prof::Scope ps(NR3D_PROF_COMPOSITE_FWD, (hipStream_t)stream);
	if (scan) { if (rgb) NR3D_COMP_FWD(pk::k_composite_fwd_scan<true>); else NR3D_COMP_FWD(pk::k_composite_fwd_scan<false>); }
	else if (lpp) { if (rgb) NR3D_COMP_FWD(pk::k_composite_fwd_lpp<true>); else NR3D_COMP_FWD(pk::k_composite_fwd_lpp<false>); }
	else     { if (rgb) NR3D_COMP_FWD(pk::k_composite_fwd<true>); else NR3D_COMP_FWD(pk::k_composite_fwd<false>); }
#undef NR3D_COMP_FWD
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_pack_composite_bwd(uint32_t P, const float *alphas, const float *vw, const float *t, const float *rgb,
                                       const int64_t *pack_infos, const int64_t *ray_index, float early_stop_eps,
                                       float alpha_thre, int normalize_depth, const float *mask, const float *depth,
                                       const float *g_mask, const float *g_depth, const float *g_rgb, const float *g_vw,
                                       float *grad_alphas, float *grad_t, float *grad_rgb, void *stream) {
	if (P == 0) return 0;
	NR3D_CHECK(alphas && vw && t && pack_infos && mask && depth && grad_alphas, "pack_composite_bwd: NULL pointer");
	const bool scan = pk::composite_scan() && P <= pk::composite_scan_max();
	const bool lpp = !scan && pk::lane_per_pack(P);
	const dim3 g = lpp ? dim3(div_up(P, pk::kBlock)) : pk::grid_for(P), b(pk::kBlock);
#define NR3D_COMP_BWD(K) hipLaunchKernelGGL(K, g, b, 0, (hipStream_t)stream, P, alphas, vw, t, rgb, pack_infos, ray_index, \
	early_stop_eps, alpha_thre, normalize_depth, mask, depth, g_mask, g_depth, g_rgb, g_vw, grad_alphas, grad_t, grad_rgb)
	prof::Scope ps(NR3D_PROF_COMPOSITE_BWD, (hipStream_t)stream);
	if (scan) { if (rgb) NR3D_COMP_BWD(pk::k_composite_bwd_scan<true>); else NR3D_COMP_BWD(pk::k_composite_bwd_scan<false>); }
	else if (lpp) { if (rgb) NR3D_COMP_BWD(pk::k_composite_bwd_lpp<true>); else NR3D_COMP_BWD(pk::k_composite_bwd_lpp<false>); }
	else     { if (rgb) NR3D_COMP_BWD(pk::k_composite_bwd<true>); else NR3D_COMP_BWD(pk::k_composite_bwd<false>); }
#undef NR3D_COMP_BWD
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_mark_pack_boundaries(uint64_t num, int dtype, const void *pack_ids, int32_t *boundaries, void *stream) {
	if (num == 0) return 0;
	const dim3 g(div_up(num, pk::kBlock)), b(pk::kBlock);
	switch (dtype) {
	case NR3D_I64: hipLaunchKernelGGL(pk::k_boundaries<int64_t>, g, b, 0, (hipStream_t)stream, num, (const int64_t *)pack_ids, boundaries); break;
	case NR3D_I32: hipLaunchKernelGGL(pk::k_boundaries<int32_t>, g, b, 0, (hipStream_t)stream, num, (const int32_t *)pack_ids, boundaries); break;
	case NR3D_I16: hipLaunchKernelGGL(pk::k_boundaries<int16_t>, g, b, 0, (hipStream_t)stream, num, (const int16_t *)pack_ids, boundaries); break;
	case NR3D_I8:  hipLaunchKernelGGL(pk::k_boundaries<int8_t>, g, b, 0, (hipStream_t)stream, num, (const int8_t *)pack_ids, boundaries); break;
	case NR3D_U8:  hipLaunchKernelGGL(pk::k_boundaries<uint8_t>, g, b, 0, (hipStream_t)stream, num, (const uint8_t *)pack_ids, boundaries); break;
	default: return ::nr3d::fail("mark_pack_boundaries: integral dtype required (got code %d)", dtype);
	}
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_octree_mark_consecutive_segments(uint32_t P, const int32_t *pidx, const int64_t *pack_infos,
                                                     const int16_t *point_hierarchies, uint8_t *mark_start, uint8_t *mark_end,
                                                     void *stream) {
	if (P == 0) return 0;
	NR3D_CHECK(pidx && pack_infos && point_hierarchies && mark_start && mark_end, "octree_mark_consecutive_segments: NULL pointer");
	hipLaunchKernelGGL(pk::k_mark_consecutive, pk::grid_for(P), dim3(pk::kBlock), 0, (hipStream_t)stream, P, pack_infos, pidx,
	                   point_hierarchies, mark_start, mark_end);
	NR3D_LAUNCH_CHECK();
	return 0;
}
