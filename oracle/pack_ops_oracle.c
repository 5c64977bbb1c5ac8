/* oracle/pack_ops_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the reference pack_ops kernels (csrc/pack_ops/pack_ops_cuda.cu), one pack
 * per loop iteration exactly like the reference's one-thread-per-pack kernels, so sequential
 * rounding order inside a pack is the reference's.  pack_infos is int64 [P,2] = (begin, length).
 *
 * Divergences from the reference, all on inputs where the reference has undefined behaviour:
 *   - empty packs (length 0) are skipped instead of reading/writing feats[begin] / feats[end-1]
 *     (packed_sum :818, cumsum/cumprod :886/:920, diff :1121-1133).
 * Reference quirks kept: exclusive cumprod leaves the first element at 0 (=> whole pack 0), :884-894;
 *   alpha_to_vw forward skips `alpha <= thre`, backward skips `alpha < thre` (:1772 vs :1838).
 *
 * Compiled with -ffp-contract=off; fmaf() marks where nvcc's default contraction fuses.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define PK_BEGIN(p) ((uint32_t)pack_infos[2 * (size_t)(p)])
#define PK_LEN(p)   ((uint32_t)pack_infos[2 * (size_t)(p) + 1])

/* ---------------- typed generic ops ---------------- */
#define DEFINE_TYPED(T, SFX)                                                                          \
/* kernel_interleave_arange :47-83 */                                                                 \
void orc_interleave_linstep_##SFX(uint32_t P, const int64_t *num_steps, const T *starts,              \
                                  const T *step_sizes, T start, T step_size, T *out, int64_t *nidx) { \
	uint32_t begin = 0;                                                                               \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t n = (uint32_t)num_steps[p];                                                          \
		T s = starts ? starts[p] : start;                                                             \
		T st = step_sizes ? step_sizes[p] : step_size;                                                \
		for (uint32_t j = 0; j < n; ++j) {                                                            \
			out[begin + j] = MULADD_##SFX((T)j, st, s);                                               \
			if (nidx) nidx[begin + j] = p;                                                            \
		}                                                                                             \
		begin += n;                                                                                   \
	}                                                                                                 \
}                                                                                                     \
/* kernel_packed_sum :798-824 */                                                                      \
void orc_packed_sum_##SFX(uint32_t P, uint32_t fd, const T *in, const int64_t *pack_infos, T *out) {  \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), e = b + PK_LEN(p);                                                  \
		if (e == b) continue;                                                                         \
		for (uint32_t j = 0; j < fd; ++j) {                                                           \
			T r = in[(size_t)b * fd + j];                                                             \
			for (uint32_t i = b + 1; i < e; ++i) r += in[(size_t)i * fd + j];                         \
			out[(size_t)p * fd + j] = r;                                                              \
		}                                                                                             \
	}                                                                                                 \
}                                                                                                     \
/* kernel_packed_cumsum[_reverse] :981-1046, kernel_packed_cumprod[_reverse] :864-929; out zero-init */\
void orc_packed_scan_##SFX(uint32_t P, uint32_t fd, const T *in, const int64_t *pack_infos,           \
                           int is_prod, int exclusive, int reverse, T *out) {                         \
	const int32_t off = exclusive ? 1 : 0;                                                            \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		int32_t b = (int32_t)PK_BEGIN(p), e = b + (int32_t)PK_LEN(p);                                 \
		if (e == b) continue;                                                                         \
		if (!reverse) {                                                                               \
			if (off == 0) for (uint32_t j = 0; j < fd; ++j) out[(size_t)b * fd + j] = in[(size_t)b * fd + j]; \
			for (uint32_t j = 0; j < fd; ++j)                                                         \
				for (int32_t i = b + 1; i < e; ++i) {                                                 \
					T a = in[(size_t)(i - off) * fd + j], c = out[(size_t)(i - 1) * fd + j];          \
					out[(size_t)i * fd + j] = is_prod ? a * c : a + c;                                \
				}                                                                                     \
		} else {                                                                                      \
			if (off == 0) for (uint32_t j = 0; j < fd; ++j) out[(size_t)(e - 1) * fd + j] = in[(size_t)(e - 1) * fd + j]; \
			for (uint32_t j = 0; j < fd; ++j)                                                         \
				for (int32_t i = e - 2; i >= b; --i) {                                                \
					T a = in[(size_t)(i + off) * fd + j], c = out[(size_t)(i + 1) * fd + j];          \
					out[(size_t)i * fd + j] = is_prod ? a * c : a + c;                                \
				}                                                                                     \
		}                                                                                             \
	}                                                                                                 \
}                                                                                                     \
/* kernel_packed_diff :1098-1141; out zero-init */                                                    \
void orc_packed_diff_##SFX(uint32_t P, uint32_t fd, const T *in, const T *appends, const T *last_fills,\
                           const int64_t *pack_infos, T *out) {                                       \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), e = b + PK_LEN(p);                                                  \
		if (e == b) continue;                                                                         \
		for (uint32_t j = 0; j < fd; ++j)                                                             \
			for (uint32_t i = b; i + 1 < e; ++i)                                                      \
				out[(size_t)i * fd + j] = in[(size_t)(i + 1) * fd + j] - in[(size_t)i * fd + j];      \
		if (appends) for (uint32_t j = 0; j < fd; ++j)                                                \
			out[(size_t)(e - 1) * fd + j] = appends[(size_t)p * fd + j] - in[(size_t)(e - 1) * fd + j]; \
		else if (last_fills) for (uint32_t j = 0; j < fd; ++j)                                        \
			out[(size_t)(e - 1) * fd + j] = last_fills[(size_t)p * fd + j];                           \
	}                                                                                                 \
}                                                                                                     \
/* kernel_packed_backward_diff :1143-1184; out zero-init */                                           \
void orc_packed_backward_diff_##SFX(uint32_t P, uint32_t fd, const T *in, const T *prepends,          \
                                    const T *first_fills, const int64_t *pack_infos, T *out) {        \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), e = b + PK_LEN(p);                                                  \
		if (e == b) continue;                                                                         \
		for (uint32_t j = 0; j < fd; ++j)                                                             \
			for (uint32_t i = b + 1; i < e; ++i)                                                      \
				out[(size_t)i * fd + j] = in[(size_t)i * fd + j] - in[(size_t)(i - 1) * fd + j];      \
		if (prepends) for (uint32_t j = 0; j < fd; ++j)                                               \
			out[(size_t)b * fd + j] = in[(size_t)b * fd + j] - prepends[(size_t)p * fd + j];          \
		else if (first_fills) for (uint32_t j = 0; j < fd; ++j)                                       \
			out[(size_t)b * fd + j] = first_fills[(size_t)p * fd + j];                                \
	}                                                                                                 \
}                                                                                                     \
/* kernel_packed_{add,sub,mul,div} :1960-2062.  op: 0 add 1 sub 2 mul 3 div */                        \
void orc_packed_binary_##SFX(uint32_t P, uint32_t fd, const T *in, const T *other,                    \
                             const int64_t *pack_infos, int op, T *out) {                             \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), e = b + PK_LEN(p);                                                  \
		for (uint32_t j = 0; j < fd; ++j) {                                                           \
			T o = other[(size_t)p * fd + j];                                                          \
			for (uint32_t i = b; i < e; ++i) {                                                        \
				T a = in[(size_t)i * fd + j];                                                         \
				out[(size_t)i * fd + j] = op == 0 ? a + o : op == 1 ? a - o : op == 2 ? a * o : a / o; \
			}                                                                                         \
		}                                                                                             \
	}                                                                                                 \
}                                                                                                     \
/* kernel_packed_{gt,geq,lt,leq,eq,neq} :2095-2249.  op: 5 gt 6 geq 7 lt 8 leq 9 eq 10 neq */         \
void orc_packed_compare_##SFX(uint32_t P, uint32_t fd, const T *in, const T *other,                   \
                              const int64_t *pack_infos, int op, uint8_t *out) {                      \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), e = b + PK_LEN(p);                                                  \
		for (uint32_t j = 0; j < fd; ++j) {                                                           \
			T o = other[(size_t)p * fd + j];                                                          \
			for (uint32_t i = b; i < e; ++i) {                                                        \
				T a = in[(size_t)i * fd + j];                                                         \
				out[(size_t)i * fd + j] = op == 5 ? a > o : op == 6 ? a >= o : op == 7 ? a < o        \
				                        : op == 8 ? a <= o : op == 9 ? a == o : a != o;               \
			}                                                                                         \
		}                                                                                             \
	}                                                                                                 \
}                                                                                                     \
/* kernel_packed_matmul :2064-2093: out[i,j] = sum_k in[i,k]*other[p,j,k] */                          \
void orc_packed_matmul_##SFX(uint32_t P, uint32_t fd, uint32_t ofd, const T *in, const T *other,      \
                             const int64_t *pack_infos, T *out) {                                     \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), e = b + PK_LEN(p);                                                  \
		const T *o = other + (size_t)p * ofd * fd;                                                    \
		for (uint32_t j = 0; j < ofd; ++j)                                                            \
			for (uint32_t i = b; i < e; ++i) {                                                        \
				T r = 0;                                                                              \
				for (uint32_t k = 0; k < fd; ++k) r = MULADD_##SFX(in[(size_t)i * fd + k], o[(size_t)j * fd + k], r); \
				out[(size_t)i * ofd + j] = r;                                                         \
			}                                                                                         \
	}                                                                                                 \
}                                                                                                     \
/* binary_search[_unsafe] :1336-1372 */                                                               \
static uint32_t bs_unsafe_##SFX(T val, const T *data, uint32_t length) {                              \
	if (length == 0) return 0;                                                                        \
	uint32_t first = 0, count = length;                                                               \
	while (count > 0) {                                                                               \
		uint32_t step = count / 2, it = first + step;                                                 \
		if (data[it] < val) { first = ++it; count -= step + 1; } else count = step;                   \
	}                                                                                                 \
	return first;                                                                                     \
}                                                                                                     \
static uint32_t bs_##SFX(T val, const T *data, uint32_t length) {                                     \
	if (length == 0) return 0;                                                                        \
	uint32_t f = bs_unsafe_##SFX(val, data, length);                                                  \
	return f < length - 1 ? f : length - 1;                                                           \
}                                                                                                     \
/* kernel_packed_searchsorted :1374-1407 (val_pack_infos NULL -> dense [P, num_to_search]) */         \
void orc_packed_searchsorted_##SFX(uint32_t P, const T *bins, const T *vals, const int64_t *pack_infos,\
                                   uint32_t num_to_search, const int64_t *val_pack_infos, int64_t *pidx) { \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), len = PK_LEN(p), ob, n = num_to_search;                             \
		if (val_pack_infos) { ob = (uint32_t)val_pack_infos[2 * p]; n = (uint32_t)val_pack_infos[2 * p + 1]; } \
		else ob = p * num_to_search;                                                                  \
		for (uint32_t i = 0; i < n; ++i) pidx[ob + i] = b + bs_##SFX(vals[ob + i], bins + b, len);    \
	}                                                                                                 \
}                                                                                                     \
/* kernel_try_merge_two_packs_sorted_aligned :1505-1571; pidx_a, pidx_b zero-init */                  \
void orc_try_merge_two_packs_sorted_aligned_##SFX(uint32_t P, const T *vals_a, const int64_t *pia,    \
        const T *vals_b, const int64_t *pib, const int64_t *pim, int b_sorted,                        \
        int64_t *pidx_a_, int64_t *pidx_b_) {                                                         \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		const uint32_t begin = (uint32_t)pia[2 * p], length = (uint32_t)pia[2 * p + 1];               \
		const uint32_t bb = (uint32_t)pib[2 * p], bl = (uint32_t)pib[2 * p + 1];                      \
		const uint32_t ob = (uint32_t)pim[2 * p];                                                     \
		const T *va = vals_a + begin, *vb = vals_b + bb;                                              \
		int64_t *pa = pidx_a_ + begin, *pb = pidx_b_ + bb;                                            \
		if (b_sorted) {                                                                               \
			int last_i = 0;                                                                           \
			for (uint32_t j = 0; j < bl; ++j) {                                                       \
				int i = (int)bs_unsafe_##SFX(vb[j], va + last_i, length - (uint32_t)last_i) + last_i; \
				pb[j] = i;                                                                            \
				if ((uint32_t)i < length) pa[i]++;                                                    \
				last_i = i;                                                                           \
			}                                                                                         \
		} else {                                                                                      \
			for (uint32_t j = 0; j < bl; ++j) {                                                       \
				int i = (int)bs_unsafe_##SFX(vb[j], va, length);                                      \
				pb[j] = i;                                                                            \
				if ((uint32_t)i < length) pa[i]++;                                                    \
			}                                                                                         \
		}                                                                                             \
		if (length > 0) pa[0] += ob;          /* reference writes pidx_a[0] unconditionally (:1558) */ \
		for (uint32_t i = 1; i < length; ++i) pa[i] += pa[i - 1] + 1;                                 \
		uint32_t acc = 1;                                                                             \
		int last_i = -1;                                                                              \
		for (uint32_t j = 0; j < bl; ++j) {                                                           \
			int i = (int)pb[j];                                                                       \
			uint32_t a_ = (i == last_i) ? (++acc) : (acc = 0);                                        \
			pb[j] = (int64_t)a_ + ((i == 0) ? (int64_t)ob : (pa[i - 1] + 1));                         \
			last_i = i;                                                                               \
		}                                                                                             \
	}                                                                                                 \
}                                                                                                     \
/* qsort_partition + kernel_packed_sort_qsort :2634-2720 (in place; ids optional) */                  \
void orc_packed_sort_qsort_##SFX(uint32_t P, T *vals_, int64_t *stack_, int64_t *ids_,                \
                                 const int64_t *pack_infos) {                                         \
	for (uint32_t p = 0; p < P; ++p) {                                                                \
		uint32_t b = PK_BEGIN(p), num = PK_LEN(p);                                                    \
		if (num == 0) continue;                                                                       \
		T *vals = vals_ + b; int64_t *stack = stack_ + b; int64_t *ids = ids_ ? ids_ + b : 0;         \
		if (num == 1) continue;                                                                       \
		int64_t l = 0, h = (int64_t)num - 1, top = -1;                                                \
		stack[++top] = l; stack[++top] = h;                                                           \
		while (top >= 0) {                                                                            \
			h = stack[top--]; l = stack[top--];                                                       \
			int64_t i = l - 1; T pivot = vals[h];                                                     \
			for (int64_t j = l; j < h; ++j)                                                           \
				if (vals[j] <= pivot) {                                                               \
					i++;                                                                              \
					{ T t = vals[j]; vals[j] = vals[i]; vals[i] = t; }                                \
					if (ids) { int64_t t = ids[j]; ids[j] = ids[i]; ids[i] = t; }                     \
				}                                                                                     \
			{ T t = vals[i + 1]; vals[i + 1] = vals[h]; vals[h] = t; }                                \
			if (ids) { int64_t t = ids[i + 1]; ids[i + 1] = ids[h]; ids[h] = t; }                     \
			int64_t pv = i + 1;                                                                       \
			if (pv - 1 > l) { stack[++top] = l; stack[++top] = pv - 1; }                              \
			if (pv + 1 < h) { stack[++top] = pv + 1; stack[++top] = h; }                              \
		}                                                                                             \
	}                                                                                                 \
}

#define MULADD_f32(a, b, c) fmaf((a), (b), (c))
#define MULADD_f64(a, b, c) fma((a), (b), (c))
#define MULADD_i64(a, b, c) ((a) * (b) + (c))
#define MULADD_i32(a, b, c) ((a) * (b) + (c))
DEFINE_TYPED(float, f32)
DEFINE_TYPED(double, f64)
DEFINE_TYPED(int64_t, i64)
DEFINE_TYPED(int32_t, i32)

/* ---------------- float-only sampling / rendering ops ---------------- */
static inline float clamp_lu(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); } /* :42-45 */

/* kernel_interleave_sample_step_wrt_depth_clamp_v3_round1 :480-506 */
void orc_sample_step_count(uint32_t P, uint32_t max_steps, float dt_gamma, float min_step, float max_step,
                           const float *nears, const float *fars, int64_t *n_per_pack) {
	for (uint32_t p = 0; p < P; ++p) {
		float t = nears[p], far = fars[p];
		uint32_t n = 0;
		while (t <= far && n < max_steps) { t += clamp_lu(t * dt_gamma, min_step, max_step); n++; }
		n_per_pack[p] = n;
	}
}
/* ..._round2 :508-545 */
void orc_sample_step_emit(uint32_t P, float dt_gamma, float min_step, float max_step, const float *nears,
                          const int64_t *pack_infos, float *t_samples, float *deltas, int64_t *nidx) {
	for (uint32_t p = 0; p < P; ++p) {
		uint32_t b = PK_BEGIN(p), n = PK_LEN(p);
		float t = nears[p];
		for (uint32_t s = 0; s < n; ++s) {
			t_samples[b + s] = t; nidx[b + s] = p;
			float dt = clamp_lu(t * dt_gamma, min_step, max_step);
			deltas[b + s] = dt; t += dt;
		}
	}
}
/* kernel_interleave_sample_step_wrt_depth_in_packed_segments_round{1,2} :606-718.
 * emit == 0: writes n_per_pack; emit == 1: pack_infos given, writes samples. */
void orc_sample_step_segments(uint32_t P, uint32_t max_steps, float dt_gamma, float min_step, float max_step,
                              const float *nears, const float *fars, const float *entries,
                              const float *exits, const int64_t *seg_pack_infos, int emit,
                              int64_t *n_per_pack, const int64_t *pack_infos, float *t_samples,
                              float *deltas, int64_t *nidx, int64_t *sidx) {
	for (uint32_t p = 0; p < P; ++p) {
		const float near = nears[p], far = fars[p];
		const uint32_t sb = (uint32_t)seg_pack_infos[2 * p], se = sb + (uint32_t)seg_pack_infos[2 * p + 1];
		const uint32_t b = emit ? PK_BEGIN(p) : 0, lim = emit ? PK_LEN(p) : max_steps;
		float t = near;
		uint32_t step = 0;
		for (uint32_t i = sb; i < se; ++i) {
			const float ce = entries[i], cx = exits[i];
			if (ce >= far || cx <= near) break;
			do { t += min_step; } while (t < ce);
			while (t <= cx && t <= far && step < lim) {
				float dt = clamp_lu(t * dt_gamma, min_step, max_step);
				if (emit) { t_samples[b + step] = t; nidx[b + step] = p; sidx[b + step] = i; deltas[b + step] = dt; }
				t += dt; step++;
			}
		}
		if (!emit) n_per_pack[p] = step;
	}
}

/* kernel_packed_invert_cdf :1633-1681 */
void orc_packed_invert_cdf(uint32_t P, const float *bins, const float *cdfs, const int64_t *pack_infos,
                           const float *u_vals, uint32_t num_to_sample, float *samples, int64_t *bin_idx) {
	const float eps = 1.0e-5f;
	for (uint32_t p = 0; p < P; ++p) {
		const uint32_t b = PK_BEGIN(p), len = PK_LEN(p), ob = p * num_to_sample;
		const float *bn = bins + b, *cd = cdfs + b;
		for (uint32_t i = 0; i < num_to_sample; ++i) {
			float u = u_vals[ob + i];
			uint32_t pos = bs_f32(u, cd, len);
			bin_idx[ob + i] = pos + b;
			if (pos == 0) samples[ob + i] = bn[0];
			else {
				uint32_t pp = pos - 1;
				float pmf = cd[pos] - cd[pp];
				samples[ob + i] = pmf < eps ? bn[pp] : fmaf((u - cd[pp]) / pmf, bn[pos] - bn[pp], bn[pp]);
			}
		}
	}
}

/* kernel_packed_alpha_to_vw_forward :1735-1793.  weights / num_steps / selector optional; zero-init */
void orc_alpha_to_vw_fwd(uint32_t P, const float *alphas, float early_stop_eps, float alpha_thre,
                         const int64_t *pack_infos, float *weights, int64_t *num_steps, uint8_t *selector) {
	for (uint32_t p = 0; p < P; ++p) {
		const uint32_t b = PK_BEGIN(p), len = PK_LEN(p);
		float T = 1.f;
		int cnt = 0;
		for (uint32_t j = 0; j < len; ++j) {
			if (T < early_stop_eps) break;
			float a = alphas[b + j];
			if (a <= alpha_thre) continue;
			const float w = a * T;
			T *= (1.f - a);
			if (weights) weights[b + j] = w;
			if (selector) selector[b + j] = 1;
			cnt += 1;
		}
		if (num_steps) num_steps[p] = cnt;
	}
}
/* kernel_packed_alpha_to_vw_backward :1795-1848; grad_alphas zero-init */
void orc_alpha_to_vw_bwd(uint32_t P, const float *alphas, const float *weights, const float *grad_weights,
                         float early_stop_eps, float alpha_thre, const int64_t *pack_infos,
                         float *grad_alphas) {
	for (uint32_t p = 0; p < P; ++p) {
		const uint32_t b = PK_BEGIN(p), len = PK_LEN(p);
		float accum = 0;
		for (uint32_t j = 0; j < len; ++j) accum = fmaf(grad_weights[b + j], weights[b + j], accum);
		float T = 1.f;
		for (uint32_t j = 0; j < len; ++j) {
			if (T < early_stop_eps) break;
			float a = alphas[b + j];
			if (a < alpha_thre) continue;
			grad_alphas[b + j] = fmaf(grad_weights[b + j], T, -accum) / fmaxf(1.f - a, 1e-10f);
			accum = fmaf(-grad_weights[b + j], weights[b + j], accum);
			T *= (1.f - a);
		}
	}
}

/* mark_pack_boundaries_cuda_kernel :2765-2782 (int64 ids) */
void orc_mark_pack_boundaries_i64(int64_t num, const int64_t *ids, int32_t *boundaries) {
	for (int64_t i = 0; i < num; ++i) boundaries[i] = (i == 0) ? 1 : (ids[i - 1] == ids[i] ? 0 : 1);
}
void orc_mark_pack_boundaries_i32(int64_t num, const int32_t *ids, int32_t *boundaries) {
	for (int64_t i = 0; i < num; ++i) boundaries[i] = (i == 0) ? 1 : (ids[i - 1] == ids[i] ? 0 : 1);
}
