// nr3d_lib_amd/csrc/mlp_half.hip -- the fused decoder in HALF precision on the f16 MFMA (gfx950), C-ABI entry points
// nr3d_mlp_half_packed_bytes / _backward_packed_bytes / _pack / _forward / _backward.
//
// What it stands in for: the reference's fast decoder is tiny-cuda-nn's `FullyFusedMLP` behind nr3d_lib/models/tcnn_adapter.py
// (:37-51 network config, :74-146 module; picked by `use_tcnn_backend`, nr3d_lib/models/blocks/__init__.py:3-15): half
// weights, half activations between the layers, half inputs / outputs.  Same contract here, with fp32 ACCUMULATION inside a
// layer (tcnn accumulates in half): y_l = half(act(W_l . x_l + b_l)) with the dot products summed in fp32 by
// v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense on MI355X, 16x the f32 MFMA that csrc/mlp.hip runs on).
//
// Structure = csrc/mlp.hip's (one wave owns a tile of 32 samples, everything transposed: H^T[feature, sample] = W . X^T,
// the C/D register map of one layer IS the B operand of the next, weights in LDS), re-derived for the K = 16 instruction:
//   * C/D map (dtype independent): lane (sample j = lane & 31, h = lane >> 5), accumulator r <-> feature 8 (r >> 2) + 4 h + (r & 3);
//   * A / B operands: 8 halfs per lane, lane (row or column = lane & 31, h) element e <-> k = 8 h + e.  Only that A and B index k
//     the SAME way matters (a dot product does not care about the order of its terms), so a step's B operand is simply eight
//     consecutive accumulator registers rounded to half -- r = 8 s + e for step s -- and the weights are packed to match:
//     A of step s, lane (out feature i, h), element e = W[i][32 it + 8 (2 s + (e >> 2)) + 4 h + (e & 3)].
//     Inputs that come from memory (x, dL/dy) are loaded straight into that register map (8-byte pieces), so ONE packing order
//     serves every layer, forward and transposed.
// Backward: the forward is recomputed from x; dH^T = W^T . dPre^T on the same maps; dW = dPre^T . H contracts over SAMPLES,
// which both operands keep in the lane index, so they pass through per-wave [feature][sample] half tiles in LDS (2-byte
// writes, one 16-byte read per MFMA operand); dW / db accumulate in fp32 registers over all of a wave's tiles, are summed over
// the workgroup in LDS and added to fp32 global buffers with one atomic per element and workgroup (the binding rounds them to
// the parameters' dtype afterwards).
#include "common.h"
#include <type_traits>

// x / dL_dy rows are fetched as 8-byte pieces, four instructions per 64-byte row, and dL/dx / y leave the same way: as streaming
// (non-temporal) accesses every piece went to L2 / HBM on its own; as plain accesses the L1 serves the three re-touches of a line and L2
// merges the pieces of a row (round 5, measured at 2^22 samples: 32 -> 64 -> 64 -> 16 fwd + bwd 0.739 -> 0.632 ms, 32 -> 32 -> 32 -> 16
// 0.489 -> 0.317, 64 -> 64 -> 64 -> 64 1.53 -> 1.22; loads alone 0.645, stores alone 0.685)
#ifndef NR3D_MLPH_NT_LOAD
#define NR3D_MLPH_NT_LOAD 0
#endif
#ifndef NR3D_MLPH_NT_STORE
#define NR3D_MLPH_NT_STORE 0
#endif

namespace nr3d {
namespace mlph {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kMaxLds = 144 * 1024;

__host__ __device__ constexpr uint32_t tiles(uint32_t d) { return (d + 31u) / 32u; }
// bytes of one packed layer: weights [NO][NI][2 steps][64 lanes][8 halfs] + bias fp32 [NO * 32]
__host__ __device__ constexpr uint32_t layer_bytes(uint32_t ni, uint32_t no) { return no * ni * 2048u + no * 128u; }

struct Shape { uint32_t n_layers, in_t, w_t, out_t; };

static bool shape_of(const nr3d_mlp_desc_t *d, Shape &s) {
	if (!d || d->n_layers < 2 || d->n_layers > NR3D_MLP_MAX_LAYERS) return false;
	uint32_t w = 0;
	for (uint32_t l = 1; l < d->n_layers; ++l) w = d->dims[l] > w ? d->dims[l] : w;
	for (uint32_t l = 0; l <= d->n_layers; ++l) if (d->dims[l] == 0 || d->dims[l] > 128) return false;
	s.n_layers = d->n_layers;
	auto round = [](uint32_t t) { return t == 3 ? 4u : t; };      // 3-tile widths run on the 4-tile instantiation
	s.in_t = round(tiles(d->dims[0])); s.w_t = round(tiles(w)); s.out_t = round(tiles(d->dims[d->n_layers]));
	return true;
}
static uint64_t packed_bytes(const Shape &s) {
	return (uint64_t)layer_bytes(s.in_t, s.w_t) + (uint64_t)(s.n_layers - 2) * layer_bytes(s.w_t, s.w_t) + layer_bytes(s.w_t, s.out_t);
}
static uint64_t transposed_bytes(const Shape &s) {
	return (uint64_t)layer_bytes(s.w_t, s.in_t) + (uint64_t)(s.n_layers - 2) * layer_bytes(s.w_t, s.w_t) + layer_bytes(s.out_t, s.w_t);
}

// ---------------------------------------------------------------------------------------------
// packing (see the header): half e of the weight block of layer l sits at
//   ((((ot * NI + it) * 2 + s) * 64 + lane) * 8 + el)  <-  W[32 ot + (lane & 31)][32 it + 8 (2 s + (el >> 2)) + 4 (lane >> 5) + (el & 3)]
// ---------------------------------------------------------------------------------------------
struct PackArgs {
	const __half *w[NR3D_MLP_MAX_LAYERS];
	const __half *b[NR3D_MLP_MAX_LAYERS];
	uint32_t in_dim[NR3D_MLP_MAX_LAYERS], out_dim[NR3D_MLP_MAX_LAYERS];   // of W as stored: [out_dim, in_dim] row-major
	uint32_t ni[NR3D_MLP_MAX_LAYERS], no[NR3D_MLP_MAX_LAYERS];
	uint32_t offset[NR3D_MLP_MAX_LAYERS + 1];                             // first BYTE of every packed layer
	uint32_t n_layers, transposed;
};

__global__ __launch_bounds__(256) void k_mlph_pack(PackArgs a, unsigned char *__restrict__ packed) {
	const uint32_t l = blockIdx.y;
	const uint32_t nw = a.no[l] * a.ni[l] * 1024u;                        // weight halfs
	__half *wdst = reinterpret_cast<__half *>(packed + a.offset[l]);
	float *bdst = reinterpret_cast<float *>(packed + a.offset[l] + (size_t)nw * 2);
	for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nw + a.no[l] * 32u; e += gridDim.x * 256) {
		if (e < nw) {
			const uint32_t el = e & 7u, lane = (e >> 3) & 63u, s = (e >> 9) & 1u, tile = e >> 10;
			const uint32_t it = tile % a.ni[l], ot = tile / a.ni[l];
			const uint32_t o = 32u * ot + (lane & 31u), f = 32u * it + 8u * (2u * s + (el >> 2)) + 4u * (lane >> 5) + (el & 3u);
			__half v = __float2half(0.0f);
			if (!a.transposed) { if (o < a.out_dim[l] && f < a.in_dim[l]) v = a.w[l][(size_t)o * a.in_dim[l] + f]; }
			else { if (o < a.in_dim[l] && f < a.out_dim[l]) v = a.w[l][(size_t)f * a.in_dim[l] + o]; }
			wdst[e] = v;
		} else {
			const uint32_t o = e - nw;
			bdst[o] = (!a.transposed && a.b[l] && o < a.out_dim[l]) ? __half2float(a.b[l][o]) : 0.0f;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// one dense layer: in = B operands (two K = 16 steps per 32-feature tile), out = fp32 accumulators on the C/D map
// ---------------------------------------------------------------------------------------------
// ReLU as ONE instruction, a signed integer maximum on the bits (v_max_i32: negative floats, -0 included, are negative integers).
// fmaxf(v, 0) costs two -- the compiler canonicalises the MFMA result first (a v_max_f32 v, v, v in front of the v_max_f32 0, v).
// Same values for every non-NaN input; a NaN with a clear sign bit stays NaN, as torch.relu keeps it (fmaxf returned 0).
__device__ __forceinline__ float relu(float v) {
	const int b = __builtin_bit_cast(int, v);
	return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ float activate(float v, int act) { return act == NR3D_MLP_ACT_RELU ? relu(v) : v; }

template <int NI, int NO, bool BIAS>
__device__ __forceinline__ void dense(const unsigned char *__restrict__ wp, const h8 (&in)[NI][2], f16v (&out)[NO], int act, int lane) {
	const float *bias = reinterpret_cast<const float *>(wp + NO * NI * 2048);
	const int h = lane >> 5;
	const h8 *wv = reinterpret_cast<const h8 *>(wp) + lane;
	constexpr bool SPLIT = (NO == 1);                  // one out tile: two accumulators over alternating steps (no dependent MFMA pair)
	// (round 5) the kernels around this are VALU bound -- ~1800 vector instructions per tile of 32 samples against 48 MFMAs -- so: an
	// accumulator that starts at zero takes the constant as the first MFMA's C operand instead of 16 moves, and the activation is
	// ONE wave-uniform branch around 16 v_max per tile instead of a select per element on the runtime code
	const f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
	f16v alt = zero;
	if (BIAS) {
#pragma unroll
		for (int ot = 0; ot < NO; ++ot)
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const f4v b4 = *reinterpret_cast<const f4v *>(bias + 32 * ot + 8 * q + 4 * h);
#pragma unroll
				for (int b = 0; b < 4; ++b) out[ot][4 * q + b] = b4[b];
			}
	}
#pragma unroll
	for (int it = 0; it < NI; ++it)
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			h8 w8[NO];
#pragma unroll
			for (int ot = 0; ot < NO; ++ot) w8[ot] = wv[((ot * NI + it) * 2 + s) * 64];
			const bool first = !BIAS && it == 0 && s == 0;          // compile time after unrolling
			if constexpr (SPLIT) {
				if (s & 1) alt = __builtin_amdgcn_mfma_f32_32x32x16_f16(w8[0], in[it][s], (it == 0) ? zero : alt, 0, 0, 0);
				else out[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w8[0], in[it][s], first ? zero : out[0], 0, 0, 0);
			} else {
#pragma unroll
				for (int ot = 0; ot < NO; ++ot) out[ot] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w8[ot], in[it][s], first ? zero : out[ot], 0, 0, 0);
			}
		}
	if constexpr (SPLIT) {
#pragma unroll
		for (int j = 0; j < 16; ++j) out[0][j] += alt[j];
	}
	if (act == NR3D_MLP_ACT_RELU) {
#pragma unroll
		for (int ot = 0; ot < NO; ++ot)
#pragma unroll
			for (int j = 0; j < 16; ++j) out[ot][j] = relu(out[ot][j]);
	}
}

// accumulators -> the next layer's B operands: step s = registers 8 s .. 8 s + 7, rounded to half (round to nearest even)
template <int NT>
__device__ __forceinline__ void to_operand(const f16v (&acc)[NT], h8 (&op)[NT][2]) {
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int s = 0; s < 2; ++s)
#pragma unroll
			for (int e = 0; e < 8; e += 2) {
				const f2v v = {acc[t][8 * s + e], acc[t][8 * s + e + 1]};
				const h2 p = __builtin_convertvector(v, h2);              // one v_cvt_pk_f16_f32 (was two conversions and a v_perm)
				op[t][s][e] = p[0]; op[t][s][e + 1] = p[1];
			}
}

// sum of the eight halfs of an operand in fp32: four v_dot2 against (1, 1) (was eight conversions and eight adds)
__device__ __forceinline__ float sum8(const h8 &v, float acc) {
	const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
	for (int e = 0; e < 8; e += 2) {
		const h2 p = {v[e], v[e + 1]};
		acc = __builtin_amdgcn_fdot2(p, ones, acc, false);
	}
	return acc;
}

// ReLU derivative as a bit mask on the operand form: 0xFFFF where the (non-negative) activation is not zero -- two packed integer
// instructions per pair of values, and an AND per pair to apply it to a gradient that has already been rounded to half (the
// bit-per-value form costs three instructions to build and three to apply, per value)
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ us8 relu_mask(const h8 &act) {
	const us8 one = {1, 1, 1, 1, 1, 1, 1, 1}, all = {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF};
	return __builtin_elementwise_min(__builtin_bit_cast(us8, act), one) * all;
}
__device__ __forceinline__ h8 apply_mask(const h8 &g, const us8 &m) { return __builtin_bit_cast(h8, (us8)(__builtin_bit_cast(us8, g) & m)); }

// rows of a half [n, dim] matrix straight into operand form: lane (sample, h) owns features 32 t + 8 q + 4 h + b = register
// r = 4 q + b of tile t, i.e. element (q & 1) * 4 + b of step q >> 1
template <int NT>
__device__ __forceinline__ void load_rows(const __half *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row, bool valid, bool vec,
                                          int lane, h8 (&r)[NT][2]) {
	const int h = lane >> 5;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t f = 32u * t + 8u * q + 4u * h;
			h4 v = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
			if (valid && f < dim) {
				const _Float16 *src = reinterpret_cast<const _Float16 *>(p) + (int64_t)row * stride + f;
				if (vec && f + 3 < dim) v = *reinterpret_cast<const h4 *>(src);
				else {
#pragma unroll
					for (int b = 0; b < 4; ++b) if (f + b < dim) v[b] = src[b];
				}
			}
#pragma unroll
			for (int b = 0; b < 4; ++b) r[t][q >> 1][(q & 1) * 4 + b] = v[b];
		}
}
// branch-free: 8-byte aligned rows whose width is a multiple of 4, row clamped by the caller; a piece beyond the width re-reads
// piece 0 (it meets zero weights)
template <int NT>
__device__ __forceinline__ void load_rows_fast(const __half *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row_clamped, int lane,
                                               h8 (&r)[NT][2]) {
	const int h = lane >> 5;
	const _Float16 *base = reinterpret_cast<const _Float16 *>(p) + (int64_t)row_clamped * stride;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t f = 32u * t + 8u * q + 4u * h;
			const h4 v = NR3D_MLPH_NT_LOAD ? __builtin_nontemporal_load(reinterpret_cast<const h4 *>(base + (f < dim ? f : 0u))) : *reinterpret_cast<const h4 *>(base + (f < dim ? f : 0u));
#pragma unroll
			for (int b = 0; b < 4; ++b) r[t][q >> 1][(q & 1) * 4 + b] = v[b];
		}
}
// feature-major input (element (row, f) at p[f * fstride + row], e.g. the [E, N] storage of the LoTD forward's half y)
template <int NT>
__device__ __forceinline__ void load_cols_fast(const __half *__restrict__ p, int64_t fstride, uint32_t dim, uint64_t row_clamped, int lane,
                                               h8 (&r)[NT][2]) {
	const int h = lane >> 5;
	const _Float16 *base = reinterpret_cast<const _Float16 *>(p) + row_clamped;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			const uint32_t f = 32u * t + 8u * (j >> 2) + 4u * h + (j & 3);
			r[t][j >> 3][j & 7] = __builtin_nontemporal_load(base + (int64_t)(f < dim ? f : 0u) * fstride);
		}
}

template <int NT>
__device__ __forceinline__ void store_rows(__half *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row, bool valid, bool vec,
                                           int lane, const f16v (&r)[NT]) {
	const int h = lane >> 5;
	if (!valid) return;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t f = 32u * t + 8u * q + 4u * h;
			if (f >= dim) continue;
			_Float16 *dst = reinterpret_cast<_Float16 *>(p) + (int64_t)row * stride + f;
			const f2v lo2 = {r[t][4 * q], r[t][4 * q + 1]}, hi2 = {r[t][4 * q + 2], r[t][4 * q + 3]};
			const h2 pl = __builtin_convertvector(lo2, h2), ph = __builtin_convertvector(hi2, h2);
			const h4 v = {pl[0], pl[1], ph[0], ph[1]};
			// rows wider than one tile (NT > 1): a lane's 8-byte pieces of a 128-byte row arrive over eight instructions -- as streaming
			// (non-temporal) stores each piece went to HBM as a partial sector write (32 -> 64 -> 64 -> 64 forward: 1.15 ms for 0.54 GB of
			// output); as plain stores L2 merges them into whole lines first
			if (vec && f + 3 < dim) { if (NT > 1 || !NR3D_MLPH_NT_STORE) *reinterpret_cast<h4 *>(dst) = v; else __builtin_nontemporal_store(v, reinterpret_cast<h4 *>(dst)); }
			else {
#pragma unroll
				for (int b = 0; b < 4; ++b) if (f + b < dim) dst[b] = v[b];
			}
		}
}
template <int NT>
__device__ __forceinline__ void store_cols(__half *__restrict__ p, int64_t fstride, uint32_t dim, uint64_t row, bool valid, int lane,
                                           const f16v (&r)[NT]) {
	const int h = lane >> 5;
	if (!valid) return;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			const uint32_t f = 32u * t + 8u * (j >> 2) + 4u * h + (j & 3);
			if (f < dim) __builtin_nontemporal_store((_Float16)r[t][j], reinterpret_cast<_Float16 *>(p) + (int64_t)f * fstride + row);
		}
}

struct FwdArgs {
	uint64_t n;
	const __half *x; int64_t xs;
	__half *y; int64_t ys;
	const unsigned char *packed; uint32_t packed_bytes;
	uint32_t n_layers, in_dim, out_dim;
	int hidden_act, out_act;
	uint32_t x_vec, y_vec;
};

__device__ __forceinline__ void stage_weights(const unsigned char *__restrict__ packed, uint32_t n_bytes, unsigned char *lds) {
	const f4v *src = reinterpret_cast<const f4v *>(packed);
	f4v *dst = reinterpret_cast<f4v *>(lds);
	for (uint32_t i = threadIdx.x; i < n_bytes / 16; i += blockDim.x) dst[i] = src[i];
	__syncthreads();
}

// XF: 0 = row-major x, any alignment / width; 1 = row-major, 8-byte aligned rows of a multiple of 4 halfs (prefetched);
// 2 = feature-major x (a.xs = feature stride, prefetched)
template <int XF, int NT>
__device__ __forceinline__ void prefetch_x(const __half *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row_clamped, int lane,
                                           h8 (&r)[NT][2]) {
	if constexpr (XF == 2) load_cols_fast<NT>(p, stride, dim, row_clamped, lane, r);
	else load_rows_fast<NT>(p, stride, dim, row_clamped, lane, r);
}

template <int IN_T, int W_T, int OUT_T, int XF>
__global__ __launch_bounds__(kThreads) void k_mlph_fwd(FwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	stage_weights(a.packed, a.packed_bytes, lds);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t n_tiles = (a.n + 31) / 32, step = (uint64_t)gridDim.x * 4;
	const uint32_t off_hidden = layer_bytes(IN_T, W_T), sz_hidden = layer_bytes(W_T, W_T);
	auto clamp_row = [&](uint64_t row) { return row < a.n ? row : a.n - 1; };
	h8 xnext[IN_T][2];
	if (XF) prefetch_x<XF, IN_T>(a.x, a.xs, a.in_dim, clamp_row(((uint64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31)), lane, xnext);
	for (uint64_t tile = (uint64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += step) {
		const uint64_t row = tile * 32 + (lane & 31);
		const bool valid = row < a.n;
		h8 xin[IN_T][2], hop[W_T][2];
		f16v hacc[W_T], yo[OUT_T];
		// wide networks: keep the compiler from hoisting every layer's weight fragments out of the tile loop (128+ registers
		// of loop-invariant LDS reads -> scratch spills): the LDS base goes through a register it cannot see through
		uint32_t opaque = 0;
		if constexpr (IN_T >= 4 || W_T >= 4 || OUT_T >= 4) asm volatile("s_mov_b32 %0, 0" : "=s"(opaque));
		const unsigned char *wl = lds + opaque;
		if (XF) {
#pragma unroll
			for (int t = 0; t < IN_T; ++t) { xin[t][0] = xnext[t][0]; xin[t][1] = xnext[t][1]; }
			prefetch_x<XF, IN_T>(a.x, a.xs, a.in_dim, clamp_row((tile + step) * 32 + (lane & 31)), lane, xnext);
		} else {
			load_rows<IN_T>(a.x, a.xs, a.in_dim, row, valid, a.x_vec != 0, lane, xin);
		}
		dense<IN_T, W_T, true>(wl, xin, hacc, a.hidden_act, lane);
		to_operand<W_T>(hacc, hop);
#pragma unroll 1
		for (uint32_t l = 1; l + 1 < a.n_layers; ++l) {
			dense<W_T, W_T, true>(wl + off_hidden + (l - 1) * sz_hidden, hop, hacc, a.hidden_act, lane);
			to_operand<W_T>(hacc, hop);
		}
		dense<W_T, OUT_T, true>(wl + off_hidden + (a.n_layers - 2) * sz_hidden, hop, yo, a.out_act, lane);
		store_rows<OUT_T>(a.y, a.ys, a.out_dim, row, valid, a.y_vec != 0, lane, yo);
	}
}

// =============================================================================================
// backward
// =============================================================================================
constexpr int kTSH = 40;                       // halfs per feature a wave's tile area is sized with (32 samples + slack: bwd_tile_halfs)

// The per-wave tiles that turn "sample in the lane index" into "sample in the K index" (round 6, second form).  They were
// [feature][sample] arrays written with one 2-byte store per value and read with one 16-byte read per MFMA operand -- and the backward
// was bound by LDS INSTRUCTIONS (profiles/r06_mlp_half_counters.txt: SQ_ACTIVE_INST_LDS over a CU's eight waves = 82-87 % of the
// kernel's cycles, two thirds of the instructions those stores).  Now [sample][feature]: a lane owns its sample's row and its operand
// registers hold runs of four consecutive features, so a 32-feature block goes out as FOUR 8-byte stores instead of sixteen 2-byte
// ones, and the operand comes back through gfx950's transposing read (ds_read_b64_tr_b16: the 16 lanes of a group hand in the
// addresses of a 4-sample x 16-feature block, four 8-byte pieces per sample, and lane c gets feature c of the four samples) -- two
// reads per operand.  Row strides 72 / 152 bytes (one / two 32-feature blocks per row): the four rows of a transposing read fall on
// disjoint banks and the half-wave's 8-byte stores on 16 different bank pairs.  The area per tile stays inside the old 80 bytes per
// feature, so the tile offsets and the host's LDS sizing did not change.
template <int NT> struct TileRow { static_assert(NT == 1 || NT == 2, "tiles of one or two 32-feature blocks"); static constexpr int kHalfs = NT == 1 ? 36 : 76; };
typedef short s4v __attribute__((ext_vector_type(4)));

// operand form (16 values per lane and 32-feature block) -> [sample][feature] tile
template <int NT>
__device__ __forceinline__ void write_tile(_Float16 *__restrict__ T, const h8 (&r)[NT][2], int lane) {
	const int s = lane & 31, h = lane >> 5;
	_Float16 *row = T + s * TileRow<NT>::kHalfs + 4 * h;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int st = 0; st < 2; ++st) {                   // values 8 st + e <-> features 32 t + 16 st + 8 (e >> 2) + 4 h + (e & 3)
			const h8 v = r[t][st];
			const h4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
			*reinterpret_cast<h4 *>(row + 32 * t + 16 * st) = lo;
			*reinterpret_cast<h4 *>(row + 32 * t + 16 * st + 8) = hi;
		}
}
// one MFMA operand of the sample contraction: lane (r = lane & 31, h = lane >> 5) gets feature 32 t + r of samples 16 st + 8 h .. + 7
template <int NT>
__device__ __forceinline__ h8 read_op(const _Float16 *__restrict__ T, int t, int st, int lane) {
	const int c = lane & 15, fb = lane & 16, h = lane >> 5;
	const _Float16 *p = T + (16 * st + 8 * h + (c >> 2)) * TileRow<NT>::kHalfs + 32 * t + fb + 4 * (c & 3);
	typedef s4v __attribute__((address_space(3))) *lds_s4;
	const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)p);
	const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p + 4 * TileRow<NT>::kHalfs));
	const h4 l4 = __builtin_bit_cast(h4, lo), h4_ = __builtin_bit_cast(h4, hi);
	const h8 v = {l4[0], l4[1], l4[2], l4[3], h4_[0], h4_[1], h4_[2], h4_[3]};
	return v;
}

struct BwdArgs {
	uint64_t n;
	const __half *x; int64_t xs;
	const __half *gy; int64_t gys;
	__half *gx; int64_t gxs;                   // NULL: dL/dx not wanted
	uint32_t x_fm, gx_fm;
	const unsigned char *packed;               // [forward layers | transposed layers]
	uint32_t fwd_bytes, total_bytes;
	float *dW[NR3D_MLP_MAX_LAYERS];            // fp32, accumulated into (atomics): zero them for plain gradients
	float *db[NR3D_MLP_MAX_LAYERS];            // may be NULL
	uint32_t dims[NR3D_MLP_MAX_LAYERS + 1];
	uint32_t n_layers;
	int hidden_act, out_act;
	uint32_t x_vec, gy_vec, gx_vec;
	uint32_t tile_halfs;                       // per wave
};

// One layer of the backward sweep.  g = dL/d(pre-activation of this layer's output) in operand form (NO tiles); TG: LDS tile that
// receives it as [feature][sample]; TB: the layer's INPUT activations as [feature][sample] (NI tiles); wT: packed transposed
// layer; hmask: which of the lane's input activations are > 0 (bit 16 t + j of tile t, value j of the operand form) -- the ReLU mask
// as ONE register per layer instead of the activations themselves (16 per 64-wide layer, live across the whole sweep).  Accumulates dW (NO x NI tiles) and the
// per-lane bias partial sums; when PREV, leaves dL/d(input of the layer) in gp (fp32, C/D map), masked when MASK.
// BITS: the mask comes as hmask (the shapes whose backward spills: one register per layer instead of 8-16 live across the sweep);
// otherwise from hin, the activations in operand form (the shapes that fit: 32 -> 64 -> 64 -> 16 is 6-10 % faster without the bit work)
template <int NO, int NI, bool PREV, bool MASK, bool BITS>
__device__ __forceinline__ void bwd_layer(const h8 (&g)[NO][2], _Float16 *__restrict__ TG, const _Float16 *__restrict__ TB,
                                          const unsigned char *__restrict__ wT, f16v (&dW)[NO][NI], float (&db)[NO], f16v (&gp)[NI],
                                          const h8 (&hin)[NI][2], uint32_t hmask, int lane) {
	const int r = lane & 31, h = lane >> 5;
	write_tile<NO>(TG, g, lane);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	h8 bv[NI][2];
#pragma unroll
	for (int it = 0; it < NI; ++it)
#pragma unroll
		for (int st = 0; st < 2; ++st) bv[it][st] = read_op<NI>(TB, it, st, lane);
#pragma unroll
	for (int ot = 0; ot < NO; ++ot)
#pragma unroll
		for (int st = 0; st < 2; ++st) {
			const h8 av = read_op<NO>(TG, ot, st, lane);
			db[ot] = sum8(av, db[ot]);
#pragma unroll
			for (int it = 0; it < NI; ++it) dW[ot][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv[it][st], dW[ot][it], 0, 0, 0);
		}
	if (PREV) {
		dense<NO, NI, false>(wT, g, gp, NR3D_MLP_ACT_NONE, lane);
		if (MASK) {
#pragma unroll
			for (int t = 0; t < NI; ++t)
#pragma unroll
				for (int j = 0; j < 16; ++j) {
					const bool on = BITS ? (((hmask >> (16 * t + j)) & 1u) != 0u) : ((float)hin[t][j >> 3][j & 7] > 0.0f);
					gp[t][j] = on ? gp[t][j] : 0.0f;
				}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// sum one layer's gradient accumulators over the waves of the workgroup (through LDS) and add them to global memory
template <int NO, int NI>
__device__ __forceinline__ void reduce_layer(const f16v (&dW)[NO][NI], const float (&db)[NO], float *__restrict__ R, float *gW, float *gb,
                                             uint32_t out_dim, uint32_t in_dim, int lane, int wave, int nw) {
	float *Rb = R + NO * NI * 1024;
	for (int w = 0; w < nw; ++w) {
		if (wave == w) {
#pragma unroll
			for (int ot = 0; ot < NO; ++ot) {
				Rb[ot * 64 + lane] = (w == 0 ? 0.0f : Rb[ot * 64 + lane]) + db[ot];
#pragma unroll
				for (int it = 0; it < NI; ++it)
#pragma unroll
					for (int j = 0; j < 16; ++j) {
						const int e = (((ot * NI + it) * 16 + j) << 6) + lane;
						R[e] = (w == 0 ? 0.0f : R[e]) + dW[ot][it][j];
					}
			}
		}
		__syncthreads();
	}
	// dW accumulator (ot, it) register j of lane ln = dW[row 32 ot + 8 (j >> 2) + 4 (ln >> 5) + (j & 3)][column 32 it + (ln & 31)]
	for (uint32_t e = threadIdx.x; e < (uint32_t)(NO * NI * 1024); e += blockDim.x) {
		const uint32_t ln = e & 63u, j = (e >> 6) & 15u, it = (e >> 10) % NI, ot = (e >> 10) / NI;
		const uint32_t k = 32u * it + (ln & 31u), o = 32u * ot + 8u * (j >> 2) + 4u * (ln >> 5) + (j & 3u);
		if (o < out_dim && k < in_dim) atomic_add_f32(gW + (size_t)o * in_dim + k, R[e]);
	}
	if (gb)
		for (uint32_t e = threadIdx.x; e < (uint32_t)(NO * 32); e += blockDim.x) {
			const uint32_t o = e;                                       // 32 ot + row: lane (row, h) summed samples 8 h .. 8 h + 7 of both steps
			if (o < out_dim) atomic_add_f32(gb + o, Rb[(e >> 5) * 64 + (e & 31u)] + Rb[(e >> 5) * 64 + 32 + (e & 31u)]);
		}
	__syncthreads();
}

template <int NT>
__device__ __forceinline__ void zero_tiles(f16v (&r)[NT]) {
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int j = 0; j < 16; ++j) r[t][j] = 0.0f;
}

// FAST: 0 = any layout (row-major or, with a.x_fm, feature-major x), 1 = prefetched row-major x and dL/dy, 2 = prefetched
// feature-major x + row-major dL/dy
// Networks of 32-wide layers leave room for EIGHT waves per workgroup, two per SIMD: the kernel is a chain of LDS round trips and
// dependent MFMAs per tile, a second wave per SIMD hides half of it (dW of such a network is <= 64 registers; the cap is then 256 per
// lane).  64-wide hidden layers keep four waves and the whole register file (at 256 registers 32 -> 64 -> 64 -> 16 spills 300-400 dwords).
template <int IN_T, int W_T, int OUT_T> struct BwdCfg { static constexpr int kMaxWaves = (IN_T == 1 && W_T == 1 && OUT_T == 1) ? 8 : 4; };
constexpr int kMaxLdsBwd = 160 * 1024;
template <int IN_T, int W_T, int OUT_T, int NH, int FAST>
__global__ __launch_bounds__((BwdCfg<IN_T, W_T, OUT_T>::kMaxWaves * 64)) void k_mlph_bwd(BwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	stage_weights(a.packed, a.total_bytes, lds);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const int r = lane & 31;
	_Float16 *tiles_ = reinterpret_cast<_Float16 *>(lds + a.total_bytes) + (size_t)wave * a.tile_halfs;
	// tile rows: X | H_1 .. H_NH | G_out
	_Float16 *TX = tiles_;
	_Float16 *TH1 = tiles_ + 32 * IN_T * kTSH;                          // H_l at TH1 + (l - 1) * 32 * W_T * kTSH
	_Float16 *TGO = TH1 + NH * 32 * W_T * kTSH;
	constexpr uint32_t f0 = layer_bytes(IN_T, W_T), fh = layer_bytes(W_T, W_T), fo = layer_bytes(W_T, OUT_T);
	constexpr uint32_t t0 = layer_bytes(W_T, IN_T), th = fh, fwd_total = f0 + (NH - 1) * fh + fo;

	f16v dW0[W_T][IN_T], dWh[NH > 1 ? NH - 1 : 1][W_T][W_T], dWo[OUT_T][W_T];
	float db0[W_T], dbh[NH > 1 ? NH - 1 : 1][W_T], dbo[OUT_T];
#pragma unroll
	for (int ot = 0; ot < W_T; ++ot) { zero_tiles<IN_T>(dW0[ot]); db0[ot] = 0.0f; }
#pragma unroll
	for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
		for (int ot = 0; ot < W_T; ++ot) { zero_tiles<W_T>(dWh[l][ot]); dbh[l][ot] = 0.0f; }
#pragma unroll
	for (int ot = 0; ot < OUT_T; ++ot) { zero_tiles<W_T>(dWo[ot]); dbo[ot] = 0.0f; }

	const uint64_t n_tiles = (a.n + 31) / 32, step = (uint64_t)gridDim.x * nw;
	auto clamp_row = [&](uint64_t row) { return row < a.n ? row : a.n - 1; };
	h8 xnext[IN_T][2], gnext[OUT_T][2];
	if (FAST) {
		const uint64_t r0 = clamp_row(((uint64_t)blockIdx.x * nw + wave) * 32 + r);
		prefetch_x<FAST, IN_T>(a.x, a.xs, a.dims[0], r0, lane, xnext);
		load_rows_fast<OUT_T>(a.gy, a.gys, a.dims[NH + 1], r0, lane, gnext);
	}
	const h8 hzero = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += step) {
		const uint64_t row = tile * 32 + r;
		const bool valid = row < a.n;
		// the weight fragments (forward + transposed layers: up to 128 registers) must NOT be hoisted out of the tile loop -- the
		// dW accumulators own the register file: the LDS base goes through a register the compiler cannot see through
		uint32_t opaque;
		asm volatile("s_mov_b32 %0, 0" : "=s"(opaque));
		const unsigned char *wf = lds + opaque, *wt = wf + fwd_total;
		h8 xin[IN_T][2], g_out[OUT_T][2], hop[NH][W_T][2];
		f16v hacc[W_T];
		if (FAST) {
#pragma unroll
			for (int t = 0; t < IN_T; ++t) { xin[t][0] = xnext[t][0]; xin[t][1] = xnext[t][1]; }
#pragma unroll
			for (int t = 0; t < OUT_T; ++t) { g_out[t][0] = gnext[t][0]; g_out[t][1] = gnext[t][1]; }
			if (!valid) {                                                   // rows past n were clamped, not zeroed
#pragma unroll
				for (int t = 0; t < OUT_T; ++t) { g_out[t][0] = hzero; g_out[t][1] = hzero; }
			}
		} else {
			if (a.x_fm) load_cols_fast<IN_T>(a.x, a.xs, a.dims[0], clamp_row(row), lane, xin);   // rows past n: dL/dy is zero there
			else load_rows<IN_T>(a.x, a.xs, a.dims[0], row, valid, a.x_vec != 0, lane, xin);
			load_rows<OUT_T>(a.gy, a.gys, a.dims[NH + 1], row, valid, a.gy_vec != 0, lane, g_out);
		}
		// ---- forward, activations kept in operand form (registers) and as [feature][sample] tiles (LDS) ----
		write_tile<IN_T>(TX, xin, lane);
		dense<IN_T, W_T, true>(wf, xin, hacc, a.hidden_act, lane);
		to_operand<W_T>(hacc, hop[0]);
		write_tile<W_T>(TH1, hop[0], lane);
#pragma unroll
		for (int l = 1; l < NH; ++l) {
			dense<W_T, W_T, true>(wf + f0 + (l - 1) * fh, hop[l - 1], hacc, a.hidden_act, lane);
			to_operand<W_T>(hacc, hop[l]);
			write_tile<W_T>(TH1 + l * 32 * W_T * kTSH, hop[l], lane);
			}
		if (a.out_act == NR3D_MLP_ACT_RELU) {
			f16v yo[OUT_T];
			dense<W_T, OUT_T, true>(wf + f0 + (NH - 1) * fh, hop[NH - 1], yo, NR3D_MLP_ACT_NONE, lane);
#pragma unroll
			for (int t = 0; t < OUT_T; ++t)
#pragma unroll
				for (int j = 0; j < 16; ++j) if (!(yo[t][j] > 0.0f)) g_out[t][j >> 3][j & 7] = (_Float16)0.0f;
		}
		// ---- backward sweep ----
		const bool relu = a.hidden_act == NR3D_MLP_ACT_RELU;
		static_assert(W_T <= 2, "one 32-bit ReLU mask per hidden layer");
		constexpr bool BITS = (IN_T + OUT_T > 2) || (W_T == 1 && NH >= 3);      // the instantiations that spill with the activations live
		uint32_t hmask[NH];
#pragma unroll
		for (int l = 0; l < NH; ++l) {
			uint32_t m = 0;
			if (BITS) {
#pragma unroll
				for (int t = 0; t < W_T; ++t)
#pragma unroll
					for (int j = 0; j < 16; ++j) m |= ((float)hop[l][t][j >> 3][j & 7] > 0.0f ? 1u : 0u) << (16 * t + j);
			}
			hmask[l] = m;
		}
		f16v g[W_T];
		if (relu) bwd_layer<OUT_T, W_T, true, true, BITS>(g_out, TGO, TH1 + (NH - 1) * 32 * W_T * kTSH, wt + t0 + (NH - 1) * th, dWo, dbo, g, hop[NH - 1], hmask[NH - 1], lane);
		else bwd_layer<OUT_T, W_T, true, false, BITS>(g_out, TGO, TH1 + (NH - 1) * 32 * W_T * kTSH, wt + t0 + (NH - 1) * th, dWo, dbo, g, hop[NH - 1], 0u, lane);
		h8 gop[W_T][2];
#pragma unroll
		for (int l = NH - 1; l >= 1; --l) {                              // hidden layer l: H_l -> H_{l+1}
			f16v gp[W_T];
			to_operand<W_T>(g, gop);
			_Float16 *TG = TH1 + l * 32 * W_T * kTSH;                    // H_{l+1}'s tile is dead: the layer above has consumed it
			const _Float16 *TB = TH1 + (l - 1) * 32 * W_T * kTSH;
			if (relu) bwd_layer<W_T, W_T, true, true, BITS>(gop, TG, TB, wt + t0 + (l - 1) * th, dWh[l - 1], dbh[l - 1], gp, hop[l - 1], hmask[l - 1], lane);
			else bwd_layer<W_T, W_T, true, false, BITS>(gop, TG, TB, wt + t0 + (l - 1) * th, dWh[l - 1], dbh[l - 1], gp, hop[l - 1], 0u, lane);
#pragma unroll
			for (int t = 0; t < W_T; ++t) g[t] = gp[t];
			}
		to_operand<W_T>(g, gop);
		if constexpr (FAST != 0) {                                       // the next tile's rows, requested before this tile's last step (as csrc/mlp.hip)
			const uint64_t rn = clamp_row((tile + step) * 32 + r);
			prefetch_x<FAST, IN_T>(a.x, a.xs, a.dims[0], rn, lane, xnext);
			load_rows_fast<OUT_T>(a.gy, a.gys, a.dims[NH + 1], rn, lane, gnext);
		}
		f16v gx[IN_T];
		if (a.gx) {
			bwd_layer<W_T, IN_T, true, false, BITS>(gop, TH1, TX, wt, dW0, db0, gx, xin, 0u, lane);
			if (a.gx_fm) store_cols<IN_T>(a.gx, a.gxs, a.dims[0], row, valid, lane, gx);
			else store_rows<IN_T>(a.gx, a.gxs, a.dims[0], row, valid, a.gx_vec != 0, lane, gx);
		} else {
			bwd_layer<W_T, IN_T, false, false, BITS>(gop, TH1, TX, wt, dW0, db0, gx, xin, 0u, lane);
		}
	}

	// ---- reduce the waves' parameter gradients in LDS (the tile area is free now), one atomic per element ----
	__syncthreads();
	float *R = reinterpret_cast<float *>(lds + a.total_bytes);
	reduce_layer<W_T, IN_T>(dW0, db0, R, a.dW[0], a.db[0], a.dims[1], a.dims[0], lane, wave, nw);
#pragma unroll
	for (int l = 1; l < NH; ++l) reduce_layer<W_T, W_T>(dWh[l - 1], dbh[l - 1], R, a.dW[l], a.db[l], a.dims[l + 1], a.dims[l], lane, wave, nw);
	reduce_layer<OUT_T, W_T>(dWo, dbo, R, a.dW[NH], a.db[NH], a.dims[NH + 1], a.dims[NH], lane, wave, nw);
}


// =============================================================================================
// backward, dW SPLIT over the waves of the workgroup (round 5; the 64-wide hidden layers)
// =============================================================================================
// k_mlph_bwd keeps every layer's whole dW in each wave's accumulators over all of its tiles: 128 registers for 32 -> 64 -> 64 -> 16,
// 192 for 64 -> 64 -> 64 -> 64 -- one wave per SIMD with nothing to hide its chain of LDS round trips and dependent MFMAs behind
// (0.138 of HBM), and 127-189 spilled dwords for every shape with a 64-wide input or output (64 -> 64 -> 64 -> 64 backward: 2.9 ms
// where its FLOPs say 1.0).  But dW = sum over (tile, K step) of dPre^T . H, and any partition of those terms over the waves is
// as good as "a wave's own tiles": both operands already pass through LDS as [feature][sample] tiles.  Here the NW waves of a
// workgroup run their chains (forward, dL/dx) on their own tiles in lockstep rounds, and for the sample contraction wave w takes
// ONE 32 x 32 block of every layer's dW -- block w % NB of the layer's NB = NO x NI blocks -- over its share of the round's 2 NW
// (K step, tile) pairs, reading the other waves' tiles: 16 accumulator registers per layer instead of 16 NB, the same number of
// MFMAs per wave, two workgroup barriers per layer and round (tiles written / tiles free again).  The registers that frees let
// EIGHT waves share a CU where LDS allows (32 -> 64 -> 64 -> 16: 34 KB of weights + 8 x 15 KB of tiles).
template <int NO, int NI, int NW>
struct DwSplit {
	static constexpr int NB = NO * NI, G = NW / NB;          // blocks of the layer; waves per block
	static_assert(NB >= 1 && NW % NB == 0, "the blocks of a layer divide the waves of the workgroup");
};

// this wave's share of one layer's sample contraction for the round: block b = wave % NB (ot = b / NI, it = b % NI), the
// (step, tile) pairs p = G k + wave / NB.  TG_off / TB_off: offsets of the layer's dPre / input tiles inside a wave's tile area.
template <int NO, int NI, int NW>
__device__ __forceinline__ void dw_round(const _Float16 *__restrict__ tiles0, uint32_t tile_halfs, uint32_t TG_off, uint32_t TB_off,
                                         f16v &dW, float &db, int wave, int lane) {
	using S = DwSplit<NO, NI, NW>;
	const int b = wave % S::NB, sub = wave / S::NB, ot = b / NI, it = b % NI;
	const int r = lane & 31, h = lane >> 5;
	h8 av[2 * S::NB], bv[2 * S::NB];
#pragma unroll
	for (int k = 0; k < 2 * S::NB; ++k) {
		const int p = S::G * k + sub, st = p / NW, t = p % NW;
		const _Float16 *T = tiles0 + (size_t)t * tile_halfs;
		av[k] = read_op<NO>(T + TG_off, ot, st, lane);
		bv[k] = read_op<NI>(T + TB_off, it, st, lane);
	}
	float sum = 0.0f;
#pragma unroll
	for (int k = 0; k < 2 * S::NB; ++k) {
		if (it == 0) sum = sum8(av[k], sum);                 // (wave uniform) one wave group per out tile carries the bias gradient
		dW = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[k], bv[k], dW, 0, 0, 0);
	}
	db += sum;
}

// the workgroup's gradient of one layer: the G waves of a block add their accumulators in LDS one after the other, then one atomic
// per element (element order of R as in reduce_layer)
template <int NO, int NI, int NW>
__device__ __forceinline__ void reduce_split(const f16v &dW, float db, float *__restrict__ R, float *gW, float *gb, uint32_t out_dim,
                                             uint32_t in_dim, int lane, int wave) {
	using S = DwSplit<NO, NI, NW>;
	const int b = wave % S::NB, sub = wave / S::NB, ot = b / NI, it = b % NI;
	float *Rb = R + S::NB * 1024;
	for (int g = 0; g < S::G; ++g) {
		if (sub == g) {
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const int e = ((b * 16 + j) << 6) + lane;
				R[e] = (g == 0 ? 0.0f : R[e]) + dW[j];
			}
			if (it == 0) Rb[ot * 64 + lane] = (g == 0 ? 0.0f : Rb[ot * 64 + lane]) + db;
		}
		__syncthreads();
	}
	for (uint32_t e = threadIdx.x; e < (uint32_t)(S::NB * 1024); e += blockDim.x) {
		const uint32_t ln = e & 63u, j = (e >> 6) & 15u, bi = (e >> 10) % NI, bo = (e >> 10) / NI;
		const uint32_t k = 32u * bi + (ln & 31u), o = 32u * bo + 8u * (j >> 2) + 4u * (ln >> 5) + (j & 3u);
		if (o < out_dim && k < in_dim) atomic_add_f32(gW + (size_t)o * in_dim + k, R[e]);
	}
	if (gb)
		for (uint32_t e = threadIdx.x; e < (uint32_t)(NO * 32); e += blockDim.x)
			if (e < out_dim) atomic_add_f32(gb + e, Rb[(e >> 5) * 64 + (e & 31u)] + Rb[(e >> 5) * 64 + 32 + (e & 31u)]);
	__syncthreads();
}

// dL/d(pre-activation of the layer below) in operand form = half(W^T . dPre), masked by the ReLU mask of that layer's output
template <int NO, int NI>
__device__ __forceinline__ void dx_layer(const h8 (&g)[NO][2], const unsigned char *__restrict__ wT, h8 (&gop)[NI][2], const us8 (&mask)[NI][2],
                                         bool relu, int lane) {
	f16v gp[NI];
	dense<NO, NI, false>(wT, g, gp, NR3D_MLP_ACT_NONE, lane);
	to_operand<NI>(gp, gop);
	if (relu) {
#pragma unroll
		for (int t = 0; t < NI; ++t) { gop[t][0] = apply_mask(gop[t][0], mask[t][0]); gop[t][1] = apply_mask(gop[t][1], mask[t][1]); }
	}
}

template <int IN_T, int W_T, int OUT_T, int NH, int FAST, int NW>
__global__ __launch_bounds__(NW * 64) void k_mlph_bwd_split(BwdArgs a) {
	static_assert(W_T == 2 && NH >= 1 && NH <= 2, "the shapes whose dW does not fit one wave: 64-wide hidden layers");
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	stage_weights(a.packed, a.total_bytes, lds);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int r = lane & 31;
	_Float16 *tiles0 = reinterpret_cast<_Float16 *>(lds + a.total_bytes);
	_Float16 *mine = tiles0 + (size_t)wave * a.tile_halfs;
	// tile rows of a wave: X | H_1 .. H_NH | G_out (dPre of hidden layer l + 1 overwrites H_{l+1}, whose readers are done by then)
	constexpr uint32_t oX = 0, oH1 = 32 * IN_T * kTSH, szH = 32 * W_T * kTSH, oGO = oH1 + NH * szH;
	constexpr uint32_t f0 = layer_bytes(IN_T, W_T), fh = layer_bytes(W_T, W_T), fo = layer_bytes(W_T, OUT_T);
	constexpr uint32_t t0 = layer_bytes(W_T, IN_T), th = fh, fwd_total = f0 + (NH - 1) * fh + fo;

	f16v dW0, dWh, dWo;                                  // ONE block of every layer (dWh unused for NH == 1)
	float db0 = 0.0f, dbh = 0.0f, dbo = 0.0f;
#pragma unroll
	for (int j = 0; j < 16; ++j) { dW0[j] = 0.0f; dWh[j] = 0.0f; dWo[j] = 0.0f; }

	const uint64_t n_tiles = (a.n + 31) / 32, per_round = (uint64_t)gridDim.x * NW;
	const uint64_t n_rounds = (n_tiles + per_round - 1) / per_round;                 // the same for every wave: the barriers below
	auto clamp_row = [&](uint64_t row) { return row < a.n ? row : a.n - 1; };
	h8 xnext[IN_T][2], gnext[OUT_T][2];
	if (FAST) {
		const uint64_t r0 = clamp_row(((uint64_t)blockIdx.x * NW + wave) * 32 + r);
		prefetch_x<FAST, IN_T>(a.x, a.xs, a.dims[0], r0, lane, xnext);
		load_rows_fast<OUT_T>(a.gy, a.gys, a.dims[NH + 1], r0, lane, gnext);
	}
	const h8 hzero = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
	const bool relu = a.hidden_act == NR3D_MLP_ACT_RELU;
	// the next round's rows are requested at the top of a round, or before its last layer where that measured faster (same-box A/B at
	// 2^22 samples, fwd+bwd ms: 32 -> 64 -> 64 -> 16 0.624 -> 0.591; 32 -> 64 -> 16 0.326 -> 0.340 and 64 -> 64 -> 64 -> 64 1.156 -> 1.206 the other way)
	constexpr bool kLate = NH == 2 && IN_T == 1 && OUT_T == 1;
	for (uint64_t round = 0; round < n_rounds; ++round) {
		const uint64_t tile = round * per_round + (uint64_t)blockIdx.x * NW + wave;
		const uint64_t row = tile * 32 + r;
		const bool valid = row < a.n;                      // a tile past the end runs on clamped rows with dL/dy = 0: it adds zeros
		uint32_t opaque;                                   // (no hoisting of the weight fragments out of the loop: see k_mlph_bwd)
		asm volatile("s_mov_b32 %0, 0" : "=s"(opaque));
		const unsigned char *wf = lds + opaque, *wt = wf + fwd_total;
		h8 xin[IN_T][2], g_out[OUT_T][2], hop[W_T][2];
		f16v hacc[W_T];
		us8 hmask[NH][W_T][2];
		if (FAST) {
#pragma unroll
			for (int t = 0; t < IN_T; ++t) { xin[t][0] = xnext[t][0]; xin[t][1] = xnext[t][1]; }
#pragma unroll
			for (int t = 0; t < OUT_T; ++t) { g_out[t][0] = gnext[t][0]; g_out[t][1] = gnext[t][1]; }
			if constexpr (!kLate) {
				const uint64_t rn = clamp_row((tile + per_round) * 32 + r);
				prefetch_x<FAST, IN_T>(a.x, a.xs, a.dims[0], rn, lane, xnext);
				load_rows_fast<OUT_T>(a.gy, a.gys, a.dims[NH + 1], rn, lane, gnext);
			}
			if (!valid) {
#pragma unroll
				for (int t = 0; t < OUT_T; ++t) { g_out[t][0] = hzero; g_out[t][1] = hzero; }
			}
		} else {
			if (a.x_fm) load_cols_fast<IN_T>(a.x, a.xs, a.dims[0], clamp_row(row), lane, xin);
			else load_rows<IN_T>(a.x, a.xs, a.dims[0], clamp_row(row), true, a.x_vec != 0, lane, xin);
			load_rows<OUT_T>(a.gy, a.gys, a.dims[NH + 1], row, valid, a.gy_vec != 0, lane, g_out);
		}
		// ---- forward on the wave's own tile: activations to LDS as [feature][sample], their signs as one word per layer ----
		write_tile<IN_T>(mine + oX, xin, lane);
		dense<IN_T, W_T, true>(wf, xin, hacc, a.hidden_act, lane);
		to_operand<W_T>(hacc, hop);
		write_tile<W_T>(mine + oH1, hop, lane);
		auto signs = [&](us8 (&m)[W_T][2]) {
#pragma unroll
			for (int t = 0; t < W_T; ++t) { m[t][0] = relu_mask(hop[t][0]); m[t][1] = relu_mask(hop[t][1]); }
		};
		signs(hmask[0]);
		if constexpr (NH == 2) {
			h8 hin[W_T][2];
#pragma unroll
			for (int t = 0; t < W_T; ++t) { hin[t][0] = hop[t][0]; hin[t][1] = hop[t][1]; }
			dense<W_T, W_T, true>(wf + f0, hin, hacc, a.hidden_act, lane);
			to_operand<W_T>(hacc, hop);
			write_tile<W_T>(mine + oH1 + szH, hop, lane);
			signs(hmask[1]);
		}
		if (a.out_act == NR3D_MLP_ACT_RELU) {
			f16v yo[OUT_T];
			dense<W_T, OUT_T, true>(wf + f0 + (NH - 1) * fh, hop, yo, NR3D_MLP_ACT_NONE, lane);
#pragma unroll
			for (int t = 0; t < OUT_T; ++t)
#pragma unroll
				for (int j = 0; j < 16; ++j) if (!(yo[t][j] > 0.0f)) g_out[t][j >> 3][j & 7] = (_Float16)0.0f;
		}
		// ---- output layer ----
		write_tile<OUT_T>(mine + oGO, g_out, lane);
		__syncthreads();                                   // every wave's G_out, H and X tiles are in LDS
		dw_round<OUT_T, W_T, NW>(tiles0, a.tile_halfs, oGO, oH1 + (NH - 1) * szH, dWo, dbo, wave, lane);
		h8 gop[W_T][2];
		dx_layer<OUT_T, W_T>(g_out, wt + t0 + (NH - 1) * th, gop, hmask[NH - 1], relu, lane);
		__syncthreads();                                   // H_NH has been read by everyone: its rows take dPre of the layer below
		if constexpr (NH == 2) {
			write_tile<W_T>(mine + oH1 + szH, gop, lane);
			__syncthreads();
			dw_round<W_T, W_T, NW>(tiles0, a.tile_halfs, oH1 + szH, oH1, dWh, dbh, wave, lane);
			h8 gin[W_T][2];
#pragma unroll
			for (int t = 0; t < W_T; ++t) { gin[t][0] = gop[t][0]; gin[t][1] = gop[t][1]; }
			dx_layer<W_T, W_T>(gin, wt + t0, gop, hmask[0], relu, lane);
			__syncthreads();
		}
		// ---- first layer ----
		if constexpr (FAST != 0 && kLate) {                              // the next round's rows
			const uint64_t rn = clamp_row((tile + per_round) * 32 + r);
			prefetch_x<FAST, IN_T>(a.x, a.xs, a.dims[0], rn, lane, xnext);
			load_rows_fast<OUT_T>(a.gy, a.gys, a.dims[NH + 1], rn, lane, gnext);
		}
		write_tile<W_T>(mine + oH1, gop, lane);
		__syncthreads();
		dw_round<W_T, IN_T, NW>(tiles0, a.tile_halfs, oH1, oX, dW0, db0, wave, lane);
		if (a.gx) {
			f16v gx[IN_T];
			dense<W_T, IN_T, false>(wt, gop, gx, NR3D_MLP_ACT_NONE, lane);
			if (a.gx_fm) store_cols<IN_T>(a.gx, a.gxs, a.dims[0], row, valid, lane, gx);
			else store_rows<IN_T>(a.gx, a.gxs, a.dims[0], row, valid, a.gx_vec != 0, lane, gx);
		}
		__syncthreads();                                   // X and H_1 are free for the next round
	}
	float *R = reinterpret_cast<float *>(lds + a.total_bytes);
	reduce_split<W_T, IN_T, NW>(dW0, db0, R, a.dW[0], a.db[0], a.dims[1], a.dims[0], lane, wave);
	if constexpr (NH == 2) reduce_split<W_T, W_T, NW>(dWh, dbh, R, a.dW[1], a.db[1], a.dims[2], a.dims[1], lane, wave);
	reduce_split<OUT_T, W_T, NW>(dWo, dbo, R, a.dW[NH], a.db[NH], a.dims[NH + 1], a.dims[NH], lane, wave);
}

}  // namespace mlph
}  // namespace nr3d

using namespace nr3d;
using namespace nr3d::mlph;

extern "C" uint64_t nr3d_mlp_half_packed_bytes(const nr3d_mlp_desc_t *desc) {
	Shape s;
	if (!shape_of(desc, s)) return 0;
	const uint64_t n = packed_bytes(s);
	return n <= (uint64_t)kMaxLds ? n : 0;
}

static bool backward_ok(const Shape &s) {        // dW of every layer lives in accumulator registers (as csrc/mlp.hip)
	if (s.w_t > 2 || s.in_t > s.w_t || s.out_t > s.w_t) return false;
	const uint32_t nh = s.n_layers - 1;
	return s.w_t == 1 ? nh <= 3 : nh <= 2;
}
// per-wave [feature][sample] tiles: X, H_1 .. H_NH, G_out
static uint32_t bwd_tile_halfs(const Shape &s) { return (32u * s.in_t + (s.n_layers - 1) * 32u * s.w_t + 32u * s.out_t) * (uint32_t)kTSH; }
static uint32_t bwd_waves(const Shape &s) {
	const uint64_t wbytes = packed_bytes(s) + transposed_bytes(s);
	const uint64_t reduce = ((uint64_t)s.w_t * s.w_t * 1024 + (uint64_t)s.w_t * 64) * 4;
	const uint32_t max_waves = (s.in_t == 1 && s.w_t == 1 && s.out_t == 1) ? 8u : 4u;      // = BwdCfg<IN_T, W_T, OUT_T>::kMaxWaves
	for (uint32_t nw = max_waves; nw >= 1; --nw) {
		const uint64_t t = (uint64_t)nw * bwd_tile_halfs(s) * 2;
		if (wbytes + (t > reduce ? t : reduce) <= (uint64_t)(nw > 4 ? kMaxLdsBwd : kMaxLds)) return nw;
	}
	return 0;
}

// k_mlph_bwd_split: eight waves when LDS holds the weights (forward + transposed) and eight waves' tiles, else four
static uint32_t split_waves(const Shape &s) {
	const uint64_t wbytes = packed_bytes(s) + transposed_bytes(s);
	const uint64_t reduce = ((uint64_t)s.w_t * s.w_t * 1024 + (uint64_t)s.w_t * 64) * 4;
	// (eight waves = 256 registers per lane: only the narrow-input, narrow-output shapes stay clear of scratch there)
#ifdef NR3D_MLPH_FORCE_NW4
	const uint32_t nw_first = 4;
#else
	const uint32_t nw_first = (s.in_t == 1 && s.out_t == 1) ? 8 : 4;
#endif
	for (uint32_t nw = nw_first; nw >= 4; nw -= 4) {
		const uint64_t t = (uint64_t)nw * bwd_tile_halfs(s) * 2;
		if (wbytes + (t > reduce ? t : reduce) <= (uint64_t)kMaxLdsBwd) return nw;
	}
	return 0;
}

extern "C" uint64_t nr3d_mlp_half_backward_packed_bytes(const nr3d_mlp_desc_t *desc) {
	Shape s;
	if (!shape_of(desc, s) || nr3d_mlp_half_packed_bytes(desc) == 0 || !backward_ok(s) || (s.w_t == 2 ? split_waves(s) : bwd_waves(s)) == 0) return 0;
	return transposed_bytes(s);
}

extern "C" int nr3d_mlp_half_pack(const nr3d_mlp_desc_t *desc, const void *const *weights, const void *const *biases, void *packed,
                                  int with_backward, void *stream) {
	Shape s;
	NR3D_CHECK(shape_of(desc, s) && nr3d_mlp_half_packed_bytes(desc) != 0, "mlp_half_pack: network outside the fused kernels' range "
	           "(2..%d linear layers, every width 1..128)", NR3D_MLP_MAX_LAYERS);
	NR3D_CHECK(weights && packed, "mlp_half_pack: NULL pointer");
	NR3D_CHECK(!with_backward || nr3d_mlp_half_backward_packed_bytes(desc) != 0, "mlp_half_pack: the fused backward does not apply to this network");
	PackArgs p;
	p.n_layers = desc->n_layers; p.transposed = 0;
	uint32_t off = 0;
	for (uint32_t l = 0; l < desc->n_layers; ++l) {
		NR3D_CHECK(weights[l] != nullptr, "mlp_half_pack: weights[%u] is NULL", l);
		p.w[l] = (const __half *)weights[l];
		p.b[l] = biases ? (const __half *)biases[l] : nullptr;
		p.in_dim[l] = desc->dims[l]; p.out_dim[l] = desc->dims[l + 1];
		p.ni[l] = l == 0 ? s.in_t : s.w_t;
		p.no[l] = l + 1 == desc->n_layers ? s.out_t : s.w_t;
		p.offset[l] = off;
		off += layer_bytes(p.ni[l], p.no[l]);
	}
	p.offset[desc->n_layers] = off;
	hipLaunchKernelGGL(k_mlph_pack, dim3(16, desc->n_layers), dim3(256), 0, (hipStream_t)stream, p, (unsigned char *)packed);
	if (with_backward) {
		PackArgs t = p;
		t.transposed = 1;
		uint32_t toff = 0;
		for (uint32_t l = 0; l < desc->n_layers; ++l) {
			t.ni[l] = p.no[l]; t.no[l] = p.ni[l];
			t.offset[l] = toff;
			toff += layer_bytes(t.ni[l], t.no[l]);
		}
		t.offset[desc->n_layers] = toff;
		hipLaunchKernelGGL(k_mlph_pack, dim3(16, desc->n_layers), dim3(256), 0, (hipStream_t)stream, t, (unsigned char *)packed + packed_bytes(s));
	}
	NR3D_LAUNCH_CHECK();
	return 0;
}

#define MLPH_DISPATCH(S, ...)                                                                                  \
	do {                                                                                                       \
		const uint32_t _i = (S).in_t, _w = (S).w_t, _o = (S).out_t;                                            \
		auto _go = [&](auto I, auto W, auto O) { constexpr int IN_T = decltype(I)::value, W_T = decltype(W)::value, OUT_T = decltype(O)::value; __VA_ARGS__; }; \
		auto _ow = [&](auto I, auto W) {                                                                       \
			if (_o == 1) _go(I, W, std::integral_constant<int, 1>{});                                          \
			else if (_o == 2) _go(I, W, std::integral_constant<int, 2>{});                                     \
			else _go(I, W, std::integral_constant<int, 4>{}); };                                               \
		auto _iw = [&](auto I) {                                                                               \
			if (_w == 1) _ow(I, std::integral_constant<int, 1>{});                                             \
			else if (_w == 2) _ow(I, std::integral_constant<int, 2>{});                                        \
			else _ow(I, std::integral_constant<int, 4>{}); };                                                  \
		if (_i == 1) _iw(std::integral_constant<int, 1>{});                                                    \
		else if (_i == 2) _iw(std::integral_constant<int, 2>{});                                               \
		else _iw(std::integral_constant<int, 4>{});                                                            \
	} while (0)

extern "C" int nr3d_mlp_half_forward(const nr3d_mlp_desc_t *desc, uint64_t n, const void *x, int64_t x_stride, int64_t x_feature_stride,
                                     const void *packed, void *y, int64_t y_stride, void *stream) {
	Shape s;
	NR3D_CHECK(shape_of(desc, s) && nr3d_mlp_half_packed_bytes(desc) != 0, "mlp_half_forward: network outside the fused kernels' range");
	if (n == 0) return 0;
	NR3D_CHECK(x && packed && y, "mlp_half_forward: NULL pointer");
	FwdArgs a;
	const bool x_fm = x_feature_stride != 1;
	NR3D_CHECK(!x_fm || x_stride == 1, "mlp_half_forward: x must be row-major (feature stride 1) or feature-major (row stride 1)");
	a.n = n; a.x = (const __half *)x; a.xs = x_fm ? x_feature_stride : x_stride; a.y = (__half *)y; a.ys = y_stride;
	a.packed = (const unsigned char *)packed;
	a.packed_bytes = (uint32_t)packed_bytes(s);
	a.n_layers = desc->n_layers; a.in_dim = desc->dims[0]; a.out_dim = desc->dims[desc->n_layers];
	a.hidden_act = (int)desc->hidden_activation; a.out_act = (int)desc->output_activation;
	a.x_vec = ((uintptr_t)x % 8 == 0 && x_stride % 4 == 0) ? 1u : 0u;
	a.y_vec = ((uintptr_t)y % 8 == 0 && y_stride % 4 == 0) ? 1u : 0u;
	const size_t lds = (size_t)a.packed_bytes;
	const uint64_t n_tiles = (n + 31) / 32;
	// 17 KB of weights per workgroup: several workgroups share a CU (the kernel streams x / y; waves hide each other's latency)
	const uint32_t grid = (uint32_t)(n_tiles / 4 + 1 < 2048 ? n_tiles / 4 + 1 : 2048);
	int rc = 0;
	MLPH_DISPATCH(s, {
		static bool attr[64] = {};
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess) { rc = ::nr3d::fail("mlp_half_forward: hipGetDevice failed"); return; }
		if (!attr[dev & 63]) {
			if (hipFuncSetAttribute((const void *)k_mlph_fwd<IN_T, W_T, OUT_T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess ||
			    hipFuncSetAttribute((const void *)k_mlph_fwd<IN_T, W_T, OUT_T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess ||
			    hipFuncSetAttribute((const void *)k_mlph_fwd<IN_T, W_T, OUT_T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
				rc = ::nr3d::fail("mlp_half_forward: cannot raise the dynamic LDS limit"); return;
			}
			attr[dev & 63] = true;
		}
		if (x_fm)
			hipLaunchKernelGGL((k_mlph_fwd<IN_T, W_T, OUT_T, 2>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
		else if (a.x_vec && a.in_dim % 4 == 0)
			hipLaunchKernelGGL((k_mlph_fwd<IN_T, W_T, OUT_T, 1>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
		else
			hipLaunchKernelGGL((k_mlph_fwd<IN_T, W_T, OUT_T, 0>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
	});
	if (rc) return rc;
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_mlp_half_backward(const nr3d_mlp_desc_t *desc, uint64_t n, const void *x, int64_t x_stride, int64_t x_feature_stride,
                                      const void *dL_dy, int64_t gy_stride, const void *packed, void *dL_dx, int64_t gx_stride,
                                      int64_t gx_feature_stride, float *const *dL_dW, float *const *dL_db, void *stream) {
	Shape s;
	NR3D_CHECK(shape_of(desc, s) && nr3d_mlp_half_backward_packed_bytes(desc) != 0, "mlp_half_backward: the fused backward does not apply to this network");
	if (n == 0) return 0;
	NR3D_CHECK(x && dL_dy && packed && dL_dW, "mlp_half_backward: NULL pointer");
	BwdArgs a;
	const bool x_fm = x_feature_stride != 1, gx_fm = dL_dx && gx_feature_stride != 1;
	NR3D_CHECK(!x_fm || x_stride == 1, "mlp_half_backward: x must be row-major (feature stride 1) or feature-major (row stride 1)");
	NR3D_CHECK(!gx_fm || gx_stride == 1, "mlp_half_backward: dL_dx must be row-major (feature stride 1) or feature-major (row stride 1)");
	a.n = n; a.x = (const __half *)x; a.xs = x_fm ? x_feature_stride : x_stride; a.gy = (const __half *)dL_dy; a.gys = gy_stride;
	a.packed = (const unsigned char *)packed;
	a.gx = (__half *)dL_dx; a.gxs = gx_fm ? gx_feature_stride : gx_stride;
	a.x_fm = x_fm ? 1u : 0u; a.gx_fm = gx_fm ? 1u : 0u;
	a.fwd_bytes = (uint32_t)packed_bytes(s);
	a.total_bytes = a.fwd_bytes + (uint32_t)transposed_bytes(s);
	for (uint32_t l = 0; l < desc->n_layers; ++l) {
		NR3D_CHECK(dL_dW[l] != nullptr, "mlp_half_backward: dL_dW[%u] is NULL", l);
		a.dW[l] = dL_dW[l];
		a.db[l] = dL_db ? dL_db[l] : nullptr;
	}
	for (uint32_t l = 0; l <= desc->n_layers; ++l) a.dims[l] = desc->dims[l];
	a.n_layers = desc->n_layers;
	a.hidden_act = (int)desc->hidden_activation; a.out_act = (int)desc->output_activation;
	a.x_vec = ((uintptr_t)x % 8 == 0 && x_stride % 4 == 0) ? 1u : 0u;
	a.gy_vec = ((uintptr_t)dL_dy % 8 == 0 && gy_stride % 4 == 0) ? 1u : 0u;
	a.gx_vec = (dL_dx && (uintptr_t)dL_dx % 8 == 0 && gx_stride % 4 == 0) ? 1u : 0u;
	a.tile_halfs = bwd_tile_halfs(s);
	const uint32_t nh = desc->n_layers - 1;
	const bool gy_fast = a.gy_vec && desc->dims[desc->n_layers] % 4 == 0;
	const int fast = !gy_fast ? 0 : x_fm ? 2 : (a.x_vec && desc->dims[0] % 4 == 0) ? 1 : 0;
	const uint64_t n_tiles = (n + 31) / 32;
	const uint64_t reduce = ((uint64_t)s.w_t * s.w_t * 1024 + (uint64_t)s.w_t * 64) * 4;
	int rc = 0;
	if (s.w_t == 2) {
		// 64-wide hidden layers: dW split over the waves of the workgroup (k_mlph_bwd_split), eight waves where LDS holds their tiles
		const uint32_t nw = split_waves(s);
		const uint64_t tbytes = (uint64_t)nw * a.tile_halfs * 2;
		const size_t lds = (size_t)a.total_bytes + (size_t)(tbytes > reduce ? tbytes : reduce);
		const uint32_t grid = (uint32_t)(n_tiles / nw + 1 < 256 ? n_tiles / nw + 1 : 256);
		auto launch = [&](auto kern) -> int {
			NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBwd));
			hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, (hipStream_t)stream, a);
			return 0;
		};
#define SPLIT_LAUNCH(I, O, H, NW_) (fast == 2 ? launch(k_mlph_bwd_split<I, 2, O, H, 2, NW_>) : fast == 1 ? launch(k_mlph_bwd_split<I, 2, O, H, 1, NW_>) : launch(k_mlph_bwd_split<I, 2, O, H, 0, NW_>))
#define SPLIT_CASE(I, O, H) if (s.in_t == I && s.out_t == O && nh == H) rc = SPLIT_LAUNCH(I, O, H, 4); else
		if (s.in_t == 1 && s.out_t == 1 && nh == 1) rc = nw == 8 ? SPLIT_LAUNCH(1, 1, 1, 8) : SPLIT_LAUNCH(1, 1, 1, 4);
		else if (s.in_t == 1 && s.out_t == 1 && nh == 2) rc = nw == 8 ? SPLIT_LAUNCH(1, 1, 2, 8) : SPLIT_LAUNCH(1, 1, 2, 4);
		else
		SPLIT_CASE(1, 2, 1) SPLIT_CASE(1, 2, 2) SPLIT_CASE(2, 1, 1) SPLIT_CASE(2, 1, 2) SPLIT_CASE(2, 2, 1) SPLIT_CASE(2, 2, 2)
		rc = ::nr3d::fail("mlp_half_backward: no kernel for this shape");
#undef SPLIT_LAUNCH
#undef SPLIT_CASE
		if (rc) return rc;
		NR3D_LAUNCH_CHECK();
		return 0;
	}
	const uint32_t nw = bwd_waves(s);
	const uint64_t tbytes = (uint64_t)nw * a.tile_halfs * 2;
	const size_t lds = (size_t)a.total_bytes + (size_t)(tbytes > reduce ? tbytes : reduce);
	const uint32_t grid = (uint32_t)(n_tiles / nw + 1 < 256 ? n_tiles / nw + 1 : 256);     // one workgroup per CU: dW lives in registers
	auto launch = [&](auto kern) -> int {
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBwd));
		hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, (hipStream_t)stream, a);
		return 0;
	};
#define BWD_CASE(I, W, O, H) if (s.in_t == I && s.w_t == W && s.out_t == O && nh == H) \
		rc = fast == 2 ? launch(k_mlph_bwd<I, W, O, H, 2>) : fast == 1 ? launch(k_mlph_bwd<I, W, O, H, 1>) : launch(k_mlph_bwd<I, W, O, H, 0>); else
	BWD_CASE(1, 1, 1, 1) BWD_CASE(1, 1, 1, 2) BWD_CASE(1, 1, 1, 3)
	rc = ::nr3d::fail("mlp_half_backward: no kernel for this shape");
#undef BWD_CASE
	if (rc) return rc;
	NR3D_LAUNCH_CHECK();
	return 0;
}
