// nr3d_lib_amd/csrc/mlp.hip -- fused fully-connected decoder (gfx950), C-ABI entry points
// nr3d_mlp_packed_floats / nr3d_mlp_pack / nr3d_mlp_forward / nr3d_mlp_backward.
//
// The step right after the encoder: nr3d_lib/models/blocks/mlp.py:27-127 (`MLP` / `FCBlock`: D hidden
// DenseLayers of width W + an output layer, nr3d_lib/models/layers.py:228-300) -- in the reference a chain of
// cuBLAS GEMMs + elementwise kernels, or tiny-cuda-nn's fused fp16 MLP behind tcnn_adapter.py.  Here: fp32 in, fp32
// accumulate on the f32 MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, so parity with an fp32 reference is
// round-off only), the whole network in ONE kernel: activations never leave registers.
//
// Layout trick: everything is computed TRANSPOSED, H^T[feature, sample] = W . X^T, per wave on a tile of 32 samples.
// The MFMA's C/D map puts sample = lane & 31 in every lane and features 8a + 4(lane >> 5) + b (a, b < 4) in its 16
// accumulator registers; its B operand wants sample = lane & 31 and one k per half-wave.  Feeding register j of the
// previous layer's accumulator as B of step j means half-wave h contributes feature 8(j>>2) + 4h + (j&3) -- a fixed
// permutation of k, absorbed into the order the weights are packed in.  So the output registers of one layer ARE the
// input operands of the next: no LDS round trip, no shuffles.  The input X and the output Y use the same map, which
// makes every lane read / write 16-byte pieces of its sample's row.
// Weights (packed once per step by k_mlp_pack into MFMA operand order, zero padded to 32-wide tiles) sit in LDS.
//
// Backward recomputes the forward from X (keeping the activations in registers), walks the layers down with
// dH^T = W^T . dY^T on the same register map, and accumulates dW = dH^T . H_prev over the wave's samples: that
// product contracts over SAMPLES, which the register map has in the lane index, so both operands go through a
// per-wave LDS transposition tile ([feature][sample] -> one sample pair per MFMA step).
#include "common.h"
#include "mlp_device.h"      // register map, dense layers (f32 MFMA / bf16 MFMA x3), loads and stores: shared with lotd_mlp.hip
#include <type_traits>

namespace nr3d {
namespace mlp {

struct Shape {
	uint32_t n_layers;                         // linear layers (hidden + output)
	uint32_t in_t, w_t, out_t;                 // 32-wide tiles of the input, the (widest) hidden layer, the output
};

static bool shape_of(const nr3d_mlp_desc_t *d, Shape &s) {
	if (!d || d->n_layers < 2 || d->n_layers > NR3D_MLP_MAX_LAYERS) return false;
	uint32_t w = 0;
	for (uint32_t l = 1; l < d->n_layers; ++l) w = d->dims[l] > w ? d->dims[l] : w;
	for (uint32_t l = 0; l <= d->n_layers; ++l) if (d->dims[l] == 0 || d->dims[l] > 128) return false;
	s.n_layers = d->n_layers;
	// 3-tile widths run on the 4-tile instantiation (one all-zero tile)
	auto round = [](uint32_t t) { return t == 3 ? 4u : t; };
	s.in_t = round(tiles(d->dims[0])); s.w_t = round(tiles(w)); s.out_t = round(tiles(d->dims[d->n_layers]));
	return true;
}

static uint64_t packed_floats(const Shape &s) {
	return (uint64_t)layer_floats(s.in_t, s.w_t) + (uint64_t)(s.n_layers - 2) * layer_floats(s.w_t, s.w_t) + layer_floats(s.w_t, s.out_t);
}

// ---------------------------------------------------------------------------------------------
// fp32 forward on the bf16 MFMA (round 5, "x3"): every fp32 value is split into three bf16 pieces, v = v1 + v2 + v3 with
// v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2) (24 significant bits; the subtractions are exact), and a product is the
// six piece products whose magnitude is above 2^-25 of it: w1 x1 + (w1 x2 + w2 x1) + (w2 x2 + w1 x3 + w3 x1), each exact in the
// multiplier and accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- the result is fp32-grade (dropped terms <= 2^-26 of a product,
// below the rounding of an fp32 multiply), at 6 MFMAs of 8 passes per K = 16 where the f32 MFMA needs 8 of 16 passes: 2.7x the
// matrix rate of v_mfma_f32_32x32x2_f32, which bounds csrc/mlp.hip's forward (0.55-0.63 of its 157 TFLOP/s).
// The weights' pieces are made once at pack time, in the operand order of csrc/mlp_half.hip (A of step s, lane (out i, h),
// element e = W[i][32 it + 8 (2 s + (e >> 2)) + 4 h + (e & 3)]); the activations' pieces per layer from the register map (a step's
// B operand = eight consecutive accumulator registers).  Region of the packed buffer: behind the f32 forward layers.
// ---------------------------------------------------------------------------------------------
static uint64_t x3_floats(const Shape &s) {
	const uint64_t n = (uint64_t)layer_x3_floats(s.in_t, s.w_t) + (uint64_t)(s.n_layers - 2) * layer_x3_floats(s.w_t, s.w_t) + layer_x3_floats(s.w_t, s.out_t);
	return n * 4 <= (uint64_t)kMaxLds ? n : 0;          // a network whose pieces do not fit LDS keeps the f32 MFMA
}
static bool x3_enabled() { return opt::on(NR3D_OPT_MLP_X3); }

// ---------------------------------------------------------------------------------------------
// packing: for layer l, packed[(((ot*NI + it)*4 + a)*64 + lane)*4 + b] = W[32 ot + (lane & 31)][32 it + 8a + 4(lane >> 5) + b]
// (no transposed layers are packed: the backward reads these transposed, mlp_device.h dense_t / dense_x3_t)
// ---------------------------------------------------------------------------------------------
struct PackArgs {
	const float *w[NR3D_MLP_MAX_LAYERS];
	const float *b[NR3D_MLP_MAX_LAYERS];
	uint32_t in_dim[NR3D_MLP_MAX_LAYERS], out_dim[NR3D_MLP_MAX_LAYERS];   // of W as stored: [out_dim, in_dim] row-major
	uint32_t ni[NR3D_MLP_MAX_LAYERS], no[NR3D_MLP_MAX_LAYERS];            // tiles of the packed layer's input / output
	uint32_t offset[NR3D_MLP_MAX_LAYERS + 1];                             // first float of every packed layer
	uint32_t n_layers;
};

__global__ __launch_bounds__(256) void k_mlp_pack(PackArgs a, float *__restrict__ packed) {
	const uint32_t l = blockIdx.y;
	const uint32_t nf = a.offset[l + 1] - a.offset[l];
	const uint32_t nw = a.no[l] * a.ni[l] * 1024u;
	for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nf; e += gridDim.x * 256) {
		float v = 0.0f;
		if (e < nw) {
			const uint32_t b = e & 3u, lane = (e >> 2) & 63u, q = (e >> 8) & 3u, tile = e >> 10;
			const uint32_t it = tile % a.ni[l], ot = tile / a.ni[l];
			const uint32_t o = 32u * ot + (lane & 31u), f = 32u * it + 8u * q + 4u * (lane >> 5) + b;
			if (o < a.out_dim[l] && f < a.in_dim[l]) v = a.w[l][(size_t)o * a.in_dim[l] + f];
		} else if (a.b[l]) {
			const uint32_t o = e - nw;
			if (o < a.out_dim[l]) v = a.b[l][o];
		}
		packed[a.offset[l] + e] = v;
	}
}

// x3 planes of layer l: bf16 index e of plane p at ((((p * NO + ot) * NI + it) * 2 + s) * 64 + lane) * 8 + el; bias fp32 behind them
__global__ __launch_bounds__(256) void k_mlp_pack_x3(PackArgs a, float *__restrict__ packed) {
	const uint32_t l = blockIdx.y;
	const uint32_t nw = a.no[l] * a.ni[l] * 1024u;                        // weights of the (padded) layer
	__bf16 *wdst = reinterpret_cast<__bf16 *>(packed + a.offset[l]);
	float *bdst = packed + a.offset[l] + a.no[l] * a.ni[l] * 1536u;
	for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nw + a.no[l] * 32u; e += gridDim.x * 256) {
		if (e < nw) {
			const uint32_t el = e & 7u, lane = (e >> 3) & 63u, st = (e >> 9) & 1u, tile = e >> 10;
			const uint32_t it = tile % a.ni[l], ot = tile / a.ni[l];
			const uint32_t o = 32u * ot + (lane & 31u), f = 32u * it + 8u * (2u * st + (el >> 2)) + 4u * (lane >> 5) + (el & 3u);
			float v = 0.0f;
			if (o < a.out_dim[l] && f < a.in_dim[l]) v = a.w[l][(size_t)o * a.in_dim[l] + f];
			const __bf16 p1 = (__bf16)v;
			const float r1 = v - (float)p1;
			const __bf16 p2 = (__bf16)r1;
			const __bf16 p3 = (__bf16)(r1 - (float)p2);
			wdst[e] = p1; wdst[nw + e] = p2; wdst[2u * nw + e] = p3;
		} else {
			const uint32_t o = e - nw;
			bdst[o] = (a.b[l] && o < a.out_dim[l]) ? a.b[l][o] : 0.0f;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// one dense layer on the register map; wp -> LDS copy of the packed layer
// ---------------------------------------------------------------------------------------------
struct FwdArgs {
	uint64_t n;
	const float *x; int64_t xs;
	float *y; int64_t ys;
	const float *packed; uint32_t packed_floats;
	uint32_t n_layers, in_dim, out_dim;
	int hidden_act, out_act;
	uint32_t x_vec, y_vec;
};

// (X3, one input and one output tile, hidden layers up to 64 wide: two workgroups per CU = two waves per SIMD -- the piece splitting is VALU work, the products MFMA work, and
// only ANOTHER wave's instructions overlap them; at 260 registers the first version ran one wave per SIMD and the two added up)
template <int IN_T, int W_T, int OUT_T, int XF, bool X3 = false>
__global__ __launch_bounds__(kThreads, (X3 && IN_T == 1 && W_T <= 2 && OUT_T == 1) ? 2 : 1) void k_mlp_fwd(FwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	stage_weights(a.packed, a.packed_floats, lds);          // (X3: a.packed points at the x3 region, a.packed_floats is its size)
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint64_t n_tiles = (a.n + 31) / 32, step = (uint64_t)gridDim.x * 4;
	const uint32_t off_hidden = X3 ? layer_x3_floats(IN_T, W_T) : layer_floats(IN_T, W_T), sz_hidden = X3 ? layer_x3_floats(W_T, W_T) : layer_floats(W_T, W_T);
	auto clamp_row = [&](uint64_t row) { return row < a.n ? row : a.n - 1; };
	f16v xnext[IN_T];
	if (XF) prefetch_x<XF, IN_T>(a.x, a.xs, a.in_dim, clamp_row(((uint64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31)), lane, xnext);
	for (uint64_t tile = (uint64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += step) {
		const uint64_t row = tile * 32 + (lane & 31);
		const bool valid = row < a.n;
		f16v xin[IN_T], hcur[W_T], yo[OUT_T];
		if (XF) {
			// software pipeline: this tile's rows were requested one iteration ago, the next tile's go out now
#pragma unroll
			for (int t = 0; t < IN_T; ++t) xin[t] = xnext[t];
			prefetch_x<XF, IN_T>(a.x, a.xs, a.in_dim, clamp_row((tile + step) * 32 + (lane & 31)), lane, xnext);
		} else {
			load_rows<IN_T>(a.x, a.xs, a.in_dim, row, valid, a.x_vec != 0, lane, xin);
		}
		// wide networks (round 4): the LDS base goes through a register the compiler cannot see through, so that it does not hoist
		// every layer's weight fragments out of the tile loop (hundreds of loop-invariant registers -> 344 scratch instructions
		// in k_mlp_fwd<4, 4, 4, 2>, round-3 review)
		uint32_t opaque = 0;
		if constexpr (IN_T >= 4 || W_T >= 4 || OUT_T >= 4) asm volatile("s_mov_b32 %0, 0" : "=s"(opaque));
		const float *wl = lds + opaque;
		if constexpr (X3) dense_x3<IN_T, W_T, true>(wl, xin, hcur, a.hidden_act, lane);
		else dense<IN_T, W_T, true>(wl, xin, hcur, a.hidden_act, lane);
#pragma unroll 1
		for (uint32_t l = 1; l + 1 < a.n_layers; ++l) {
			f16v hn[W_T];
			if constexpr (X3) dense_x3<W_T, W_T, true>(wl + off_hidden + (l - 1) * sz_hidden, hcur, hn, a.hidden_act, lane);
			else dense<W_T, W_T, true>(wl + off_hidden + (l - 1) * sz_hidden, hcur, hn, a.hidden_act, lane);
#pragma unroll
			for (int t = 0; t < W_T; ++t) hcur[t] = hn[t];
		}
		if constexpr (X3) dense_x3<W_T, OUT_T, true>(wl + off_hidden + (a.n_layers - 2) * sz_hidden, hcur, yo, a.out_act, lane);
		else dense<W_T, OUT_T, true>(wl + off_hidden + (a.n_layers - 2) * sz_hidden, hcur, yo, a.out_act, lane);
		store_rows<OUT_T>(a.y, a.ys, a.out_dim, row, valid, a.y_vec != 0, lane, yo);
	}
}

// =============================================================================================
// backward: dL/dx (optional), dL/dW_l, dL/db_l from x and dL/dy, forward recomputed in registers
// =============================================================================================
constexpr int kTS = 36;                        // row stride (floats) of the per-wave [feature][sample] LDS tiles:
                                               // 16-byte aligned rows, and 8 consecutive rows cover all 32 banks

// register map -> [feature][sample] tile (rows of kTS floats)
template <int MAXT>
__device__ __forceinline__ void write_tile(float *__restrict__ T, int nt, const f16v (&r)[MAXT], int lane) {
	const int s = lane & 31, h = lane >> 5;
#pragma unroll
	for (int t = 0; t < MAXT; ++t) {
		if (t >= nt) continue;
#pragma unroll
		for (int j = 0; j < 16; ++j) T[(32 * t + 8 * (j >> 2) + 4 * h + (j & 3)) * kTS + s] = r[t][j];
	}
}

// 16 samples (half-wave h: samples 16h .. 16h+15) of row `row` -> MFMA operand values of the sample contraction
__device__ __forceinline__ void read_row16(const float *__restrict__ T, int row, int h, float (&v)[16]) {
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const f4v t4 = *reinterpret_cast<const f4v *>(T + row * kTS + 16 * h + 4 * q);
#pragma unroll
		for (int b = 0; b < 4; ++b) v[4 * q + b] = t4[b];
	}
}

struct BwdArgs {
	uint64_t n;
	const float *x; int64_t xs;
	const float *gy; int64_t gys;
	float *gx; int64_t gxs;                    // NULL: dL/dx not wanted
	uint32_t x_fm, gx_fm;                      // x is read / dL/dx is stored feature-major (xs / gxs = feature stride)
	const float *packed;                       // the forward layers: f32, or their x3 planes
	uint32_t total_floats;                     // of their padded LDS copy
	float *dW[NR3D_MLP_MAX_LAYERS];            // accumulated into (atomics): zero them for plain gradients
	float *db[NR3D_MLP_MAX_LAYERS];            // may be NULL
	uint32_t dims[NR3D_MLP_MAX_LAYERS + 1];
	uint32_t n_layers;
	int hidden_act, out_act;
	uint32_t x_vec, gy_vec, gx_vec;
	uint32_t tile_floats;                      // per wave
};

// One layer of the backward sweep.  g = dL/d(pre-activation of this layer's output) on the register map (NO tiles);
// TG: LDS tile that receives g as [feature][sample]; TB: the layer's INPUT activations as [feature][sample] (NI tiles);
// wT: the padded LDS copy of THIS layer (read transposed).  Accumulates dW (NO x NI tiles) and the per-lane bias partial sums; when PREV, leaves
// dL/d(input of the layer) in gp, masked with the ReLU derivative of the input activations when MASK.
// eight consecutive samples of a [feature][sample] tile row as the three bf16 pieces of an MFMA operand (mlp_device.h split3)
__device__ __forceinline__ void split3_row8(const float *__restrict__ src, bf8 (&p)[3], float &sum) {
	const f4v lo = *reinterpret_cast<const f4v *>(src), hi = *reinterpret_cast<const f4v *>(src + 4);
	const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
	for (int e = 0; e < 8; e += 2) {
		split3_pair<false>(v[e], v[e + 1], e, p);
		sum += v[e] + v[e + 1];
	}
}

template <int NO, int NI, bool PREV, bool MASK, bool X3 = false>
__device__ __forceinline__ void bwd_layer(const f16v (&g)[NO], float *__restrict__ TG, const float *__restrict__ TB,
                                          const float *__restrict__ wT, f16v (&dW)[NO][NI], float (&db)[NO], f16v (&gp)[NI],
                                          int lane) {
	const int r = lane & 31, h = lane >> 5;
	write_tile<NO>(TG, NO, g, lane);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	if constexpr (X3) {
		// dW += dPre^T . H on the bf16 MFMA (round 6): the contraction runs over the tile's 32 samples = two K = 16 steps; lane (r, h) holds
		// samples 16 s + 8 h .. + 7 of row r of both operands, each split into three bf16 pieces, six piece products per step -- the
		// same fp32-grade sum as dense_x3 (smallest terms first), 12 MFMAs of 8 passes where the f32 MFMA takes 16 of 16 passes
		constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
		for (int st = 0; st < 2; ++st) {
			bf8 ap[NO][3], bp[NI][3];
			float dummy = 0.0f;
#pragma unroll
			for (int ot = 0; ot < NO; ++ot) split3_row8(TG + (32 * ot + r) * kTS + 16 * st + 8 * h, ap[ot], db[ot]);
#pragma unroll
			for (int it = 0; it < NI; ++it) split3_row8(TB + (32 * it + r) * kTS + 16 * st + 8 * h, bp[it], dummy);
#pragma unroll
			for (int t = 0; t < 6; ++t)
#pragma unroll
				for (int ot = 0; ot < NO; ++ot)
#pragma unroll
					for (int it = 0; it < NI; ++it)
						dW[ot][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[ot][PW[t]], bp[it][PX[t]], dW[ot][it], 0, 0, 0);
		}
	} else {
		float bv[NI][16];
#pragma unroll
		for (int it = 0; it < NI; ++it) read_row16(TB, 32 * it + r, h, bv[it]);
#pragma unroll
		for (int ot = 0; ot < NO; ++ot) {
			float av[16];
			read_row16(TG, 32 * ot + r, h, av);
			float sum = 0.0f;
#pragma unroll
			for (int t = 0; t < 16; ++t) sum += av[t];
			db[ot] += sum;
#pragma unroll
			for (int it = 0; it < NI; ++it)
#pragma unroll
				for (int t = 0; t < 16; ++t) dW[ot][it] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[it][t], dW[ot][it], 0, 0, 0);
		}
	}
	if (PREV) {
		if constexpr (X3) dense_x3_t<NO, NI>(wT, g, gp, lane);          // wT: the padded copy of the FORWARD layer's x3 planes
		else dense_t<NO, NI>(wT, g, gp, lane);                          //     ... of the forward layer
		if (MASK) {
#pragma unroll
			for (int t = 0; t < NI; ++t)
#pragma unroll
				for (int j = 0; j < 16; ++j)
					gp[t][j] = TB[(32 * t + 8 * (j >> 2) + 4 * h + (j & 3)) * kTS + r] > 0.0f ? gp[t][j] : 0.0f;
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
	for (uint32_t e = threadIdx.x; e < (uint32_t)(NO * NI * 1024); e += blockDim.x) {
		const uint32_t ln = e & 63u, j = (e >> 6) & 15u, it = (e >> 10) % NI, ot = (e >> 10) / NI;
		const uint32_t k = 32u * it + (ln & 31u), o = 32u * ot + 8u * (j >> 2) + 4u * (ln >> 5) + (j & 3u);
		if (o < out_dim && k < in_dim) atomic_add_f32(gW + (size_t)o * in_dim + k, R[e]);
	}
	if (gb)
		for (uint32_t e = threadIdx.x; e < (uint32_t)(NO * 32); e += blockDim.x) {
			const uint32_t o = e;                                       // 32 ot + row
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

// FAST: x and dL/dy rows are 16-byte aligned with widths that are multiples of 4 -> branch-free loads, the next tile's
// rows requested while this tile is processed (one wave per SIMD: nothing else hides the memory latency)
// FAST: 0 = any layout (row-major or, with a.x_fm, feature-major x), 1 = prefetched row-major x and dL/dy, 2 = prefetched
// feature-major x + row-major dL/dy
// Networks of 32-wide layers with <= 2 hidden layers leave room for EIGHT waves per workgroup, two per SIMD (round 4, as csrc/mlp_half.hip):
// the kernel is a chain of LDS round trips and dependent MFMAs per tile, a second wave per SIMD hides part of it.
template <int IN_T, int W_T, int OUT_T, int NH> struct BwdCfg { static constexpr int kMaxWaves = (IN_T == 1 && W_T == 1 && OUT_T == 1 && NH <= 2) ? 8 : 4; };
constexpr int kMaxLdsBwd = 160 * 1024;
// X3 (round 6): the forward recomputation, the dH = W^T dPre chain and the sample contraction dW = dPre^T H all run on the bf16 MFMA with
// three-piece splits (dense_x3 / dense_x3_t / bwd_layer<..., true>); a.packed then points at the x3 planes of the forward layers.
// The ReLU masks come from the SAME forward arithmetic as nr3d_mlp_forward's x3 route.
template <int IN_T, int W_T, int OUT_T, int NH, int FAST, bool X3 = false>
__global__ __launch_bounds__((BwdCfg<IN_T, W_T, OUT_T, NH>::kMaxWaves * 64)) void k_mlp_bwd(BwdArgs a) {
	extern __shared__ __attribute__((aligned(16))) float lds[];
	// FAST: the next tile's rows are requested before a tile's LAST step, not at its top: their 32 - 64 registers then overlap one
	// layer's work instead of five (k_mlp_bwd<2,2,2,2>: 183 -> 139 / 220 -> 187 spilled dwords, 64 -> 64 -> 64 -> 64 4.38 -> 3.86 ms), and
	// one step still covers the latency (every measured shape equal or faster, 32 -> 32 -> 32 -> 16 0.66 -> 0.62 ms)
	{
		// ONE padded copy of the forward layers (f32, or their x3 planes) serves the forward recomputation (16-byte reads) and the
		// dH = W^T dPre chain (dense_t's 4-byte reads / dense_x3_t's transposing reads) -- no second, transposed copy: its LDS goes to
		// the waves' tiles.  On the f32 MFMA 64 -> 64 -> 64 -> 64 runs four waves per CU where two fitted; the x3 planes of
		// 32 -> 64 -> 64 -> 16 leave room for four waves where both orientations left two
		constexpr int GPP = X3 ? 6 : 4;
		constexpr uint32_t s0 = IN_T * W_T * GPP * 256 + W_T * 32, sh = W_T * W_T * GPP * 256 + W_T * 32;      // packed layers
		constexpr uint32_t GS = X3 ? kGS3 : kGS;
		constexpr uint32_t d0 = IN_T * W_T * GPP * GS + W_T * 32, dh = W_T * W_T * GPP * GS + W_T * 32;        // padded copies
		stage_layer_padded<IN_T, W_T, GPP>(a.packed, lds);
#pragma unroll
		for (int l = 1; l < NH; ++l) stage_layer_padded<W_T, W_T, GPP>(a.packed + s0 + (l - 1) * sh, lds + d0 + (l - 1) * dh);
		stage_layer_padded<W_T, OUT_T, GPP>(a.packed + s0 + (NH - 1) * sh, lds + d0 + (NH - 1) * dh);
		__syncthreads();
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const int r = lane & 31;
	float *tiles = lds + a.total_floats + (size_t)wave * a.tile_floats;
	// tile rows: X or G_out | H_1 .. H_NH.  G_out (dL/dy as [feature][sample]) lives from the output layer's step of the sweep to its
	// end, X from the first layer's step on -- the forward reads x from registers --, so they share their rows (round 6: the tile area
	// of 64 -> 64 -> 64 -> 64 drops from 36.9 to 27.6 KB per wave and TWO waves fit next to 98 KB of weights where one did;
	// 32 -> 64 -> 64 -> 16: four instead of three)
	constexpr int XG_T = IN_T > OUT_T ? IN_T : OUT_T;
	float *TX = tiles;
	float *TGO = tiles;
	float *TH1 = tiles + 32 * XG_T * kTS;                               // H_l at TH1 + (l - 1) * 32 * W_T * kTS
	// the padded layers: [0 | hidden ... | out]; the dH chain reads the same copies (wt + t0 + l th = layer l + 1)
	constexpr uint32_t f0 = X3 ? layer_x3_floats_pad(IN_T, W_T) : layer_floats_pad(IN_T, W_T);
	constexpr uint32_t fh = X3 ? layer_x3_floats_pad(W_T, W_T) : layer_floats_pad(W_T, W_T);
	constexpr uint32_t t0 = f0, th = fh;
	const float *wf = lds, *wt = lds;

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
	f16v xnext[IN_T], gnext[OUT_T];
	if (FAST) {
		const uint64_t r0 = clamp_row(((uint64_t)blockIdx.x * nw + wave) * 32 + r);
		prefetch_x<FAST, IN_T>(a.x, a.xs, a.dims[0], r0, lane, xnext);
		load_rows_fast<OUT_T>(a.gy, a.gys, a.dims[NH + 1], r0, lane, gnext);
	}
	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += step) {
		const uint64_t row = tile * 32 + r;
		const bool valid = row < a.n;
		f16v xin[IN_T], g_out[OUT_T], hcur[W_T];
		if (FAST) {
#pragma unroll
			for (int t = 0; t < IN_T; ++t) xin[t] = xnext[t];
#pragma unroll
			for (int t = 0; t < OUT_T; ++t) g_out[t] = gnext[t];
		} else {
			if (a.x_fm) load_cols_fast<IN_T>(a.x, a.xs, a.dims[0], clamp_row(row), lane, xin);   // rows past n: dL/dy is zero there
			else load_rows<IN_T>(a.x, a.xs, a.dims[0], row, valid, a.x_vec != 0, lane, xin);
			load_rows<OUT_T>(a.gy, a.gys, a.dims[NH + 1], row, valid, a.gy_vec != 0, lane, g_out);
		}
		// ---- forward, activations kept as [feature][sample] tiles ----
		if constexpr (X3) dense_x3<IN_T, W_T, true, true>(wf, xin, hcur, a.hidden_act, lane);
		else dense<IN_T, W_T, true, true>(wf, xin, hcur, a.hidden_act, lane);
		write_tile<W_T>(TH1, W_T, hcur, lane);
#pragma unroll
		for (int l = 1; l < NH; ++l) {
			f16v hn[W_T];
			if constexpr (X3) dense_x3<W_T, W_T, true, true>(wf + f0 + (l - 1) * fh, hcur, hn, a.hidden_act, lane);
			else dense<W_T, W_T, true, true>(wf + f0 + (l - 1) * fh, hcur, hn, a.hidden_act, lane);
#pragma unroll
			for (int t = 0; t < W_T; ++t) hcur[t] = hn[t];
			write_tile<W_T>(TH1 + l * 32 * W_T * kTS, W_T, hcur, lane);
		}
		if (FAST && !valid) zero_tiles<OUT_T>(g_out);                    // rows past n were clamped, not zeroed
		if (a.out_act == NR3D_MLP_ACT_RELU) {
			f16v yo[OUT_T];
			if constexpr (X3) dense_x3<W_T, OUT_T, true, true>(wf + f0 + (NH - 1) * fh, hcur, yo, NR3D_MLP_ACT_NONE, lane);
			else dense<W_T, OUT_T, true, true>(wf + f0 + (NH - 1) * fh, hcur, yo, NR3D_MLP_ACT_NONE, lane);
#pragma unroll
			for (int t = 0; t < OUT_T; ++t)
#pragma unroll
				for (int j = 0; j < 16; ++j) g_out[t][j] = yo[t][j] > 0.0f ? g_out[t][j] : 0.0f;
		}
		// ---- backward sweep ----
		const bool relu = a.hidden_act == NR3D_MLP_ACT_RELU;
		f16v g[W_T];
		if (relu) bwd_layer<OUT_T, W_T, true, true, X3>(g_out, TGO, TH1 + (NH - 1) * 32 * W_T * kTS, wt + t0 + (NH - 1) * th, dWo, dbo, g, lane);
		else bwd_layer<OUT_T, W_T, true, false, X3>(g_out, TGO, TH1 + (NH - 1) * 32 * W_T * kTS, wt + t0 + (NH - 1) * th, dWo, dbo, g, lane);
		write_tile<IN_T>(TX, IN_T, xin, lane);                           // into G_out's rows: their reader, the step above, is done; x leaves the registers here
#pragma unroll
		for (int l = NH - 1; l >= 1; --l) {                              // hidden layer l: H_l -> H_{l+1}
			f16v gp[W_T];
			float *TG = TH1 + l * 32 * W_T * kTS;                       // H_{l+1} is dead once its mask has been applied
			const float *TB = TH1 + (l - 1) * 32 * W_T * kTS;
			if (relu) bwd_layer<W_T, W_T, true, true, X3>(g, TG, TB, wt + t0 + (l - 1) * th, dWh[l - 1], dbh[l - 1], gp, lane);
			else bwd_layer<W_T, W_T, true, false, X3>(g, TG, TB, wt + t0 + (l - 1) * th, dWh[l - 1], dbh[l - 1], gp, lane);
#pragma unroll
			for (int t = 0; t < W_T; ++t) g[t] = gp[t];
		}
		if constexpr (FAST != 0) {                                       // the next tile's rows (see above)
			const uint64_t rn = clamp_row((tile + step) * 32 + r);
			prefetch_x<FAST, IN_T>(a.x, a.xs, a.dims[0], rn, lane, xnext);
			load_rows_fast<OUT_T>(a.gy, a.gys, a.dims[NH + 1], rn, lane, gnext);
		}
		f16v gx[IN_T];
		if (a.gx) {
			bwd_layer<W_T, IN_T, true, false, X3>(g, TH1, TX, wt, dW0, db0, gx, lane);
			if (a.gx_fm) store_cols<IN_T>(a.gx, a.gxs, a.dims[0], row, valid, lane, gx);
			else store_rows<IN_T>(a.gx, a.gxs, a.dims[0], row, valid, a.gx_vec != 0, lane, gx);
		} else {
			bwd_layer<W_T, IN_T, false, false, X3>(g, TH1, TX, wt, dW0, db0, gx, lane);
		}
	}

	// ---- reduce the waves' parameter gradients in LDS (the tile area is free now), one atomic per element ----
	__syncthreads();
	float *R = lds + a.total_floats;
	reduce_layer<W_T, IN_T>(dW0, db0, R, a.dW[0], a.db[0], a.dims[1], a.dims[0], lane, wave, nw);
#pragma unroll
	for (int l = 1; l < NH; ++l) reduce_layer<W_T, W_T>(dWh[l - 1], dbh[l - 1], R, a.dW[l], a.db[l], a.dims[l + 1], a.dims[l], lane, wave, nw);
	reduce_layer<OUT_T, W_T>(dWo, dbo, R, a.dW[NH], a.db[NH], a.dims[NH + 1], a.dims[NH], lane, wave, nw);
}

}  // namespace mlp
}  // namespace nr3d

using namespace nr3d;
using namespace nr3d::mlp;

// the forward part of the packed buffer: [f32 layers | x3 planes of the same layers (0 floats when they do not fit LDS)]
static uint64_t forward_floats(const Shape &s) { return packed_floats(s) + x3_floats(s); }

namespace nr3d {
namespace mlp {
bool forward_region(const nr3d_mlp_desc_t *desc, const float *packed, bool &x3, const float *&region, uint32_t &region_floats,
                    uint32_t &in_t, uint32_t &w_t, uint32_t &out_t) {
	Shape s;
	if (!shape_of(desc, s) || packed_floats(s) * 4 > (uint64_t)kMaxLds) return false;
	x3 = x3_enabled() && x3_floats(s) != 0;
	region = x3 ? packed + packed_floats(s) : packed;
	region_floats = (uint32_t)(x3 ? x3_floats(s) : packed_floats(s));
	in_t = s.in_t; w_t = s.w_t; out_t = s.out_t;
	return true;
}
}  // namespace mlp
}  // namespace nr3d

extern "C" uint64_t nr3d_mlp_packed_floats(const nr3d_mlp_desc_t *desc) {
	Shape s;
	if (!shape_of(desc, s)) return 0;
	const uint64_t n = packed_floats(s);
	return n * 4 <= (uint64_t)kMaxLds ? forward_floats(s) : 0;      // 0: the fused kernels do not apply to this network
}

static int fill_pack(const nr3d_mlp_desc_t *d, const Shape &s, const float *const *weights, const float *const *biases, PackArgs &p) {
	p.n_layers = d->n_layers;
	uint32_t off = 0;
	for (uint32_t l = 0; l < d->n_layers; ++l) {
		NR3D_CHECK(weights[l] != nullptr, "mlp_pack: weights[%u] is NULL", l);
		p.w[l] = weights[l];
		p.b[l] = biases ? biases[l] : nullptr;
		p.in_dim[l] = d->dims[l]; p.out_dim[l] = d->dims[l + 1];
		p.ni[l] = l == 0 ? s.in_t : s.w_t;
		p.no[l] = l + 1 == d->n_layers ? s.out_t : s.w_t;
		p.offset[l] = off;
		off += layer_floats(p.ni[l], p.no[l]);
	}
	p.offset[d->n_layers] = off;
	return 0;
}

// the fused backward keeps every layer's dW in accumulator registers: hidden width <= 64, at most 2 hidden layers of
// width > 32 (3 of width <= 32), input / output no wider (in tiles) than the hidden layers
static bool backward_ok(const Shape &s) {
	if (s.w_t > 2 || s.in_t > s.w_t || s.out_t > s.w_t) return false;
	const uint32_t nh = s.n_layers - 1;
	return s.w_t == 1 ? nh <= 3 : nh <= 2;
}

// the f32 backward's LDS copy of the forward layers (mlp_device.h kGS / kHS)
static uint64_t padded_floats(const Shape &s) {
	return (uint64_t)layer_floats_pad(s.in_t, s.w_t) + (uint64_t)(s.n_layers - 2) * layer_floats_pad(s.w_t, s.w_t) + layer_floats_pad(s.w_t, s.out_t);
}

static uint64_t padded_x3_floats(const Shape &s) {
	return (uint64_t)layer_x3_floats_pad(s.in_t, s.w_t) + (uint64_t)(s.n_layers - 2) * layer_x3_floats_pad(s.w_t, s.w_t) + layer_x3_floats_pad(s.w_t, s.out_t);
}

// per-wave [feature][sample] tiles: X, H_1 .. H_NH, G_out
static uint32_t bwd_tile_floats(const Shape &s) { return (32u * (s.in_t > s.out_t ? s.in_t : s.out_t) + (s.n_layers - 1) * 32u * s.w_t) * (uint32_t)kTS; }

// weights the backward keeps in LDS: one padded copy of the forward layers, f32 or (x3) their bf16 planes
static uint64_t bwd_weight_floats(const Shape &s, bool x3) { return x3 ? padded_x3_floats(s) : padded_floats(s); }

static uint32_t bwd_waves(const Shape &s, bool x3 = false) {
	if (x3 && x3_floats(s) == 0) return 0;
	const uint64_t wbytes = bwd_weight_floats(s, x3) * 4;
	const uint64_t reduce = ((uint64_t)s.w_t * s.w_t * 1024 + (uint64_t)s.w_t * 64) * 4;     // one layer at a time
	const uint32_t max_waves = (s.in_t == 1 && s.w_t == 1 && s.out_t == 1 && s.n_layers - 1 <= 2) ? 8u : 4u;      // = BwdCfg<...>::kMaxWaves
	for (uint32_t nw = max_waves; nw >= 1; --nw) {
		const uint64_t t = (uint64_t)nw * bwd_tile_floats(s) * 4;
		// the whole 160 KB (the kernel has no static LDS)
		if (wbytes + (t > reduce ? t : reduce) <= (uint64_t)kMaxLdsBwd) return nw;
	}
	return 0;
}

// Does the x3 option put the backward on the bf16 MFMA?  The kernel is not bound by its MFMAs -- counters of 32 -> 64 -> 64 -> 16
// (profiles/r06_mlp_counters.txt): VALU active 45 % of a wave's cycles (the piece splitting), MFMA pipe 31 %, one wave per SIMD
// overlaps little of it -- so a wave lost to the bigger planes costs more than the cheaper products bring: x3 where its planes
// (1.5 x the f32 bytes + padding) leave as many waves as the f32 copy.  Same-box A/B at 2^22 samples, fwd+bwd ms, x3 / f32:
// 32->64->64->16 1.90 / 2.33, 32->64->16 1.00 / 1.18, 32->32->16 0.60 / 0.63, 18->32->3 0.57 / 0.75; 64->64->64->64 x3 has three waves
// against four: 5.49 / 3.96 -> f32 (so do 64->64->64 and 32->64->64->64).  Until the transposing read (dense_x3_t) the small shapes
// kept a second, transposed set of planes in LDS (every weight read 16 bytes): equal within 2 % now, and gone.
static bool backward_x3(const Shape &s) {
	const uint32_t w0 = bwd_waves(s, false), w3 = bwd_waves(s, true);
	// no x3 kernel is built (BWD_CASE) for the 64-wide shapes it does not win on: two hidden layers with a 64-wide input or output
	// (w3 < w0), and 64 -> 64 -> 64 (same waves, backward equal within 3 %: the splits of its 64-wide tiles eat what the cheaper
	// products bring, and the f32 kernel spills less)
	if (s.w_t == 2 && ((s.n_layers == 3 && s.in_t + s.out_t >= 3) || s.in_t + s.out_t >= 4)) return false;
	return w3 != 0 && w3 >= w0;
}

// behind the forward part: kBwdHeader floats (non-zero size = "the fused backward applies"; the backward reads the forward layers)
constexpr uint32_t kBwdHeader = 4;
extern "C" uint64_t nr3d_mlp_backward_packed_floats(const nr3d_mlp_desc_t *desc) {
	Shape s;
	if (!shape_of(desc, s) || nr3d_mlp_packed_floats(desc) == 0 || !backward_ok(s) || bwd_waves(s) == 0) return 0;
	return kBwdHeader;
}

extern "C" int nr3d_mlp_pack(const nr3d_mlp_desc_t *desc, const float *const *weights, const float *const *biases, float *packed,
                             int with_backward, void *stream) {
	Shape s;
	NR3D_CHECK(shape_of(desc, s) && nr3d_mlp_packed_floats(desc) != 0, "mlp_pack: network outside the fused kernels' range "
	           "(2..%d linear layers, every width 1..128, packed weights <= %d KB)", NR3D_MLP_MAX_LAYERS, kMaxLds / 1024);
	NR3D_CHECK(weights && packed, "mlp_pack: NULL pointer");
	NR3D_CHECK(!with_backward || nr3d_mlp_backward_packed_floats(desc) != 0, "mlp_pack: the fused backward does not apply to this network");
	PackArgs p;
	if (int rc = fill_pack(desc, s, weights, biases, p)) return rc;
	hipLaunchKernelGGL(k_mlp_pack, dim3(16, desc->n_layers), dim3(256), 0, (hipStream_t)stream, p, packed);
	if (x3_floats(s)) {
		PackArgs x = p;
		uint32_t off = 0;
		for (uint32_t l = 0; l < desc->n_layers; ++l) { x.offset[l] = off; off += layer_x3_floats(p.ni[l], p.no[l]); }
		x.offset[desc->n_layers] = off;
		hipLaunchKernelGGL(k_mlp_pack_x3, dim3(16, desc->n_layers), dim3(256), 0, (hipStream_t)stream, x, packed + packed_floats(s));
	}
	NR3D_LAUNCH_CHECK();
	return 0;
}

#define MLP_DISPATCH(S, ...)                                                                                   \
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

extern "C" int nr3d_mlp_forward(const nr3d_mlp_desc_t *desc, uint64_t n, const float *x, int64_t x_stride, int64_t x_feature_stride,
                                const float *packed, float *y, int64_t y_stride, void *stream) {
	Shape s;
	NR3D_CHECK(shape_of(desc, s) && nr3d_mlp_packed_floats(desc) != 0, "mlp_forward: network outside the fused kernels' range");
	if (n == 0) return 0;
	NR3D_CHECK(x && packed && y, "mlp_forward: NULL pointer");
	FwdArgs a;
	const bool x_fm = x_feature_stride != 1;
	NR3D_CHECK(!x_fm || x_stride == 1, "mlp_forward: x must be row-major (feature stride 1) or feature-major (row stride 1)");
	a.n = n; a.x = x; a.xs = x_fm ? x_feature_stride : x_stride; a.y = y; a.ys = y_stride; a.packed = packed;
	a.packed_floats = (uint32_t)packed_floats(s);
	const bool x3 = x3_enabled() && x3_floats(s) != 0;
	if (x3) { a.packed = packed + packed_floats(s); a.packed_floats = (uint32_t)x3_floats(s); }
	a.n_layers = desc->n_layers; a.in_dim = desc->dims[0]; a.out_dim = desc->dims[desc->n_layers];
	a.hidden_act = (int)desc->hidden_activation; a.out_act = (int)desc->output_activation;
	a.x_vec = ((uintptr_t)x % 16 == 0 && x_stride % 4 == 0) ? 1u : 0u;
	a.y_vec = ((uintptr_t)y % 16 == 0 && y_stride % 4 == 0) ? 1u : 0u;
	const size_t lds = (size_t)a.packed_floats * 4;
	const uint64_t n_tiles = (n + 31) / 32;
	const uint32_t grid = (uint32_t)(n_tiles / 4 + 1 < 1024 ? n_tiles / 4 + 1 : 1024);
	int rc = 0;
	MLP_DISPATCH(s, {
		static bool attr[64] = {};
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess) { rc = ::nr3d::fail("mlp_forward: hipGetDevice failed"); return; }
		if (!attr[dev & 63]) {
			if (hipFuncSetAttribute((const void *)k_mlp_fwd<IN_T, W_T, OUT_T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess ||
			    hipFuncSetAttribute((const void *)k_mlp_fwd<IN_T, W_T, OUT_T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess ||
			    hipFuncSetAttribute((const void *)k_mlp_fwd<IN_T, W_T, OUT_T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
				rc = ::nr3d::fail("mlp_forward: cannot raise the dynamic LDS limit"); return;
			}
			attr[dev & 63] = true;
		}
		if (x3) {
			static bool attr3[64] = {};
			if (!attr3[dev & 63]) {
				if (hipFuncSetAttribute((const void *)k_mlp_fwd<IN_T, W_T, OUT_T, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess ||
				    hipFuncSetAttribute((const void *)k_mlp_fwd<IN_T, W_T, OUT_T, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess ||
				    hipFuncSetAttribute((const void *)k_mlp_fwd<IN_T, W_T, OUT_T, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds) != hipSuccess) {
					rc = ::nr3d::fail("mlp_forward: cannot raise the dynamic LDS limit"); return;
				}
				attr3[dev & 63] = true;
			}
			if (x_fm)
				hipLaunchKernelGGL((k_mlp_fwd<IN_T, W_T, OUT_T, 2, true>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
			else if (a.x_vec && a.in_dim % 4 == 0)
				hipLaunchKernelGGL((k_mlp_fwd<IN_T, W_T, OUT_T, 1, true>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
			else
				hipLaunchKernelGGL((k_mlp_fwd<IN_T, W_T, OUT_T, 0, true>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
		} else if (x_fm)
			hipLaunchKernelGGL((k_mlp_fwd<IN_T, W_T, OUT_T, 2>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
		else if (a.x_vec && a.in_dim % 4 == 0)
			hipLaunchKernelGGL((k_mlp_fwd<IN_T, W_T, OUT_T, 1>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
		else
			hipLaunchKernelGGL((k_mlp_fwd<IN_T, W_T, OUT_T, 0>), dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, a);
	});
	if (rc) return rc;
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_mlp_backward(const nr3d_mlp_desc_t *desc, uint64_t n, const float *x, int64_t x_stride, int64_t x_feature_stride,
                                 const float *dL_dy, int64_t gy_stride, const float *packed, float *dL_dx, int64_t gx_stride,
                                 int64_t gx_feature_stride, float *const *dL_dW, float *const *dL_db, void *stream) {
	Shape s;
	NR3D_CHECK(shape_of(desc, s) && nr3d_mlp_backward_packed_floats(desc) != 0, "mlp_backward: the fused backward does not apply to this network");
	if (n == 0) return 0;
	NR3D_CHECK(x && dL_dy && packed && dL_dW, "mlp_backward: NULL pointer");
	BwdArgs a;
	const bool x_fm = x_feature_stride != 1, gx_fm = dL_dx && gx_feature_stride != 1;
	NR3D_CHECK(!x_fm || x_stride == 1, "mlp_backward: x must be row-major (feature stride 1) or feature-major (row stride 1)");
	NR3D_CHECK(!gx_fm || gx_stride == 1, "mlp_backward: dL_dx must be row-major (feature stride 1) or feature-major (row stride 1)");
	a.n = n; a.x = x; a.xs = x_fm ? x_feature_stride : x_stride; a.gy = dL_dy; a.gys = gy_stride; a.packed = packed;
	a.gx = dL_dx; a.gxs = gx_fm ? gx_feature_stride : gx_stride;
	a.x_fm = x_fm ? 1u : 0u; a.gx_fm = gx_fm ? 1u : 0u;
	// round 6: on the bf16 MFMA with three-piece splits (forward recomputation, dH chain, dW) when the option is on and the planes fit
	const bool x3 = x3_enabled() && backward_x3(s);
	a.total_floats = (uint32_t)bwd_weight_floats(s, x3);                 // of the LDS copy (the kernel pads the packed layers itself)
	if (x3) a.packed = packed + packed_floats(s);                        // the x3 planes of the forward layers
	for (uint32_t l = 0; l < desc->n_layers; ++l) {
		NR3D_CHECK(dL_dW[l] != nullptr, "mlp_backward: dL_dW[%u] is NULL", l);
		a.dW[l] = dL_dW[l];
		a.db[l] = dL_db ? dL_db[l] : nullptr;
	}
	for (uint32_t l = 0; l <= desc->n_layers; ++l) a.dims[l] = desc->dims[l];
	a.n_layers = desc->n_layers;
	a.hidden_act = (int)desc->hidden_activation; a.out_act = (int)desc->output_activation;
	a.x_vec = ((uintptr_t)x % 16 == 0 && x_stride % 4 == 0) ? 1u : 0u;
	a.gy_vec = ((uintptr_t)dL_dy % 16 == 0 && gy_stride % 4 == 0) ? 1u : 0u;
	a.gx_vec = (dL_dx && (uintptr_t)dL_dx % 16 == 0 && gx_stride % 4 == 0) ? 1u : 0u;
	a.tile_floats = bwd_tile_floats(s);
	const uint32_t nw = bwd_waves(s, x3);
	const uint64_t reduce = ((uint64_t)s.w_t * s.w_t * 1024 + (uint64_t)s.w_t * 64) * 4;
	const uint64_t tbytes = (uint64_t)nw * a.tile_floats * 4;
	const size_t lds = (size_t)a.total_floats * 4 + (size_t)(tbytes > reduce ? tbytes : reduce);
	const uint64_t n_tiles = (n + 31) / 32;
	const uint32_t grid = (uint32_t)(n_tiles / nw + 1 < 256 ? n_tiles / nw + 1 : 256);     // one workgroup per CU: dW lives in registers
	const uint32_t nh = desc->n_layers - 1;
	const bool gy_fast = a.gy_vec && desc->dims[desc->n_layers] % 4 == 0;
	const int fast = !gy_fast ? 0 : x_fm ? 2 : (a.x_vec && desc->dims[0] % 4 == 0) ? 1 : 0;
	auto launch = [&](auto kern) -> int {
		NR3D_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBwd));
		hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, (hipStream_t)stream, a);
		return 0;
	};
	int rc = 0;
#define BWD_CASE(I, W, O, H) if (s.in_t == I && s.w_t == W && s.out_t == O && nh == H) { \
		if (x3) { if constexpr (!(W == 2 && ((H == 2 && I + O >= 3) || I + O >= 4))) /* (= backward_x3()) */ \
		              rc = fast == 2 ? launch(k_mlp_bwd<I, W, O, H, 2, true>) : fast == 1 ? launch(k_mlp_bwd<I, W, O, H, 1, true>) : launch(k_mlp_bwd<I, W, O, H, 0, true>); \
		          else rc = ::nr3d::fail("mlp_backward: no bf16 MFMA backward for this shape"); } \
		else rc = fast == 2 ? launch(k_mlp_bwd<I, W, O, H, 2>) : fast == 1 ? launch(k_mlp_bwd<I, W, O, H, 1>) : launch(k_mlp_bwd<I, W, O, H, 0>); } else
	BWD_CASE(1, 1, 1, 1) BWD_CASE(1, 1, 1, 2) BWD_CASE(1, 1, 1, 3)
	BWD_CASE(1, 2, 1, 1) BWD_CASE(1, 2, 1, 2) BWD_CASE(1, 2, 2, 1) BWD_CASE(1, 2, 2, 2)
	BWD_CASE(2, 2, 1, 1) BWD_CASE(2, 2, 1, 2) BWD_CASE(2, 2, 2, 1) BWD_CASE(2, 2, 2, 2)
	rc = ::nr3d::fail("mlp_backward: no kernel for this shape");
#undef BWD_CASE
	if (rc) return rc;
	NR3D_LAUNCH_CHECK();
	return 0;
}
