// nr3d_lib_amd/csrc/mlp_device.h -- device side of the fused fp32 decoder shared between translation units (round 6): the register
// map, the dense layers on the f32 MFMA and on the bf16 MFMA with three-piece splits, row / column loads and stores.  mlp.hip holds
// the kernels and the host side of the decoder; lotd_mlp.hip runs the same layers behind the LoTD encoder inside one kernel.  See
// mlp.hip's header comment for the layout.
#pragma once
#include "common.h"
#include <type_traits>

// 1: streaming (non-temporal) row loads / stores as in rounds 3-4; 0: plain (see mlp_half.hip: the 16-byte pieces of a row arrive over
// four instructions, L1 / L2 serve the re-touches of a line only for plain accesses)
#ifndef NR3D_MLP_NT
#define NR3D_MLP_NT 0
#endif

namespace nr3d {
namespace mlp {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;                  // 4 waves per workgroup, one 32-sample tile per wave at a time
constexpr int kMaxLds = 144 * 1024;            // of the CU's 160 KB

__host__ __device__ constexpr uint32_t tiles(uint32_t d) { return (d + 31u) / 32u; }
// floats of one packed layer: weights [NO][NI][4][64][4] + bias [NO * 32]
__host__ __device__ constexpr uint32_t layer_floats(uint32_t ni, uint32_t no) { return no * ni * 1024u + no * 32u; }

// the backward kernel's padded LDS copy of a packed layer (dense<..., PAD = true> / dense_t): a group of 64 lanes x 4 floats starts
// every kGS floats and its second half-wave kHS floats in -- 8 and 4 floats of padding that put the transposed 4-byte reads of
// dense_t on different banks while every quarter wave of the forward's 16-byte reads stays contiguous
constexpr int kGS = 264, kHS = 132;
__host__ __device__ constexpr uint32_t layer_floats_pad(uint32_t ni, uint32_t no) { return no * ni * 4u * kGS + no * 32u; }
// (the x3 planes have groups of the same size, 64 lanes x 8 bf16: six groups per tile pair; their padding is tuned for the
// transposing read of dense_x3_t instead -- 32 and 16 floats per group and half-wave: the four rows of a 16-lane read, the two
// halves of a row and the two 16-lane groups of a half-wave all fall on different banks)
constexpr int kGS3 = 288, kHS3 = 144;
__host__ __device__ constexpr uint32_t layer_x3_floats_pad(uint32_t ni, uint32_t no) { return no * ni * 6u * kGS3 + no * 32u; }

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
// floats of one x3 layer: three planes of [NO][NI][2 steps][64 lanes][8 bf16] + bias fp32 [NO * 32]
__host__ __device__ constexpr uint32_t layer_x3_floats(uint32_t ni, uint32_t no) { return no * ni * 1536u + no * 32u; }
// ---------------------------------------------------------------------------------------------
// one dense layer on the register map; wp -> LDS copy of the packed layer
// ---------------------------------------------------------------------------------------------
// ReLU as ONE instruction, a signed integer maximum on the bits (v_max_i32: negative floats, -0 included, are negative integers).
// fmaxf(v, 0) costs two -- the compiler canonicalises the MFMA result first (a v_max_f32 v, v, v in front of the v_max_f32 0, v).
// Same values for every non-NaN input; a NaN with a clear sign bit stays NaN, as torch.relu keeps it (fmaxf returned 0).
__device__ __forceinline__ float relu(float v) {
	const int b = __builtin_bit_cast(int, v);
	return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ float activate(float v, int act) { return act == NR3D_MLP_ACT_RELU ? relu(v) : v; }

// Scheduling of the MFMA stream (measured rules, MI355X_MICROARCH.md): an instruction issued between two MFMAs on the
// SAME accumulator costs ~43 cycles (a cliff), between MFMAs on DIFFERENT accumulators ~6.  So consecutive MFMAs
// alternate accumulators -- the out tiles of the layer, or, for a single out tile, two partial sums over the even / odd
// k-steps that are added at the end -- and the weight fragments (one 16-byte LDS read per lane and 4 MFMA steps) are
// read kWPF groups ahead of their use: with one or two waves per SIMD nothing else would cover the LDS latency.
constexpr int kWPF = 6;

// PAD: wp is the backward kernel's padded copy (kGS / kHS below) instead of the packed layout
template <int NI, int NO, bool BIAS, bool PAD = false>
__device__ __forceinline__ void dense(const float *__restrict__ wp, const f16v (&in)[NI], f16v (&out)[NO], int act, int lane) {
	const float *bias = wp + NO * NI * (PAD ? 4 * kGS : 1024);
	const int h = lane >> 5;
	constexpr bool SPLIT = (NO == 1);              // one out tile: two accumulators over alternating k-steps
	constexpr int G = NO * NI * 4;                 // weight groups, consumed in (it, q, ot) order
	constexpr int PF = kWPF < G ? kWPF : G;
	const f4v *wv = reinterpret_cast<const f4v *>(wp) + (PAD ? h * (kHS / 4) + (lane & 31) : lane);
	auto lds_index = [](int g) { const int ot = g % NO, s = g / NO; return ((ot * NI + s / 4) * 4 + (s % 4)) * (PAD ? kGS / 4 : 64); };   // [ot][it][q] in LDS
	f4v ring[PF];
#pragma unroll
	for (int g = 0; g < PF; ++g) ring[g] = wv[lds_index(g)];
	f16v alt;                                      // SPLIT: the odd k-steps' partial sum
#pragma unroll
	for (int j = 0; j < 16; ++j) alt[j] = 0.0f;
#pragma unroll
	for (int ot = 0; ot < NO; ++ot)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			f4v b4 = {0.0f, 0.0f, 0.0f, 0.0f};
			if (BIAS) b4 = *reinterpret_cast<const f4v *>(bias + 32 * ot + 8 * q + 4 * h);
#pragma unroll
			for (int b = 0; b < 4; ++b) out[ot][4 * q + b] = b4[b];
		}
#pragma unroll
	for (int s = 0; s < NI * 4; ++s) {
		const int it = s / 4, q = s % 4;
		f4v w4[NO];
#pragma unroll
		for (int ot = 0; ot < NO; ++ot) {
			const int g = s * NO + ot;
			w4[ot] = ring[g % PF];
			if (g + PF < G) ring[g % PF] = wv[lds_index(g + PF)];
		}
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			if constexpr (SPLIT) {
				if (b & 1) alt = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[0][b], in[it][4 * q + b], alt, 0, 0, 0);
				else out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[0][b], in[it][4 * q + b], out[0], 0, 0, 0);
			} else {
#pragma unroll
				for (int ot = 0; ot < NO; ++ot)
					out[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[ot][b], in[it][4 * q + b], out[ot], 0, 0, 0);
			}
		}
	}
#pragma unroll
	for (int ot = 0; ot < NO; ++ot)
#pragma unroll
		for (int j = 0; j < 16; ++j) out[ot][j] = activate(SPLIT ? out[ot][j] + alt[j] : out[ot][j], act);
}

// out = W^T in from the PADDED copy of the forward layer W (NI tiles of W's outputs come in, NO tiles of W's inputs go out; round 6:
// the backward keeps ONE copy of the weights in LDS).  Lane (r, h) of the A operand of step (it, q, b) holds
// W[32 it + 8 q + 4 h + b][32 ot + r], which the packed layout keeps at group (it * NO + ot) * 4 + (r >> 3), half-wave (r >> 2) & 1,
// lane 8 q + 4 h + b, element r & 3: four 4-byte reads 16 bytes apart where dense() takes one 16-byte read.  Unpadded, the lanes'
// addresses would differ by multiples of 512 bytes (r >> 2) -> 8 lanes per bank; with the padding the 32 lanes of a half-wave fall
// on 32 different banks ((r >> 3) * 8 + ((r >> 2) & 1) * 4 + (r & 3)).
template <int NI, int NO>
__device__ __forceinline__ void dense_t(const float *__restrict__ wp, const f16v (&in)[NI], f16v (&out)[NO], int lane) {
	const int r = lane & 31, h = lane >> 5;
	constexpr bool SPLIT = (NO == 1);
	constexpr int G = NO * NI * 4;
	constexpr int PF = kWPF < G ? kWPF : G;
	const float *wl = wp + (r >> 3) * kGS + ((r >> 2) & 1) * kHS + (r & 3) + 16 * h;
	auto fetch = [&](int g) {
		const int ot = g % NO, s = g / NO;                            // s = it * 4 + q
		const float *p = wl + ((s / 4) * NO + ot) * 4 * kGS + 32 * (s % 4);
		const f4v v = {p[0], p[4], p[8], p[12]};
		return v;
	};
	f4v ring[PF];
#pragma unroll
	for (int g = 0; g < PF; ++g) ring[g] = fetch(g);
	f16v alt;
#pragma unroll
	for (int j = 0; j < 16; ++j) alt[j] = 0.0f;
#pragma unroll
	for (int ot = 0; ot < NO; ++ot)
#pragma unroll
		for (int j = 0; j < 16; ++j) out[ot][j] = 0.0f;
#pragma unroll
	for (int s = 0; s < NI * 4; ++s) {
		const int it = s / 4, q = s % 4;
		f4v w4[NO];
#pragma unroll
		for (int ot = 0; ot < NO; ++ot) {
			const int g = s * NO + ot;
			w4[ot] = ring[g % PF];
			if (g + PF < G) ring[g % PF] = fetch(g + PF);
		}
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			if constexpr (SPLIT) {
				if (b & 1) alt = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[0][b], in[it][4 * q + b], alt, 0, 0, 0);
				else out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[0][b], in[it][4 * q + b], out[0], 0, 0, 0);
			} else {
#pragma unroll
				for (int ot = 0; ot < NO; ++ot)
					out[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[ot][b], in[it][4 * q + b], out[ot], 0, 0, 0);
			}
		}
	}
	if constexpr (SPLIT) {
#pragma unroll
		for (int j = 0; j < 16; ++j) out[0][j] += alt[j];
	}
}

// one packed layer -> its padded LDS copy (all threads of the workgroup; the caller synchronises)
// (GPP = groups per tile pair: 4 for the f32 layout, 6 for the x3 planes)
template <int NI, int NO, int GPP = 4>
__device__ __forceinline__ void stage_layer_padded(const float *__restrict__ src, float *__restrict__ dst) {
	constexpr int GS = GPP == 4 ? kGS : kGS3, HS = GPP == 4 ? kHS : kHS3;
	const f4v *s4 = reinterpret_cast<const f4v *>(src);
	f4v *d4 = reinterpret_cast<f4v *>(dst);
	for (uint32_t i = threadIdx.x; i < (uint32_t)(NO * NI * GPP * 64); i += blockDim.x)
		d4[(i >> 6) * (GS / 4) + ((i >> 5) & 1u) * (HS / 4) + (i & 31u)] = s4[i];
	for (uint32_t i = threadIdx.x; i < (uint32_t)(NO * 32); i += blockDim.x) dst[NO * NI * GPP * GS + i] = src[NO * NI * GPP * 256 + i];
}

// two values -> their three bf16 pieces at elements e, e + 1 of the operands.  The conversions are packed (v_cvt_pk_bf16_f32).  The
// residuals: SCALAR subtractions in the forward kernels (two waves per SIMD: beside MFMAs a v_pk_add_f32 costs about 13 cycles beyond
// its issue slot, MI355X_MICROARCH.md "price of one filler beside MFMAs", two v_sub_f32 do not -- the empty asm statements keep the
// SLP vectoriser from packing them again), PACKED in the backward kernels (one wave per SIMD, VALU-bound: the instruction count decides).
// Same-box A/B at 2^22 samples, scalar against packed: forward 64 -> 64 -> 64 -> 64 0.710 -> 0.656 ms, 32 -> 64 -> 16 0.228 -> 0.220;
// backward 32 -> 64 -> 64 -> 16 1.45 -> 1.52, 32 -> 32 -> 32 -> 16 0.64 -> 0.68.
template <bool SCALAR>
__device__ __forceinline__ void split3_pair(float a0, float a1, int e, bf8 (&p)[3]) {
	const f2v a = {a0, a1};
	const bf2 p1 = __builtin_convertvector(a, bf2);
	f2v r1;
	if constexpr (SCALAR) {
		float r0 = a0 - (float)p1[0], rb = a1 - (float)p1[1];
		asm volatile("" : "+v"(r0));
		asm volatile("" : "+v"(rb));
		r1[0] = r0; r1[1] = rb;
	} else {
		r1 = a - __builtin_convertvector(p1, f2v);
	}
	const bf2 p2 = __builtin_convertvector(r1, bf2);
	f2v r2;
	if constexpr (SCALAR) {
		float q0 = r1[0] - (float)p2[0], q1 = r1[1] - (float)p2[1];
		asm volatile("" : "+v"(q0));
		asm volatile("" : "+v"(q1));
		r2[0] = q0; r2[1] = q1;
	} else {
		r2 = r1 - __builtin_convertvector(p2, f2v);
	}
	const bf2 p3 = __builtin_convertvector(r2, bf2);
	p[0][e] = p1[0]; p[0][e + 1] = p1[1];
	p[1][e] = p2[0]; p[1][e + 1] = p2[1];
	p[2][e] = p3[0]; p[2][e + 1] = p3[1];
}

// the three bf16 pieces of registers 8 s .. 8 s + 7 of a register-map tile (a K = 16 step's B operand)
template <bool SCALAR = false>
__device__ __forceinline__ void split3(const f16v &v, int s, bf8 (&p)[3]) {
#pragma unroll
	for (int e = 0; e < 8; e += 2) split3_pair<SCALAR>(v[8 * s + e], v[8 * s + e + 1], e, p);
}

// one dense layer in fp32 on the bf16 MFMA; wp -> LDS copy of the layer's x3 planes (+ bias)
template <int NI, int NO, bool BIAS, bool PAD = false>
__device__ __forceinline__ void dense_x3(const float *__restrict__ wp, const f16v (&in)[NI], f16v (&out)[NO], int act, int lane) {
	const float *bias = wp + NO * NI * (PAD ? 6 * kGS3 : 1536);
	const int h = lane >> 5;
	constexpr int GU = PAD ? kGS3 / 4 : 64;             // bf8 units per group
	constexpr int PLANE = NO * NI * 2 * GU;             // bf8 units per plane
	const bf8 *wv = reinterpret_cast<const bf8 *>(wp) + (PAD ? h * (kHS3 / 4) + (lane & 31) : lane);
	constexpr bool SPLIT = (NO == 1);                  // one out tile: the small terms go to a second accumulator (no dependent MFMA chain)
	const f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
	f16v alt = zero;
#pragma unroll
	for (int ot = 0; ot < NO; ++ot)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			f4v b4 = {0.0f, 0.0f, 0.0f, 0.0f};
			if (BIAS) b4 = *reinterpret_cast<const f4v *>(bias + 32 * ot + 8 * q + 4 * h);
#pragma unroll
			for (int b = 0; b < 4; ++b) out[ot][4 * q + b] = b4[b];
		}
	// (weight piece, input piece) of the six kept products, smallest first
	constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
	for (int it = 0; it < NI; ++it)
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			bf8 xs[3];
			split3<!PAD>(in[it], s, xs);            // (PAD = the backward kernels)
			bf8 w[3][NO];
#pragma unroll
			for (int pl = 0; pl < 3; ++pl)
#pragma unroll
				for (int ot = 0; ot < NO; ++ot) w[pl][ot] = wv[pl * PLANE + ((ot * NI + it) * 2 + s) * GU];
#pragma unroll
			for (int t = 0; t < 6; ++t) {
				if constexpr (SPLIT) {
					if (t < 5) alt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][0], xs[PX[t]], alt, 0, 0, 0);
					else out[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][0], xs[PX[t]], out[0], 0, 0, 0);
				} else {
#pragma unroll
					for (int ot = 0; ot < NO; ++ot) out[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][ot], xs[PX[t]], out[ot], 0, 0, 0);
				}
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

// out = W^T in on the bf16 MFMA from the PADDED x3 planes of the forward layer W (dense_t's counterpart: NI tiles of W's outputs come in,
// NO tiles of W's inputs go out).  Element e = 0 .. 7 of lane (r, h) of the A operand of K = 16 step (it, s) is
// W_pl[32 it + o][32 ot + r] with o = 16 s + 8 (e >> 2) + 4 h + (e & 3) (the register map's rows): for a 16-lane group that is a
// block of 4 consecutive rows o x 16 consecutive inputs r, which the planes hold as four 8-byte pieces per row (lane o of either
// half-wave, elements 0 .. 3 or 4 .. 7) -- exactly what gfx950's transposing read takes: ds_read_b64_tr_b16 hands lane c column c of
// the 4 x 16 block whose pieces the group's lanes point at, so an operand is TWO reads (e >> 2 = 0, 1).  (The first version read
// eight 2-byte values and packed them: four times the LDS instructions plus the packing, slower than the planes in both
// orientations wherever those fitted.)  Same piece products in the same order as dense_x3.
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
template <int NI, int NO>
__device__ __forceinline__ void dense_x3_t(const float *__restrict__ wp, const f16v (&in)[NI], f16v (&out)[NO], int lane) {
	const int c = lane & 15, sf = (lane >> 4) & 1, h = lane >> 5;
	constexpr int PLANE = NO * NI * 2 * kGS3 * 2;      // 2-byte units per plane
	// this lane's piece of the block: row c >> 2, piece q = c & 3 = inputs 4 q .. 4 q + 3 of the group's 16 -> half-wave q & 1, elements 4 (q >> 1) ..
	const unsigned short *wl = reinterpret_cast<const unsigned short *>(wp) + sf * (kGS3 * 2) + (c & 1) * (kHS3 * 2) + (4 * h + (c >> 2)) * 8 + 4 * ((c >> 1) & 1);
	typedef s4v __attribute__((address_space(3))) *lds_s4;
	constexpr bool SPLIT = (NO == 1);
	const f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
	f16v alt = zero;
#pragma unroll
	for (int ot = 0; ot < NO; ++ot) out[ot] = zero;
	constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
	for (int it = 0; it < NI; ++it)
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			bf8 xs[3];
			split3(in[it], s, xs);
			bf8 w[3][NO];
#pragma unroll
			for (int pl = 0; pl < 3; ++pl)
#pragma unroll
				for (int ot = 0; ot < NO; ++ot) {
					const unsigned short *p = wl + pl * PLANE + (it * NO + ot) * 2 * (kGS3 * 2) + 128 * s;
					const bf4 lo = __builtin_bit_cast(bf4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)p));
					const bf4 hi = __builtin_bit_cast(bf4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(p + 64)));
					const bf8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
					w[pl][ot] = v;
				}
#pragma unroll
			for (int t = 0; t < 6; ++t) {
				if constexpr (SPLIT) {
					if (t < 5) alt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][0], xs[PX[t]], alt, 0, 0, 0);
					else out[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][0], xs[PX[t]], out[0], 0, 0, 0);
				} else {
#pragma unroll
					for (int ot = 0; ot < NO; ++ot) out[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PW[t]][ot], xs[PX[t]], out[ot], 0, 0, 0);
				}
			}
		}
	if constexpr (SPLIT) {
#pragma unroll
		for (int j = 0; j < 16; ++j) out[0][j] += alt[j];
	}
}

// rows of a [n, dim] matrix on the register map: lane (s = lane & 31, h = lane >> 5) owns features 32t + 8q + 4h + b
template <int NT>
__device__ __forceinline__ void load_rows(const float *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row, bool valid,
                                          bool vec, int lane, f16v (&r)[NT]) {
	const int h = lane >> 5;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t f = 32u * t + 8u * q + 4u * h;
			f4v v = {0.0f, 0.0f, 0.0f, 0.0f};
			if (valid && f < dim) {
				const float *src = p + (int64_t)row * stride + f;
				if (vec && f + 3 < dim) v = NR3D_MLP_NT ? __builtin_nontemporal_load(reinterpret_cast<const f4v *>(src)) : *reinterpret_cast<const f4v *>(src);
				else {
#pragma unroll
					for (int b = 0; b < 4; ++b) if (f + b < dim) v[b] = src[b];
				}
			}
#pragma unroll
			for (int b = 0; b < 4; ++b) r[t][4 * q + b] = v[b];
		}
}

// Branch-free variant for 16-byte aligned rows whose width is a multiple of 4: every lane loads from a valid address
// (row clamped to the last row, a piece beyond the width re-reads piece 0) and nothing is selected afterwards -- padded
// features meet zero weights and rows beyond n are never stored -- so no value is consumed before the first MFMA and
// the loads of the NEXT tile can stay in flight under this tile's arithmetic (a load under a branch makes the compiler
// drain the memory counter at the join).
template <int NT>
__device__ __forceinline__ void load_rows_fast(const float *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row_clamped,
                                               int lane, f16v (&r)[NT]) {
	const int h = lane >> 5;
	const float *base = p + (int64_t)row_clamped * stride;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t f = 32u * t + 8u * q + 4u * h;
			const f4v v = NR3D_MLP_NT ? __builtin_nontemporal_load(reinterpret_cast<const f4v *>(base + (f < dim ? f : 0u))) : *reinterpret_cast<const f4v *>(base + (f < dim ? f : 0u));
#pragma unroll
			for (int b = 0; b < 4; ++b) r[t][4 * q + b] = v[b];
		}
}

// Feature-major input (element (row, f) at p[f * fstride + row], e.g. the [E, N] storage the LoTD kernels write): a
// half-wave reads 32 consecutive samples of one feature, 128 contiguous bytes per request -- the natural layout of the
// B operand of H^T = W X^T.  Branch-free like load_rows_fast: the row is clamped, a feature beyond the width re-reads
// feature 0 (it meets zero weights).  No alignment requirement.
template <int NT>
__device__ __forceinline__ void load_cols_fast(const float *__restrict__ p, int64_t fstride, uint32_t dim, uint64_t row_clamped,
                                               int lane, f16v (&r)[NT]) {
	const int h = lane >> 5;
	const float *base = p + row_clamped;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			const uint32_t f = 32u * t + 8u * (j >> 2) + 4u * h + (j & 3);
			r[t][j] = __builtin_nontemporal_load(base + (int64_t)(f < dim ? f : 0u) * fstride);
		}
}

template <int NT>
__device__ __forceinline__ void store_cols(float *__restrict__ p, int64_t fstride, uint32_t dim, uint64_t row, bool valid, int lane,
                                           const f16v (&r)[NT]) {
	const int h = lane >> 5;
	if (!valid) return;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			const uint32_t f = 32u * t + 8u * (j >> 2) + 4u * h + (j & 3);
			if (f < dim) __builtin_nontemporal_store(r[t][j], p + (int64_t)f * fstride + row);
		}
}

template <int NT>
__device__ __forceinline__ void store_rows(float *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row, bool valid, bool vec,
                                           int lane, const f16v (&r)[NT]) {
	const int h = lane >> 5;
	if (!valid) return;
#pragma unroll
	for (int t = 0; t < NT; ++t)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint32_t f = 32u * t + 8u * q + 4u * h;
			if (f >= dim) continue;
			float *dst = p + (int64_t)row * stride + f;
			if (vec && f + 3 < dim) {
				const f4v v = {r[t][4 * q], r[t][4 * q + 1], r[t][4 * q + 2], r[t][4 * q + 3]};
				// (rows wider than one tile: plain stores, L2 merges the row's 16-byte pieces into whole lines -- mlp_half.hip, store_rows)
				if (NT > 1 || !NR3D_MLP_NT) *reinterpret_cast<f4v *>(dst) = v; else __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(dst));
			} else {
#pragma unroll
				for (int b = 0; b < 4; ++b) if (f + b < dim) dst[b] = r[t][4 * q + b];
			}
		}
}

__device__ __forceinline__ void stage_weights(const float *__restrict__ packed, uint32_t n_floats, float *lds) {
	const f4v *src = reinterpret_cast<const f4v *>(packed);
	f4v *dst = reinterpret_cast<f4v *>(lds);
	for (uint32_t i = threadIdx.x; i < n_floats / 4; i += kThreads) dst[i] = src[i];
	__syncthreads();
}

// XF: 0 = row-major input, any alignment / width; 1 = row-major, 16-byte aligned rows of a multiple of 4 floats
// (prefetched); 2 = feature-major input (a.xs = feature stride, prefetched)
template <int XF, int NT>
__device__ __forceinline__ void prefetch_x(const float *__restrict__ p, int64_t stride, uint32_t dim, uint64_t row_clamped, int lane,
                                           f16v (&r)[NT]) {
	if constexpr (XF == 2) load_cols_fast<NT>(p, stride, dim, row_clamped, lane, r);
	else load_rows_fast<NT>(p, stride, dim, row_clamped, lane, r);
}

// host side (mlp.hip): the region of a packed decoder (nr3d_mlp_pack) the forward kernels stage in LDS -- the x3 planes when the bf16
// route is on and they exist, else the f32 layers -- and the decoder's tile counts.  false: outside the fused kernels' range.
bool forward_region(const nr3d_mlp_desc_t *desc, const float *packed, bool &x3, const float *&region, uint32_t &region_floats,
                    uint32_t &in_t, uint32_t &w_t, uint32_t &out_t);

}  // namespace mlp
}  // namespace nr3d
