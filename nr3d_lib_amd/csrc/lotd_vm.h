// nr3d_lib_amd/csrc/lotd_vm.h -- the per-component update arithmetic and the table geometry of a VectorMatrix level, shared by the
// record path / k_vm_direct (lotd_bin.hip) and the sorted-points path (lotd_sorted.hip).
#pragma once
#include "lotd_device.h"

namespace nr3d {
namespace lotd {


// ONE component d of a VM level's updates (records 0..3: the plane's four entries, 4..5: the line's two), for the
// three-threads-per-point form of stage A (bin_body, SPLIT == 3), in SEPARABLE form: with (A, B) the plane's dims, m its corner,
// wo_m = w_A w_B, LI = lerp_d(line), PI = sum_m wo_m plane_m:
//     first order    plane_m += g wo_m LI                              line_s += g w_d(s) PI
//     second order   plane_m += g (wo_m a_d (line_1 - line_0) + C_m LI) line_s += g (a_d sgn(s) PI + w_d(s) PC)
//                    C_m = a_A sgn_A w_B + a_B sgn_B w_A,  PC = sum_m C_m plane_m,  a = scale w' v
// -- what emit_plane_line sums corner by corner from the eight corner weights ((g w_k0) line_0 + (g w_k1) line_1, ...): the
// same polynomial, other association, a third of the arithmetic and no corner-weight array (stage A of the VM levels was
// half VALU after the split, profiles/r03m_c4_counters.txt).
// the arithmetic of one component, given the four plane values pv[m] (m bit 0: the plane's first dim, bit 1: its second) and the two line
// values lv[s] of every feature
template <int G, int DC, int NRT, bool SECOND>
__device__ __forceinline__ void vm_component_math(const Cell<3> &c, const float (&a)[3], const float (&grad)[G], const float (&pv)[4][G],
                                                  const float (&lv)[2][G], float (&val)[NRT][G]) {
	constexpr int DA = DC == 0 ? 1 : 0, DB = DC == 2 ? 1 : 2;          // the plane's dims, ascending: bit 0 / bit 1 of m
	const float wd1 = c.w[DC], wd0 = 1.0f - wd1;
	float wo[4], Cm[4];
#pragma unroll
	for (uint32_t m = 0; m < 4; ++m) {
		const float wa = (m & 1u) ? c.w[DA] : 1.0f - c.w[DA], wb = (m & 2u) ? c.w[DB] : 1.0f - c.w[DB];
		wo[m] = wa * wb;
		Cm[m] = SECOND ? __fmaf_rn((m & 1u) ? a[DA] : -a[DA], wb, ((m & 2u) ? a[DB] : -a[DB]) * wa) : 0.0f;
	}
#pragma unroll
	for (int f = 0; f < G; ++f) {
		const float LI = __fmaf_rn(wd1, lv[1][f], wd0 * lv[0][f]);
		float PI = 0.0f, PC = 0.0f;
#pragma unroll
		for (uint32_t m = 0; m < 4; ++m) { PI = __fmaf_rn(wo[m], pv[m][f], PI); if (SECOND) PC = __fmaf_rn(Cm[m], pv[m][f], PC); }
		if (!SECOND) {
			const float gl = grad[f] * LI, gp = grad[f] * PI;
#pragma unroll
			for (uint32_t m = 0; m < 4; ++m) val[m][f] = gl * wo[m];
			val[4][f] = gp * wd0;
			val[5][f] = gp * wd1;
		} else {
			const float dl = a[DC] * (lv[1][f] - lv[0][f]);
#pragma unroll
			for (uint32_t m = 0; m < 4; ++m) val[m][f] = grad[f] * __fmaf_rn(wo[m], dl, Cm[m] * LI);
			const float ap = a[DC] * PI;
			val[4][f] = grad[f] * __fmaf_rn(wd0, PC, -ap);
			val[5][f] = grad[f] * __fmaf_rn(wd1, PC, ap);
		}
	}
}

template <int G, int DC, int NRT, bool SECOND, typename TB>
__device__ __forceinline__ uint32_t emit_vm_component(const Lvl &L, const Cell<3> &c, const float (&a)[3], const float (&grad)[G],
                                                      TB grid, uint32_t foff, uint32_t (&ent)[NRT], float (&val)[NRT][G]) {
	static_assert(NRT >= 6, "six records per VM component");
	// (round 4, measured: one 8-byte load per feature pair behind a run-time alignment test instead of the two 4-byte loads below is
	// SLOWER -- k_vm_direct 1.28 -> 1.39 ms, k_bin_vm3 1.03 -> 1.07 -- the pair's second word is an L1 hit either way)
	float pv[4][G], lv[2][G];
	uint32_t pe[4], le[2];
#pragma unroll
	for (uint32_t m = 0; m < 4; ++m) {
		uint32_t p[3], pl[3], ln[3];
		corner_pos<3>(c, insert_zero(m, DC), p);
		entry_vm(L, p, pl, ln);
		pe[m] = pl[DC];
		if (m == 0) le[0] = ln[DC];
#pragma unroll
		for (int f = 0; f < G; ++f) pv[m][f] = grid[pe[m] * L.F + foff + f];
	}
	le[1] = le[0] + 1u;
#pragma unroll
	for (uint32_t sl = 0; sl < 2; ++sl)
#pragma unroll
		for (int f = 0; f < G; ++f) lv[sl][f] = grid[le[sl] * L.F + foff + f];
	vm_component_math<G, DC, NRT, SECOND>(c, a, grad, pv, lv, val);
#pragma unroll
	for (uint32_t m = 0; m < 4; ++m) ent[m] = pe[m];
	ent[4] = le[0]; ent[5] = le[1];
	return 6u;
}

// The same for an INTERIOR cell of a level whose entries are exactly one 2-feature pseudo level (L.F == 2): the two corners along the
// plane's second dim are neighbouring entries, and so are the line's two -- three loads (two plane rows + the line) of two whole entries
// each instead of twelve scalar ones.  Same values, same arithmetic: bit-identical to emit_vm_component.
struct __attribute__((aligned(4))) VmHalf4 { __half2 a, b; };     // two half entries: 4-byte aligned 8-byte load
struct __attribute__((aligned(8))) VmFloat4 { float x, y, z, w; };   // two float entries: 8-byte aligned 16-byte load
__device__ __forceinline__ void vm_ld_pair(const float *g, uint32_t entry, float (&lo)[2], float (&hi)[2]) {
	const VmFloat4 t = *reinterpret_cast<const VmFloat4 *>(g + (size_t)entry * 2u);
	lo[0] = t.x; lo[1] = t.y; hi[0] = t.z; hi[1] = t.w;
}
__device__ __forceinline__ void vm_ld_pair(HalfTab g, uint32_t entry, float (&lo)[2], float (&hi)[2]) {
	const VmHalf4 t = *reinterpret_cast<const VmHalf4 *>(g.p + (size_t)entry * 2u);
	const float2 u = __half22float2(t.a), v = __half22float2(t.b);
	lo[0] = u.x; lo[1] = u.y; hi[0] = v.x; hi[1] = v.y;
}
template <int DC, int NRT, bool SECOND, typename TB>
__device__ __forceinline__ uint32_t emit_vm_component_f2(const Lvl &L, const Cell<3> &c, const float (&a)[3], const float (&grad)[2],
                                                         TB grid, uint32_t (&ent)[NRT], float (&val)[NRT][2]) {
	static_assert(NRT >= 6, "six records per VM component");
	float pv[4][2], lv[2][2];
	uint32_t p[3], pl[3], ln[3];
	corner_pos<3>(c, 0u, p);
	entry_vm(L, p, pl, ln);
	const uint32_t pe0 = pl[DC], le0 = ln[DC];
	constexpr int DA = DC == 0 ? 1 : 0, DB = DC == 2 ? 1 : 2;
	const uint32_t Rb = L.res[DB];                                  // plane_d is row-major over (DA, DB): m bit 0 steps a whole row
	vm_ld_pair(grid, pe0, pv[0], pv[2]);
	vm_ld_pair(grid, pe0 + Rb, pv[1], pv[3]);
	vm_ld_pair(grid, le0, lv[0], lv[1]);
	vm_component_math<2, DC, NRT, SECOND>(c, a, grad, pv, lv, val);
	ent[0] = pe0; ent[1] = pe0 + Rb; ent[2] = pe0 + 1u; ent[3] = pe0 + Rb + 1u;
	ent[4] = le0; ent[5] = le0 + 1u;
	return 6u;
}

struct VmGeom { uint32_t Ra, Rb, plane_lo, line_lo, Rd; int a; };
__host__ __device__ inline VmGeom vm_geom(const uint32_t (&res)[NR3D_LOTD_MAX_DIMS], int d) {
	VmGeom gm;
	if (d == 3) {
		// a DENSE level in the same terms (lotd_sorted.hip serves a forest's small Dense levels next to its VM levels): rows = the x_0
		// slices of the table [R0][R1][R2], a row = one slice of R1 R2 entries, no line
		gm.a = 0; gm.Ra = res[0]; gm.Rb = res[1] * res[2]; gm.Rd = 0u; gm.plane_lo = 0u; gm.line_lo = 0u;
		return gm;
	}
	gm.a = d == 0 ? 1 : 0;
	const int b = d == 2 ? 1 : 2;
	gm.Ra = res[gm.a]; gm.Rb = res[b]; gm.Rd = res[d];
	const uint32_t lines = res[0] + res[1] + res[2];
	const uint32_t psz[3] = {res[1] * res[2], res[0] * res[2], res[0] * res[1]};
	gm.plane_lo = lines + (d > 0 ? psz[0] : 0u) + (d > 1 ? psz[1] : 0u);
	gm.line_lo = (d > 0 ? res[0] : 0u) + (d > 1 ? res[1] : 0u);
	return gm;
}

}  // namespace lotd
}  // namespace nr3d
