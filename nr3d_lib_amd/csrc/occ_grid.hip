// nr3d_lib_amd/csrc/occ_grid.hip -- occupancy-grid ray marching (gfx950), C-ABI entry points
// nr3d_ray_marching_count / nr3d_ray_marching_emit.
//
// Replaces nr3d_lib.bindings._occ_grid.{ray_marching, batched_ray_marching}:
//   csrc/occ_grid/src/ray_marching.cu:17-244, csrc/occ_grid/src/batched_marching.cu:18-287,
//   helpers csrc/occ_grid/include/occ_grid/helpers_march.h:11-77, helpers_contraction.h:10-125.
//
// The per-ray t-sequence is a serial fp32 recurrence whose rounding decides which voxel a sample lands
// in, so each lane walks one ray with exactly the reference's operation order (origin + t*dir is an
// explicit fmaf, matching nvcc's contraction; division and sqrt are IEEE-correct).  What changes vs. the
// reference is everything around the loop: the per-ray counts are scanned ON THE DEVICE into
// packed_info (no host cumsum/stack), the grand total is left in a device word for the caller's
// single readback, and single/batched marching share one kernel.
#include "common.h"
#include "scan.h"
#include "compact.h"
#include "pack_launch.h"
#include <stdlib.h>

namespace nr3d {
namespace occ {

constexpr int kBlock = 128;   // rays per workgroup: more workgroups in flight for small ray counts

struct f3 { float x, y, z; };

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
__device__ __forceinline__ float calc_dt(float t, float g, float lo, float hi) { return clampf(t * g, lo, hi); }

struct Grid {
	f3 mn, mx;
	int rx, ry, rz;
	const uint8_t *cells;
	int type;
	// POW2 fast path: when the ROI extents and the resolution are powers of two, x / d == x * (1 / d) bit for bit
	// (1 / d is exact and scaling by a power of two does not round), so the ~12-instruction IEEE division
	// sequences of the probe and of the voxel-exit distance become single multiplies.
	f3 inv_ext, inv_res;
};

__device__ __forceinline__ bool is_pow2f(float v) {
	const uint32_t b = __float_as_uint(v);
	const uint32_t e = (b >> 23) & 0xFFu;
	return (b & 0x807FFFFFu) == 0u && e >= 64u && e <= 190u;    // positive, mantissa 0, comfortably normal (and so is 1/v)
}

template <bool POW2>
__device__ __forceinline__ f3 to_unit(const Grid &g, f3 p) {
	if (POW2) return {(p.x - g.mn.x) * g.inv_ext.x, (p.y - g.mn.y) * g.inv_ext.y, (p.z - g.mn.z) * g.inv_ext.z};
	return {(p.x - g.mn.x) / (g.mx.x - g.mn.x), (p.y - g.mn.y) / (g.mx.y - g.mn.y), (p.z - g.mn.z) / (g.mx.z - g.mn.z)};
}

template <bool POW2>
__device__ __forceinline__ f3 contract(const Grid &g, f3 p) {
	f3 u = to_unit<POW2>(g, p);
	if (g.type == NR3D_CONTRACT_UN_BOUNDED_TANH) {
		u = {tanhf(u.x - 0.5f) * 0.5f + 0.5f, tanhf(u.y - 0.5f) * 0.5f + 0.5f, tanhf(u.z - 0.5f) * 0.5f + 0.5f};
	} else if (g.type == NR3D_CONTRACT_UN_BOUNDED_SPHERE) {
		u = {u.x * 2.0f - 1.0f, u.y * 2.0f - 1.0f, u.z * 2.0f - 1.0f};
		const float n2 = __fmaf_rn(u.z, u.z, __fmaf_rn(u.y, u.y, u.x * u.x));
		const float n = sqrtf(n2);
		if (n > 1.0f) {
			const float s = 2.0f - 1.0f / n;
			u = {s * (u.x / n), s * (u.y / n), s * (u.z / n)};
		}
		u = {u.x * 0.25f + 0.5f, u.y * 0.25f + 0.5f, u.z * 0.25f + 0.5f};
	}
	return u;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return max(lo, min(v, hi)); }

// returns occupancy; *cell receives the flat voxel index (z contiguous)
template <bool POW2>
__device__ __forceinline__ bool probe(const Grid &g, f3 p, int *cell) {
	if (g.type == NR3D_CONTRACT_AABB &&
	    (p.x < g.mn.x || p.x > g.mx.x || p.y < g.mn.y || p.y > g.mx.y || p.z < g.mn.z || p.z > g.mx.z))
		return false;
	const f3 u = contract<POW2>(g, p);
	const int ix = clampi((int)(u.x * (float)g.rx), 0, g.rx - 1);
	const int iy = clampi((int)(u.y * (float)g.ry), 0, g.ry - 1);
	const int iz = clampi((int)(u.z * (float)g.rz), 0, g.rz - 1);
	const int idx = ix * (g.ry * g.rz) + iy * g.rz + iz;
	*cell = idx;
	return g.cells[idx] != 0;
}

template <bool POW2>
__device__ __forceinline__ float axis_exit(float q, float dirsign, float inv, float r, float inv_r, float extent) {
	const float d = (floorf(q + 0.5f + 0.5f * dirsign) - q) * inv;
	return (POW2 ? d * inv_r : d / r) * extent;
}

// DDA distance to the next voxel boundary, then advance in dt_min multiples (helpers_march.h:47-77)
template <bool POW2>
__device__ __forceinline__ float skip_voxel(const Grid &g, float t, float dt_min, f3 p, f3 dir, f3 inv) {
	const f3 u = to_unit<POW2>(g, p);
	const float tx = axis_exit<POW2>(u.x * (float)g.rx, copysignf(1.0f, dir.x), inv.x, (float)g.rx, g.inv_res.x, g.mx.x - g.mn.x);
	const float ty = axis_exit<POW2>(u.y * (float)g.ry, copysignf(1.0f, dir.y), inv.y, (float)g.ry, g.inv_res.y, g.mx.y - g.mn.y);
	const float tz = axis_exit<POW2>(u.z * (float)g.rz, copysignf(1.0f, dir.z), inv.z, (float)g.rz, g.inv_res.z, g.mx.z - g.mn.z);
	const float target = t + fmaxf(fminf(fminf(tx, ty), tz), 0.0f);
	float tt = t;
	do { tt += dt_min; } while (tt < target);
	return tt;
}

// the serial march of one ray (one lane); returns the number of samples
template <bool EMIT, bool CACHE, bool POW2>
__device__ __forceinline__ uint32_t march_ray(const Grid &g, uint32_t i, uint32_t b, int32_t grid_offset, f3 o, f3 dir, f3 inv,
                                              float near, float far, float dt_min, float dt_max, float dt_gamma,
                                              uint32_t max_steps, uint32_t base, float *__restrict__ t_starts,
                                              float *__restrict__ t_ends, int32_t *__restrict__ ridx,
                                              int32_t *__restrict__ bidx, int32_t *__restrict__ gidx,
                                              uint32_t *__restrict__ cache) {
	const int type = g.type;
	uint32_t j = 0;
	float t0 = near;
	float dt = calc_dt(t0, dt_gamma, dt_min, dt_max);
	float t1 = t0 + dt;
	float tm = (t0 + t1) * 0.5f;
	while (tm < far && j < max_steps) {
		const f3 p = {__fmaf_rn(tm, dir.x, o.x), __fmaf_rn(tm, dir.y, o.y), __fmaf_rn(tm, dir.z, o.z)};
		int cell = -1;
		if (probe<POW2>(g, p, &cell)) {
			if (EMIT) {
				t_starts[base + j] = t0;
				t_ends[base + j] = t1;
				ridx[base + j] = (int32_t)i;
				if (bidx) bidx[base + j] = (int32_t)b;
				if (gidx) gidx[base + j] = cell + grid_offset;
			}
			if (CACHE) {
				uint32_t *c = cache + ((size_t)i * max_steps + j) * 3;
				c[0] = __float_as_uint(t0);
				c[1] = __float_as_uint(t1);
				c[2] = (uint32_t)(cell + grid_offset);
			}
			++j;
			t0 = t1;
			t1 = t0 + calc_dt(t0, dt_gamma, dt_min, dt_max);
			tm = (t0 + t1) * 0.5f;
		} else if (type == NR3D_CONTRACT_AABB) {
			tm = skip_voxel<POW2>(g, tm, dt_min, p, dir, inv);
			dt = calc_dt(tm, dt_gamma, dt_min, dt_max);
			t0 = tm - dt * 0.5f;
			t1 = tm + dt * 0.5f;
		} else {
			t0 = t1;
			t1 = t0 + calc_dt(t0, dt_gamma, dt_min, dt_max);
			tm = (t0 + t1) * 0.5f;
		}
	}
	return j;
}

// CACHE (count pass only): every emitted sample is also stored in a per-ray slot of a caller-provided cache
// ([n_rays][max_steps] x {t_start, t_end, cell}); the emit pass then is a parallel compaction (k_emit_cached) instead
// of a second serial march.  A ray is one serial chain of dependent VALU ops and one dependent byte load per probe
// (~1300 cycles per probe for a lone wave; measured: 4..64 rays per wave take the same time), so for small ray
// counts the op is bound by its longest ray and skipping the second march halves it.
template <bool EMIT, bool CACHE>
__global__ __launch_bounds__(kBlock) void k_march(uint32_t n_rays, const float *__restrict__ rays_o,
                                                  const float *__restrict__ rays_d, const float *__restrict__ t_min,
                                                  const float *__restrict__ t_max, const float *__restrict__ roi,
                                                  int rx, int ry, int rz, const uint8_t *__restrict__ cells, int type,
                                                  float step_size, float max_step_size, float dt_gamma,
                                                  uint32_t max_steps, int batched, const int32_t *__restrict__ batch_inds,
                                                  uint32_t batch_data_size, const int32_t *__restrict__ packed_info,
                                                  int32_t *__restrict__ counts, float *__restrict__ t_starts,
                                                  float *__restrict__ t_ends, int32_t *__restrict__ ridx,
                                                  int32_t *__restrict__ bidx, int32_t *__restrict__ gidx,
                                                  uint32_t *__restrict__ cache) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_rays) return;
	uint32_t b = 0;
	if (batched) {
		if (batch_inds) {
			const int32_t v = batch_inds[i];
			if (v < 0) { if (!EMIT) counts[i] = 0; return; }   // reference leaves the count uninitialised here
			b = (uint32_t)v;
		} else if (batch_data_size) {
			b = i / batch_data_size;
		}
	}
	const uint32_t vol = (uint32_t)(rx * ry * rz);
	Grid g;
	const float *r6 = roi + 6 * (size_t)b;
	g.mn = {r6[0], r6[1], r6[2]};
	g.mx = {r6[3], r6[4], r6[5]};
	g.rx = rx; g.ry = ry; g.rz = rz;
	g.cells = cells + (size_t)b * vol;
	g.type = type;
	const int32_t grid_offset = batched ? (int32_t)(b * vol) : 0;

	uint32_t base = 0;
	if (EMIT) {
		base = (uint32_t)packed_info[2 * (size_t)i];
		max_steps = (uint32_t)packed_info[2 * (size_t)i + 1];
	}
	const f3 o = {rays_o[3 * (size_t)i], rays_o[3 * (size_t)i + 1], rays_o[3 * (size_t)i + 2]};
	const f3 dir = {rays_d[3 * (size_t)i], rays_d[3 * (size_t)i + 1], rays_d[3 * (size_t)i + 2]};
	const f3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	const float far = t_max[i];
	const float dt_min = step_size, dt_max = max_step_size;

	g.inv_ext = {1.0f / (g.mx.x - g.mn.x), 1.0f / (g.mx.y - g.mn.y), 1.0f / (g.mx.z - g.mn.z)};
	g.inv_res = {1.0f / (float)rx, 1.0f / (float)ry, 1.0f / (float)rz};
	const bool pow2 = is_pow2f(g.mx.x - g.mn.x) && is_pow2f(g.mx.y - g.mn.y) && is_pow2f(g.mx.z - g.mn.z) &&
	                  is_pow2f((float)rx) && is_pow2f((float)ry) && is_pow2f((float)rz);
	uint32_t j;
	if (__all(pow2))      // wave-uniform: no divergence between the two instantiations
		j = march_ray<EMIT, CACHE, true>(g, i, b, grid_offset, o, dir, inv, t_min[i], far, dt_min, dt_max, dt_gamma, max_steps,
		                                 base, t_starts, t_ends, ridx, bidx, gidx, cache);
	else
		j = march_ray<EMIT, CACHE, false>(g, i, b, grid_offset, o, dir, inv, t_min[i], far, dt_min, dt_max, dt_gamma, max_steps,
		                                  base, t_starts, t_ends, ridx, bidx, gidx, cache);
	if (!EMIT) counts[i] = (int32_t)j;
}

// ---------------------------------------------------------------------------------------------------
// G lanes per ray (count pass with the sample cache), for ray counts that leave the chip mostly idle with one lane per
// ray: there the op is bound by the dependent chain of its longest ray -- ~100 dependent VALU ops and one dependent byte
// load per probe.  Only three of those ops ARE the recurrence (t0 <- t1, t1 <- t0 + dt(t0), tm <- (t0 + t1) / 2 after a
// hit or, outside AABB grids, after any probe); position, contraction, cell index and the load hang off tm.  So the group
// runs the recurrence `len` steps ahead (every lane computes the same serial chain, lane l keeps step l: same operations
// in the same order as the one-lane march, bit for bit), probes all `len` positions at once and keeps the prefix up to the
// first miss; a miss in an AABB grid re-centres the interval on the voxel exit (skip_voxel), which the whole group
// computes once before the next round (each lane works out its own exit while its probe is in flight).  After a miss the
// next round looks only 8 steps ahead, after an all-hit round G steps.
template <int G>
__device__ __forceinline__ unsigned long long group_ballot(bool v) {
	const unsigned long long m = __ballot(v);
	if (G == 64) return m;
	return (m >> ((threadIdx.x & 63u) / G * G)) & ((1ull << (G & 63)) - 1ull);
}

template <int G, bool POW2>
__device__ __forceinline__ uint32_t march_ray_group(const Grid &g, uint32_t lane, int32_t grid_offset, f3 o, f3 dir, f3 inv,
                                                    float near, float far, float dt_min, float dt_max, float dt_gamma,
                                                    uint32_t max_steps, uint32_t *__restrict__ cache_ray) {
	const bool aabb = g.type == NR3D_CONTRACT_AABB;
	constexpr int kShort = G < 8 ? G : 8;
	// after kMissRun consecutive empty voxels the exit recurrence is run ahead too, kMissAhead steps at a time.  Measured
	// on 4096 rays through a 128^3 grid (us per march, one lane per ray = 166 / 145): random occupancy 0.5 -- runs of ~2
	// voxels, the worst case for any look-ahead -- 131 with (4, 8), 171 with (3, 16), 126 with none; a solid ball (long
	// empty and long occupied runs) 67, 67, 180.
	constexpr int kMissRun = 4, kMissAhead = G < 8 ? G : 8;
	uint32_t j = 0, miss_run = 0;
	float t0 = near;
	float t1 = t0 + calc_dt(t0, dt_gamma, dt_min, dt_max);
	float tm = (t0 + t1) * 0.5f;
	int len = aabb ? kShort : G;
	const unsigned long long below = (1ull << lane) - 1ull;
	while (true) {
		float c0 = t0, c1 = t1, cm = tm, m0 = t0, m1 = t1, mm = tm;
		for (int k = 1; k < len; ++k) {
			c0 = c1;
			c1 = c0 + calc_dt(c0, dt_gamma, dt_min, dt_max);
			cm = (c0 + c1) * 0.5f;
			if ((int)lane >= k) { m0 = c0; m1 = c1; mm = cm; }
		}
		const float n0 = c1, n1 = n0 + calc_dt(n0, dt_gamma, dt_min, dt_max), nm = (n0 + n1) * 0.5f;   // after `len` advances
		const bool in = (int)lane < len && mm < far;
		const f3 p = {__fmaf_rn(mm, dir.x, o.x), __fmaf_rn(mm, dir.y, o.y), __fmaf_rn(mm, dir.z, o.z)};
		int cell = -1;
		bool hit;
		float sk = 0.0f;
		if (aabb) {
			// every lane also works out where ITS position would skip to if it misses (independent of the probe: the byte
			// load is in flight meanwhile); only the first missing lane's value is used
			const bool inside = in && !(p.x < g.mn.x || p.x > g.mx.x || p.y < g.mn.y || p.y > g.mx.y || p.z < g.mn.z || p.z > g.mx.z);
			uint8_t occ = 0;
			if (inside) {
				const f3 u = contract<POW2>(g, p);
				const int ix = clampi((int)(u.x * (float)g.rx), 0, g.rx - 1);
				const int iy = clampi((int)(u.y * (float)g.ry), 0, g.ry - 1);
				const int iz = clampi((int)(u.z * (float)g.rz), 0, g.rz - 1);
				cell = ix * (g.ry * g.rz) + iy * g.rz + iz;
				occ = g.cells[cell];
			}
			if (in) sk = skip_voxel<POW2>(g, mm, dt_min, p, dir, inv);
			hit = inside && occ != 0;
		} else {
			hit = in && probe<POW2>(g, p, &cell);
		}
		if (aabb) {
			const bool ok = hit && (j + lane < max_steps);
			const unsigned long long okm = group_ballot<G>(ok);
			const uint32_t F = (~okm == 0ull) ? 64u : (uint32_t)__ffsll((long long)~okm) - 1u;       // first lane that did not emit (<= len)
			if (lane < F) {
				uint32_t *c = cache_ray + (size_t)(j + lane) * 3;
				c[0] = __float_as_uint(m0); c[1] = __float_as_uint(m1); c[2] = (uint32_t)(cell + grid_offset);
			}
			j += F;
			if ((int)F >= len) { t0 = n0; t1 = n1; tm = nm; len = G; continue; }
			const float fm = __shfl(mm, (int)F, G);           // the position that missed (or ran out of range / of steps)
			if (!(fm < far) || !(j < max_steps)) break;
			tm = __shfl(sk, (int)F, G);                       // its voxel exit, computed by lane F while the probe was in flight
			miss_run = F == 0 ? miss_run + 1 : 1;
			if (miss_run >= (uint32_t)kMissRun) {
				// a run of empty voxels: the exit recurrence tm <- skip_voxel(tm) does not depend on what is probed either, so
				// the group now runs IT ahead (lane k keeps the position after k more misses) and probes the candidates at
				// once; the first occupied one (or the end of the ray) ends the run.  Positions before it emit nothing.
				bool done = false;
				while (true) {
					float my = tm, cur = tm;
					for (int k = 1; k < kMissAhead; ++k) {
						const f3 pc = {__fmaf_rn(cur, dir.x, o.x), __fmaf_rn(cur, dir.y, o.y), __fmaf_rn(cur, dir.z, o.z)};
						cur = skip_voxel<POW2>(g, cur, dt_min, pc, dir, inv);
						if ((int)lane >= k) my = cur;
					}
					const bool inm = (int)lane < kMissAhead && my < far;
					const f3 pm = {__fmaf_rn(my, dir.x, o.x), __fmaf_rn(my, dir.y, o.y), __fmaf_rn(my, dir.z, o.z)};
					int cm2 = -1;
					const bool hitm = inm && probe<POW2>(g, pm, &cm2);
					const float nxt = skip_voxel<POW2>(g, my, dt_min, pm, dir, inv);       // lane kMissAhead-1: where the run goes on
					const unsigned long long stop = group_ballot<G>((int)lane < kMissAhead && (hitm || !inm));
					if (stop == 0ull) { tm = __shfl(nxt, kMissAhead - 1, G); continue; }
					tm = __shfl(my, __ffsll((long long)stop) - 1, G);
					done = !(tm < far);
					break;
				}
				if (done) break;
				miss_run = 0;
			}
			const float dt = calc_dt(tm, dt_gamma, dt_min, dt_max);
			t0 = tm - dt * 0.5f;
			t1 = tm + dt * 0.5f;
			len = kShort;
		} else {
			const unsigned long long hm = group_ballot<G>(hit);
			const uint32_t before = (uint32_t)__popcll(hm & below);
			const bool exec = in && (j + before < max_steps);
			if (exec && hit) {
				uint32_t *c = cache_ray + (size_t)(j + before) * 3;
				c[0] = __float_as_uint(m0); c[1] = __float_as_uint(m1); c[2] = (uint32_t)(cell + grid_offset);
			}
			const unsigned long long em = group_ballot<G>(exec);
			j += (uint32_t)__popcll(hm & em);
			if (em != ((G == 64) ? ~0ull : ((1ull << (G & 63)) - 1ull))) break;
			t0 = n0; t1 = n1; tm = nm;
		}
	}
	return j;
}

template <int G>
__global__ __launch_bounds__(256) void k_march_group(uint32_t n_rays, const float *__restrict__ rays_o,
                                                     const float *__restrict__ rays_d, const float *__restrict__ t_min,
                                                     const float *__restrict__ t_max, const float *__restrict__ roi, int rx,
                                                     int ry, int rz, const uint8_t *__restrict__ cells, int type,
                                                     float step_size, float max_step_size, float dt_gamma, uint32_t max_steps,
                                                     int batched, const int32_t *__restrict__ batch_inds,
                                                     uint32_t batch_data_size, int32_t *__restrict__ counts,
                                                     uint32_t *__restrict__ cache) {
	const uint32_t i = (blockIdx.x * 256u + threadIdx.x) / G, lane = threadIdx.x % G;
	if (i >= n_rays) return;
	uint32_t b = 0;
	if (batched) {
		if (batch_inds) {
			const int32_t v = batch_inds[i];
			if (v < 0) { if (lane == 0) counts[i] = 0; return; }
			b = (uint32_t)v;
		} else if (batch_data_size) {
			b = i / batch_data_size;
		}
	}
	const uint32_t vol = (uint32_t)(rx * ry * rz);
	Grid g;
	const float *r6 = roi + 6 * (size_t)b;
	g.mn = {r6[0], r6[1], r6[2]};
	g.mx = {r6[3], r6[4], r6[5]};
	g.rx = rx; g.ry = ry; g.rz = rz;
	g.cells = cells + (size_t)b * vol;
	g.type = type;
	const int32_t grid_offset = batched ? (int32_t)(b * vol) : 0;
	const f3 o = {rays_o[3 * (size_t)i], rays_o[3 * (size_t)i + 1], rays_o[3 * (size_t)i + 2]};
	const f3 dir = {rays_d[3 * (size_t)i], rays_d[3 * (size_t)i + 1], rays_d[3 * (size_t)i + 2]};
	const f3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	g.inv_ext = {1.0f / (g.mx.x - g.mn.x), 1.0f / (g.mx.y - g.mn.y), 1.0f / (g.mx.z - g.mn.z)};
	g.inv_res = {1.0f / (float)rx, 1.0f / (float)ry, 1.0f / (float)rz};
	const bool pow2 = is_pow2f(g.mx.x - g.mn.x) && is_pow2f(g.mx.y - g.mn.y) && is_pow2f(g.mx.z - g.mn.z) &&
	                  is_pow2f((float)rx) && is_pow2f((float)ry) && is_pow2f((float)rz);
	uint32_t *cache_ray = cache + (size_t)i * max_steps * 3;
	uint32_t j;
	if (__all(pow2))
		j = march_ray_group<G, true>(g, lane, grid_offset, o, dir, inv, t_min[i], t_max[i], step_size, max_step_size, dt_gamma,
		                             max_steps, cache_ray);
	else
		j = march_ray_group<G, false>(g, lane, grid_offset, o, dir, inv, t_min[i], t_max[i], step_size, max_step_size, dt_gamma,
		                              max_steps, cache_ray);
	if (lane == 0) counts[i] = (int32_t)j;
}

// one wave per ray: copy the ray's cached samples to their packed position
__global__ __launch_bounds__(256) void k_emit_cached(uint32_t n_rays, uint32_t stride, int batched,
                                                     const int32_t *__restrict__ batch_inds, uint32_t batch_data_size,
                                                     const int32_t *__restrict__ packed_info,
                                                     const uint32_t *__restrict__ cache, float *__restrict__ t_starts,
                                                     float *__restrict__ t_ends, int32_t *__restrict__ ridx,
                                                     int32_t *__restrict__ bidx, int32_t *__restrict__ gidx,
                                                     const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                     int64_t *__restrict__ ridx64, float *__restrict__ deltas,
                                                     float *__restrict__ samples) {
	const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (i >= n_rays) return;
	const uint32_t base = (uint32_t)packed_info[2 * (size_t)i], cnt = (uint32_t)packed_info[2 * (size_t)i + 1];
	int32_t b = 0;
	if (batched) b = batch_inds ? batch_inds[i] : (batch_data_size ? (int32_t)(i / batch_data_size) : 0);
	const uint32_t *c = cache + (size_t)i * stride * 3;
	float o[3] = {0.0f, 0.0f, 0.0f}, d[3] = {0.0f, 0.0f, 0.0f};
	if (samples) {
#pragma unroll
		for (int k = 0; k < 3; ++k) { o[k] = rays_o[3 * (size_t)i + k]; d[k] = rays_d[3 * (size_t)i + k]; }
	}
	for (uint32_t j = lane; j < cnt; j += 64) {
		const float a = __uint_as_float(c[3 * j]), e = __uint_as_float(c[3 * j + 1]);
		t_starts[base + j] = a;
		t_ends[base + j] = e;
		ridx[base + j] = (int32_t)i;
		if (bidx) bidx[base + j] = b;
		if (gidx) gidx[base + j] = (int32_t)c[3 * j + 2];
		// the per-sample epilogue of nr3d_march_finish_samples (ray_glue.hip: the same expressions), when asked for
		if (ridx64) ridx64[base + j] = (int64_t)i;
		if (deltas) deltas[base + j] = e - a;
		if (samples) {
#pragma unroll
			for (int k = 0; k < 3; ++k) samples[(size_t)(base + j) * 3 + k] = __fmaf_rn(d[k], a, o[k]);
		}
	}
}

// ---------------------------------------------------------------------------------------------------
// Forest of occupancy grids (forest_marching.cu:16-143): the ray's block segments (block, entry, exit) are walked in
// order; inside a segment the march is the single-grid one against the block's own grid and ROI
// (world_origin + k * world_block_size, one fused multiply-add as nvcc contracts it).  No ROI test on the probe
// (block_grid_occupied_at :16-25) and one unconditional `t_mid += step_size` at every segment start (:99-101).
// ---------------------------------------------------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(kBlock) void k_forest_march(const int16_t *__restrict__ block_ks, f3 world_origin,
                                                         f3 world_block_size, uint32_t n_rays,
                                                         const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                         const float *__restrict__ t_min, const float *__restrict__ t_max,
                                                         const int32_t *__restrict__ seg_block_inds,
                                                         const float *__restrict__ seg_entries,
                                                         const float *__restrict__ seg_exits,
                                                         const int32_t *__restrict__ seg_pack_infos, int rx, int ry, int rz,
                                                         const uint8_t *__restrict__ cells, float step_size,
                                                         float max_step_size, float dt_gamma, uint32_t max_steps,
                                                         const int32_t *__restrict__ packed_info,
                                                         int32_t *__restrict__ counts, float *__restrict__ t_starts,
                                                         float *__restrict__ t_ends, int32_t *__restrict__ ridx,
                                                         int32_t *__restrict__ blidx, int32_t *__restrict__ gidx) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n_rays) return;
	const uint32_t seg_begin = (uint32_t)seg_pack_infos[2 * (size_t)i], seg_len = (uint32_t)seg_pack_infos[2 * (size_t)i + 1];
	const uint32_t vol = (uint32_t)(rx * ry * rz);
	uint32_t base = 0;
	if (EMIT) {
		base = (uint32_t)packed_info[2 * (size_t)i];
		max_steps = (uint32_t)packed_info[2 * (size_t)i + 1];
	}
	const f3 o = {rays_o[3 * (size_t)i], rays_o[3 * (size_t)i + 1], rays_o[3 * (size_t)i + 2]};
	const f3 dir = {rays_d[3 * (size_t)i], rays_d[3 * (size_t)i + 1], rays_d[3 * (size_t)i + 2]};
	const f3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	const float near = t_min[i], far = t_max[i];
	const float dt_min = step_size, dt_max = max_step_size;
	Grid g;
	g.rx = rx; g.ry = ry; g.rz = rz;
	g.type = NR3D_CONTRACT_AABB;
	g.inv_ext = {0.0f, 0.0f, 0.0f};
	g.inv_res = {0.0f, 0.0f, 0.0f};

	uint32_t j = 0;
	float t0 = near;
	float dt = calc_dt(t0, dt_gamma, dt_min, dt_max);
	float t1 = t0 + dt;
	float tm = (t0 + t1) * 0.5f;
	for (uint32_t s = 0; s < seg_len; ++s) {
		const float entry = seg_entries[seg_begin + s], exit = seg_exits[seg_begin + s];
		const uint32_t b = (uint32_t)seg_block_inds[seg_begin + s];
		const int16_t *k = block_ks + 3 * (size_t)b;
		g.mn = {__fmaf_rn((float)k[0], world_block_size.x, world_origin.x), __fmaf_rn((float)k[1], world_block_size.y, world_origin.y),
		        __fmaf_rn((float)k[2], world_block_size.z, world_origin.z)};
		g.mx = {g.mn.x + world_block_size.x, g.mn.y + world_block_size.y, g.mn.z + world_block_size.z};
		const uint32_t grid_offset = b * vol;
		g.cells = cells + grid_offset;
		if (entry >= far || exit <= near) break;
		do { tm += step_size; } while (tm < entry);
		dt = calc_dt(tm, dt_gamma, dt_min, dt_max);
		t0 = tm - dt * 0.5f;
		t1 = tm + dt * 0.5f;
		while (tm <= exit && tm <= far && j < max_steps) {
			const f3 p = {__fmaf_rn(tm, dir.x, o.x), __fmaf_rn(tm, dir.y, o.y), __fmaf_rn(tm, dir.z, o.z)};
			const f3 u = to_unit<false>(g, p);
			const int ix = clampi((int)(u.x * (float)rx), 0, rx - 1);
			const int iy = clampi((int)(u.y * (float)ry), 0, ry - 1);
			const int iz = clampi((int)(u.z * (float)rz), 0, rz - 1);
			const int cell = ix * (ry * rz) + iy * rz + iz;
			if (g.cells[cell] != 0) {
				if (EMIT) {
					t_starts[base + j] = t0;
					t_ends[base + j] = t1;
					ridx[base + j] = (int32_t)i;
					blidx[base + j] = (int32_t)b;
					if (gidx) gidx[base + j] = cell + (int32_t)grid_offset;
				}
				++j;
				t0 = t1;
				t1 = t0 + calc_dt(t0, dt_gamma, dt_min, dt_max);
				tm = (t0 + t1) * 0.5f;
			} else {
				tm = skip_voxel<false>(g, tm, dt_min, p, dir, inv);
				dt = calc_dt(tm, dt_gamma, dt_min, dt_max);
				t0 = tm - dt * 0.5f;
				t1 = tm + dt * 0.5f;
			}
		}
	}
	if (!EMIT) counts[i] = (int32_t)j;
}

// ---------------------------------------------------------------------------------------------------
// Occupancy-value grid maintenance (the producer side of the marcher's input):
//   new[v] = max(ema_decay * old[v], max over the samples that fall into voxel v)   for touched voxels, old[v] otherwise
// (update_occ_val_grid[_idx]_ / update_batched_*, nr3d_lib/models/accelerations/occgrid/utils.py:80-125, there built on
// torch_scatter.scatter_max with `out = ema_decay * grid`).  Two kernels so that a sharded caller can all-reduce(MAX)
// the per-voxel sample maxima before the decay is applied once.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_max_f32(float *addr, float v) {
	// order-preserving integer views: non-negative floats compare like signed ints, negative ones inversely as uints.
	// -0.0f (sign bit only = INT_MIN as an int) would lose against the -inf sentinel: canonicalise it to +0.0f
	v += 0.0f;
	if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
	else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void k_occ_scatter_max(uint64_t n, const int64_t *__restrict__ gidx,
                                                         const float *__restrict__ pts, const int64_t *__restrict__ bidx,
                                                         uint64_t per_batch, const float *__restrict__ val, int rx, int ry,
                                                         int rz, uint64_t n_batches, float *__restrict__ vmax) {
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	int64_t ix, iy, iz;
	if (gidx) {
		ix = gidx[3 * i]; iy = gidx[3 * i + 1]; iz = gidx[3 * i + 2];
	} else {
		// ((pts / 2 + 0.5) * resolution).long().clamp(0, resolution - 1): fp32, truncation toward zero
		ix = (int64_t)((pts[3 * i] / 2.0f + 0.5f) * (float)rx);
		iy = (int64_t)((pts[3 * i + 1] / 2.0f + 0.5f) * (float)ry);
		iz = (int64_t)((pts[3 * i + 2] / 2.0f + 0.5f) * (float)rz);
		ix = max((int64_t)0, min(ix, (int64_t)rx - 1));
		iy = max((int64_t)0, min(iy, (int64_t)ry - 1));
		iz = max((int64_t)0, min(iz, (int64_t)rz - 1));
	}
	const uint64_t vol = (uint64_t)rx * ry * rz;
	const int64_t bs = bidx ? bidx[i] : (int64_t)(per_batch ? i / per_batch : 0);
	// caller-supplied voxel / batch indices outside the grid are dropped (the reference's index_put_ / scatter_max would
	// raise; a raw kernel must not write out of bounds)
	if (ix < 0 || iy < 0 || iz < 0 || ix >= rx || iy >= ry || iz >= rz || bs < 0 || (uint64_t)bs >= n_batches) return;
	const uint64_t b = (uint64_t)bs;
	atomic_max_f32(vmax + b * vol + (uint64_t)ix * ((uint64_t)ry * rz) + (uint64_t)iy * rz + (uint64_t)iz, val[i]);
}

__global__ __launch_bounds__(256) void k_occ_apply_max(uint64_t n_voxels, float ema_decay, const float *__restrict__ vmax,
                                                       float *__restrict__ grid) {
	const uint64_t v = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (v >= n_voxels) return;
	const float m = vmax[v];
	if (__float_as_uint(m) != 0xFF800000u) grid[v] = fmaxf(ema_decay * grid[v], m);
}

}  // namespace occ
}  // namespace nr3d

using namespace nr3d;

extern "C" int nr3d_occ_scatter_max(uint64_t n, const int64_t *gidx, const float *pts, const int64_t *bidx,
                                    uint64_t per_batch, const float *occ_val, const int32_t grid_res[3], uint32_t n_batches,
                                    float *vmax, void *stream) {
	NR3D_CHECK(vmax != nullptr, "occ_scatter_max: NULL scratch grid");
	NR3D_CHECK(n == 0 || (gidx != nullptr) != (pts != nullptr), "occ_scatter_max: pass exactly one of gidx / pts");
	const uint64_t total = (uint64_t)grid_res[0] * grid_res[1] * grid_res[2] * (n_batches ? n_batches : 1);
	NR3D_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)vmax, (int)0xFF800000u, total, (hipStream_t)stream));   // -inf
	if (n == 0) return 0;
	NR3D_CHECK(occ_val != nullptr, "occ_scatter_max: NULL occ_val");
	hipLaunchKernelGGL(occ::k_occ_scatter_max, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, gidx, pts, bidx,
	                   per_batch, occ_val, grid_res[0], grid_res[1], grid_res[2], (uint64_t)(n_batches ? n_batches : 1), vmax);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_occ_apply_max(uint64_t n_voxels, float ema_decay, const float *vmax, float *occ_val_grid, void *stream) {
	if (n_voxels == 0) return 0;
	NR3D_CHECK(vmax && occ_val_grid, "occ_apply_max: NULL pointer");
	hipLaunchKernelGGL(occ::k_occ_apply_max, dim3(div_up(n_voxels, 256)), dim3(256), 0, (hipStream_t)stream, n_voxels, ema_decay,
	                   vmax, occ_val_grid);
	NR3D_LAUNCH_CHECK();
	return 0;
}

// scratch layout: [ int32 counts[n] (padded to 8 B) | tile sums ]
extern "C" uint64_t nr3d_scan_tmp_bytes(uint64_t n) { return ((n * sizeof(int64_t) + 7) / 8) * 8 + scan::tmp_bytes(n); }

extern "C" uint64_t nr3d_ray_marching_cache_bytes(uint32_t n_rays, uint32_t max_steps) {
	return (uint64_t)n_rays * max_steps * 12u;
}

// count pass + the scan of the counts.  ridx_hit == NULL: packed_info and total_steps[0] (nr3d_ray_marching_count); else the
// same scan also compacts the rays that got samples (nr3d_march_finish_rays' outputs; total_steps is then {S, n_hit})
static int march_count(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                       const float *t_max, const float *roi, const int32_t grid_res[3],
                       const uint8_t *grid_binary, int type, float step_size, float max_step_size,
                       float dt_gamma, uint32_t max_steps, int batched, const int32_t *batch_inds,
                       uint32_t batch_data_size, int32_t *packed_info, int64_t *total_steps,
                       void *scan_tmp, void *sample_cache, uint64_t sample_cache_bytes, int64_t *ridx_hit,
                       int64_t *pack_infos64, void *stream) {
	NR3D_CHECK(total_steps && scan_tmp, "ray_marching: NULL scratch pointer");
	hipStream_t st = (hipStream_t)stream;
	if (n_rays == 0) { NR3D_HIP_CHECK(hipMemsetAsync(total_steps, 0, (ridx_hit ? 2 : 1) * sizeof(int64_t), st)); return 0; }
	NR3D_CHECK(rays_o && rays_d && t_min && t_max && roi && grid_binary && packed_info, "ray_marching: NULL tensor pointer");
	NR3D_CHECK(type >= 0 && type <= 2, "ray_marching: invalid contraction type %d", type);
	int32_t *counts = (int32_t *)scan_tmp;
	void *tiles = (char *)scan_tmp + (((uint64_t)n_rays * sizeof(int64_t) + 7) / 8) * 8;
	const bool cached = sample_cache && sample_cache_bytes >= nr3d_ray_marching_cache_bytes(n_rays, max_steps);
	auto launch = [&](auto kern) {
		hipLaunchKernelGGL(kern, dim3(div_up(n_rays, occ::kBlock)), dim3(occ::kBlock), 0, st, n_rays, rays_o, rays_d, t_min,
		                   t_max, roi, grid_res[0], grid_res[1], grid_res[2], grid_binary, type, step_size, max_step_size,
		                   dt_gamma, max_steps, batched, batch_inds, batch_data_size, (const int32_t *)nullptr, counts,
		                   (float *)nullptr, (float *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
		                   (uint32_t *)sample_cache);
	};
	// lanes per ray: 32 / 16 while that still leaves the chip short of work, else one (NR3D_OPT_MARCH_GROUP = 1 | 16 | 32 | 64 forces)
	int group = n_rays <= 8192u ? 32 : (n_rays <= 32768u ? 16 : 1);
	{ const int64_t v = opt::get(NR3D_OPT_MARCH_GROUP); if (v == 1 || v == 16 || v == 32 || v == 64) group = (int)v; }
	if (!cached) group = 1;
	auto launch_group = [&](auto kern, int G) {
		hipLaunchKernelGGL(kern, dim3(div_up((uint64_t)n_rays * G, 256)), dim3(256), 0, st, n_rays, rays_o, rays_d, t_min, t_max, roi,
		                   grid_res[0], grid_res[1], grid_res[2], grid_binary, type, step_size, max_step_size, dt_gamma, max_steps,
		                   batched, batch_inds, batch_data_size, counts, (uint32_t *)sample_cache);
	};
	{
		prof::Scope ps(NR3D_PROF_MARCH, st);
		if (group == 64) launch_group(occ::k_march_group<64>, 64);
		else if (group == 32) launch_group(occ::k_march_group<32>, 32);
		else if (group == 16) launch_group(occ::k_march_group<16>, 16);
		else if (cached) launch(occ::k_march<false, true>);
		else launch(occ::k_march<false, false>);
	}
	NR3D_LAUNCH_CHECK();
	if (ridx_hit) {
		NR3D_CHECK(pack_infos64 != nullptr, "ray_marching_count_finished: NULL output pointer");
		glue::PackWriter<int32_t, 1> w{counts, nullptr, nullptr, ridx_hit, pack_infos64, packed_info};
		return glue::compact_packs<int32_t, 1>(n_rays, w, total_steps, tiles, st);
	}
	return scan::pack_infos_from_counts<int32_t, int32_t>(n_rays, counts, packed_info, total_steps, tiles, st);
}

extern "C" int nr3d_ray_marching_count(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                                       const float *t_max, const float *roi, const int32_t grid_res[3],
                                       const uint8_t *grid_binary, int type, float step_size, float max_step_size,
                                       float dt_gamma, uint32_t max_steps, int batched, const int32_t *batch_inds,
                                       uint32_t batch_data_size, int32_t *packed_info, int64_t *total_steps,
                                       void *scan_tmp, void *sample_cache, uint64_t sample_cache_bytes,
                                       void *stream) {
	return march_count(n_rays, rays_o, rays_d, t_min, t_max, roi, grid_res, grid_binary, type, step_size, max_step_size, dt_gamma,
	                   max_steps, batched, batch_inds, batch_data_size, packed_info, total_steps, scan_tmp, sample_cache,
	                   sample_cache_bytes, nullptr, nullptr, stream);
}

extern "C" int nr3d_ray_marching_count_finished(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                                                const float *t_max, const float *roi, const int32_t grid_res[3],
                                                const uint8_t *grid_binary, int type, float step_size, float max_step_size,
                                                float dt_gamma, uint32_t max_steps, int batched, const int32_t *batch_inds,
                                                uint32_t batch_data_size, int32_t *packed_info, int64_t *ridx_hit,
                                                int64_t *pack_infos, int64_t *totals, void *scan_tmp, void *sample_cache,
                                                uint64_t sample_cache_bytes, void *stream) {
	NR3D_CHECK(n_rays == 0 || (ridx_hit && pack_infos), "ray_marching_count_finished: NULL output pointer");
	static int64_t dummy;       // n_rays == 0: only the totals are touched
	return march_count(n_rays, rays_o, rays_d, t_min, t_max, roi, grid_res, grid_binary, type, step_size, max_step_size, dt_gamma,
	                   max_steps, batched, batch_inds, batch_data_size, packed_info, totals, scan_tmp, sample_cache,
	                   sample_cache_bytes, ridx_hit ? ridx_hit : &dummy, pack_infos, stream);
}

extern "C" int nr3d_ray_marching_emit(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                                      const float *t_max, const float *roi, const int32_t grid_res[3],
                                      const uint8_t *grid_binary, int type, float step_size, float max_step_size,
                                      float dt_gamma, int batched, const int32_t *batch_inds, uint32_t batch_data_size,
                                      const int32_t *packed_info, float *t_starts, float *t_ends, int32_t *ridx,
                                      int32_t *bidx, int32_t *gidx, const void *sample_cache,
                                      uint32_t cache_max_steps, void *stream) {
	if (n_rays == 0) return 0;
	NR3D_CHECK(rays_o && rays_d && t_min && t_max && roi && grid_binary && packed_info, "ray_marching: NULL tensor pointer");
	NR3D_CHECK(t_starts && t_ends && ridx, "ray_marching: NULL output pointer");
	prof::Scope ps(NR3D_PROF_MARCH, (hipStream_t)stream);
	if (sample_cache) {   // filled by nr3d_ray_marching_count with the same rays and max_steps == cache_max_steps
		hipLaunchKernelGGL(occ::k_emit_cached, dim3(div_up(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, n_rays,
		                   cache_max_steps, batched, batch_inds, batch_data_size, packed_info, (const uint32_t *)sample_cache,
		                   t_starts, t_ends, ridx, bidx, gidx, (const float *)nullptr, (const float *)nullptr, (int64_t *)nullptr,
		                   (float *)nullptr, (float *)nullptr);
		NR3D_LAUNCH_CHECK();
		return 0;
	}
	hipLaunchKernelGGL((occ::k_march<true, false>), dim3(div_up(n_rays, occ::kBlock)), dim3(occ::kBlock), 0, (hipStream_t)stream,
	                   n_rays, rays_o, rays_d, t_min, t_max, roi, grid_res[0], grid_res[1], grid_res[2], grid_binary,
	                   type, step_size, max_step_size, dt_gamma, 0u, batched, batch_inds, batch_data_size, packed_info,
	                   (int32_t *)nullptr, t_starts, t_ends, ridx, bidx, gidx, (uint32_t *)nullptr);
	NR3D_LAUNCH_CHECK();
	return 0;
}

extern "C" int nr3d_ray_marching_emit_finished(uint32_t n_rays, const float *rays_o, const float *rays_d, int batched,
                                               const int32_t *batch_inds, uint32_t batch_data_size, const int32_t *packed_info,
                                               const void *sample_cache, uint32_t cache_max_steps, float *t_starts,
                                               float *t_ends, int32_t *ridx, int32_t *bidx, int32_t *gidx, int64_t *ridx64,
                                               float *deltas, float *samples, void *stream) {
	if (n_rays == 0) return 0;
	NR3D_CHECK(packed_info && sample_cache && t_starts && t_ends && ridx, "ray_marching_emit_finished: NULL tensor pointer");
	NR3D_CHECK(!samples || (rays_o && rays_d), "ray_marching_emit_finished: samples need rays_o / rays_d");
	prof::Scope ps(NR3D_PROF_MARCH, (hipStream_t)stream);
	hipLaunchKernelGGL(occ::k_emit_cached, dim3(div_up(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, n_rays, cache_max_steps,
	                   batched, batch_inds, batch_data_size, packed_info, (const uint32_t *)sample_cache, t_starts, t_ends, ridx,
	                   bidx, gidx, rays_o, rays_d, ridx64, deltas, samples);
	NR3D_LAUNCH_CHECK();
	return 0;
}

// ---------------------------------------------------------------------------------------------------
// configs[2] in one call (round 6): count (+ sample cache) -> scan (+ hit-ray compaction, totals) -> [cached emit (+ per-sample
// epilogue) + alpha = 1 - exp(-sigma * delta) + alpha composite in ONE launch], enqueued back to back on `stream`.  Nothing here waits for the
// device: every per-sample buffer has the capacity of the bound n_rays * max_steps, and the composite runs one wave per ray over ALL
// rays (rays without samples write their own zeros), so neither S nor n_hit is needed on the host before the last launch.  The
// caller reads `totals` back AFTER this returns (and after enqueueing the backward, if it wants one) and uses it only to slice
// views.  At 4096 rays the two-phase path spent ~80 us of host time behind the count readback (profiles/r05final_*).
// ---------------------------------------------------------------------------------------------------
extern "C" int nr3d_march_composite_fwd(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                                        const float *t_max, const float *roi, const int32_t grid_res[3],
                                        const uint8_t *grid_binary, int type, float step_size, float max_step_size,
                                        float dt_gamma, uint32_t max_steps, int32_t *packed_info, int64_t *ridx_hit,
                                        int64_t *pack_infos, int64_t *totals, void *scan_tmp, void *sample_cache,
                                        uint64_t sample_cache_bytes, uint64_t rows, float *t_starts, float *t_ends, int32_t *ridx,
                                        int32_t *gidx, int64_t *ridx64, float *deltas, float *samples, const float *sigma,
                                        uint64_t sigma_rows, const float *rgb, float early_stop_eps, float alpha_thre,
                                        int normalize_depth, float *alphas, float *vw, float *mask, float *depth, float *rgb_out,
                                        void *stream) {
	NR3D_CHECK(totals != nullptr, "march_composite_fwd: NULL totals");
	hipStream_t st = (hipStream_t)stream;
	if (n_rays == 0) { NR3D_HIP_CHECK(hipMemsetAsync(totals, 0, 2 * sizeof(int64_t), st)); return 0; }
	NR3D_CHECK(rows >= (uint64_t)n_rays * max_steps, "march_composite_fwd: per-sample buffers hold %llu rows, the bound n_rays * max_steps is %llu",
	           (unsigned long long)rows, (unsigned long long)n_rays * max_steps);
	NR3D_CHECK(sample_cache && sample_cache_bytes >= nr3d_ray_marching_cache_bytes(n_rays, max_steps),
	           "march_composite_fwd: needs the sample cache (nr3d_ray_marching_cache_bytes)");
	NR3D_CHECK(packed_info && ridx_hit && pack_infos && t_starts && t_ends && ridx && deltas, "march_composite_fwd: NULL march output");
	NR3D_CHECK(sigma && alphas && vw && mask && depth && (!rgb || rgb_out), "march_composite_fwd: NULL composite tensor");
	if (int rc = march_count(n_rays, rays_o, rays_d, t_min, t_max, roi, grid_res, grid_binary, type, step_size, max_step_size, dt_gamma,
	                         max_steps, 0, nullptr, 0u, packed_info, totals, scan_tmp, sample_cache, sample_cache_bytes, ridx_hit,
	                         pack_infos, stream))
		return rc;
	return pk::launch_emit_composite_rays_fwd(n_rays, packed_info, sample_cache, max_steps, rays_o, rays_d, t_starts, t_ends, ridx, gidx, ridx64,
	                                          deltas, samples, sigma, sigma_rows, rgb, early_stop_eps, alpha_thre, normalize_depth, alphas, vw,
	                                          mask, depth, rgb_out, st);
}

extern "C" int nr3d_march_composite_bwd(uint32_t n_rays, const int32_t *packed_info, const float *alphas, const float *vw,
                                        const float *t, const float *rgb, float early_stop_eps, float alpha_thre,
                                        int normalize_depth, const float *mask, const float *depth, const float *g_mask,
                                        const float *g_depth, const float *g_rgb, float *grad_alphas, float *grad_t,
                                        float *grad_rgb, const float *sigma, const float *deltas, uint64_t sigma_rows,
                                        float *grad_sigma, void *stream) {
	if (n_rays == 0) return 0;
	NR3D_CHECK(packed_info && alphas && vw && t && mask && depth && grad_alphas, "march_composite_bwd: NULL pointer");
	NR3D_CHECK(!grad_sigma || (sigma && deltas), "march_composite_bwd: grad_sigma needs sigma and deltas");
	return pk::launch_composite_rays_bwd(n_rays, packed_info, alphas, vw, t, rgb, early_stop_eps, alpha_thre, normalize_depth, mask,
	                                     depth, g_mask, g_depth, g_rgb, grad_alphas, grad_t, grad_rgb, sigma, deltas, sigma_rows,
	                                     grad_sigma, (hipStream_t)stream);
}

static int forest_march_check(const nr3d_forest_meta_t *forest, const void *a, const void *b, const void *c, const void *d,
                              const void *e, const void *f, const void *g, const void *h, const void *grid) {
	NR3D_CHECK(forest && forest->block_ks, "forest_ray_marching: forest meta or block_ks is NULL");
	NR3D_CHECK(a && b && c && d && h && grid, "forest_ray_marching: NULL tensor pointer");
	(void)e; (void)f; (void)g;       // segment arrays may be empty (NULL) when no ray crosses a block
	return 0;
}

extern "C" int nr3d_forest_ray_marching_count(const nr3d_forest_meta_t *forest, uint32_t n_rays, const float *rays_o,
                                              const float *rays_d, const float *t_min, const float *t_max,
                                              const int32_t *seg_block_inds, const float *seg_entries,
                                              const float *seg_exits, const int32_t *seg_pack_infos,
                                              const int32_t grid_res[3], const uint8_t *grid_binary, float step_size,
                                              float max_step_size, float dt_gamma, uint32_t max_steps,
                                              int32_t *packed_info, int64_t *total_steps, void *scan_tmp, void *stream) {
	NR3D_CHECK(total_steps && scan_tmp, "forest_ray_marching: NULL scratch pointer");
	hipStream_t st = (hipStream_t)stream;
	if (n_rays == 0) { NR3D_HIP_CHECK(hipMemsetAsync(total_steps, 0, sizeof(int64_t), st)); return 0; }
	if (int rc = forest_march_check(forest, rays_o, rays_d, t_min, t_max, seg_block_inds, seg_entries, seg_exits,
	                                seg_pack_infos, grid_binary)) return rc;
	NR3D_CHECK(packed_info, "forest_ray_marching: NULL packed_info");
	int32_t *counts = (int32_t *)scan_tmp;
	void *tiles = (char *)scan_tmp + (((uint64_t)n_rays * sizeof(int64_t) + 7) / 8) * 8;
	const occ::f3 wo = {forest->world_origin[0], forest->world_origin[1], forest->world_origin[2]};
	const occ::f3 wb = {forest->world_block_size[0], forest->world_block_size[1], forest->world_block_size[2]};
	hipLaunchKernelGGL((occ::k_forest_march<false>), dim3(div_up(n_rays, occ::kBlock)), dim3(occ::kBlock), 0, st, forest->block_ks,
	                   wo, wb, n_rays, rays_o, rays_d, t_min, t_max, seg_block_inds, seg_entries, seg_exits, seg_pack_infos,
	                   grid_res[0], grid_res[1], grid_res[2], grid_binary, step_size, max_step_size, dt_gamma, max_steps,
	                   (const int32_t *)nullptr, counts, (float *)nullptr, (float *)nullptr, (int32_t *)nullptr,
	                   (int32_t *)nullptr, (int32_t *)nullptr);
	NR3D_LAUNCH_CHECK();
	return scan::pack_infos_from_counts<int32_t, int32_t>(n_rays, counts, packed_info, total_steps, tiles, st);
}

extern "C" int nr3d_forest_ray_marching_emit(const nr3d_forest_meta_t *forest, uint32_t n_rays, const float *rays_o,
                                             const float *rays_d, const float *t_min, const float *t_max,
                                             const int32_t *seg_block_inds, const float *seg_entries,
                                             const float *seg_exits, const int32_t *seg_pack_infos,
                                             const int32_t grid_res[3], const uint8_t *grid_binary, float step_size,
                                             float max_step_size, float dt_gamma, const int32_t *packed_info,
                                             float *t_starts, float *t_ends, int32_t *ridx, int32_t *blidx, int32_t *gidx,
                                             void *stream) {
	if (n_rays == 0) return 0;
	if (int rc = forest_march_check(forest, rays_o, rays_d, t_min, t_max, seg_block_inds, seg_entries, seg_exits,
	                                seg_pack_infos, grid_binary)) return rc;
	NR3D_CHECK(packed_info && t_starts && t_ends && ridx && blidx, "forest_ray_marching: NULL output pointer");
	const occ::f3 wo = {forest->world_origin[0], forest->world_origin[1], forest->world_origin[2]};
	const occ::f3 wb = {forest->world_block_size[0], forest->world_block_size[1], forest->world_block_size[2]};
	hipLaunchKernelGGL((occ::k_forest_march<true>), dim3(div_up(n_rays, occ::kBlock)), dim3(occ::kBlock), 0, (hipStream_t)stream,
	                   forest->block_ks, wo, wb, n_rays, rays_o, rays_d, t_min, t_max, seg_block_inds, seg_entries, seg_exits,
	                   seg_pack_infos, grid_res[0], grid_res[1], grid_res[2], grid_binary, step_size, max_step_size, dt_gamma,
	                   0u, packed_info, (int32_t *)nullptr, t_starts, t_ends, ridx, blidx, gidx);
	NR3D_LAUNCH_CHECK();
	return 0;
}
