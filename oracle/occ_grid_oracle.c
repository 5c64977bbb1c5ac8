/* oracle/occ_grid_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C fp32 restatement of the reference occupancy-grid ray marcher
 *   csrc/occ_grid/src/ray_marching.cu:17-244          (single grid)
 *   csrc/occ_grid/src/batched_marching.cu:18-287      (batched grids)
 *   csrc/occ_grid/include/occ_grid/helpers_march.h:11-77
 *   csrc/occ_grid/include/occ_grid/helpers_contraction.h:10-125
 *   csrc/occ_grid/include/occ_grid/helpers_math.h:177-180 (int() truncation), :1167-1170 (clamp),
 *                                                 :1343 (floorf), :1471-1474 (sign = copysignf)
 *
 * FMA contract (nvcc --fmad=true): `origin + t_mid * dir` is fmaf(t_mid, dir, origin).  The other
 * mul+add sites in the marcher multiply by 0.5f / 2.0f (exact), so fusing does not change them;
 * they are written unfused.  Compiled with -ffp-contract=off.
 *
 * PARITY UNPINNED by the reference's own tests (it has no assertions for this path): this file is a
 * restatement from source only.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } f3;

static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
static inline int clampi(int f, int a, int b) { int m = f < b ? f : b; return a > m ? a : m; }

static inline float calc_dt(float t, float dt_gamma, float dt_min, float dt_max) {
	return clampf(t * dt_gamma, dt_min, dt_max);                 /* helpers_march.h:11-14 */
}

static inline f3 roi_to_unit(f3 p, f3 mn, f3 mx) {               /* helpers_contraction.h:10-15 */
	f3 r = {(p.x - mn.x) / (mx.x - mn.x), (p.y - mn.y) / (mx.y - mn.y), (p.z - mn.z) / (mx.z - mn.z)};
	return r;
}

static inline f3 apply_contraction(f3 p, f3 mn, f3 mx, int type) { /* helpers_contraction.h:93-108 */
	f3 u = roi_to_unit(p, mn, mx);
	if (type == 1) {             /* UN_BOUNDED_TANH :24-35 */
		u.x -= 0.5f; u.y -= 0.5f; u.z -= 0.5f;
		f3 r = {fmaf(tanhf(u.x), 0.5f, 0.5f), fmaf(tanhf(u.y), 0.5f, 0.5f), fmaf(tanhf(u.z), 0.5f, 0.5f)};
		return r;
	} else if (type == 2) {      /* UN_BOUNDED_SPHERE :56-73 */
		u.x = u.x * 2.0f - 1.0f; u.y = u.y * 2.0f - 1.0f; u.z = u.z * 2.0f - 1.0f;
		float norm_sq = fmaf(u.z, u.z, fmaf(u.y, u.y, u.x * u.x));
		float norm = sqrtf(norm_sq);
		if (norm > 1.0f) {
			float s = 2.0f - 1.0f / norm;
			u.x = s * (u.x / norm); u.y = s * (u.y / norm); u.z = s * (u.z / norm);
		}
		u.x = u.x * 0.25f + 0.5f; u.y = u.y * 0.25f + 0.5f; u.z = u.z * 0.25f + 0.5f;
		return u;
	}
	return u;                    /* AABB */
}

static inline int grid_idx_at(f3 u, const int *res) {            /* helpers_march.h:16-25 */
	int ix = (int)(u.x * (float)res[0]), iy = (int)(u.y * (float)res[1]), iz = (int)(u.z * (float)res[2]);
	ix = clampi(ix, 0, res[0] - 1); iy = clampi(iy, 0, res[1] - 1); iz = clampi(iz, 0, res[2] - 1);
	return ix * (res[1] * res[2]) + iy * res[2] + iz;
}

static inline int grid_occupied_at(f3 p, f3 mn, f3 mx, int type, const int *res,
                                   const uint8_t *grid, int *idx_out) { /* helpers_march.h:27-44 */
	if (type == 0 && (p.x < mn.x || p.x > mx.x || p.y < mn.y || p.y > mx.y || p.z < mn.z || p.z > mx.z))
		return 0;
	f3 u = apply_contraction(p, mn, mx, type);
	int idx = grid_idx_at(u, res);
	*idx_out = idx;
	return grid[idx] != 0;
}

static inline float dist_next_voxel(f3 p, f3 dir, f3 inv, f3 mn, f3 mx, const int *res) {
	/* helpers_march.h:47-56 */
	float rx = (float)res[0], ry = (float)res[1], rz = (float)res[2];
	f3 u = roi_to_unit(p, mn, mx);
	float qx = u.x * rx, qy = u.y * ry, qz = u.z * rz;
	float tx = ((floorf(qx + 0.5f + 0.5f * copysignf(1.0f, dir.x)) - qx) * inv.x) / rx * (mx.x - mn.x);
	float ty = ((floorf(qy + 0.5f + 0.5f * copysignf(1.0f, dir.y)) - qy) * inv.y) / ry * (mx.y - mn.y);
	float tz = ((floorf(qz + 0.5f + 0.5f * copysignf(1.0f, dir.z)) - qz) * inv.z) / rz * (mx.z - mn.z);
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t, 0.0f);
}

static inline float advance_next_voxel(float t, float dt_min, f3 p, f3 dir, f3 inv, f3 mn, f3 mx,
                                       const int *res) {         /* helpers_march.h:58-77 */
	float t_target = t + dist_next_voxel(p, dir, inv, mn, mx, res);
	float _t = t;
	do { _t += dt_min; } while (_t < t_target);
	return _t;
}

/* One ray; returns the number of emitted samples.  When t_starts != NULL writes (pass 2).
 * (ray_marching.cu:67-132 / batched_marching.cu:84-150) */
static uint32_t march_one(const float *o3, const float *d3, float near, float far, const float *roi,
                          const int *res, const uint8_t *grid, int type, float step_size,
                          float max_step_size, float dt_gamma, uint32_t max_steps, int32_t ray_id,
                          int32_t batch_ind, int32_t grid_offset, float *t_starts, float *t_ends,
                          int32_t *ridx, int32_t *bidx, int32_t *gidx, uint64_t *n_probes) {
	const f3 origin = {o3[0], o3[1], o3[2]};
	const f3 dir = {d3[0], d3[1], d3[2]};
	const f3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
	const f3 mn = {roi[0], roi[1], roi[2]}, mx = {roi[3], roi[4], roi[5]};
	const float dt_min = step_size, dt_max = max_step_size;
	uint32_t j = 0;
	float t0 = near;
	float dt = calc_dt(t0, dt_gamma, dt_min, dt_max);
	float t1 = t0 + dt;
	float t_mid = (t0 + t1) * 0.5f;
	while ((t_mid < far) && (j < max_steps)) {
		const f3 p = {fmaf(t_mid, dir.x, origin.x), fmaf(t_mid, dir.y, origin.y), fmaf(t_mid, dir.z, origin.z)};
		int gi = -1;
		if (n_probes) ++*n_probes;
		if (grid_occupied_at(p, mn, mx, type, res, grid, &gi)) {
			if (t_starts) {
				t_starts[j] = t0; t_ends[j] = t1; ridx[j] = ray_id;
				if (bidx) bidx[j] = batch_ind;
				if (gidx) gidx[j] = gi + grid_offset;
			}
			++j;
			t0 = t1;
			t1 = t0 + calc_dt(t0, dt_gamma, dt_min, dt_max);
			t_mid = (t0 + t1) * 0.5f;
		} else if (type == 0) {
			t_mid = advance_next_voxel(t_mid, dt_min, p, dir, inv, mn, mx, res);
			dt = calc_dt(t_mid, dt_gamma, dt_min, dt_max);
			t0 = t_mid - dt * 0.5f;
			t1 = t_mid + dt * 0.5f;
		} else {
			t0 = t1;
			t1 = t0 + calc_dt(t0, dt_gamma, dt_min, dt_max);
			t_mid = (t0 + t1) * 0.5f;
		}
	}
	return j;
}

/* Pass 1: per-ray counts (+ optional probe count for the roofline byte model).
 * batched: batch_inds (int32, <0 skips the ray: count forced to 0 here -- the reference leaves it
 * uninitialised, batched_marching.cu:55/:212) or batch_data_size; roi is [B,6]; grid is [B,Rx,Ry,Rz]. */
void orc_march_count(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                     const float *t_max, const float *roi, const int *res, const uint8_t *grid, int type,
                     float step_size, float max_step_size, float dt_gamma, uint32_t max_steps,
                     int batched, const int32_t *batch_inds, uint32_t batch_data_size,
                     int32_t *num_steps, uint64_t *n_probes_total) {
	uint64_t probes = 0;
	const uint32_t vol = (uint32_t)(res[0] * res[1] * res[2]);
	for (uint32_t i = 0; i < n_rays; ++i) {
		uint32_t b = 0;
		if (batched) {
			if (batch_inds) {
				if (batch_inds[i] < 0) { num_steps[i] = 0; continue; }
				b = (uint32_t)batch_inds[i];
			} else if (batch_data_size) b = i / batch_data_size;
		}
		num_steps[i] = (int32_t)march_one(rays_o + 3 * (size_t)i, rays_d + 3 * (size_t)i, t_min[i], t_max[i],
		                                  roi + 6 * (size_t)b, res, grid + (size_t)b * vol, type, step_size,
		                                  max_step_size, dt_gamma, max_steps, (int32_t)i, (int32_t)b, 0,
		                                  NULL, NULL, NULL, NULL, NULL, &probes);
	}
	if (n_probes_total) *n_probes_total = probes;
}

/* Pass 2: packed_info[i] = (base, count) int32 (host cumsum, ray_marching.cu:205-206). */
void orc_march_emit(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                    const float *t_max, const float *roi, const int *res, const uint8_t *grid, int type,
                    float step_size, float max_step_size, float dt_gamma,
                    int batched, const int32_t *batch_inds, uint32_t batch_data_size,
                    const int32_t *packed_info, float *t_starts, float *t_ends, int32_t *ridx,
                    int32_t *bidx, int32_t *gidx) {
	const uint32_t vol = (uint32_t)(res[0] * res[1] * res[2]);
	for (uint32_t i = 0; i < n_rays; ++i) {
		uint32_t b = 0;
		if (batched) {
			if (batch_inds) {
				if (batch_inds[i] < 0) continue;
				b = (uint32_t)batch_inds[i];
			} else if (batch_data_size) b = i / batch_data_size;
		}
		const uint32_t base = (uint32_t)packed_info[2 * i], cnt = (uint32_t)packed_info[2 * i + 1];
		march_one(rays_o + 3 * (size_t)i, rays_d + 3 * (size_t)i, t_min[i], t_max[i], roi + 6 * (size_t)b, res,
		          grid + (size_t)b * vol, type, step_size, max_step_size, dt_gamma, cnt, (int32_t)i,
		          (int32_t)b, batched ? (int32_t)(b * vol) : 0, t_starts + base, t_ends + base, ridx + base,
		          bidx ? bidx + base : NULL, gidx ? gidx + base : NULL, NULL);
	}
}

/* ------------------------------------------------------------------------------------------------
 * Forest of occupancy grids  (csrc/occ_grid/src/forest_marching.cu:16-143).  The block segments of every ray
 * (block index, entry / exit depth; packed per ray) come from the octree ray trace (kaolin, not restated).
 * One call = one pass: packed_info == NULL counts (num_steps), otherwise emits.
 * `local_roi_min = world_origin + k * world_block_size` is one expression: FMA under nvcc's contraction.
 * Per segment the marcher first steps `t_mid += step_size` at least once (do-while, :99-101) -- kept.
 * ---------------------------------------------------------------------------------------------- */
void orc_forest_march(uint32_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                      const float *t_max, const int32_t *seg_block_inds, const float *seg_entries,
                      const float *seg_exits, const int32_t *seg_pack_infos, const int *res,
                      const uint8_t *grid /*[B,Rx,Ry,Rz]*/, const int16_t *block_ks, const float *world_origin,
                      const float *world_block_size, float step_size, float max_step_size, float dt_gamma,
                      uint32_t max_steps_in, const int32_t *packed_info, int32_t *num_steps, float *t_starts,
                      float *t_ends, int32_t *ridx, int32_t *blidx, int32_t *gidx) {
	const uint32_t vol = (uint32_t)(res[0] * res[1] * res[2]);
	for (uint32_t i = 0; i < n_rays; ++i) {
		const uint32_t seg_begin = (uint32_t)seg_pack_infos[2 * i], seg_len = (uint32_t)seg_pack_infos[2 * i + 1];
		uint32_t max_steps = max_steps_in, base = 0;
		if (packed_info) { base = (uint32_t)packed_info[2 * i]; max_steps = (uint32_t)packed_info[2 * i + 1]; }
		const f3 origin = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]};
		const f3 dir = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
		const f3 inv = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
		const float near = t_min[i], far = t_max[i];
		const float dt_min = step_size, dt_max = max_step_size;
		uint32_t j = 0;
		float t0 = near;
		float dt = calc_dt(t0, dt_gamma, dt_min, dt_max);
		float t1 = t0 + dt;
		float t_mid = (t0 + t1) * 0.5f;
		for (uint32_t s = 0; s < seg_len; ++s) {
			const float cur_entry = seg_entries[seg_begin + s], cur_exit = seg_exits[seg_begin + s];
			const uint32_t b = (uint32_t)seg_block_inds[seg_begin + s];
			const int16_t *k = block_ks + 3 * (size_t)b;
			const f3 mn = {fmaf((float)k[0], world_block_size[0], world_origin[0]),
			               fmaf((float)k[1], world_block_size[1], world_origin[1]),
			               fmaf((float)k[2], world_block_size[2], world_origin[2])};
			const f3 mx = {mn.x + world_block_size[0], mn.y + world_block_size[1], mn.z + world_block_size[2]};
			const uint32_t grid_offset = b * vol;
			if (cur_entry >= far || cur_exit <= near) break;
			do { t_mid += step_size; } while (t_mid < cur_entry);
			dt = calc_dt(t_mid, dt_gamma, dt_min, dt_max);
			t0 = t_mid - dt * 0.5f;
			t1 = t_mid + dt * 0.5f;
			while (t_mid <= cur_exit && t_mid <= far && j < max_steps) {
				const f3 p = {fmaf(t_mid, dir.x, origin.x), fmaf(t_mid, dir.y, origin.y), fmaf(t_mid, dir.z, origin.z)};
				const int gi = grid_idx_at(roi_to_unit(p, mn, mx), res);     /* block_grid_occupied_at :16-25: no ROI test */
				if (grid[(size_t)grid_offset + (uint32_t)gi]) {
					if (packed_info) {
						t_starts[base + j] = t0; t_ends[base + j] = t1; ridx[base + j] = (int32_t)i;
						blidx[base + j] = (int32_t)b;
						if (gidx) gidx[base + j] = gi + (int32_t)grid_offset;
					}
					++j;
					t0 = t1;
					t1 = t0 + calc_dt(t0, dt_gamma, dt_min, dt_max);
					t_mid = (t0 + t1) * 0.5f;
				} else {
					t_mid = advance_next_voxel(t_mid, dt_min, p, dir, inv, mn, mx, res);
					dt = calc_dt(t_mid, dt_gamma, dt_min, dt_max);
					t0 = t_mid - dt * 0.5f;
					t1 = t_mid + dt * 0.5f;
				}
			}
		}
		if (!packed_info) num_steps[i] = (int32_t)j;
	}
}
