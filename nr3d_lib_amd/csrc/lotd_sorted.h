// nr3d_lib_amd/csrc/lotd_sorted.h -- the interface of the sorted-points dL/dparam path of large VM levels (lotd_sorted.hip) towards
// its caller, the binned backward (lotd_bin.hip).
#pragma once
#include "lotd_vm.h"

namespace nr3d {
namespace lotd {

constexpr uint32_t kVsAcc = 10112;            // LDS accumulators, in entries of two features: plane band + line d (158 KiB of fp64)
constexpr uint32_t kVsThreads = 1024, kVsMaxQD = 96, kVsMaxSc = 32, kVsPmax = 32768, kVsSlot = 2u * kVsAcc;
constexpr int kVsLds = (int)(kVsAcc * 2u * 8u);

struct VsPlan {
	uint32_t n_qd, n_blocks, n_items, n_groups, w_max, rb_max, rd_max, m_slots;
	float thr_max;                         // forest: points nearer than this to a face of their block are boundary candidates
	uint32_t item_base[kVsMaxQD + 1];      // items of (pseudo level, component) pair k: [block][band]
	float thr[kVsMaxQD];                   // forest: a point nearer than this to a face of its block may sit in a boundary cell
	uint16_t q[kVsMaxQD], rows[kVsMaxQD], n_bands[kVsMaxQD];     // pseudo level, cell rows per band, bands per plane
	uint8_t d[kVsMaxQD];
	// sort keys: the distinct row scales of the served (level, component) pairs along x_o (o = 0, 1), the clamp of each one's row,
	// the width and the all-ones value ("no row": negative / NaN coordinate) of the row field
	uint32_t n_sc[2], row_bits[2], row_none[2];
	float sc[2][kVsMaxSc], cap[2][kVsMaxSc];
};
struct VsScratch { uint64_t stats, multi_list, plan, key_in[2], key_out[2], keyb_in[2], keyb_out[2], idxb, cand_cnt, n_cand, perm[2], permb[2], xm[2], tmp, xs[2], vs[2], gts[2], items, irec, units, item_tmax, cinfo[2], handoff, lines, slots, total; };

// which pseudo levels the sorted path serves (mask; 0: none) and its plan; the scratch layout; the run on `st`
uint64_t vm_sorted_plan(const nr3d_lotd_meta_t *m, uint32_t n, uint32_t n_blocks, bool forest, int32_t min_level, int32_t max_level,
                        uint64_t skip, VsPlan &vp);
void vm_sorted_scratch(const VsPlan &vp, uint32_t n, uint32_t E, bool second, bool forest, VsScratch &s);
int vm_sorted_run(bool second, const VsPlan &vp, const nr3d_lotd_meta_t *meta, const nr3d_lotd_meta_t *md, uint32_t n, const float *x,
                  const float *vin, const float *g, int64_t g_sn, int64_t g_se, const void *params, bool p_half, const Batch &ba,
                  const ForestDev *forest, float *dparam, char *scratch, const VsScratch &s, hipStream_t st);

}  // namespace lotd
}  // namespace nr3d
