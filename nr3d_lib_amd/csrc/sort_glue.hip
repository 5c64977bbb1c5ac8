// nr3d_lib_amd/csrc/sort_glue.hip -- the device radix sort (rocPRIM through hipCUB) behind plain functions, in its own
// translation unit: the only library code of libnr3d_hip.so.  Used by lotd_sorted.inc to order the POINTS of a dL/dparam pass
// (by table block and coordinate) -- plumbing in front of the hand-written accumulation kernels, stable and deterministic.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace nr3d {
namespace sortglue {

size_t pairs_tmp_bytes(uint32_t n) {
	size_t b64 = 0, b32 = 0;
	(void)hipcub::DeviceRadixSort::SortPairs((void *)nullptr, b64, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
	                                   (uint32_t *)nullptr, (int)n, 0, 64, (hipStream_t)0);
	(void)hipcub::DeviceRadixSort::SortPairs((void *)nullptr, b32, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
	                                   (uint32_t *)nullptr, (int)n, 0, 32, (hipStream_t)0);
	return ((b64 > b32 ? b64 : b32) + 255) / 256 * 256;
}

int pairs_u64(void *tmp, size_t tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, uint32_t n, int bits,
              hipStream_t st) {
	NR3D_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)n, 0, bits, st));
	return 0;
}

int pairs_u32(void *tmp, size_t tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, uint32_t n,
              hipStream_t st) {
	NR3D_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)n, 0, 32, st));
	return 0;
}

}  // namespace sortglue
}  // namespace nr3d
