// nr3d_lib_amd/csrc/common.h -- shared host/device helpers for libnr3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nr3d_hip.h"
#include "options.h"

namespace nr3d {

// thread-local error string behind nr3d_last_error()
char *err_buf();
int fail(const char *fmt, ...);

#define NR3D_CHECK(cond, ...) do { if (!(cond)) return ::nr3d::fail(__VA_ARGS__); } while (0)
#define NR3D_HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
	return ::nr3d::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)
#define NR3D_LAUNCH_CHECK() NR3D_HIP_CHECK(hipGetLastError())

static inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Optional HIP-event timing of individual kernels (nr3d_prof_enable / nr3d_prof_read, include/nr3d_hip.h): a Scope records
// an event pair on the launch stream around the launches in its lifetime when its id is enabled, and costs one load
// and a branch when it is not.
namespace prof {
extern uint32_t g_mask;
void begin(int id, hipStream_t st);
void end(int id, hipStream_t st);
struct Scope {
	int id; hipStream_t st; bool on;
	Scope(int id_, hipStream_t st_) : id(id_), st(st_), on((g_mask >> id_) & 1u) { if (on) begin(id, st); }
	~Scope() { if (on) end(id, st); }
};
}  // namespace prof

// the library's own stable LSD radix sort of (u32 key, u32 value) pairs (rsort.hip): `batch` (1 or 2) independent sorts of the same
// length per call, by key bits [0, bits); vin[b] == NULL: the values are the element indices; n_dev (optional): the element count
// in device memory (<= n_max).  Inputs untouched, outputs must not alias them.
namespace rsort {
size_t tmp_bytes(uint32_t n_max, int batch);
int sort_pairs(void *tmp, int batch, const uint32_t *const *kin, const uint32_t *const *vin, uint32_t *const *kout, uint32_t *const *vout,
               uint32_t n_max, const uint32_t *n_dev, int bits, hipStream_t st);
}  // namespace rsort

// ---- dtype <-> float conversions used by kernels templated on storage type ----
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half(v); }

// Hardware fp32 add atomic at agent scope (global_atomic_add_f32); memory is coarse-grained device
// memory allocated by the caller, for which the hardware instruction is valid.
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

}  // namespace nr3d
