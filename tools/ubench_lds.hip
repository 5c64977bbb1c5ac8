// tools/ubench_lds.hip -- LDS atomic / RMW rates at random addresses (prices the LDS-privatised scatter).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// KIND: 7 native ds_add_f32 (__builtin_amdgcn_ds_faddf), 8 the same x2 adjacent (round 6); 0 f32 add, 1 u32 add, 2 u64 add, 3 f64 add, 4 plain (non-atomic) f32 rmw, 5 f32 add x2 adjacent, 6 u64 add x2 adjacent
template <int KIND>
__global__ void k(float *out, uint32_t per_thread, uint32_t mask) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float *f = (float *)smem; uint32_t *u = (uint32_t *)smem; unsigned long long *q = (unsigned long long *)smem; double *d = (double *)smem;
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) u[i] = 0;
	__syncthreads();
	uint32_t s = rnd((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 5u);
	for (uint32_t k2 = 0; k2 < per_thread; ++k2) {
		s = rnd(s + k2);
		const uint32_t i = s & mask;
		if (KIND == 0) atomicAdd(&f[i], 1.0f);
		else if (KIND == 1) atomicAdd(&u[i], 1u);
		else if (KIND == 2) atomicAdd(&q[i >> 1], 1ull);
		else if (KIND == 3) atomicAdd(&d[i >> 1], 1.0);
		else if (KIND == 4) f[i] = f[i] + 1.0f;
		else if (KIND == 5) { atomicAdd(&f[i & ~1u], 1.0f); atomicAdd(&f[i | 1u], 1.0f); }
		else if (KIND == 6) { atomicAdd(&q[(i >> 2) * 2], 1ull); atomicAdd(&q[(i >> 2) * 2 + 1], 1ull); }
		else if (KIND == 7) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)&f[i], 1.0f, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
		else { __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)&f[i & ~1u], 1.0f, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
		       __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)&f[i | 1u], 1.0f, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false); }
	}
	__syncthreads();
	if (f[threadIdx.x] == -1.f) out[0] = 1.f;
}

int main() {
	float *out; CK(hipMalloc(&out, 4096));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const char *nm[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_add_f64", "plain f32 rmw", "ds_add_f32 x2 adjacent", "ds_add_u64 x2 adjacent", "native ds_add_f32", "native ds_add_f32 x2 adj"};
	for (int wg : {256, 1024}) for (int kind = 0; kind < 9; ++kind) {
		const uint32_t blocks = 1024, per = 512;
		float best = 1e9f;
		for (int rep = 0; rep < 3; ++rep) {
			CK(hipEventRecord(a));
			const dim3 g(blocks), bl(wg);
#define L(K) hipLaunchKernelGGL(k<K>, g, bl, 131072, 0, out, per, 32767u)
			switch (kind) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 5: L(5); break; case 6: L(6); break; case 7: L(7); break; default: L(8); }
			CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			if (ms < best) best = ms;
		}
		const double ops = (double)blocks * wg * per * ((kind == 5 || kind == 6 || kind == 8) ? 2 : 1);
		printf("wg=%4d  %-24s %8.3f ms  %8.2f Gops/s\n", wg, nm[kind], best, ops / best / 1e6);
	}
	return 0;
}
