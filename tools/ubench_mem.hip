// tools/ubench_mem.hip -- micro-benchmarks that price the LoTD design choices on MI355X:
//   random fp32 atomic adds (agent vs workgroup scope) and random 8-byte gathers, for table sizes that
//   fit one XCD L2 (4 MiB), the Infinity Cache (46 MiB) or neither (1 GiB).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mem.hip -o tools/ubench_mem ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// variants of ONE atomic per random entry (8-byte aligned slots): which opcode is fast?
template <int KIND>   // 0 f32, 1 u32, 2 u64, 3 f64, 4 pk_f16(half2), 5 f32 returning
__global__ void k_atomic1(float *tab, uint32_t mask, uint32_t per_thread) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t s = rnd(gid * 2654435761u + 999u);
	float sink = 0.f;
	for (uint32_t k = 0; k < per_thread; ++k) {
		s = rnd(s + k);
		float *p = tab + 2 * (size_t)(s & mask);
		if (KIND == 0) unsafeAtomicAdd(p, 1.0f);
		else if (KIND == 1) atomicAdd((unsigned int *)p, 1u);
		else if (KIND == 2) atomicAdd((unsigned long long *)p, 1ull);
		else if (KIND == 3) unsafeAtomicAdd((double *)p, 1.0);
		else if (KIND == 4) unsafeAtomicAdd((__half2 *)p, __half2{(__half)1.0f, (__half)1.0f});
		else sink += unsafeAtomicAdd(p, 1.0f);
	}
	if (sink == 123.f) tab[0] = sink;
}

// LDS atomics: every lane adds into a 32K-entry (128 KB) LDS table at random addresses
__global__ void k_lds_atomic(float *out, uint32_t per_thread) {
	extern __shared__ float lds[];
	for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 0.f;
	__syncthreads();
	uint32_t s = rnd((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 5u);
	for (uint32_t k = 0; k < per_thread; ++k) {
		s = rnd(s + k);
		atomicAdd(&lds[s & 32767u], 1.0f);
	}
	__syncthreads();
	if (lds[threadIdx.x] == -1.f) out[0] = 1.f;
}

template <int MODE>   // 0 agent-scope hw atomic, 1 workgroup-scope atomic, 2 plain store (no atomic; upper bound)
__global__ void k_atomic(float *tab, uint32_t mask, uint32_t per_thread, uint32_t xcd_local) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t s = rnd(gid * 2654435761u + 12345u);
	// xcd_local: restrict each XCD (blockIdx % 8) to its own 1/8 slice of the table
	const uint32_t slice = (mask + 1) >> 3;
	for (uint32_t k = 0; k < per_thread; ++k) {
		s = rnd(s + k);
		uint32_t i = s & mask;
		if (xcd_local) i = (blockIdx.x & 7u) * slice + (i & (slice - 1));
		float *p = tab + 2 * (size_t)i;
		if (MODE == 0) { unsafeAtomicAdd(p, 1.0f); unsafeAtomicAdd(p + 1, 1.0f); }
		else if (MODE == 1) {
			__hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			__hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else { p[0] = 1.0f; p[1] = 1.0f; }
	}
}

__global__ void k_gather(const float2 *tab, uint32_t mask, uint32_t per_thread, float *out, uint32_t xcd_local) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t s = rnd(gid * 2654435761u + 777u);
	const uint32_t slice = (mask + 1) >> 3;
	float acc = 0.f;
	for (uint32_t k = 0; k < per_thread; k += 8) {
		float2 v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			s = rnd(s + k + u);
			uint32_t i = s & mask;
			if (xcd_local) i = (blockIdx.x & 7u) * slice + (i & (slice - 1));
			v[u] = tab[i];
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y;
	}
	if (acc == 123.456f) out[gid] = acc;
}

int main() {
	const uint32_t threads = 1u << 22, per = 32;   // 2^27 entries touched (2 floats each)
	float *out; CK(hipMalloc(&out, threads * 4));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	{
		const size_t entries = (size_t)1 << 19;
		float *tab; CK(hipMalloc(&tab, entries * 8)); CK(hipMemset(tab, 0, entries * 8));
		const char *nm[] = {"f32", "u32", "u64", "f64", "pk_f16", "f32 returning"};
		for (int kind = 0; kind < 6; ++kind) {
			float best = 1e9f;
			for (int rep = 0; rep < 3; ++rep) {
				CK(hipEventRecord(a));
				const dim3 g(threads / 256), bl(256);
				const uint32_t m = (uint32_t)(entries - 1);
				switch (kind) {
				case 0: hipLaunchKernelGGL(k_atomic1<0>, g, bl, 0, 0, tab, m, per); break;
				case 1: hipLaunchKernelGGL(k_atomic1<1>, g, bl, 0, 0, tab, m, per); break;
				case 2: hipLaunchKernelGGL(k_atomic1<2>, g, bl, 0, 0, tab, m, per); break;
				case 3: hipLaunchKernelGGL(k_atomic1<3>, g, bl, 0, 0, tab, m, per); break;
				case 4: hipLaunchKernelGGL(k_atomic1<4>, g, bl, 0, 0, tab, m, per); break;
				default: hipLaunchKernelGGL(k_atomic1<5>, g, bl, 0, 0, tab, m, per); break;
				}
				CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
				float ms; CK(hipEventElapsedTime(&ms, a, b));
				if (ms < best) best = ms;
			}
			printf("4 MiB table, 1 atomic/entry  %-14s %8.3f ms  %8.2f Gops/s\n", nm[kind], best, (double)threads * per / best / 1e6);
		}
		CK(hipFuncSetAttribute((const void *)k_lds_atomic, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
		float best = 1e9f;
		const uint32_t lds_threads = 256 * 1024, lds_per = 256;
		for (int rep = 0; rep < 3; ++rep) {
			CK(hipEventRecord(a));
			hipLaunchKernelGGL(k_lds_atomic, dim3(lds_threads / 1024), dim3(1024), 131072, 0, out, lds_per);
			CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			if (ms < best) best = ms;
		}
		printf("LDS f32 atomic add, 128 KB table/CU, 256 WGs x 1024 thr   %8.3f ms  %8.2f Gops/s\n", best, (double)lds_threads * lds_per / best / 1e6);
		CK(hipFree(tab));
	}
	for (uint32_t log2e : {19u}) {          // entries of 8 B: 4 MiB, 32 MiB, 1 GiB
		const size_t entries = (size_t)1 << log2e;
		float *tab; CK(hipMalloc(&tab, entries * 8)); CK(hipMemset(tab, 0, entries * 8));
		for (int xl = 0; xl < 2; ++xl) {
			for (int mode = 0; mode < 4; ++mode) {
				float best = 1e9f;
				for (int rep = 0; rep < 3; ++rep) {
					CK(hipEventRecord(a));
					if (mode == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(threads / 256), dim3(256), 0, 0, tab, (uint32_t)(entries - 1), per, xl);
					else if (mode == 1) hipLaunchKernelGGL(k_atomic<1>, dim3(threads / 256), dim3(256), 0, 0, tab, (uint32_t)(entries - 1), per, xl);
					else if (mode == 2) hipLaunchKernelGGL(k_atomic<2>, dim3(threads / 256), dim3(256), 0, 0, tab, (uint32_t)(entries - 1), per, xl);
					else hipLaunchKernelGGL(k_gather, dim3(threads / 256), dim3(256), 0, 0, (const float2 *)tab, (uint32_t)(entries - 1), per, out, xl);
					CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
					float ms; CK(hipEventElapsedTime(&ms, a, b));
					if (ms < best) best = ms;
				}
				const double ops = (double)threads * per * (mode == 3 ? 1 : 2);
				const char *nm[] = {"atomic_f32 agent", "atomic_f32 workgroup", "plain store x2", "gather float2"};
				printf("table %7.1f MiB  xcd_local=%d  %-22s %8.3f ms  %8.2f Gops/s\n", entries * 8 / 1048576.0, xl, nm[mode], best, ops / best / 1e6);
			}
		}
		CK(hipFree(tab));
	}
	return 0;
}
