// tools/ubench_gather.hip -- what bounds the forward's corner gathers on MI355X (round 3):
//   (a) random 8-byte gathers from a 4 MiB table under every cache policy (default / sc0 / sc1 / sc0 sc1 / nt), 4- and 16-byte
//       gathers: is there a request flavour the L2 serves faster than one 128-byte line per lane?
//   (b) the same gathers with a slow load mixed in (one coalesced 12-byte-per-lane read of a 1 GiB array per `every` gather
//       instructions -- the forward's x read, an L2 miss): does a miss in the vector L1's in-order return queue hold the
//       hits of every other wave up?  (c) the slow load moved to the scalar cache (s_load).
//   (d) with the forward's output stores (4 non-temporal dwords per lane per 4 gathers).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o tools/ubench_gather ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int POL> __device__ __forceinline__ float2 ld8(const float2 *p) {
	float2 v;
	if (POL == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
	else if (POL == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
	else if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	else if (POL == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	else asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
	return v;
}

// (a) policy / width
template <int POL, int WIDTH>   // WIDTH 4, 8, 16 bytes
__global__ void k_gather(const char *tab, uint32_t mask, uint32_t per_thread, float *out) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t s = rnd(gid * 2654435761u + 777u);
	float acc = 0.f;
	for (uint32_t k = 0; k < per_thread; k += 4) {
		float2 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			s = rnd(s + k + u);
			const uint32_t i = s & mask;
			if (WIDTH == 8) v[u] = ld8<POL>((const float2 *)tab + i);
			else if (WIDTH == 4) { v[u].x = ((const float *)tab)[2 * i]; v[u].y = 0.f; }
			else { const float4 q = ((const float4 *)tab)[i >> 1]; v[u] = make_float2(q.x + q.z, q.y + q.w); }
		}
		if (WIDTH == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
		for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
	}
	if (acc == 123.456f) out[gid] = acc;
}

// (b, c, d) the forward's structure: a wave lives for ONE round = [slow load] -> 4 gathers -> [4 NT stores]; blocks of 256
// SLOW: 0 none, 1 vector load of a 12-byte-per-lane-pair stream (L2 miss), 2 the same bytes through s_load
template <int SLOW, bool STORES, int ROUNDS>
__global__ void k_mix(const float2 *tab, uint32_t mask, const float *big, float *out, uint32_t n_big) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	float acc = 0.f;
	for (int r = 0; r < ROUNDS; ++r) {
		const uint32_t item = (blockIdx.x * ROUNDS + r) * blockDim.x + threadIdx.x;
		uint32_t s = rnd(item * 2654435761u + 777u);
		float xs = 0.f;
		if (SLOW == 1) {
			const float *p = big + (size_t)((item >> 1) % n_big) * 3;
			xs = p[0] + p[1] + p[2];
		} else if (SLOW == 2) {
			const uint32_t w0 = __builtin_amdgcn_readfirstlane((item >> 1) % n_big);
			const float *p = big + (size_t)w0 * 3;
			float t = 0.f;
#pragma unroll
			for (int k = 0; k < 96; ++k) t += p[k];
			xs = t;
		}
		s ^= (uint32_t)(xs == 77.f);                 // the gathers depend on the slow load, as the cell locator does
		float2 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) { s = rnd(s + u); v[u] = tab[s & mask]; }
#pragma unroll
		for (int u = 0; u < 4; ++u) acc += v[u].x * v[u].y;
		if (STORES) {
#pragma unroll
			for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(acc + u, &out[(size_t)u * gridDim.x * blockDim.x * ROUNDS + item]);
		}
	}
	if (!STORES && acc == 123.456f) out[gid] = acc;
}

int main() {
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const size_t entries = (size_t)1 << 19;       // 4 MiB of float2
	char *tab; CK(hipMalloc(&tab, entries * 8)); CK(hipMemset(tab, 0, entries * 8));
	const uint32_t n_big = 1u << 26;              // 2^26 points * 12 B = 768 MiB: never cached
	float *big; CK(hipMalloc(&big, (size_t)n_big * 12 + 1024)); CK(hipMemset(big, 0, (size_t)n_big * 12 + 1024));
	const uint32_t threads = 1u << 22;
	float *out; CK(hipMalloc(&out, (size_t)threads * 4 * 4 * 4));
	auto timeit = [&](auto launch, const char *name, double ops) {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			if (ms < best) best = ms;
		}
		printf("%-72s %8.3f ms  %8.2f G gathers/s\n", name, best, ops / best / 1e6);
		return 0;
	};
	const uint32_t per = 32, m = (uint32_t)(entries - 1);
	const dim3 g(threads / 256), bl(256);
#define POLICY(P, W, NAME) timeit([&] { hipLaunchKernelGGL((k_gather<P, W>), g, bl, 0, 0, tab, m, per, out); }, NAME, (double)threads * per)
	POLICY(0, 8, "a: 8-byte gather, default policy");
	POLICY(1, 8, "a: 8-byte gather, sc0");
	POLICY(2, 8, "a: 8-byte gather, sc1");
	POLICY(3, 8, "a: 8-byte gather, sc0 sc1");
	POLICY(4, 8, "a: 8-byte gather, nt");
	POLICY(0, 4, "a: 4-byte gather");
	POLICY(0, 16, "a: 16-byte gather (8-byte pairs)");
	const uint32_t items = 1u << 24;              // lane items; 4 gathers each
#define MIX(S, ST, R, NAME) timeit([&] { hipLaunchKernelGGL((k_mix<S, ST, R>), dim3(items / 256 / R), bl, 0, 0, (const float2 *)tab, m, big, out, n_big); }, NAME, (double)items * 4)
	MIX(0, false, 1, "b: one round per wave, 4 gathers, no slow load, no stores");
	MIX(1, false, 1, "b: + 12-byte x read per lane pair through the vector L1 (L2 miss)");
	MIX(2, false, 1, "c: + the same x read through the scalar cache");
	MIX(0, true, 1, "d: no slow load, 4 NT dword stores per lane");
	MIX(1, true, 1, "d: vector x read + stores (the forward's mix)");
	MIX(2, true, 1, "d: scalar x read + stores");
	MIX(0, false, 4, "b4: four rounds per wave, no slow load, no stores");
	MIX(1, false, 4, "b4: + vector x read");
	MIX(2, false, 4, "c4: + scalar x read");
	MIX(1, true, 4, "d4: vector x read + stores");
	MIX(2, true, 4, "d4: scalar x read + stores");
	return 0;
}
