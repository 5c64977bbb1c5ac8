// tools/ubench_stream.hip -- HBM streaming ceilings on MI355X for the access shapes the binned dL/dparam path uses.
//   read  A: linear, 16 B / lane           B: linear, 12 B / lane (dwordx3)
//         C: 768-byte runs, one run per wave-load, runs 48 KiB apart (stage-B's shape: run b of every slot)
//   write D: linear 16 B / lane            E: 768-byte runs 48 KiB apart
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_stream.hip -o tools/ubench_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void rd16(const uint4 *__restrict__ p, size_t n4, uint32_t *out) {
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
		const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
struct __attribute__((packed, aligned(4))) R3 { uint32_t a, b, c; };
__global__ __launch_bounds__(1024) void rd12(const R3 *__restrict__ p, size_t n, uint32_t *out) {
	uint32_t acc = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const R3 v = p[i]; acc += v.a ^ v.b ^ v.c;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
// slots of `slot_recs` records; bucket b (of nb) owns records [b*64, b*64+64) of every slot.  Workgroup = bucket.
template <int UNROLL>
__global__ __launch_bounds__(1024) void rd_runs(const R3 *__restrict__ p, uint32_t n_slots, uint32_t slot_recs, uint32_t *out) {
	p += (size_t)blockIdx.y * n_slots * slot_recs;
	const uint32_t b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t acc = 0;
	for (uint32_t s0 = wave * UNROLL; s0 < n_slots; s0 += 16 * UNROLL) {
		R3 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) v[u] = p[(size_t)(s0 + u) * slot_recs + b * 64 + lane];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) acc += v[u].a ^ v[u].b ^ v[u].c;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
// as rd_runs but the run a wave reads per slot is RL x 64 records (slots hold nb runs of RL x 64 records)
template <int RL>
__global__ __launch_bounds__(1024) void rd_runs_long(const R3 *__restrict__ p, uint32_t n_slots, uint32_t slot_recs, uint32_t *out) {
	p += (size_t)blockIdx.y * n_slots * slot_recs;
	const uint32_t b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	constexpr int U = 8 / RL;
	uint32_t acc = 0;
	for (uint32_t s0 = wave * U; s0 < n_slots; s0 += 16 * U) {
		R3 v[U][RL];
#pragma unroll
		for (int u = 0; u < U; ++u)
#pragma unroll
			for (int k = 0; k < RL; ++k) v[u][k] = p[(size_t)(s0 + u) * slot_recs + b * (64 * RL) + k * 64 + lane];
#pragma unroll
		for (int u = 0; u < U; ++u)
#pragma unroll
			for (int k = 0; k < RL; ++k) acc += v[u][k].a ^ v[u][k].b ^ v[u][k].c;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(1024) void wr16(uint4 *__restrict__ p, size_t n4) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
		p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// workgroup = slot: writes its whole slot (48 KiB) linearly, 16 B / lane
__global__ __launch_bounds__(512) void wr_slot(uint4 *__restrict__ p, uint32_t slot_v4) {
	uint4 *d = p + (size_t)blockIdx.x * slot_v4;
	for (uint32_t i = threadIdx.x; i < slot_v4; i += 512) d[i] = make_uint4(i, 1, 2, 3);
}

// workgroup = point block i: writes its 64 runs of 64 records, run b to bucket stream b at position i (bucket-major)
__global__ __launch_bounds__(512) void wr_bucket_major(uint4 *__restrict__ p, uint32_t n_blk) {
	const uint32_t i = blockIdx.x % n_blk, lvl = blockIdx.x / n_blk;
	uint4 *base = p + (size_t)lvl * n_blk * 64 * 48;                  // 48 uint4 = 768 B per run
	for (uint32_t v = threadIdx.x; v < 64 * 48; v += 512) {
		const uint32_t b = v / 48, w = v - b * 48;
		base[((size_t)b * n_blk + i) * 48 + w] = make_uint4(v, 1, 2, 3);
	}
}

int main() {
	const uint32_t slot_recs = 4096, n_slots = 2048 * 16;        // 2^20 points x 16 levels / 512 points per slot
	const size_t n_rec = (size_t)slot_recs * n_slots, bytes = n_rec * 12;
	void *buf; uint32_t *out;
	CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 1, bytes));
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	auto timeit = [&](const char *nm, auto fn) {
		fn(); hipDeviceSynchronize();
		hipEventRecord(e0); for (int i = 0; i < 5; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
		printf("%-52s %8.3f ms  %7.2f TB/s\n", nm, ms, bytes / ms * 1e-9);
	};
	printf("buffer %.2f GB\n", bytes * 1e-9);
	timeit("read linear 16 B/lane, 2048 WG x 1024", [&] { hipLaunchKernelGGL(rd16, dim3(2048), dim3(1024), 0, 0, (const uint4 *)buf, bytes / 16, out); });
	timeit("read linear 16 B/lane, 256 WG x 1024", [&] { hipLaunchKernelGGL(rd16, dim3(256), dim3(1024), 0, 0, (const uint4 *)buf, bytes / 16, out); });
	timeit("read linear 12 B/lane, 2048 WG x 1024", [&] { hipLaunchKernelGGL(rd12, dim3(2048), dim3(1024), 0, 0, (const R3 *)buf, n_rec, out); });
	// 64 buckets per level -> per level 2048 slots; emulate with all 32768 slots and 64 buckets (each WG reads 1/64 of every slot)
	timeit("read 768-B runs, only 64 WG (all slots)", [&] { hipLaunchKernelGGL(rd_runs<8>, dim3(64, 1), dim3(1024), 0, 0, (const R3 *)buf, n_slots, slot_recs, out); });
	timeit("read 768-B runs, 1024 WG (64 buckets x 16 levels) u=8", [&] {
		// level l = slots [l*2048, (l+1)*2048): launch as 16 grids back to back is equivalent to one grid of 1024 WGs
		hipLaunchKernelGGL(rd_runs<8>, dim3(64, 16), dim3(1024), 0, 0, (const R3 *)buf, 2048u, slot_recs, out); });
	timeit("write linear 16 B/lane, 2048 WG x 1024", [&] { hipLaunchKernelGGL(wr16, dim3(2048), dim3(1024), 0, 0, (uint4 *)buf, bytes / 16); });
	timeit("write one 48-KiB slot per WG (512 thr), 32768 WG", [&] { hipLaunchKernelGGL(wr_slot, dim3(n_slots), dim3(512), 0, 0, (uint4 *)buf, slot_recs * 12 / 16); });
	timeit("write 768-B runs bucket-major (64 streams x 16 lvls)", [&] { hipLaunchKernelGGL(wr_bucket_major, dim3(n_slots), dim3(512), 0, 0, (uint4 *)buf, 2048u); });
	timeit("read 1536-B runs (slots of 8192 rec), 64 WG x 16", [&] {
		hipLaunchKernelGGL(rd_runs_long<2>, dim3(64, 16), dim3(1024), 0, 0, (const R3 *)buf, 1024u, 8192u, out); });
	timeit("read 3072-B runs (slots of 16384 rec), 64 WG x 16", [&] {
		hipLaunchKernelGGL(rd_runs_long<4>, dim3(64, 16), dim3(1024), 0, 0, (const R3 *)buf, 512u, 16384u, out); });
	timeit("read 6144-B runs (slots of 32768 rec), 64 WG x 16", [&] {
		hipLaunchKernelGGL(rd_runs_long<8>, dim3(64, 16), dim3(1024), 0, 0, (const R3 *)buf, 256u, 32768u, out); });
	// same runs, but slots padded so consecutive slots do not map to the same HBM channel
	for (uint32_t pad : {0u, 16u, 32u, 64u, 96u, 128u, 344u, 352u, 1024u}) {
		const uint32_t stride = slot_recs + pad, ns = (uint32_t)(n_rec / stride / 16) / 128 * 128;   // slots per level that fit
		char nm[96]; snprintf(nm, sizeof nm, "read 768-B runs, slot stride %u B (pad %u rec)", stride * 12, pad);
		const double scale = (double)ns * 16 * slot_recs / n_rec;
		hipLaunchKernelGGL(rd_runs<8>, dim3(64), dim3(1024), 0, 0, (const R3 *)buf, ns, stride, out); hipDeviceSynchronize();
		hipEventRecord(e0);
		for (int it = 0; it < 5; ++it)
			hipLaunchKernelGGL(rd_runs<8>, dim3(64, 16), dim3(1024), 0, 0, (const R3 *)buf, ns, stride, out);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
		printf("%-52s %8.3f ms  %7.2f TB/s\n", nm, ms, bytes * scale / ms * 1e-9);
	}
	return 0;
}
