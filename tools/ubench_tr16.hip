// tools/ubench_tr16.hip -- what gfx950 ds_read_b64_tr_b16 returns for two per-lane address patterns (run on the GPU box: hipcc --offload-arch=gfx950 tools/ubench_tr16.hip -o /tmp/tr16 && /tmp/tr16)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int *addr, float *out) {
	__shared__ __attribute__((aligned(16))) _Float16 lds[4096];
	for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)(float)i;
	__syncthreads();
	s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + addr[threadIdx.x]));
	h4 f = __builtin_bit_cast(h4, v);
	for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)f[j];
}
int main() {
	int h_addr[64]; float h_out[256];
	int *d_addr; float *d_out;
	hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
	for (int variant = 0; variant < 2; ++variant) {
		for (int l = 0; l < 64; ++l) {
			const int i = l & 15, g = l >> 4;
			h_addr[l] = variant == 0 ? g * 1024 + i * 40        /* lane i of a group points at row i (stride 40 halfs), 4 contiguous */
			                         : g * 1024 + (i / 4) * 40 + (i % 4) * 4;
		}
		(void)hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
		(void)hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
		printf("variant %d\n", variant);
		for (int l = 0; l < 20; ++l) printf("lane %2d addr %4d -> %g %g %g %g\n", l, h_addr[l], h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
	}
	return 0;
}
