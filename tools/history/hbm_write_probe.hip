// HBM write bandwidth of plain coalesced 8-byte stores (the access pattern of the direction planes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void fill8(uint2 *p, size_t per_block, unsigned v) {
	uint2 *q = p + (size_t)blockIdx.x * per_block;
	for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) q[i] = make_uint2(v + (unsigned)i, v);
}
__global__ void fill16(uint4 *p, size_t per_block, unsigned v) {
	uint4 *q = p + (size_t)blockIdx.x * per_block;
	for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) q[i] = make_uint4(v + (unsigned)i, v, v, v);
}
int main() {
	const size_t bytes = 8ull << 30;
	void *buf; CHECK(hipMalloc(&buf, bytes));
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	for (int blocks : {1024, 2048, 4096, 16384}) {
		for (int mode = 0; mode < 2; ++mode) {
			const size_t elems = bytes / (mode ? 16 : 8);
			const size_t per_block = elems / blocks;
			float best = 1e9f;
			for (int rep = 0; rep < 5; ++rep) {
				CHECK(hipEventRecord(a));
				if (mode) hipLaunchKernelGGL(fill16, dim3(blocks), dim3(256), 0, 0, (uint4 *)buf, per_block, rep);
				else hipLaunchKernelGGL(fill8, dim3(blocks), dim3(256), 0, 0, (uint2 *)buf, per_block, rep);
				CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
				float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
			}
			printf("%s stores, %5d blocks x 256 threads: %.3f ms for 8 GiB = %.0f GB/s\n", mode ? "16-byte" : " 8-byte", blocks, best, bytes / (best * 1e-3) / 1e9);
		}
	}
	CHECK(hipMemset(buf, 0, bytes));
	CHECK(hipEventRecord(a)); CHECK(hipMemsetAsync(buf, 1, bytes, 0)); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
	float ms; CHECK(hipEventElapsedTime(&ms, a, b));
	printf("hipMemsetAsync 8 GiB: %.3f ms = %.0f GB/s\n", ms, bytes / (ms * 1e-3) / 1e9);
	return 0;
}
