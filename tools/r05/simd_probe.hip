// Which SIMD do the waves of a THREE-wave workgroup land on?  (Real mitochondrial pairs are three strips at three words per lane: if the
// dispatcher started every workgroup at SIMD 0, a quarter of the chip would never see a fill wave.)
// hipcc --offload-arch=gfx950 -O2 tools/r05/simd_probe.hip -o build/var/simd_probe && build/var/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned *out, int spin)
{
	const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9)
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
	if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
}

int main()
{
	for (int waves : {3, 4, 5}) {
		for (int wgs : {198, 792, 2376}) {
			unsigned *d;
			const int n = wgs * waves;
			hipMalloc(&d, n * sizeof(unsigned));
			hipLaunchKernelGGL(probe, dim3(wgs), dim3(64 * waves), 0, 0, d, 200000);
			std::vector<unsigned> h(n);
			hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
			long simd[4] = {0, 0, 0, 0};
			for (unsigned v : h) simd[(v >> 4) & 3]++;
			// first wave of each workgroup
			long first[4] = {0, 0, 0, 0};
			for (int w = 0; w < wgs; ++w) first[(h[w * waves] >> 4) & 3]++;
			printf("%d-wave workgroups x %4d: waves per SIMD id  %ld %ld %ld %ld   (first wave of a workgroup: %ld %ld %ld %ld)\n", waves, wgs, simd[0], simd[1], simd[2],
			       simd[3], first[0], first[1], first[2], first[3]);
			hipFree(d);
		}
	}
	return 0;
}
