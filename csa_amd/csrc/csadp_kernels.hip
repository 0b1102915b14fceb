/*
 * csadp_kernels.hip -- column statistics of a finished alignment on the device (tools.c:194-293,
 * CalculateSumOfPairsScore): SURVEY 8(f-4).  gfx950, wave64.  The matrix fills live in csadp_bits.hip
 * (bit-parallel first fills) and csadp_cells.hip (every other fill); the tiled 32-bit and packed-16
 * families of rounds 1-2 were reachable only through environment switches and were retired in round 3.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

/*
 * K3.  Column statistics of a finished alignment, tools.c:259-281 (CalculateSumOfPairsScore):
 * per column the number of gaps, whether all sequences carry the same character, and the
 * sum over sequence pairs of {gap/gap: 0, equal: +1, different: -1}.  One thread per column,
 * strings stored sequence-major so a wave reads 64 consecutive characters of one sequence.
 */
__global__ __launch_bounds__(256) void sp_columns(const uint8_t *__restrict__ chars, int nseq, int length,
                                                  long long *__restrict__ out /* gaps, conserved, score */)
{
	__shared__ long long part[3];
	if (threadIdx.x < 3) part[threadIdx.x] = 0;
	__syncthreads();
	long long gaps = 0, cons = 0, score = 0;
	for (int col = blockIdx.x * blockDim.x + threadIdx.x; col < length; col += gridDim.x * blockDim.x) {
		const uint8_t c0 = chars[col];
		bool same = true;
		for (int i = 0; i < nseq; ++i) {
			const uint8_t a = chars[(size_t)i * length + col];
			gaps += (a == '-');
			same = same && (a == c0);
			for (int j = i + 1; j < nseq; ++j) {
				const uint8_t b = chars[(size_t)j * length + col];
				if (a == '-' && b == '-') continue;
				score += (a == b) ? 1 : -1;
			}
		}
		cons += same;
	}
	atomicAdd((unsigned long long *)&part[0], (unsigned long long)gaps);
	atomicAdd((unsigned long long *)&part[1], (unsigned long long)cons);
	atomicAdd((unsigned long long *)&part[2], (unsigned long long)score);
	__syncthreads();
	if (threadIdx.x < 3) atomicAdd((unsigned long long *)&out[threadIdx.x], (unsigned long long)part[threadIdx.x]);
}

/* host -> device copy by the compute units: the inputs of a profile step are a few hundred KB in pinned host memory, which a
 * kernel reads over PCIe directly; the copy engine's path costs ~25 us before the first byte (hipMemcpyAsync), this one ~10 */
__global__ __launch_bounds__(256) void pull_pinned(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t units)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < units; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

hipError_t launch_pull_pinned(void *dst, const void *pinned_src, size_t bytes, hipStream_t st)
{
	const size_t units = (bytes + 15) / 16;                  /* both buffers are allocated in multiples of 256 bytes */
	if (units == 0) return hipSuccess;
	const int blocks = (int)std::min<size_t>((units + 255) / 256, 512);
	hipLaunchKernelGGL(pull_pinned, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint4 *>(dst), reinterpret_cast<const uint4 *>(pinned_src), units);
	return hipGetLastError();
}

hipError_t launch_sp_columns(const uint8_t *chars, int nseq, int length, long long *out, hipStream_t st)
{
	const int blocks = (length + 255) / 256 < 2048 ? (length + 255) / 256 : 2048;
	hipLaunchKernelGGL(sp_columns, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, chars, nseq, length, out);
	return hipGetLastError();
}

/* ---- launch wrappers (host) ----------------------------------------------------------- */

}  // namespace csadp
