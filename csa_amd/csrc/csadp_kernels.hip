/*
 * csadp_kernels.hip -- column statistics of a finished alignment on the device (tools.c:194-293,
 * CalculateSumOfPairsScore): SURVEY 8(f-4).  gfx950, wave64.  The matrix fills live in csadp_bits.hip
 * (bit-parallel first fills) and csadp_cells.hip (every other fill); the tiled 32-bit and packed-16
 * families of rounds 1-2 were reachable only through environment switches and were retired in round 3.
 */
#include <hip/hip_runtime.h>
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

hipError_t launch_sp_columns(const uint8_t *chars, int nseq, int length, long long *out, hipStream_t st)
{
	const int blocks = (length + 255) / 256 < 2048 ? (length + 255) / 256 : 2048;
	hipLaunchKernelGGL(sp_columns, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, chars, nseq, length, out);
	return hipGetLastError();
}

/* ---- launch wrappers (host) ----------------------------------------------------------- */

}  // namespace csadp
