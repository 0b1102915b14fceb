/*
 * csadp_pairio.hip -- device-side input packing and output expansion of 2-sequence tasks, so that
 * a pair batch crosses PCIe as raw letters in and finished rows out.  gfx950, wave64.
 *
 *   nw_pack_planes   CharAt (alignment.c:16-20) + CharCodeFromSeq (dynamicprogramming.c:57-71) for
 *                    every letter of both regions: rotated, linearised reads of the circular text
 *                    (idx = rotation + start + p, wrapped once by subtraction) -> the two bit planes
 *                    nw_fill_bits consumes.  Letters other than A,C,G,T raise the job's input status
 *                    (the reference indexes scorevector[][-1] there: survey Q4).
 *   nw_expand_rows   the traceback application of dynamicprogramming.c:1066-1138 for numberofseqs = 2:
 *                    walks the op list of the traceback kernel and writes the two aligned rows, the
 *                    leftover rows / columns of :1115-1138 first, plus the DP score as the sum of the
 *                    move scores along the path (:993-998 for i = 1).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

namespace {

/* A 0, C 1, G 2, T 3 (dynamicprogramming.c:63-69); anything else: -1 */
__device__ __forceinline__ int letter_code(uint32_t c)
{
	const uint32_t h = (c >> 1) & 3u;                  /* A 0, C 1, T 2, G 3 */
	const bool ok = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
	return ok ? (int)(h ^ (h >> 1)) : -1;
}

/* text index of region position p (CharAt: one wrap by subtraction) */
__device__ __forceinline__ int wrap_index(int first, int p, int size)
{
	int i = first + p;
	if (i >= size) i -= size;
	return i;
}

}  // namespace

/*
 * grid (jobs, 2, kPackSlices): y = 0 packs the column sequence into colplanes[2][nwords_pad], y = 1 the row
 * sequence into rowplanes[2][rowwords].  A wave takes 64 consecutive letters per group: one coalesced byte load,
 * two ballots, one 16-byte store by lane 0; it requests the letters of four groups before it uses the first.
 * Words beyond the region are zero (the fill needs no bounds).  The kernel sits between two fills of its
 * stream, on a chip full of fill waves: as 2 workgroups of 64 dependent loads per wave it took 0.13-0.18 ms there
 * (45 us alone) with the stream's next fill waiting behind it -- hence many short waves (rocprofv3 kernel trace).
 */
constexpr int kPackSlices = 8;
constexpr int kPackAhead = 4;

__global__ __launch_bounds__(256) void nw_pack_planes(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs)
{
	const BitJob &J = jobs[blockIdx.x];
	if (J.text[0] == 0) return;                          /* the host wrote this job's planes */
	const int which = blockIdx.y;
	const int n = which ? J.nrows : J.ncols;
	const int nwords = which ? J.rowwords : J.nwords_pad;
	uint32_t *planes = reinterpret_cast<uint32_t *>(arena + (which ? J.rowplanes : J.colplanes));
	const uint8_t *text = arena + J.text[which];
	const int size = J.size[which], first = J.first[which];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
	const int stride = nwaves * gridDim.z;
	bool bad = false;
	for (int g0 = blockIdx.z * nwaves + wave; 2 * g0 < nwords; g0 += kPackAhead * stride) {
		uint32_t c[kPackAhead];
#pragma unroll
		for (int u = 0; u < kPackAhead; ++u) {
			const int p = 64 * (g0 + u * stride) + lane;
			c[u] = p < n ? text[wrap_index(first, p, size)] : 0u;     /* p < n <= 32 * nwords: inside the region */
		}
#pragma unroll
		for (int u = 0; u < kPackAhead; ++u) {
			const int g = g0 + u * stride;
			if (2 * g >= nwords) break;
			int code = 0;
			if (64 * g + lane < n) {
				code = letter_code(c[u]);
				if (code < 0) { bad = true; code = 0; }
			}
			const unsigned long long b0 = __ballot(code & 1), b1 = __ballot(code & 2);
			if (lane == 0) {
				planes[2 * g] = (uint32_t)b0;
				planes[nwords + 2 * g] = (uint32_t)b1;
				if (2 * g + 1 < nwords) {
					planes[2 * g + 1] = (uint32_t)(b0 >> 32);
					planes[nwords + 2 * g + 1] = (uint32_t)(b1 >> 32);
				}
			}
		}
	}
	if (__ballot(bad) != 0 && lane == 0) atomicOr(reinterpret_cast<int *>(arena + J.istatus), 1);
}

/*
 * One workgroup per job.  ops[t] is the move taken from the t-th cell of the walk (t = 0 at cell
 * (nrows, ncols)); column m = consensus - 1 - t of the alignment belongs to it.  The letter an op
 * consumes is found from the number of row / column consuming ops before it: a block-wide prefix sum
 * over chunks of kOpsPerThread x 256 ops.
 */
constexpr int kOpsPerThread = 8;

__global__ __launch_bounds__(256) void nw_expand_rows(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs)
{
	__shared__ uint32_t wave_tot[4];
	__shared__ int red[4];
	const BitJob &J = jobs[blockIdx.x];
	if (J.text[0] == 0) return;
	int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
	const uint8_t *ops = arena + J.ops;
	const uint8_t *tc = arena + J.text[0], *tr = arena + J.text[1];
	uint8_t *oc = arena + J.out[0], *orow = arena + J.out[1];
	const int n = summary[0], remj = summary[1], remk = summary[2];
	const int cons = n + remj + remk;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int sizec = J.size[0], sizer = J.size[1], firstc = J.first[0], firstr = J.first[1];

	/* :1115-1138: what the walk left over sits at the left end: first remk column letters against
	 * gaps, then remj row letters against gaps (at most one of the two is non-zero) */
	for (int m = tid; m < remk; m += blockDim.x) {
		oc[m] = tc[wrap_index(firstc, m, sizec)];
		orow[m] = '-';
	}
	for (int m = tid; m < remj; m += blockDim.x) {
		orow[remk + m] = tr[wrap_index(firstr, m, sizer)];
		oc[remk + m] = '-';
	}
	if (tid == 0) {
		oc[cons] = 0;
		orow[cons] = 0;
	}

	int score = 0;
	int rows_done = 0, cols_done = 0;
	for (int base = 0; base < n; base += 256 * kOpsPerThread) {
		const int t0 = base + tid * kOpsPerThread;
		uint8_t op[kOpsPerThread];
		{
			const uint2 v = (t0 < n) ? *reinterpret_cast<const uint2 *>(ops + t0) : make_uint2(0x03030303u, 0x03030303u);
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				op[u] = (uint8_t)(v.x >> (8 * u));
				op[4 + u] = (uint8_t)(v.y >> (8 * u));
			}
		}
		uint32_t mine = 0;                              /* rows | cols << 16 consumed by this thread's ops */
#pragma unroll
		for (int u = 0; u < kOpsPerThread; ++u) {
			if (t0 + u >= n) op[u] = 3;                 /* past the end: consumes nothing */
			mine += (op[u] == DIR_D ? 0x00010001u : op[u] == DIR_U ? 0x00000001u : op[u] == DIR_L ? 0x00010000u : 0u);
		}
		/* exclusive prefix over the block: inside the wave by shuffles, across the 4 waves through LDS */
		uint32_t incl = mine;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t up = __shfl_up(incl, d);
			if (lane >= d) incl += up;
		}
		if (lane == 63) wave_tot[wave] = incl;
		__syncthreads();
		uint32_t before = incl - mine;
		uint32_t chunk_total = 0;
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			if (w < wave) before += wave_tot[w];
			chunk_total += wave_tot[w];
		}
		__syncthreads();
		int rc = rows_done + (int)(before & 0xffffu), cc = cols_done + (int)(before >> 16);
#pragma unroll
		for (int u = 0; u < kOpsPerThread; ++u) {
			const int t = t0 + u;
			if (t >= n) break;
			const int m = cons - 1 - t;
			const uint8_t o = op[u];
			uint8_t a = '-', b = '-';                    /* column sequence letter, row sequence letter */
			if (o != DIR_U) { a = tc[wrap_index(firstc, J.ncols - 1 - cc, sizec)]; ++cc; }
			if (o != DIR_L) { b = tr[wrap_index(firstr, J.nrows - 1 - rc, sizer)]; ++rc; }
			oc[m] = a;
			orow[m] = b;
			score += (o == DIR_D && a == b) ? 1 : -1;   /* MATCHSCORE, else MISMATCHSCORE = INDELSCORE = -1 (:16-19) */
		}
		rows_done += (int)(chunk_total & 0xffffu);
		cols_done += (int)(chunk_total >> 16);
	}
	/* block sum of the move scores; + the border cell the walk stopped on: H[j][0] = -j, H[0][k] = -k (:967, :972) */
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) score += __shfl_down(score, d);
	if (lane == 0) red[wave] = score;
	__syncthreads();
	if (tid == 0) {
		summary[3] = red[0] + red[1] + red[2] + red[3] - remj - remk;
		int st = *reinterpret_cast<const int *>(arena + J.istatus);
		if (rows_done + remj != J.nrows || cols_done + remk != J.ncols) st |= 2;    /* the walk does not span the matrix */
		summary[4] = st;
	}
}

hipError_t launch_pack_planes(uint8_t *arena, const BitJob *jobs, int njobs, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	hipLaunchKernelGGL(nw_pack_planes, dim3(njobs, 2, kPackSlices), dim3(256), 0, st, arena, jobs);
	return hipGetLastError();
}

hipError_t launch_expand_rows(uint8_t *arena, const BitJob *jobs, int njobs, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	hipLaunchKernelGGL(nw_expand_rows, dim3(njobs), dim3(256), 0, st, arena, jobs);
	return hipGetLastError();
}

}  // namespace csadp
