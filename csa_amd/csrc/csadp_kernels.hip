/*
 * csadp_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the DP hot path.
 *
 *   nw_fill_tiles   K1: the fill of dynamicprogramming.c:990-1029 (int32 max-of-3 with the
 *                   reference's D >= L >= U tie-break), directions packed 2 bit/cell.
 *   nw_traceback    K2: the direction walk of dynamicprogramming.c:1037-1047 / :1072-1114,
 *                   emitting one op per visited cell; the host applies the ops to the
 *                   strings and the profile (csadp_progressive.c).
 *
 * Mapping.  A wave owns a strip of 64*C columns; lane l keeps the C cells of its columns of
 * the previous row in registers.  The wave is skewed: at global step T lane L (global lane
 * index over all strips) computes row T-L+1, so the left neighbour's value of the same row
 * was produced one step earlier and arrives with ONE cross-lane instruction
 * (v_mov_b32_dpp wave_shr:1); lane 0 takes the value of the previous strip (or the border
 * column) through the DPP "old" operand, pre-loaded from LDS.  A tile is TR consecutive
 * steps of one strip -- a parallelogram in (row, column) space, so there is no per-tile
 * pipeline ramp; the ramp exists once per matrix.  Tile (a, s) needs tiles (a-1, s) [lane
 * registers, via FillJob::state] and (a, s-1), (a-1, s-1) [right edge, via
 * FillJob::handoff]: the host launches one grid per tile anti-diagonal a+s, for all tasks
 * of a batch at once; no workgroup waits on another inside a launch.
 *
 * Per cell: v_bfe_u32 (profile field) + v_lshl_add_u32 (diag) + 2 v_add (up, left) +
 * v_min3_i32 + v_alignbit_b32 (shift the 2-bit tag into the direction word) + v_and (clear
 * the tag) = 7 VALU instructions; see csadp_device.h for the cost/tag representation.
 * Directions are stored in the order they are produced (strip, step, lane): one coalesced
 * 256-byte store per wave and step; the traceback kernel addresses the same layout.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

#define DPP_WAVE_SHR1 0x138

template <int C, int TR>
__global__ __launch_bounds__(64) void nw_fill_tiles(uint8_t *__restrict__ arena,
                                                    const FillJob *__restrict__ jobs,
                                                    const TileRef *__restrict__ tiles)
{
	static_assert(C % 16 == 0, "a lane-step must fill whole direction words");
	static_assert(TR % 64 == 0, "hand-off words are flushed every 64 steps");
	constexpr int W = C / 16;

	__shared__ int32_t feed[TR];                                     /* lane-0 input per step   */
	__shared__ __attribute__((aligned(16))) uint8_t rsh[TR + 64];    /* 6*code of the tile rows */

	const TileRef tr = tiles[blockIdx.x];
	const FillJob &J = jobs[tr.job];
	const int lane = threadIdx.x;
	const int s = tr.s;
	const int T0 = tr.a * TR;
	const int L = s * kLanes + lane;
	const int nrows = J.nrows;
	const int upc = J.upc;

	/* ---- stage the tile inputs in LDS ------------------------------------------------ */
	if (s == 0) {
		const int lm = J.leftmul;
		for (int t = lane; t < TR; t += kLanes) feed[t] = lm * (T0 + t + 1);
	} else {
		const int32_t *h = reinterpret_cast<const int32_t *>(arena + J.handoff) + (size_t)(s - 1) * J.hpitch + T0;  /* value after step T-1 */
		for (int t = lane; t < TR; t += kLanes) feed[t] = h[t];
	}
	{
		/* rows r0 = T - L (0-based) for T in [T0, T0+TR), lanes 0..63:  rsh[j] holds row
		 * T0 - 64*s - 64 + j, lane l at local step t reads j = t + 64 - l */
		const uint32_t *src = reinterpret_cast<const uint32_t *>(arena + J.rowshift + (J.padl + T0 - s * kLanes - 64));
		uint32_t *dst = reinterpret_cast<uint32_t *>(rsh);
		for (int j = lane; j < (TR + 64) / 4; j += kLanes) dst[j] = src[j];
	}

	/* ---- per-lane column tables and the row above, in registers ----------------------- */
	uint32_t tab[C];
	int32_t leftc[C];
	int32_t hup[C];
	int32_t in_left, last_clean;
	{
		const uint32_t *ct = reinterpret_cast<const uint32_t *>(arena + J.coltab) + (size_t)L * C;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			tab[c] = ct[c];
			leftc[c] = (int32_t)(((tab[c] >> 24) & 63u) << 2) + 1;
		}
	}
	int32_t *st = reinterpret_cast<int32_t *>(arena + J.state) + (size_t)s * (C + 2) * kLanes + lane;
	if (tr.first) {
		const int32_t *tp = reinterpret_cast<const int32_t *>(arena + J.top) + (size_t)L * C;     /* tp[0] = column left of the lane's first */
		in_left = tp[0];
#pragma unroll
		for (int c = 0; c < C; ++c) hup[c] = tp[c + 1];
		last_clean = hup[C - 1];
	} else {
#pragma unroll
		for (int c = 0; c < C; ++c) hup[c] = st[c * kLanes];
		in_left = st[C * kLanes];
		last_clean = st[(C + 1) * kLanes];
	}
	__syncthreads();

	uint32_t *dirs = reinterpret_cast<uint32_t *>(arena + J.dirs) + ((size_t)s * J.steps_pad + T0) * (W * kLanes) + lane;
	int32_t *hand = reinterpret_cast<int32_t *>(arena + J.handoff) + (size_t)s * J.hpitch + T0 + 1 + lane;
	const uint8_t *myrsh = rsh + 64 - lane;
	const int tf_local = J.tf - T0;
	int32_t coll = 0;

#pragma unroll 2
	for (int t = 0; t < TR; ++t) {
		const int r = T0 + t - L;                       /* 0-based row of this lane at this step */
		const int32_t in_diag = in_left;
		in_left = __builtin_amdgcn_update_dpp(feed[t], last_clean, DPP_WAVE_SHR1, 0xf, 0xf, false);
		uint32_t acc[W];
#pragma unroll
		for (int w = 0; w < W; ++w) acc[w] = 0;

		if ((unsigned)r < (unsigned)nrows) {
			const uint32_t sh = myrsh[t];
			int32_t cd = in_diag;
			int32_t cl = in_left;
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const uint32_t f = __builtin_amdgcn_ubfe(tab[c], sh, 6);
				const int32_t dg = (int32_t)(f << 3) + cd;
				const int32_t up = hup[c] + upc;
				const int32_t lf = cl + leftc[c];
				int32_t h = min(min(dg, up), lf);
				acc[c / 16] = __builtin_amdgcn_alignbit((uint32_t)h, acc[c / 16], 2);
				cd = hup[c];
				h &= ~3;
				hup[c] = h;
				cl = h;
			}
			last_clean = cl;
		}
#pragma unroll
		for (int w = 0; w < W; ++w) dirs[(size_t)t * (W * kLanes) + w * kLanes] = acc[w];

		/* right edge of the strip: collect lane 63's value of 64 steps, store coalesced */
		{
			const int32_t edge = __builtin_amdgcn_readlane(last_clean, 63);
			coll = (lane == (t & 63)) ? edge : coll;
		}
		if ((t & 63) == 63) hand[t - 63] = coll;

		if (t == tf_local && L == J.lf) {
			int32_t *fr = reinterpret_cast<int32_t *>(arena + J.final_row);
#pragma unroll
			for (int c = 0; c < C; ++c) fr[c] = hup[c];
		}
	}

#pragma unroll
	for (int c = 0; c < C; ++c) st[c * kLanes] = hup[c];
	st[C * kLanes] = in_left;
	st[(C + 1) * kLanes] = last_clean;
}

/*
 * K2.  One wave per fill.  The wave pulls a window of 64 steps x 64 lanes of direction
 * words of the current strip into LDS with coalesced loads and all lanes replay the same
 * serial walk on it (uniform control flow, LDS broadcast reads).  Each visited cell emits
 * one op byte; 64 ops are gathered in registers and stored with one coalesced store.
 */
template <int C>
__global__ __launch_bounds__(64) void nw_traceback(uint8_t *__restrict__ arena,
                                                   const FillJob *__restrict__ jobs)
{
	constexpr int W = C / 16;
	constexpr int WIN = 64;
	__shared__ uint32_t win[WIN * W * kLanes];

	const FillJob &J = jobs[blockIdx.x];
	uint8_t *ops = arena + J.ops;
	int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
	const int lane = threadIdx.x;
	int r = J.nrows, k = J.ncols;
	int n = 0;
	uint32_t myop = 0;

	while (r > 0 && k > 0) {
		const int Lg = (k - 1) / C;
		const int s = Lg >> 6;
		const int Ttop = r - 1 + Lg;                           /* newest step in the window */
		const uint32_t *src = reinterpret_cast<const uint32_t *>(arena + J.dirs) + (size_t)s * J.steps_pad * (W * kLanes) + lane;
		for (int j = 0; j < WIN; ++j) {
			const int T = Ttop - j;
#pragma unroll
			for (int w = 0; w < W; ++w)
				win[(j * W + w) * kLanes + lane] = (T >= 0) ? src[((size_t)T * W + w) * kLanes] : 0u;
		}
		__syncthreads();
		while (r > 0 && k > 0) {
			const int kc = k - 1;
			const int Lc = kc / C;
			const int nc = kc - Lc * C;
			if ((Lc >> 6) != s) break;
			const int j = Ttop - (r - 1 + Lc);
			if (j >= WIN) break;
			const uint32_t word = win[(j * W + (nc >> 4)) * kLanes + (Lc & 63)];
			const uint32_t code = (word >> (2 * (nc & 15))) & 3u;
			if (lane == (n & 63)) myop = code;
			++n;
			if ((n & 63) == 0) ops[n - 64 + lane] = (uint8_t)myop;
			if (code == DIR_D) { --r; --k; }
			else if (code == DIR_L) { --k; }
			else { --r; }
		}
		__syncthreads();
	}
	if (lane < (n & 63)) ops[(n & ~63) + lane] = (uint8_t)myop;
	if (lane == 0) {
		/* cost of the final cell -> H[nrows][ncols] (csadp_device.h) */
		const int nf = (J.ncols - 1) % C;
		const int i8 = J.upc - 2;                                /* 8*i */
		const int cost = (J.ncols > 0 && J.nrows > 0) ? reinterpret_cast<const int32_t *>(arena + J.final_row)[nf] : 0;
		summary[0] = n;
		summary[1] = r;
		summary[2] = k;
		summary[3] = ((i8 >> 1) * J.nrows - cost) / 4;
	}
}

/* ---- launch wrappers (host) ----------------------------------------------------------- */

template <int C, int TR>
static hipError_t launch_fill_t(uint8_t *arena, const FillJob *jobs, const TileRef *tiles, int ntiles, hipStream_t st)
{
	hipLaunchKernelGGL((nw_fill_tiles<C, TR>), dim3(ntiles), dim3(kLanes), 0, st, arena, jobs, tiles);
	return hipGetLastError();
}

hipError_t launch_fill(int C, int TR, uint8_t *arena, const FillJob *jobs, const TileRef *tiles, int ntiles,
                       hipStream_t st)
{
	if (ntiles <= 0) return hipSuccess;
	if (C == 16 && TR == 64) return launch_fill_t<16, 64>(arena, jobs, tiles, ntiles, st);
	if (C == 16 && TR == 128) return launch_fill_t<16, 128>(arena, jobs, tiles, ntiles, st);
	if (C == 16 && TR == 256) return launch_fill_t<16, 256>(arena, jobs, tiles, ntiles, st);
	if (C == 32 && TR == 64) return launch_fill_t<32, 64>(arena, jobs, tiles, ntiles, st);
	if (C == 32 && TR == 128) return launch_fill_t<32, 128>(arena, jobs, tiles, ntiles, st);
	if (C == 32 && TR == 256) return launch_fill_t<32, 256>(arena, jobs, tiles, ntiles, st);
	return hipErrorInvalidValue;
}

hipError_t launch_traceback(int C, uint8_t *arena, const FillJob *jobs, int njobs, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	if (C == 16) hipLaunchKernelGGL((nw_traceback<16>), dim3(njobs), dim3(kLanes), 0, st, arena, jobs);
	else if (C == 32) hipLaunchKernelGGL((nw_traceback<32>), dim3(njobs), dim3(kLanes), 0, st, arena, jobs);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

}  // namespace csadp
