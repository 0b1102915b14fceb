/*
 * csadp_cells_tb.hip -- the direction walk of dynamicprogramming.c:1037-1047 over the 2-bit tags nw_fill_cells
 * (csadp_cells.hip) leaves in HBM.  gfx950, wave64.
 *
 * The walk is a chain of nrows + ncols dependent look-ups; one wave doing it alone took 0.6-1.25 ms for the
 * 16.5 k x 16.8 k profile steps of the reference's Set3 (75 ns per op), 40 % of such a step.  It is cut into bands of
 * kBandRows rows whose walks are independent once the column in which the path ENTERS each band is known:
 *
 *   nw_tb_scout     K2e  every band but the bottom one is walked from a start column every kBandStride columns, by one
 *                        lane each, out of direction words staged in LDS.  Walks do not cross (from a cell the path is
 *                        a function of the cell) and stay merged once they meet: when the two starts that flank a
 *                        column leave the band in the same column, so does the walk from that column.
 *   nw_tb_resolve   K2f  one workgroup per matrix follows the path band by band: a look-up in the scouts' table
 *                        where the flanks merged (the scouted corridor of up to 512 bands staged in LDS), a run-batched
 *                        exact walk through a band where they did not, and through the bottom band.  Matrices too
 *                        small for bands are walked here in one go, ops included.
 *   nw_tb_emit      K2g  one wave per band walks from the known entry column and writes the band's ops where an
 *                        upper bound of the ops below puts them: (rows below) + (columns to the right).
 *   nw_tb_gather    K2h  closes the holes: suffix sums of the bands' op counts, one copy.
 *
 * Cell (j, k), 1-based: column c = k - 1 lives in strip S = c / 128, lane (c % 128) / 2, half h = c % 2 (column A or B of
 * the lane), at local step l = (j - 1) + lane.  Word l / 16 of "virtual strip" 2S + h holds its tag at bits 2 * (l % 16).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

namespace {

static_assert(kCellStripCols == 128, "the walk's shifts assume 128 columns per strip");
static_assert(kBandRows % 16 == 0 && (kBandStride & (kBandStride - 1)) == 0, "bands start on word boundaries; the stride is a shift");

/* ---- K2d: the run-batched window walk ------------------------------------------------------------------------------
 * Along a diagonal move the row falls by one and the lane by one every second column, so while the path crosses one
 * strip (128 columns) l falls by ~192 = 12 words: the LDS window holds, for each of the kTbStrips strips left of the
 * current cell and both halves, the kTbWords words around the expected crossing -- 4 x 2 x 32 x 64 words = 64 KiB,
 * loaded as whole 256-byte rows by the four waves.  Wave 0 then walks run-batched (lane i looks at cell (j-i, k-i), a
 * ballot finds the end of the run of 'D'); leaving the window just reloads it around the current cell. */
constexpr int kTbStrips = 4;
constexpr int kTbWords = 32;
constexpr int kTbSlack = 8;          /* words above the expected entry point of a strip */
constexpr int kTbExtra = 2;          /* rounds taken from the fetched words after a look-up's own */
constexpr int kTbWinWords = kTbStrips * kCellCols * kTbWords * kLanes;

/* Walks from (j, k) until row jstop or column 0 is reached; rows <= jstop are never looked at.  All 256 threads call it;
 * on return every thread holds the new j, k, n.  EMIT: ops[n...] receive the moves. */
template <bool EMIT>
__device__ void walk_window(uint32_t *win, int *wlo, int *pos, const uint32_t *__restrict__ dirs, int wpitch, uint8_t *ops,
                            int &j, int &k, int &n, int jstop)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	while (j > jstop && k > 0) {
		const int s0 = (k - 1) >> 7;                              /* kCellStripCols = 128 */
		if (tid < kTbStrips) {
			/* strip s0 - tid: the path is expected at its right edge (column 128*s + 127, lane 63) in row
			 * j - (k - 1 - that column); for the current strip that point is extrapolated to the right */
			const int sB = s0 - tid;
			const int jedge = j - ((k - 1) - (kCellStripCols * sB + kCellStripCols - 1));
			wlo[tid] = ((jedge - 1 + (kLanes - 1)) >> 4) + kTbSlack - (kTbWords - 1);
		}
		__syncthreads();
		/* kTbStrips * 2 * kTbWords rows of 256 bytes = 16 uint4 per row; slot = strip block * 2 + half */
		for (int e = tid; e < kTbStrips * kCellCols * kTbWords * 16; e += 256) {
			const int slot = e / (kTbWords * 16), u = (e / 16) % kTbWords, q = e % 16;
			const int sB = s0 - slot / kCellCols, w = wlo[slot / kCellCols] + u;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (sB >= 0 && w >= 0 && w < wpitch)
				v = *reinterpret_cast<const uint4 *>(dirs + ((size_t)(sB * kCellCols + slot % kCellCols) * wpitch + w) * kLanes + 4 * q);
			reinterpret_cast<uint4 *>(win)[e] = v;
		}
		__syncthreads();
		if (wave == 0) {
			/* lane b < kTbStrips keeps wlo[b] in a register: the 64 cells of an iteration touch at most two
			 * strips, whose window origins are fetched with v_readlane instead of a second LDS round trip */
			const int wreg = lane < kTbStrips ? wlo[lane] : 0;
			for (;;) {
				/* one iteration = one LDS look-up per lane, straight-line: lane i looks at cell (j - i, k - i) */
				const int kc = k - 1;                           /* 0-based column of lane 0's cell */
				const int Bk = s0 - (kc >> 7);                  /* its strip block (wave-uniform) */
				if (Bk >= kTbStrips) break;
				const int wloA = __builtin_amdgcn_readlane(wreg, Bk);
				const int wloB = __builtin_amdgcn_readlane(wreg, Bk + 1 < kTbStrips ? Bk + 1 : Bk);
				const int ri = j - lane, kz = kc - lane;        /* row (1-based), column (0-based): outside when <= jstop / < 0 */
				const int sc = kz >> 7;                         /* arithmetic: negative columns give a strip that fails the tests below */
				const int B = s0 - sc;
				const int ln = (kz & (kCellStripCols - 1)) >> 1;   /* the lane that owns the column */
				const int l = ri - 1 + ln;                      /* local step of the cell in its strip */
				const int u = (l >> 4) - (sc == (kc >> 7) ? wloA : wloB);
				const bool ok = (ri > jstop) & (kz >= 0) & (B < kTbStrips) & ((unsigned)u < (unsigned)kTbWords);
				const uint32_t w = win[ok ? ((B * kCellCols + (kz & 1)) * kTbWords + u) * kLanes + ln : 0];
				const uint32_t code = ok ? (w >> (2 * (l & 15))) & 3u : 3u;     /* 3 = stop: border, band top or outside the window */
				/* a run of 'D' and the gap move that ends it are taken in ONE iteration, written by ONE store */
				const unsigned long long stop = __ballot(code != DIR_D);
				const int run = stop ? __builtin_ctzll(stop) : kLanes;
				const uint32_t c0 = run < kLanes ? (uint32_t)__builtin_amdgcn_readlane((int)code, run) : 3u;
				const int gap = c0 != 3u;
				if (EMIT && lane < run + gap) ops[n + lane] = (uint8_t)(lane < run ? (uint32_t)DIR_D : c0);
				n += run + gap;
				j -= run + (gap & (c0 != DIR_L));
				k -= run + (c0 == DIR_L);
				if (run + gap == 0) break;                      /* border reached or window left: the outer loop decides */
				/* More rounds out of the SAME words: after a gap move the cells of the new diagonal are the fetched
				 * columns one row up (after U) or down (after L) -- in the same word 15 times out of 16, since a word
				 * holds 16 consecutive rows of its column.  Lane q >= p looks at its column again, p = columns consumed,
				 * delta = rows the diagonal has drifted.  No look-up, ~1/3 of an iteration's cost; kTbExtra rounds at
				 * most (every further one finds fewer of its tags in the fetched words). */
				int p = run + (c0 == DIR_L), delta = 0;
				uint32_t last = c0;
				bool more = gap != 0;
#pragma unroll
				for (int extra = 0; extra < kTbExtra; ++extra) {
					if (!more || p >= kLanes) break;
					delta += last == DIR_L ? 1 : -1;
					const int l2 = l + delta;
					const bool ok2 = ok & (lane >= p) & (ri + delta > jstop) & ((l2 >> 4) == (l >> 4));
					const uint32_t code2 = ok2 ? (w >> (2 * (l2 & 15))) & 3u : 3u;
					const unsigned long long stop2 = __ballot(code2 != DIR_D) >> p;
					const int left = kLanes - p;
					const int run2 = stop2 ? __builtin_ctzll(stop2) : left;
					const uint32_t c2 = run2 < left ? (uint32_t)__builtin_amdgcn_readlane((int)code2, p + run2) : 3u;
					const int gap2 = c2 != 3u;
					const int i2 = lane - p;
					if (EMIT && i2 >= 0 && i2 < run2 + gap2) ops[n + i2] = (uint8_t)(i2 < run2 ? (uint32_t)DIR_D : c2);
					n += run2 + gap2;
					j -= run2 + (gap2 & (c2 != DIR_L));
					k -= run2 + (c2 == DIR_L);
					p += run2 + (c2 == DIR_L);
					last = c2;
					more = gap2 != 0;
				}
			}
			if (lane == 0) {
				pos[0] = j;
				pos[1] = k;
				pos[2] = n;
			}
		}
		__syncthreads();
		j = pos[0];
		k = pos[1];
		n = pos[2];
		__syncthreads();
	}
}

/* ---- the direction words of one band, staged in LDS: [nS strips][2 halves][kBandWords][64 lanes], strips sLo .. sLo + nS - 1
 * (sLo may be negative: those strips hold zeros nobody reads), words from wLo on ------------------------------------------------ */

__device__ __forceinline__ void stage_band(uint32_t *lds, const uint32_t *__restrict__ dirs, int wpitch, int nstrips, int sLo, int nS, int wLo,
                                           int tid, int nthreads)
{
	/* uint4 units, 16 per row of 64 lanes; kStageDepth loads of a thread are in flight before the first is stored (one at
	 * a time, each waited for, made the staging the longest part of all three kernels that use it) */
	constexpr int kStageDepth = 8;
	const int total = nS * kCellCols * kBandWords * 16;
	for (int e0 = tid; e0 < total; e0 += nthreads * kStageDepth) {
		uint4 v[kStageDepth];
#pragma unroll
		for (int x = 0; x < kStageDepth; ++x) {
			const int e = e0 + x * nthreads;
			const int slot = e / (kBandWords * 16), u = (e / 16) % kBandWords, q = e % 16;
			const int sB = sLo + slot / kCellCols, w = wLo + u;
			v[x] = make_uint4(0, 0, 0, 0);
			if (e < total && sB >= 0 && sB < nstrips && w < wpitch)
				v[x] = *reinterpret_cast<const uint4 *>(dirs + ((size_t)(sB * kCellCols + slot % kCellCols) * wpitch + w) * kLanes + 4 * q);
		}
#pragma unroll
		for (int x = 0; x < kStageDepth; ++x) {
			const int e = e0 + x * nthreads;
			if (e < total) reinterpret_cast<uint4 *>(lds)[e] = v[x];
		}
	}
}

/* The scouts of band b cover J.tb_groups groups of kScoutStarts start columns around the straight line between the
 * matrix' corners (all groups when there are no more): first group of the band.  Starts outside count as unknown. */
__device__ __forceinline__ int scout_first_group(const CellJob &J, int b)
{
	const int ngroups = (J.ncols / kBandStride + 1 + kScoutStarts - 1) / kScoutStarts;
	if (J.tb_groups >= ngroups) return 0;
	const int centre = (int)((long long)(b + 1) * kBandRows * J.ncols / J.nrows) / (kScoutStarts * kBandStride);
	return min(max(centre - J.tb_groups / 2, 0), ngroups - J.tb_groups);
}

/* ---- the run-batched walk through ONE band (rows jstop + 1 .. jstop + kBandRows at most) out of a staged window of nS
 * strips: lane i looks at cell (j - i, k - i), a ballot finds the end of the run of 'D', the gap move that ends it is
 * taken in the same iteration (as K2d above, without its per-strip window origins: a band's words are the same for every
 * strip).  Called by all NT threads of the workgroup; wave 0 walks.  Leaving the window restages it around the cell. */
template <bool EMIT, int NT>
__device__ void walk_band(uint32_t *lds, int nS, int *pos, const uint32_t *__restrict__ dirs, int wpitch, int nstrips, uint8_t *out,
                          int &j, int &k, int &n, int jstop)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wLo = jstop >> 4;
	while (j > jstop && k > 0) {
		const int sLo = ((k - 1) >> 7) - (nS - 1);
		stage_band(lds, dirs, wpitch, nstrips, sLo, nS, wLo, tid, NT);
		__syncthreads();
		if (wave == 0) {
			for (;;) {
				const int kc = k - 1;
				if ((kc >> 7) < sLo) break;                      /* wave-uniform: lane 0's cell lies left of the window */
				const int ri = j - lane, kz = kc - lane;
				const int sc = kz >> 7;
				const int ln = (kz & (kCellStripCols - 1)) >> 1;
				const int l = ri - 1 + ln;
				const unsigned ds = (unsigned)(sc - sLo), dw = (unsigned)((l >> 4) - wLo);
				const bool ok = (ri > jstop) & (kz >= 0) & (ds < (unsigned)nS) & (dw < (unsigned)kBandWords);
				const uint32_t w = lds[ok ? ((ds * kCellCols + (kz & 1)) * kBandWords + dw) * kLanes + ln : 0];
				const uint32_t code = ok ? (w >> (2 * (l & 15))) & 3u : 3u;
				const unsigned long long stop = __ballot(code != DIR_D);
				const int run = stop ? __builtin_ctzll(stop) : kLanes;
				const uint32_t c0 = run < kLanes ? (uint32_t)__builtin_amdgcn_readlane((int)code, run) : 3u;
				const int gap = c0 != 3u;
				if (EMIT && lane < run + gap) out[n + lane] = (uint8_t)(lane < run ? (uint32_t)DIR_D : c0);
				n += run + gap;
				j -= run + (gap & (c0 != DIR_L));
				k -= run + (c0 == DIR_L);
				if (run + gap == 0) break;
				int p = run + (c0 == DIR_L), delta = 0;
				uint32_t last = c0;
				bool more = gap != 0;
#pragma unroll
				for (int extra = 0; extra < kTbExtra; ++extra) {
					if (!more || p >= kLanes) break;
					delta += last == DIR_L ? 1 : -1;
					const int l2 = l + delta;
					const bool ok2 = ok & (lane >= p) & (ri + delta > jstop) & ((l2 >> 4) == (l >> 4));
					const uint32_t code2 = ok2 ? (w >> (2 * (l2 & 15))) & 3u : 3u;
					const unsigned long long stop2 = __ballot(code2 != DIR_D) >> p;
					const int left = kLanes - p;
					const int run2 = stop2 ? __builtin_ctzll(stop2) : left;
					const uint32_t c2 = run2 < left ? (uint32_t)__builtin_amdgcn_readlane((int)code2, p + run2) : 3u;
					const int gap2 = c2 != 3u;
					const int i2 = lane - p;
					if (EMIT && i2 >= 0 && i2 < run2 + gap2) out[n + i2] = (uint8_t)(i2 < run2 ? (uint32_t)DIR_D : c2);
					n += run2 + gap2;
					j -= run2 + (gap2 & (c2 != DIR_L));
					k -= run2 + (c2 == DIR_L);
					p += run2 + (c2 == DIR_L);
					last = c2;
					more = gap2 != 0;
				}
			}
			if (lane == 0) {
				pos[0] = j;
				pos[1] = k;
				pos[2] = n;
			}
		}
		__syncthreads();
		j = pos[0];
		k = pos[1];
		n = pos[2];
		__syncthreads();
	}
}

/* ---- K2e ---------------------------------------------------------------------------------------------------------- */

__global__ __launch_bounds__(256) void nw_tb_scout(uint8_t *__restrict__ arena, const CellJob *__restrict__ jobs)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];     /* guard + (1 + kScoutStrips * 128) columns x kScoutPitch words + 4 per strip */
	const CellJob &J = jobs[blockIdx.z];
	const int b = blockIdx.y;
	if (!J.banded || b >= J.nbands - 1 || (int)blockIdx.x >= J.tb_groups) return;   /* the bottom band is entered in column ncols: walked exactly */
	const int g = scout_first_group(J, b) + blockIdx.x;
	const int nstarts = J.ncols / kBandStride + 1;                     /* start i sits in column i * kBandStride <= ncols */
	const int i0 = g * kScoutStarts;
	if (i0 >= nstarts) return;
	const uint32_t *dirs = reinterpret_cast<const uint32_t *>(arena + J.dirs);
	const int wpitch = J.steps_pad / 16;
	const int ilast = min(i0 + kScoutStarts, nstarts) - 1;
	const int sLo = ((ilast * kBandStride - 1) >> 7) - (kScoutStrips - 1);   /* the strip of the last start's cell and those left of it */
	const int wLo = (b * kBandRows) >> 4;                                     /* first direction word of the band's rows */
#ifdef CSADP_TB_STATS
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
	/* The scouts' window is column-major with a strip skew: word w of the band (0 .. kBandWords - 1) of window column
	 * x = c - 128 sLo sits at [kScoutGuard + (x + 1) * kScoutPitch + 4 (x >> 7) + w].  With l' = j - jtop - 1 + (x >> 1) --
	 * the local step plus 64 per strip -- that is (x + 1) * pitch + (l' >> 4): the step's address is a shift, an add, a shift and
	 * a multiply-add.  Bank of start t (16 columns apart, lock-step): 16 (t & 1) + ((t & 7) >> 1) + 4 (t >> 3) + const -- the
	 * 32 lanes of a read group on 32 banks (round 3: rows of 128 columns per strip and word, 64 lanes on 4 banks, 19.5
	 * conflict cycles per LDS instruction; profiles/r03_pmc_summary.json).
	 * The step has no validity tests: every tag a scout must not follow reads as 3 = "stay" -- rows at or above the band's
	 * top (masked while staging: a lane that arrives in row jtop parks there = success), strips left of the matrix, the
	 * slots between columns, and a guard column at x = -1 that lanes leaving the window on the left are clamped to
	 * (parked below jtop = unknown).  Round 3's step carried five boolean masks through VCC and back: ~45 instructions,
	 * 480 cycles; this one is 10 dependent instructions and the read. */
	{
		constexpr int kDepth = 8;
		const int total = kScoutStrips * kCellCols * kBandWords * 16;     /* uint4 units of 4 lanes of one half */
		for (int e = threadIdx.x; e < kScoutGuard + kScoutPitch; e += 256) lds[e] = ~0u;   /* guard column and the slots below column 0 */
		for (int e0 = threadIdx.x; e0 < total; e0 += 256 * kDepth) {
			uint4 v[kDepth];
#pragma unroll
			for (int x = 0; x < kDepth; ++x) {
				const int e = e0 + x * 256;
				const int slot = e / (kBandWords * 16), u = (e / 16) % kBandWords, q = e % 16;
				const int sB = sLo + slot / kCellCols, w = wLo + u;
				v[x] = make_uint4(~0u, ~0u, ~0u, ~0u);                       /* left of the matrix: stay */
				if (e < total && sB >= 0) v[x] = make_uint4(0, 0, 0, 0);
				if (e < total && sB >= 0 && sB < J.nstrips && w < wpitch)
					v[x] = *reinterpret_cast<const uint4 *>(dirs + ((size_t)(sB * kCellCols + slot % kCellCols) * wpitch + w) * kLanes + 4 * q);
			}
#pragma unroll
			for (int x = 0; x < kDepth; ++x) {
				const int e = e0 + x * 256;
				if (e >= total) continue;
				const int slot = e / (kBandWords * 16), u = (e / 16) % kBandWords, q = e % 16;
				const int strip = slot / kCellCols;
				/* lanes 4 q .. 4 q + 3 of half h own columns 128 strip + 2 lane + h; tags of rows <= jtop: local steps
				 * l - 16 wLo <= lane - 1, i.e. the first (lane - 16 u) tags of word u */
				uint32_t *dst = lds + kScoutGuard + (strip * kCellStripCols + 8 * q + (slot % kCellCols) + 1) * kScoutPitch + 4 * strip + u;
				const uint32_t vv[4] = {v[x].x, v[x].y, v[x].z, v[x].w};
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int cnt = min(max(4 * q + i - 16 * u, 0), 16);
					const uint32_t stay = cnt >= 16 ? ~0u : (1u << (2 * cnt)) - 1u;
					dst[2 * i * kScoutPitch] = vv[i] | stay;
					if (u == kBandWords - 1) dst[2 * i * kScoutPitch + 1] = ~0u;        /* the slot behind the column's last word */
				}
				if (u == 0 && q == 0 && (slot % kCellCols) == 0) {                 /* the four slots the strip skew leaves below a strip's first column */
					dst[-1] = ~0u;
					dst[-2] = ~0u;
					dst[-3] = ~0u;
					dst[-4] = ~0u;
				}
			}
		}
	}
	__syncthreads();
#ifdef CSADP_TB_STATS
	const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
	int nsteps = 0;
#endif
	const int t = threadIdx.x;
	if (t >= kScoutStarts) return;                                     /* one wave walks, a start per lane */
	const int i = i0 + t;
	const int jtop = b * kBandRows;
	/* j = row, x = window column = (column - 1) - 128 sLo.  A lane parks -- its tag reads 3 -- in row jtop (the band is crossed:
	 * its result), in column 0 of the matrix, or in the guard column (it left the staged strips: unknown).  kScoutCap steps are
	 * kBandRows rows and kScoutMaxLeft L moves: a start right of the path on its slow way to it is still below jtop then: unknown */
	const int xoff = kCellStripCols * sLo;
	int j = jtop + kBandRows, x = (i < nstarts ? i * kBandStride : 0) - 1 - xoff;
	const int jb = -1 - jtop;
	for (int steps = 0; steps < kScoutCap; steps += 16) {
		uint32_t moved = 0;
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			const int xc = max(x, -1);
			const int l = j + jb + (xc >> 1);                              /* local step relative to the band's first word, + 64 per window strip */
			const uint32_t word = lds[kScoutGuard + kScoutPitch + __mul24(xc, kScoutPitch) + (l >> 4)];   /* 24-bit multiply-add: v_mul_lo_u32 is quarter rate */
			const uint32_t tag = (word >> (2 * (l & 15))) & 3u;
			j -= (int)((5u >> tag) & 1u);                                 /* U, D: a row up;  L, stay: not */
			x -= (int)((6u >> tag) & 1u);                                 /* L, D: a column left */
			moved |= tag ^ 3u;
		}
#ifdef CSADP_TB_STATS
		nsteps += 16;
#endif
		if (!__any(moved != 0)) break;
	}
	const int c = x + xoff;
#ifdef CSADP_TB_STATS
	if (t == 0 && blockIdx.x == 1 && b == J.nbands / 2 && J.nbands > 100)
		printf("scout: staged in %.1f us, %d steps in %.1f us\n", (double)(t1 - t0) / 100.0, nsteps, (double)(__builtin_amdgcn_s_memrealtime() - t1) / 100.0);
#endif
	const int k = c + 1;
	const unsigned moved = (unsigned)(i * kBandStride - k);
	uint16_t *tab = reinterpret_cast<uint16_t *>(arena + J.tb_tab) + (size_t)b * J.tb_pitch + blockIdx.x * kScoutStarts;
	tab[t] = (uint16_t)((j == jtop && k > 0 && moved < kBandMoved) ? moved : kBandUnknown);
}

/* ---- K2f ---------------------------------------------------------------------------------------------------------- */

constexpr int kResolveTabBytes = 64 * 1024;
constexpr int kResolveChunk = 512;

__global__ __launch_bounds__(256) void nw_tb_resolve(uint8_t *__restrict__ arena, const CellJob *__restrict__ jobs)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
	uint32_t *win = dyn;                                               /* kTbWinWords */
	uint16_t *tab = reinterpret_cast<uint16_t *>(dyn + kTbWinWords);   /* kResolveTabBytes, banded jobs only */
	__shared__ int wlo[kTbStrips];
	__shared__ int pos[3];
	__shared__ int first[kResolveChunk];                               /* first scouted start of each loaded band */

	const CellJob &J = jobs[blockIdx.x];
	const uint32_t *dirs = reinterpret_cast<const uint32_t *>(arena + J.dirs);
	const int wpitch = J.steps_pad / 16;
	const int tid = threadIdx.x;
	int j = J.nrows, k = J.ncols, n = 0;
	if (!J.banded) {
		walk_window<true>(win, wlo, pos, dirs, wpitch, arena + J.ops, j, k, n, 0);
		if (tid == 0) {
			int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
			summary[0] = n;
			summary[1] = j;
			summary[2] = k;
			summary[3] = 0;
		}
		return;
	}
	const int nb = J.nbands, pitch = J.tb_pitch;
	int32_t *ent = reinterpret_cast<int32_t *>(arena + J.tb_ent);
	const uint16_t *gtab = reinterpret_cast<const uint16_t *>(arena + J.tb_tab);
	for (int b = tid; b < nb; b += 256) ent[b] = -1;
	const int chunk = min(kResolveChunk, max(1, kResolveTabBytes / (pitch * 2)));      /* bands whose table rows fit the LDS */
	int cLo = nb, cHi = -1;                                            /* loaded bands */
#ifdef CSADP_TB_STATS
	int walked = 0, loads = 0, outside = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long twalk = 0;
#endif
	__syncthreads();
	while (j > 0 && k > 0) {
		const int b = (j - 1) / kBandRows;                               /* j is the bottom row of band b: (b+1)*R, or nrows */
		if (tid == 0) ent[b] = k;
		if (b < nb - 1) {
			if (b < cLo || b > cHi) {
				__syncthreads();                                           /* nobody still reads the old rows */
				cHi = b;
				cLo = max(0, b - chunk + 1);
#ifdef CSADP_TB_STATS
				++loads;
#endif
				/* pitch is a multiple of 8 entries: 16-byte units */
				const uint4 *src = reinterpret_cast<const uint4 *>(gtab + (size_t)cLo * pitch);
				const int units = (cHi - cLo + 1) * (pitch / 8);
				for (int e = tid; e < units; e += 256) reinterpret_cast<uint4 *>(tab)[e] = src[e];
				for (int e = tid; e <= cHi - cLo; e += 256) first[e] = scout_first_group(J, cLo + e) * kScoutStarts;
				__syncthreads();
			}
			if (tid == 0) {
				int jj = j, kk = k, bb = b;
				int f = first[bb - cLo];
				const uint16_t *row = tab + (bb - cLo) * pitch;
				for (;;) {
					const int fnext = first[max(bb - 1 - cLo, 0)];         /* read beside the row's entry, not after it */
					const int lo = kk / kBandStride, hi = lo + ((kk % kBandStride) != 0);
					const unsigned xlo = (unsigned)(lo - f), xhi = (unsigned)(hi - f);
					if (xlo >= (unsigned)pitch || xhi >= (unsigned)pitch) break;   /* outside the scouted corridor (negative: huge) */
					const unsigned a = row[xlo], c = row[xhi];             /* both reads in flight together: one LDS latency per band */
					if (a == kBandUnknown || c == kBandUnknown) break;     /* (start columns beyond ncols are written as unknown) */
					const int ea = lo * kBandStride - (int)a;
					if (ea != hi * kBandStride - (int)c) break;            /* the flanks have not merged inside the band */
					kk = ea;
					jj = bb * kBandRows;
					--bb;
					if (jj == 0) break;
					ent[bb] = kk;
					if (bb < cLo) break;
					f = fnext;
					row -= pitch;
				}
				pos[0] = jj;
				pos[1] = kk;
			}
			__syncthreads();
			const int jn = pos[0], kn = pos[1];
			__syncthreads();
			if (jn != j) {
				j = jn;
				k = kn;
				continue;
			}
		}
#ifdef CSADP_TB_STATS
		++walked;
		{
			const int g0 = scout_first_group(J, b), lo = k / kBandStride;
			if (b < nb - 1 && (lo / kScoutStarts < g0 || (lo + 1) / kScoutStarts >= g0 + J.tb_groups)) ++outside;
		}
		const unsigned long long tw = __builtin_amdgcn_s_memrealtime();
#endif
		walk_band<false, 256>(win, kEmitStrips, pos, dirs, wpitch, J.nstrips, nullptr, j, k, n, b * kBandRows);
#ifdef CSADP_TB_STATS
		twalk += __builtin_amdgcn_s_memrealtime() - tw;
#endif
	}
#ifdef CSADP_TB_STATS
	if (tid == 0 && nb > 16)
		printf("resolve %d x %d: %d bands, %d walked (%d outside the corridor), %d table loads, %.1f us of %.1f walking\n", J.nrows, J.ncols, nb,
		       walked, outside, loads, (double)twalk / 100.0, (double)(__builtin_amdgcn_s_memrealtime() - t0) / 100.0);
#endif
	if (tid == 0) {
		ent[nb] = j;
		ent[nb + 1] = k;
	}
}

/* ---- K2g ---------------------------------------------------------------------------------------------------------- */

__global__ __launch_bounds__(64) void nw_tb_emit(uint8_t *__restrict__ arena, const CellJob *__restrict__ jobs)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[kEmitStrips * kCellCols * kBandWords * kLanes];
	const CellJob &J = jobs[blockIdx.y];
	const int b = blockIdx.x;
	if (!J.banded || b >= J.nbands) return;
	const int32_t *ent = reinterpret_cast<const int32_t *>(arena + J.tb_ent);
	int32_t *cnt = reinterpret_cast<int32_t *>(arena + J.tb_cnt);
	const int kin = ent[b];
	if (kin <= 0) {                                                     /* the path ended below this band */
		if (threadIdx.x == 0) cnt[b] = 0;
		return;
	}
	const uint32_t *dirs = reinterpret_cast<const uint32_t *>(arena + J.dirs);
	const int wpitch = J.steps_pad / 16;
	const int jtop = b * kBandRows, jin = min(jtop + kBandRows, J.nrows);
	__shared__ int pos[3];
	uint8_t *out = arena + J.tb_scratch + (size_t)(J.nrows - jin) + (size_t)(J.ncols - kin);
	int j = jin, k = kin, n = 0;
	walk_band<true, 64>(lds, kEmitStrips, pos, dirs, wpitch, J.nstrips, out, j, k, n, jtop);
	if (threadIdx.x == 0) cnt[b] = n;
}

/* ---- K2h ---------------------------------------------------------------------------------------------------------- */

__global__ __launch_bounds__(256) void nw_tb_gather(uint8_t *__restrict__ arena, const CellJob *__restrict__ jobs)
{
	const CellJob &J = jobs[blockIdx.y];
	if (!J.banded) return;
	const int nb = J.nbands, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int b = blockIdx.x * 4 + wave;                               /* a wave per band */
	if (b >= nb) return;
	const int32_t *ent = reinterpret_cast<const int32_t *>(arena + J.tb_ent);
	const int32_t *cnt = reinterpret_cast<const int32_t *>(arena + J.tb_cnt);
	/* walk order runs from the bottom band up: the ops of band b follow those of all bands below it (b' > b) */
	int at = 0;
	for (int bb = b + 1 + lane; bb < nb; bb += kLanes) at += cnt[bb];
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) at += __shfl_xor(at, d);
	const int kin = ent[b], m = cnt[b];
	if (kin > 0) {
		const int jin = min((b + 1) * kBandRows, J.nrows);
		const uint8_t *src = arena + J.tb_scratch + (size_t)(J.nrows - jin) + (size_t)(J.ncols - kin);
		uint8_t *ops = arena + J.ops;
		for (int e = lane; e < m; e += kLanes) ops[at + e] = src[e];
	}
	if (b == 0 && lane == 0) {
		int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
		summary[0] = at + m;
		summary[1] = ent[nb];
		summary[2] = ent[nb + 1];
		summary[3] = 0;
	}
}

}  // namespace

int traceback_cells_lds_bytes(bool banded) { return kTbWinWords * 4 + (banded ? kResolveTabBytes : 0); }

constexpr int kScoutLdsBytes = (kScoutGuard + (kScoutStrips * kCellStripCols + 1) * kScoutPitch + 4 * kScoutStrips) * 4;

hipError_t configure_traceback_cells()
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(nw_tb_resolve), hipFuncAttributeMaxDynamicSharedMemorySize,
	                                   traceback_cells_lds_bytes(true));
	if (e != hipSuccess) return e;
	return hipFuncSetAttribute(reinterpret_cast<const void *>(nw_tb_scout), hipFuncAttributeMaxDynamicSharedMemorySize, kScoutLdsBytes);
}

hipError_t launch_traceback_cells(uint8_t *arena, const CellJob *jobs, int njobs, int max_bands, int max_groups, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	const bool banded = max_bands > 0;
	/* the banded kernels take their job from the grid's y / z, which end at 65535: more jobs than that go in several launches */
	constexpr int kGridYZ = 65535;
	if (banded && max_bands > 1 && max_groups > 0)
		for (int j0 = 0; j0 < njobs; j0 += kGridYZ)
			hipLaunchKernelGGL(nw_tb_scout, dim3(max_groups, max_bands - 1, std::min(njobs - j0, kGridYZ)), dim3(256), kScoutLdsBytes, st, arena, jobs + j0);
	hipLaunchKernelGGL(nw_tb_resolve, dim3(njobs), dim3(256), traceback_cells_lds_bytes(banded), st, arena, jobs);
	if (banded)
		for (int j0 = 0; j0 < njobs; j0 += kGridYZ) {
			const int nj = std::min(njobs - j0, kGridYZ);
			hipLaunchKernelGGL(nw_tb_emit, dim3(max_bands, nj), dim3(64), 0, st, arena, jobs + j0);
			hipLaunchKernelGGL(nw_tb_gather, dim3((max_bands + 3) / 4, nj), dim3(256), 0, st, arena, jobs + j0);
		}
	return hipGetLastError();
}

}  // namespace csadp
