/*
 * csadp_cells.hip -- the matrix fill of dynamicprogramming.c:990-1029 for ANY progressive step (i >= 1,
 * stale borders included) as a persistent cell-per-lane wavefront, and its direction walk
 * (dynamicprogramming.c:1037-1047).  gfx950, wave64.
 *
 *   nw_fill_cells<WIDE>    K1c: a lane owns one column, a wave 64 columns, a workgroup kCellWaves strips
 *   nw_traceback_cells     K2d: the run-batched walk over K1c's direction words
 *
 * Why this shape.  The reference's own use of the DP (mode N) is a handful of wide gaps, each a chain
 * of up to 63 strictly sequential profile fills: what counts is the LATENCY of one fill, and a fill's
 * critical path is its nrows + ncols anti-diagonals.  The tiled kernel (csadp_kernels.hip) gives a
 * lane 16 columns x 2 rows per step -- ~200 dependent-issue-bound instructions -- and needs
 * nrows/2 + ncols/16 such steps; here a step is ONE cell (8 VALU instructions: two DPP moves, the
 * table lookup, two additions, max3, the direction shift, the tag mask) and a matrix takes
 * nrows + ncols of them, on one wave per SIMD so that nothing else competes for the issue slot.
 * Gain form and tie-break as in csadp_device.h: X = 4*H + 4*i*r, candidates tagged U 0 / L 1 / D 2,
 * one v_max3_i32 yields the reference's H and the reference's direction (D >= L >= U, :1014-1025).
 *
 * Data flow.  At local step l lane L of a strip works on row l - L + 1; the value and the letter
 * offset of that row come from lane L-1 (v_mov_b32_dpp wave_shr:1), which had the row one step
 * earlier.  Lane 0 takes them from LDS (`inject`, prepared per block of 32 steps): the border column
 * and the row letters for the first strip of a job, else the words the previous strip's lane 63
 * left in the LDS ring 63 steps earlier (same workgroup) or in `hand` in HBM (previous chunk,
 * published block by block through an agent-scope counter).  Directions: 16 steps of 2-bit tags per
 * word and lane, one coalesced 256-byte store per wave every 16 steps = 0.25 B/cell, the algorithmic
 * figure of SURVEY 8(d).
 *
 * Waits.  Inside a workgroup all waves are resident, so ring waits always end.  Across workgroups the
 * launch relies on the work list (a job's chunks in ascending order) being dispatched in order -- for
 * speed only: every spin is bounded, a time-out raises the abort word and the host repeats the pass
 * chunk by chunk (csadp_engine.cpp: check_abort), where every producer has finished before its
 * consumer starts.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

namespace {

constexpr int kRing = 8;                          /* hand-off blocks buffered per strip boundary */
constexpr int kRingSteps = kRing * kCellBlock;
constexpr int kSpinMax = 1 << 22;
constexpr int DPP_WAVE_SHR1 = 0x138;

__device__ __forceinline__ bool wait_lds(const int *counter, int need)
{
	int spins = 0;
	while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
		__builtin_amdgcn_s_sleep(1);
		if (++spins > kSpinMax) return false;
	}
	return true;
}

__device__ __forceinline__ bool wait_hbm(const int *counter, int need)
{
	int spins = 0;
	while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
		__builtin_amdgcn_s_sleep(4);
		if (++spins > kSpinMax) return false;
	}
	return true;
}

struct CellState {
	int32_t hup;        /* X of the cell above (this column, previous row)              */
	int32_t diag;       /* X of the cell above-left = what came from the left last step */
	int32_t outv;       /* this lane's X of the current row, tag cleared                */
	uint32_t outs;      /* the letter offset of the row this lane has just worked on    */
};

/* 32 steps.  inject[t] = (X, letter offset) entering lane 0 at step t; every lane stores what it hands
 * to the right to lanebuf[t] (the ring for lane 63, a scrap area for all others: no EXEC change);
 * dirs points at this lane's word of the block's first 16 steps. */
template <bool WIDE, bool RAMP>
__device__ __forceinline__ void cell_block(CellState &S, uint32_t tab, int32_t leftc, const uint2 *inject, uint2 *lanebuf,
                                           uint32_t *dirs, int l0, int lane)
{
	uint32_t ioff = 0;
	asm volatile("" : "+v"(ioff));                     /* keep the address in a VGPR: one broadcast LDS read per step */
	uint2 cur = inject[ioff];
	uint32_t acc = 0;
#pragma unroll
	for (int t = 0; t < kCellBlock; ++t) {
		const uint2 nxt = inject[ioff + (t + 1 < kCellBlock ? t + 1 : t)];
		const int32_t in = __builtin_amdgcn_update_dpp((int)cur.x, S.outv, DPP_WAVE_SHR1, 0xf, 0xf, false);
		const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp((int)cur.y, (int)S.outs, DPP_WAVE_SHR1, 0xf, 0xf, false);
		cur = nxt;
		int32_t dg;
		if (WIDE) dg = S.diag + 2 + (int32_t)(__builtin_amdgcn_ubfe(tab, sh, 6) << 3);    /* 6-bit counts: gain = 8*sv + 2 */
		else dg = S.diag + (int32_t)__builtin_amdgcn_ubfe(tab, sh, 8);                    /* pre-scaled byte 8*sv + 2      */
		const int32_t lf = in + leftc;
		int32_t h = max(max(dg, S.hup), lf);
		if (RAMP) {
			/* rows above the matrix: the lane keeps its border values until its first row arrives */
			const bool live = l0 + t >= lane;
			acc = __builtin_amdgcn_alignbit((uint32_t)h, acc, 2);
			h &= ~3;
			S.diag = live ? in : S.diag;
			S.hup = live ? h : S.hup;
			S.outv = S.hup;
		} else {
			acc = __builtin_amdgcn_alignbit((uint32_t)h, acc, 2);
			h &= ~3;
			S.diag = in;
			S.hup = h;
			S.outv = h;
		}
		S.outs = sh;
		lanebuf[t] = make_uint2((uint32_t)S.outv, S.outs);
		if ((t & 15) == 15) dirs[(t >> 4) * kLanes] = acc;
	}
}

}  // namespace

template <bool WIDE>
__global__ __launch_bounds__(kCellWaves *kLanes) void nw_fill_cells(uint8_t *__restrict__ arena, const CellJob *__restrict__ jobs,
                                                                    const TileRef *__restrict__ work, int *__restrict__ abort_word)
{
	__shared__ __attribute__((aligned(16))) uint2 ring[kCellWaves][kRingSteps];
	__shared__ __attribute__((aligned(16))) uint2 inject[kCellWaves][kCellBlock];
	__shared__ uint2 scrap[kCellWaves][kLanes + kCellBlock];   /* lane l, step t -> slot l + t: conflict-free */
	__shared__ int made[kCellWaves], taken[kCellWaves];

	const TileRef item = work[blockIdx.x];
	const CellJob &J = jobs[item.job];
	const int chunk = item.a;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	const int s = chunk * kCellWaves + wv;                    /* this wave's strip */
	if (threadIdx.x < kCellWaves) {
		made[threadIdx.x] = 0;
		taken[threadIdx.x] = 0;
	}
	__syncthreads();
	if (s >= J.nstrips) return;

	const int nb = J.steps_pad / kCellBlock;
	const int col = s * kLanes + lane;                         /* 0-based column of this lane */
	const uint32_t tab = reinterpret_cast<const uint32_t *>(arena + J.coltab)[col];
	const int32_t leftc = reinterpret_cast<const int32_t *>(arena + J.leftc)[col];
	const int32_t *top = reinterpret_cast<const int32_t *>(arena + J.top);
	const uint8_t *rsh = arena + J.rowshift;
	uint32_t *dirs = reinterpret_cast<uint32_t *>(arena + J.dirs) + (size_t)s * (J.steps_pad / 16) * kLanes + lane;
	uint2 *hand_out = reinterpret_cast<uint2 *>(arena + J.hand) + (size_t)chunk * J.steps_pad;
	const uint2 *hand_in = reinterpret_cast<const uint2 *>(arena + J.hand) + (size_t)(chunk > 0 ? chunk - 1 : 0) * J.steps_pad;
	int *progress = reinterpret_cast<int *>(arena + J.progress);
	const bool feeds = wv + 1 < kCellWaves && s + 1 < J.nstrips;        /* a wave of this workgroup reads my ring */
	const bool publishes = wv + 1 == kCellWaves && s + 1 < J.nstrips;   /* the next chunk reads my hand-off words  */
	const bool first_strip = s == 0;
	const bool from_chunk = wv == 0 && chunk > 0;

	CellState S;
	S.hup = top[col + 1];
	S.diag = top[col];
	S.outv = S.hup;
	S.outs = 0;
	/* row letters of the job's first strip, fetched two blocks ahead (lane t: row 32b + t + 1) */
	uint32_t let0 = 0, let1 = 0;
	if (first_strip) {
		let0 = rsh[lane & 31];
		let1 = rsh[kCellBlock + (lane & 31)];
	}
	for (int b = 0; b < nb; ++b) {
		/* what enters lane 0 during this block: lane t prepares step t */
		uint2 word = make_uint2(0u, 0u);
		const int t = lane & 31;
		const int ps = b * kCellBlock + 63 + t;                 /* the producer's lane 63 is 63 steps ahead */
		const int need = (b + 3 < nb) ? b + 3 : nb;
		if (first_strip) {
			word.x = (uint32_t)(J.leftmul * (b * kCellBlock + t + 1));      /* border column: X[r][0] = leftmul * r */
			word.y = let0;
			let0 = let1;
			let1 = rsh[(b + 2) * kCellBlock + t];                             /* rowshift is padded by 64 bytes */
		} else if (wv > 0) {
			if (!wait_lds(&made[wv - 1], need)) { if (lane == 0) atomicExch(abort_word, 1); return; }
			if (ps < J.steps_pad) word = ring[wv - 1][ps % kRingSteps];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) __hip_atomic_store(&taken[wv], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else if (from_chunk) {
			if (!wait_hbm(&progress[chunk - 1], need)) { if (lane == 0) atomicExch(abort_word, 1); return; }
			if (ps < J.steps_pad) {
				const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(hand_in + ps), __ATOMIC_RELAXED,
				                                               __HIP_MEMORY_SCOPE_AGENT);
				word = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
			}
		}
		if (lane < kCellBlock) inject[wv][lane] = word;
		const bool ringer = lane == kLanes - 1 && (feeds || publishes);
		uint2 *lanebuf = ringer ? &ring[wv][(b * kCellBlock) % kRingSteps] : &scrap[wv][lane];
		if (feeds) {
			/* the ring slots of this block last held block b - kRing, which the consumer reads while
			 * preparing its blocks b - kRing - 2 and b - kRing - 1 */
			if (!wait_lds(&taken[wv + 1], b - kRing)) { if (lane == 0) atomicExch(abort_word, 1); return; }
		}
		uint32_t *d = dirs + (size_t)b * (kCellBlock / 16) * kLanes;
		if (b < 2) cell_block<WIDE, true>(S, tab, leftc, inject[wv], lanebuf, d, b * kCellBlock, lane);
		else cell_block<WIDE, false>(S, tab, leftc, inject[wv], lanebuf, d, b * kCellBlock, lane);
		if (feeds) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == kLanes - 1) __hip_atomic_store(&made[wv], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		if (publishes) {
			/* this block's 32 hand-off words: LDS -> HBM with agent-scope stores, drained, then the counter */
			if (lane < kCellBlock) {
				const uint2 v = ring[wv][(b * kCellBlock + lane) % kRingSteps];
				__hip_atomic_store(reinterpret_cast<unsigned long long *>(hand_out + b * kCellBlock + lane),
				                   (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (lane == 0) __hip_atomic_store(&progress[chunk], b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

/*
 * K2d.  Cell (j, k), 1-based, lives in strip s = (k-1)/64, lane (k-1)%64, at local step
 * l = (j-1) + lane, i.e. at "global step" g = l + 64*s = j + k - 2, its anti-diagonal.  Word
 * l/16 of (strip, lane) holds its tag at bits 2*(l%16).  A diagonal move lowers g by 2, a left or up
 * move by 1, so while the path crosses one strip (64 columns) g falls by ~128 = 8 words: the LDS window
 * holds, for each of the kTbStrips strips left of the current cell, the kTbWords words around the
 * expected crossing -- 16 x 16 x 64 words = 64 KiB, loaded as whole 256-byte rows by the four waves.
 * Wave 0 then walks run-batched (lane i looks at cell (j-i, k-i), a ballot finds the end of the
 * run of 'D'); leaving the window just reloads it around the current cell.
 */
constexpr int kTbStrips = 16;
constexpr int kTbWords = 16;
constexpr int kTbSlack = 3;          /* words above the expected entry point of a strip */

__global__ __launch_bounds__(256) void nw_traceback_cells(uint8_t *__restrict__ arena, const CellJob *__restrict__ jobs)
{
	__shared__ __attribute__((aligned(16))) uint32_t win[kTbStrips * kTbWords * kLanes];
	__shared__ int wlo[kTbStrips];
	__shared__ int pos[3];

	const CellJob &J = jobs[blockIdx.x];
	uint8_t *ops = arena + J.ops;
	int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
	const uint32_t *dirs = reinterpret_cast<const uint32_t *>(arena + J.dirs);
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wpitch = J.steps_pad / 16;                      /* words per (strip, lane) */
	int j = J.nrows, k = J.ncols;
	int n = 0;

	while (j > 0 && k > 0) {
		const int s0 = (k - 1) >> 6;
		const int g0 = j + k - 2;
		if (tid < kTbStrips) {
			/* strip s0 - tid: the path is expected at its right edge (column 64*s + 64) on anti-diagonal
			 * g0 - 2*(k - 64*s - 64); for the current strip that point is extrapolated to the right */
			const int sB = s0 - tid;
			const int gedge = g0 - 2 * (k - 64 * sB - 64);
			wlo[tid] = ((gedge - 64 * sB) >> 4) + kTbSlack - (kTbWords - 1);
		}
		__syncthreads();
		/* kTbStrips * kTbWords rows of 256 bytes = 16 uint4 per row */
		for (int e = tid; e < kTbStrips * kTbWords * 16; e += 256) {
			const int B = e / (kTbWords * 16), u = (e / 16) % kTbWords, q = e % 16;
			const int sB = s0 - B, w = wlo[B] + u;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (sB >= 0 && w >= 0 && w < wpitch) v = *reinterpret_cast<const uint4 *>(dirs + ((size_t)sB * wpitch + w) * kLanes + 4 * q);
			reinterpret_cast<uint4 *>(win)[e] = v;
		}
		__syncthreads();
		if (wave == 0) {
			for (;;) {
				const int ri = j - lane, ki = k - lane;
				uint32_t code = 3;                              /* 3 = stop: border or outside the window */
				if (ri > 0 && ki > 0) {
					const int sc = (ki - 1) >> 6;
					const int B = s0 - sc;
					if (B < kTbStrips) {
						const int l = ri + ki - 2 - 64 * sc;
						const int u = (l >> 4) - wlo[B];
						if (u >= 0 && u < kTbWords) code = (win[(B * kTbWords + u) * kLanes + ((ki - 1) & 63)] >> (2 * (l & 15))) & 3u;
					}
				}
				const unsigned long long stop = __ballot(code != DIR_D);
				const int run = stop ? __builtin_ctzll(stop) : kLanes;
				if (run > 0) {
					if (lane < run) ops[n + lane] = (uint8_t)DIR_D;
					n += run;
					j -= run;
					k -= run;
					continue;
				}
				const uint32_t c0 = __builtin_amdgcn_readfirstlane(code);
				if (c0 == 3) break;
				if (lane == 0) ops[n] = (uint8_t)c0;
				++n;
				if (c0 == DIR_L) --k; else --j;
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
	if (tid == 0) {
		summary[0] = n;
		summary[1] = j;
		summary[2] = k;
		summary[3] = 0;
	}
}

hipError_t launch_fill_cells(bool wide, uint8_t *arena, const CellJob *jobs, const TileRef *work, int nwork, int *abort_word, hipStream_t st)
{
	if (nwork <= 0) return hipSuccess;
	if (wide) hipLaunchKernelGGL(nw_fill_cells<true>, dim3(nwork), dim3(kCellWaves * kLanes), 0, st, arena, jobs, work, abort_word);
	else hipLaunchKernelGGL(nw_fill_cells<false>, dim3(nwork), dim3(kCellWaves * kLanes), 0, st, arena, jobs, work, abort_word);
	return hipGetLastError();
}

hipError_t launch_traceback_cells(uint8_t *arena, const CellJob *jobs, int njobs, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	hipLaunchKernelGGL(nw_traceback_cells, dim3(njobs), dim3(256), 0, st, arena, jobs);
	return hipGetLastError();
}

}  // namespace csadp
