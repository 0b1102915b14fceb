/*
 * csadp_bits.hip -- bit-parallel form of the pairwise fill (dynamicprogramming.c:990-1029 for a
 * profile of ONE sequence: i = 1, scores +1 match / -1 mismatch / -1 gap, fresh borders) and its
 * traceback (dynamicprogramming.c:1037-1047).  gfx950, wave64.
 *
 *   nw_fill_bits<W, WAVES, WORK>   K1b: a lane owns W words of 32 columns, a wave a strip of 2048 W columns, a
 *                                  workgroup WAVES strips; WORK = false: one workgroup per matrix, WORK = true:
 *                                  a workgroup per chunk of WAVES strips of a matrix (work list)
 *   nw_traceback_windows<W>        K2c: replays the 16-lane x 32-step pieces the path crosses into windowed LDS tiles, walks them
 *
 * Why it is exact.  Let u = H[r][k-1] - H[r-1][k-1] (vertical step left of the cell), w =
 * H[r-1][k] - H[r-1][k-1] (horizontal step above it), both in {-1,0,1,2}.  The reference's cell
 *     H[r][k] = max(diag + s, left - 1, up - 1),  ties D >= L >= U  (:1014-1025)
 * gives c = H[r][k] - H[r-1][k-1] = 1 on a match and max(u, w, 0) - 1 on a mismatch; the new
 * steps are c - w (vertical) and c - u (horizontal); the direction is D iff match or c = -1,
 * else L iff c - u = -1, else U.  Along a row the vertical step is a 4-state machine driven by
 * (match, w); with thermometer planes (">= 0", ">= 1", ">= 2") its ">= 2" and ">= 1" planes are
 * carry chains (generate / propagate), which an integer addition resolves for 32 columns at once,
 * and the ">= 0" plane is a shift.  tools/bitproto.py checks these formulas against the plain
 * recurrence, tie-breaks included; tools/subco_probe.hip checks the instruction sequence below
 * against the plain word recurrence on the device.
 *
 * How a lane gets its three carries (round 3).  What crosses from a lane to its right neighbour is one
 * bit per plane and step: the top bit of the neighbour's outgoing plane.  Each lane keeps the COMPLEMENT
 * of its outgoing planes; `v_sub_co_u32_dpp vcc <- nO(left lane) - 0x80000000` borrows exactly where the
 * left lane's top bit was set, for all 64 lanes in one instruction, and `v_addc_co_u32` takes the chain's
 * carry from VCC.  The first lane of a wave (of a DPP row in the replay) has no left lane: the DPP operand
 * reads 0 there (bound_ctrl) and its second operand is 0 or 1, loaded per step from LDS -- 0 - 1 borrows.
 * With W words per lane the carry between a lane's own words is the carry-out of the previous addc: no
 * instruction.  The row letter travels the same way: x = B ^ R (column-letter plane xor row-letter mask) of
 * a lane's first word is the left lane's x of the step before xor a per-lane constant (v_xor_b32_dpp; the
 * first lane keeps the value loaded for it).  The round-2 form packed everything into one hand-off word per
 * step (2 v_perm, merge, DPP move, 4 v_bfe, v_alignbit, 2 v_add3) and every lane stored that word to LDS
 * every step; here nothing is stored per step.  Every lane accumulates its outgoing carries
 * (acc = 2 acc + carry: one v_addc_co_u32 per plane on the carry the chain has just left in VCC); the three
 * words are saved with the lane's checkpoint every 32 steps, so the traceback can restart a replay at ANY
 * lane, and lane 63's three words per block are all the next strip needs.
 * Work per lane and step (tools/count_valu.py): 31 VALU for W = 1, 52 for W = 2 (0.81 per cell), 73 for W = 3 (0.76), 94 for W = 4 (0.73).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

namespace {

constexpr int kRing = 8;                       /* blocks of hand-off words buffered per strip boundary (a power of two) */
constexpr unsigned long long kSpinTicks = 50000000ull;   /* bound of every wait: 0.5 s of the 100 MHz s_memrealtime clock */

/* v_bitop3_b32: any boolean function of three words in one instruction; the table is the function
 * applied to these three constants */
constexpr uint32_t LA = 0xF0, LB = 0xCC, LC = 0xAA;
#define BITOP3(a, b, c, expr) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), (unsigned char)((expr) & 0xff)))

/* the same as a statement the compiler keeps in program order with the DPP statements below (all `asm volatile`): a register
 * written by a vector instruction must not be read through DPP by one of the next two instructions, and the compiler's
 * hazard recogniser does not look into inline assembly -- left to its scheduler, the complemented planes were computed
 * right in front of the borrow that reads them (tools/check_dpp_hazards.py gates the build) */
#define BITOP3_ORDERED(dst, a, b, c, expr) \
	asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(dst) : "v"(a), "v"(b), "v"(c), "n"((int)((expr) & 0xff)))

constexpr uint32_t kNoCarry = 0x80000000u;     /* complemented outgoing plane whose top bit is clear; also the subtrahend of every lane but the first */

template <int W>            /* W words of 32 columns per lane */
struct BitState {
	uint32_t nH0[W], H1[W], H2[W];   /* horizontal steps of the row above: NOT ">= 0", ">= 1", ">= 2" */
	uint32_t x0, x1;                 /* word 0: column-letter planes xor the row letter's masks (the right lane chains from them) */
	uint32_t nO2, nO1, nO0;          /* complements of the last word's outgoing vertical-step planes */
	uint32_t acc2, acc1, acc0;       /* the lane's outgoing carries, one bit per step, first step in bit 31 after 32 steps */
};

/* per-lane constants of the letter chain: D = this lane's word 0 xor the left lane's word 0, E[h] = word 0 xor word h */
template <int W>
struct LaneConst {
	uint32_t D0, D1;
	uint32_t E0[W], E1[W];
};

#define CSADP_DPP_WAVE " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define CSADP_DPP_ROW " row_shr:1 row_mask:0xf bank_mask:0xf"

/* x of this lane's first word: the left lane's x of the step before, xor D.  The first lane of the wave (ROWS: of each
 * row of 16 lanes) has no source lane and keeps `first` -- the register the instruction writes. */
template <bool ROWS>
__device__ __forceinline__ uint32_t chain_x(uint32_t first, uint32_t left, uint32_t d)
{
	if (ROWS) asm volatile("v_xor_b32_dpp %0, %1, %2" CSADP_DPP_ROW : "+v"(first) : "v"(left), "v"(d));
	else asm volatile("v_xor_b32_dpp %0, %1, %2" CSADP_DPP_WAVE : "+v"(first) : "v"(left), "v"(d));
	return first;
}

/*
 * One carry chain over the W words of a lane: vcc <- borrow of (left lane's nO) - z  [= the left lane's top bit;
 * first lane: z = 1 borrows, z = 0 does not]; s[h] = a[h] + b[h] + carry, the carry running through the words;
 * ACC: acc = 2 acc + the carry that leaves the lane.  The instructions of one chain stay together (VCC).
 */
#define CSADP_SUB(dpp) "v_sub_co_u32_dpp %[j], vcc, %[no], %[z]" dpp " bound_ctrl:0\n\t"
#define CSADP_ACC "v_addc_co_u32 %[acc], vcc, %[acc], %[acc], vcc"

template <bool ROWS, bool ACC>
__device__ __forceinline__ void chain_w1(uint32_t &s0, uint32_t &acc, uint32_t no, uint32_t z, uint32_t a0, uint32_t b0)
{
	uint32_t j;
	if (ROWS && ACC)
		asm volatile(CSADP_SUB(CSADP_DPP_ROW) "v_addc_co_u32 %[s0], vcc, %[a0], %[b0], vcc\n\t" CSADP_ACC
		    : [j] "=&v"(j), [s0] "=&v"(s0), [acc] "+v"(acc) : [no] "v"(no), [z] "v"(z), [a0] "v"(a0), [b0] "v"(b0) : "vcc");
	else if (ROWS)
		asm volatile(CSADP_SUB(CSADP_DPP_ROW) "v_addc_co_u32 %[s0], vcc, %[a0], %[b0], vcc"
		    : [j] "=&v"(j), [s0] "=&v"(s0) : [no] "v"(no), [z] "v"(z), [a0] "v"(a0), [b0] "v"(b0) : "vcc");
	else if (ACC)
		asm volatile(CSADP_SUB(CSADP_DPP_WAVE) "v_addc_co_u32 %[s0], vcc, %[a0], %[b0], vcc\n\t" CSADP_ACC
		    : [j] "=&v"(j), [s0] "=&v"(s0), [acc] "+v"(acc) : [no] "v"(no), [z] "v"(z), [a0] "v"(a0), [b0] "v"(b0) : "vcc");
	else
		asm volatile(CSADP_SUB(CSADP_DPP_WAVE) "v_addc_co_u32 %[s0], vcc, %[a0], %[b0], vcc"
		    : [j] "=&v"(j), [s0] "=&v"(s0) : [no] "v"(no), [z] "v"(z), [a0] "v"(a0), [b0] "v"(b0) : "vcc");
}

template <bool ROWS, bool ACC>
__device__ __forceinline__ void chain_w2(uint32_t &s0, uint32_t &s1, uint32_t &acc, uint32_t no, uint32_t z, uint32_t a0, uint32_t b0, uint32_t a1,
                                         uint32_t b1)
{
	uint32_t j;
#define CSADP_W2 "v_addc_co_u32 %[s0], vcc, %[a0], %[b0], vcc\n\tv_addc_co_u32 %[s1], vcc, %[a1], %[b1], vcc"
#define CSADP_W2_OUT [j] "=&v"(j), [s0] "=&v"(s0), [s1] "=&v"(s1)
#define CSADP_W2_IN [no] "v"(no), [z] "v"(z), [a0] "v"(a0), [b0] "v"(b0), [a1] "v"(a1), [b1] "v"(b1)
	if (ROWS && ACC) asm volatile(CSADP_SUB(CSADP_DPP_ROW) CSADP_W2 "\n\t" CSADP_ACC : CSADP_W2_OUT, [acc] "+v"(acc) : CSADP_W2_IN : "vcc");
	else if (ROWS) asm volatile(CSADP_SUB(CSADP_DPP_ROW) CSADP_W2 : CSADP_W2_OUT : CSADP_W2_IN : "vcc");
	else if (ACC) asm volatile(CSADP_SUB(CSADP_DPP_WAVE) CSADP_W2 "\n\t" CSADP_ACC : CSADP_W2_OUT, [acc] "+v"(acc) : CSADP_W2_IN : "vcc");
	else asm volatile(CSADP_SUB(CSADP_DPP_WAVE) CSADP_W2 : CSADP_W2_OUT : CSADP_W2_IN : "vcc");
#undef CSADP_W2
#undef CSADP_W2_OUT
#undef CSADP_W2_IN
}

template <bool ROWS, bool ACC>
__device__ __forceinline__ void chain_w3(uint32_t (&s)[3], uint32_t &acc, uint32_t no, uint32_t z, const uint32_t (&a)[3], const uint32_t (&b)[3])
{
	uint32_t j;
#define CSADP_W3                                                                                                        \
	"v_addc_co_u32 %[s0], vcc, %[a0], %[b0], vcc\n\tv_addc_co_u32 %[s1], vcc, %[a1], %[b1], vcc\n\t"                    \
	"v_addc_co_u32 %[s2], vcc, %[a2], %[b2], vcc"
#define CSADP_W3_OUT [j] "=&v"(j), [s0] "=&v"(s[0]), [s1] "=&v"(s[1]), [s2] "=&v"(s[2])
#define CSADP_W3_IN [no] "v"(no), [z] "v"(z), [a0] "v"(a[0]), [b0] "v"(b[0]), [a1] "v"(a[1]), [b1] "v"(b[1]), [a2] "v"(a[2]), [b2] "v"(b[2])
	if (ROWS && ACC) asm volatile(CSADP_SUB(CSADP_DPP_ROW) CSADP_W3 "\n\t" CSADP_ACC : CSADP_W3_OUT, [acc] "+v"(acc) : CSADP_W3_IN : "vcc");
	else if (ROWS) asm volatile(CSADP_SUB(CSADP_DPP_ROW) CSADP_W3 : CSADP_W3_OUT : CSADP_W3_IN : "vcc");
	else if (ACC) asm volatile(CSADP_SUB(CSADP_DPP_WAVE) CSADP_W3 "\n\t" CSADP_ACC : CSADP_W3_OUT, [acc] "+v"(acc) : CSADP_W3_IN : "vcc");
	else asm volatile(CSADP_SUB(CSADP_DPP_WAVE) CSADP_W3 : CSADP_W3_OUT : CSADP_W3_IN : "vcc");
#undef CSADP_W3
#undef CSADP_W3_OUT
#undef CSADP_W3_IN
}

template <bool ROWS, bool ACC>
__device__ __forceinline__ void chain_w4(uint32_t (&s)[4], uint32_t &acc, uint32_t no, uint32_t z, const uint32_t (&a)[4], const uint32_t (&b)[4])
{
	uint32_t j;
#define CSADP_W4                                                                                                        \
	"v_addc_co_u32 %[s0], vcc, %[a0], %[b0], vcc\n\tv_addc_co_u32 %[s1], vcc, %[a1], %[b1], vcc\n\t"                    \
	"v_addc_co_u32 %[s2], vcc, %[a2], %[b2], vcc\n\tv_addc_co_u32 %[s3], vcc, %[a3], %[b3], vcc"
#define CSADP_W4_OUT [j] "=&v"(j), [s0] "=&v"(s[0]), [s1] "=&v"(s[1]), [s2] "=&v"(s[2]), [s3] "=&v"(s[3])
#define CSADP_W4_IN                                                                                                     \
	[no] "v"(no), [z] "v"(z), [a0] "v"(a[0]), [b0] "v"(b[0]), [a1] "v"(a[1]), [b1] "v"(b[1]), [a2] "v"(a[2]), [b2] "v"(b[2]), [a3] "v"(a[3]), \
	    [b3] "v"(b[3])
	if (ROWS && ACC) asm volatile(CSADP_SUB(CSADP_DPP_ROW) CSADP_W4 "\n\t" CSADP_ACC : CSADP_W4_OUT, [acc] "+v"(acc) : CSADP_W4_IN : "vcc");
	else if (ROWS) asm volatile(CSADP_SUB(CSADP_DPP_ROW) CSADP_W4 : CSADP_W4_OUT : CSADP_W4_IN : "vcc");
	else if (ACC) asm volatile(CSADP_SUB(CSADP_DPP_WAVE) CSADP_W4 "\n\t" CSADP_ACC : CSADP_W4_OUT, [acc] "+v"(acc) : CSADP_W4_IN : "vcc");
	else asm volatile(CSADP_SUB(CSADP_DPP_WAVE) CSADP_W4 : CSADP_W4_OUT : CSADP_W4_IN : "vcc");
#undef CSADP_W4
#undef CSADP_W4_OUT
#undef CSADP_W4_IN
}

template <int W, bool ROWS, bool ACC>
__device__ __forceinline__ void chain(uint32_t (&s)[W], uint32_t &acc, uint32_t no, uint32_t z, const uint32_t (&a)[W], const uint32_t (&b)[W])
{
	static_assert(W >= 1 && W <= 4, "words per lane");
	if constexpr (W == 1) chain_w1<ROWS, ACC>(s[0], acc, no, z, a[0], b[0]);
	else if constexpr (W == 2) chain_w2<ROWS, ACC>(s[0], s[1], acc, no, z, a[0], b[0], a[1], b[1]);
	else if constexpr (W == 3) chain_w3<ROWS, ACC>(s, acc, no, z, a, b);
	else chain_w4<ROWS, ACC>(s, acc, no, z, a, b);
}

/* What the first lane of a wave (fill) / of a DPP row (replay) is given per step, 8 words in LDS: its x0, x1, and
 * 1 / 0 = carry / no carry into its planes ">= 2", ">= 1", ">= 0".  Every other lane reads the same offsets of
 * a constant row block whose subtrahend slots hold 0x80000000. */
constexpr int kInjWords = 8;
enum : int { INJ_X0 = 0, INJ_X1 = 1, INJ_Z2 = 2, INJ_Z1 = 3, INJ_Z0 = 4 };

enum : int { OUT_NONE = 0, OUT_WIN = 2 };

/* OUT_WIN (the windowed traceback, K2c below): a piece keeps only the words of a window of kWinWords words of 32 columns around the
 * planned path.  A tile is [32 rows][kWinPitch cells]; row 31 - t holds step t (rows run UP the matrix, like the walk), cell
 * gw % kWinWords holds global word gw, the odd pitch's last cell takes the words of a lane outside the window (never read). */
constexpr int kWinWords = 8;
constexpr int kWinPitch = kWinWords + 1;
template <int W>
struct WinOut {
	uint2 *cell[W];          /* this lane's cell of word h in row 0 of the piece's tile (the pad cell for a word outside the window) */
	bool any;                /* some word of the lane lies in the window */
};

/*
 * 32 steps.  ip: this lane's source of per-step inputs in LDS (the inject rows for a first lane, the constant
 * rows for all others), read PF steps ahead.  RAMPIN: lanes whose row index is still negative keep an empty
 * row above.  OUT_NONE (fill): ACC is on, nothing is stored.  OUT_WIN (traceback replay): the wave is four independent
 * pieces of 16 lanes (DPP stays inside a row) and every step's (not-diagonal, left) masks of the words in the piece's
 * window go to its LDS tile (`win`).
 */
template <int W, bool RAMPIN, int OUT, int PF>
__device__ __forceinline__ void bits_block(BitState<W> &S, const LaneConst<W> &K, const uint32_t *ip, int l0, int lane, const WinOut<W> *win = nullptr)
{
	constexpr bool ROWS = (OUT != OUT_NONE);
	constexpr bool ACC = (OUT == OUT_NONE);
	uint4 qa[PF];
	uint32_t qb[PF];
	asm volatile("s_nop 1");                               /* whatever the block loop moved into the state registers has landed */
#pragma unroll
	for (int p = 0; p < PF; ++p) {
		qa[p] = *reinterpret_cast<const uint4 *>(ip + p * kInjWords);
		qb[p] = ip[p * kInjWords + INJ_Z0];
	}
#pragma unroll
	for (int t = 0; t < kBitBlock; ++t) {
		const uint4 in = qa[t % PF];
		const uint32_t z0 = qb[t % PF];
		if (t + PF < kBitBlock) {
			qa[t % PF] = *reinterpret_cast<const uint4 *>(ip + (t + PF) * kInjWords);
			qb[t % PF] = ip[(t + PF) * kInjWords + INJ_Z0];
		}
		const uint32_t x0 = chain_x<ROWS>(in.x, S.x0, K.D0);
		const uint32_t x1 = chain_x<ROWS>(in.y, S.x1, K.D1);
		S.x0 = x0;
		S.x1 = x1;
		[[maybe_unused]] const uint32_t live = RAMPIN ? ((l0 + t >= lane) ? ~0u : 0u) : ~0u;
		uint32_t nE[W], g2[W], s2[W], G2[W], g1[W], A1[W], s1[W], G1[W], O0[W], G0[W], nH0[W];
		[[maybe_unused]] uint32_t wnd[W], wlf[W];
#pragma unroll
		for (int h = 0; h < W; ++h) {
			nH0[h] = S.nH0[h];
			nE[h] = (h == 0) ? (x0 | x1) : ((x0 ^ K.E0[h]) | (x1 ^ K.E1[h]));           /* 1 = mismatch */
			/* vertical step >= 2: generated by a match over w = -1, carried through mismatches over w = -1 */
			g2[h] = BITOP3(nE[h], nH0[h], nH0[h], ~LA & LB);
		}
		chain<W, ROWS, ACC>(s2, S.acc2, S.nO2, in.z, nH0, g2);
#pragma unroll
		for (int h = 0; h < W; ++h) {
			G2[h] = BITOP3(s2[h], nH0[h], g2[h], LA ^ LB ^ LC);                         /* incoming: u >= 2 */
			if (h == W - 1) BITOP3_ORDERED(S.nO2, g2[h], nH0[h], G2[h], ~(LA | (LB & LC)));   /* NOT outgoing: between this step's two other chains */
			/* >= 1: match over w <= 0, or mismatch over w = 0 with u >= 2; carried over w = -1 */
			const uint32_t t1 = BITOP3(nE[h], nH0[h], G2[h], ~LA | (~LB & LC));
			g1[h] = BITOP3(t1, S.H1[h], S.H1[h], LA & ~LB);
			A1[h] = BITOP3(g1[h], nE[h], nH0[h], LA | (LB & LC));
		}
		chain<W, ROWS, ACC>(s1, S.acc1, S.nO1, in.w, A1, g1);
#pragma unroll
		for (int h = 0; h < W; ++h) {
			G1[h] = BITOP3(s1[h], A1[h], g1[h], LA ^ LB ^ LC);
			if (h == W - 1) BITOP3_ORDERED(S.nO1, g1[h], A1[h], G1[h], ~(LA | (LB & LC)));
			/* >= 0: no chain.  match: w <= 1; mismatch: w = -1, or w = 0 and u >= 1, or w = 1 and u >= 2 */
			const uint32_t v = BITOP3(S.H1[h], G2[h], G1[h], (LA & LB) | (~LA & LC));
			const uint32_t w = BITOP3(nE[h], v, S.H2[h], ~LC & (~LA | LB));
			O0[h] = BITOP3(w, nE[h], nH0[h], LA | (LB & LC));
		}
		/* G0 = (O0 << 1) | carry = O0 + O0 + carry, the carry running through the lane's words like the others */
		chain<W, ROWS, ACC>(G0, S.acc0, S.nO0, z0, O0, O0);
		BITOP3_ORDERED(S.nO0, O0[W - 1], O0[W - 1], O0[W - 1], ~LA);                   /* behind its own chain: read again a step later */
#pragma unroll
		for (int h = 0; h < W; ++h) {
			/* c = H[r][k] - H[r-1][k-1]: C1 = (c = 1), C0 = (c >= 0); new horizontal steps c - u */
			const uint32_t C1 = BITOP3(nE[h], G2[h], S.H2[h], ~LA | LB | LC);
			const uint32_t C0 = BITOP3(nE[h], G1[h], S.H1[h], ~LA | LB | LC);
			uint32_t T2 = BITOP3(C1, G0[h], G0[h], LA & ~LB);
			const uint32_t a1 = BITOP3(C1, G1[h], G1[h], LA & ~LB);
			uint32_t T1 = BITOP3(G0[h], a1, C0, (LA & LB) | (~LA & LC));
			const uint32_t b0 = BITOP3(C0, G1[h], G0[h], LC & (~LA | LB));
			uint32_t nT0 = BITOP3(b0, C1, G2[h], LA & (~LB | LC));
			if (OUT == OUT_WIN) {
				wnd[h] = C0 & nE[h];
				wlf[h] = wnd[h] & nT0;
			}
			if (RAMPIN) {
				nT0 |= ~live;
				T1 &= live;
				T2 &= live;
			}
			S.nH0[h] = nT0;
			S.H1[h] = T1;
			S.H2[h] = T2;
		}
		if (OUT == OUT_WIN) {
			if (win->any) {
#pragma unroll
				for (int h = 0; h < W; ++h) win->cell[h][(kBitBlock - 1 - t) * kWinPitch] = make_uint2(wnd[h], wlf[h]);
			}
		}
	}
}

/*
 * The same 32 steps as ONE inline-assembly statement (tools/gen_bits_block.py -> csadp_bits_block.inc), for the launches that run
 * ONE wave per SIMD with one word per lane -- a single matrix, the first fills of a whole-genome profile alignment.  A lone wave
 * pays ~4.2 cycles for every instruction it issues, whatever it is: here no `s_nop` follows the chains (the compiler puts one
 * behind every asm statement whose output the next instruction reads: three per step), there is one counted wait per step, and the
 * block is 8-byte aligned with its 8-byte instructions at 0 mod 8, the placement a lone wave runs fastest: a 16 kbp pair fills in 1.43
 * instead of 1.59 ms, a 200 kbp pair in 17.3 instead of 18.6 ms.  With several waves per SIMD the block gains nothing over the C++ form
 * and the other placement is the fast one (measured both: tools/subco_probe.hip, profiles/r03_ab_asm_block.txt), so those kernels keep
 * the C++ form.  tools/subco_probe.hip holds the generated block to the plain word recurrence on the device; the DPP hazard check of the
 * build covers it.
 */
#include "csadp_bits_block.inc"
__device__ __forceinline__ void bits_block_lone(BitState<1> &S, const LaneConst<1> &K, const uint32_t *ip)
{
	const uint32_t ipa = (uint32_t)(uintptr_t)ip;             /* LDS byte address = low half of the generic pointer */
	asm volatile(BITS_BLOCK_ASM_W1_LONE
	             : [nh0_0] "+v"(S.nH0[0]), [h1_0] "+v"(S.H1[0]), [h2_0] "+v"(S.H2[0]), [x0] "+v"(S.x0), [x1] "+v"(S.x1), [no2] "+v"(S.nO2),
	               [no1] "+v"(S.nO1), [no0] "+v"(S.nO0), [a2] "+v"(S.acc2), [a1] "+v"(S.acc1), [a0] "+v"(S.acc0)
	             : [d0] "v"(K.D0), [d1] "v"(K.D1), [ip] "v"(ipa)
	             : BITS_BLOCK_CLOBBERS_W1);
}

/* checkpoint of a lane after block b of strip s: planes per word h at ck[((s nb + b) W + h) 64 + lane] (.w of word 0: acc2),
 * the other two accumulators at hand[(s nb + b) 64 + lane] */
template <int W>
__device__ __forceinline__ void save_state(uint4 *ck, uint2 *hand, size_t blk, int lane, const BitState<W> &S)
{
#pragma unroll
	for (int h = 0; h < W; ++h) ck[(blk * W + h) * kLanes + lane] = make_uint4(S.nH0[h], S.H1[h], S.H2[h], h == 0 ? S.acc2 : 0u);
	hand[blk * kLanes + lane] = make_uint2(S.acc1, S.acc0);
}

template <int W>
__device__ __forceinline__ void fresh_state(BitState<W> &S)
{
#pragma unroll
	for (int h = 0; h < W; ++h) {
		S.nH0[h] = ~0u;
		S.H1[h] = S.H2[h] = 0;
	}
	S.x0 = S.x1 = 0;
	S.nO2 = S.nO1 = S.nO0 = kNoCarry;
	S.acc2 = S.acc1 = S.acc0 = 0;
}

/* Counters in LDS that order LDS data only: the LDS executes one wave's accesses in the order they were issued
 * and is coherent inside the compute unit, so relaxed accesses suffice.  TIGHT: poll without sleeping (the
 * one-workgroup-per-job launches and the one-wave-per-SIMD launches; the chunked launches with 2 or 4 waves per SIMD
 * sleep between polls: measured both ways in round 2).  Every wait is bounded in TIME (s_memrealtime, 100 MHz). */
template <bool TIGHT>
__device__ __forceinline__ bool wait_at_least(const int *counter, int need)
{
	if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) return true;
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	int spins = 0;
	while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
		if (!TIGHT) __builtin_amdgcn_s_sleep(2);
		if ((++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) return false;
	}
	return true;
}

/* the row planes are written before the launch (host, or nw_pack_planes in an earlier kernel) and only read here:
 * through the constant address space they become scalar loads, counted apart from the vector memory accesses */
typedef const __attribute__((address_space(4))) uint32_t *ConstWords;

/* bit for step t (0..31) out of the accumulators of two consecutive blocks of the lane that feeds: `older` holds the
 * step just before the first one wanted in its bit 0, `newer` the following 31 in bits 31..1 */
__device__ __forceinline__ uint32_t carry_bit(uint32_t older, uint32_t newer, int t)
{
	const uint32_t w = __builtin_amdgcn_alignbit(older, newer, 1);
	return (w >> (31 - t)) & 1u;
}

/* hand-off between the chunks of a matrix (WORK): three 8-byte granules per block, {accumulator, epoch}, each written
 * by ONE write-through store and valid exactly when it carries the launch's epoch (MI355X_MICROARCH.md, data-tagged
 * granules).  The region is zeroed when the batch is laid out and epochs are never 0. */
__device__ __forceinline__ unsigned long long granule(uint32_t v, uint32_t epoch) { return ((unsigned long long)epoch << 32) | v; }

}  // namespace

/*
 * K1b.  One wave per strip.  Strip s consumes, for every row, the three carries that leave lane 63 of strip s-1
 * (which works on row r at its step r + 63).  They travel per BLOCK of 32 steps: lane 63's three accumulators, through an
 * LDS ring inside a workgroup (`made` = blocks the producer has finished, `taken` = blocks whose inputs the consumer has
 * fetched, for back-pressure), through epoch-tagged granules in HBM between the workgroups of a chunked matrix.  The work
 * list puts a matrix' chunks in ascending order, so the chunk a workgroup waits for was dispatched before it -- for speed
 * only: every wait is bounded in time, a time-out raises the abort word and the host repeats the pass chunk by chunk.
 * LONE (WORK with 4 waves: one wave per SIMD, launches of few strips): inputs are read further ahead and polls do not sleep.
 */
/* PACK (the launches of whole jobs only): jobs narrower than the workgroup SHARE it -- four waves, one per SIMD of the compute unit, where a
 * workgroup of one to three waves leaves SIMDs to chance (round 5).  `work` then is a table of one entry per wave, {job, strip} (job < 0: the
 * place is empty), a job's strips on consecutive waves.  The jobs of a workgroup have nothing to do with each other: a wave's ring neighbour
 * is the wave before it only inside its own job. */
template <int W, int WAVES, bool WORK, int PACK = 1>
__global__ __launch_bounds__(WAVES *kLanes) void nw_fill_bits(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs, int njobs,
                                                              const TileRef *__restrict__ work, uint32_t epoch, int *__restrict__ abort_word)
{
	constexpr bool LONE = WORK && WAVES == 4;
#ifdef CSADP_VARIANT_SLEEP
	constexpr bool TIGHT = LONE;
#else
	constexpr bool TIGHT = !WORK || LONE;
#endif
#ifndef CSADP_PF_MANY
#define CSADP_PF_MANY 1
#endif
#ifndef CSADP_LONE_PF
#define CSADP_LONE_PF 3
#endif
	constexpr int PF = LONE ? CSADP_LONE_PF : CSADP_PF_MANY;
	__shared__ __attribute__((aligned(16))) uint32_t ring[WAVES][kRing][4];
	__shared__ __attribute__((aligned(16))) uint32_t inject[WAVES][kBitBlock * kInjWords];
	/* 4 words further than a multiple of the 64 banks from the inject rows: the first lane's 16 bytes and everybody else's never share a bank */
	__shared__ __attribute__((aligned(16))) uint32_t konst[kBitBlock * kInjWords + 4];
	__shared__ int made[WAVES], taken[WAVES];
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	int sw = wv;                                                 /* this wave's place among the waves of its job's chunk */
	int chunk = 0;
	bool present = true;
	const BitJob *jp;
	if (WORK && PACK > 1) {
		/* chunked launch with shared workgroups: whole chunks of wide jobs as before, their last, partial chunks and the narrow jobs of the
		 * batch side by side in workgroups of their own; the table's strip index says which chunk a wave belongs to and where in it */
		const TileRef item = work[(size_t)blockIdx.x * WAVES + wv];
		present = item.job >= 0;
		jp = &jobs[(size_t)blockIdx.y * njobs + (present ? item.job : 0)];
		chunk = item.a / WAVES;
		sw = item.a % WAVES;
	} else if (WORK) {
		const TileRef item = work[blockIdx.x];                   /* x: the work list of one pass, y: the pass */
		jp = &jobs[(size_t)blockIdx.y * njobs + item.job];
		chunk = item.a;
	} else if (PACK > 1) {
		const TileRef item = work[(size_t)blockIdx.x * WAVES + wv];   /* x: the workgroups of one pass, y: the pass */
		present = item.job >= 0;
		jp = &jobs[(size_t)blockIdx.y * njobs + (present ? item.job : 0)];
		sw = item.a;
	} else {
		jp = &jobs[blockIdx.x];
	}
	const BitJob &J = *jp;
	const int s = chunk * WAVES + sw;                            /* this wave's strip */
	if (threadIdx.x < WAVES) {
		made[threadIdx.x] = 0;
		taken[threadIdx.x] = 0;
	}
	for (int i = threadIdx.x; i < kBitBlock * kInjWords + 4; i += blockDim.x) konst[i] = kNoCarry;
	__syncthreads();
	if (!present || s >= J.nstrips) return;

	const int nb = J.steps_pad / kBitBlock;
	const uint32_t *cp = reinterpret_cast<const uint32_t *>(arena + J.colplanes);
	ConstWords rp = (ConstWords)(uintptr_t)(arena + J.rowplanes);
	uint32_t a0n = rp[0], a1n = rp[J.rowwords];                  /* requested one block ahead */
	uint32_t B0[W], B1[W];
	const size_t w0 = ((size_t)s * kLanes + lane) * W;
#pragma unroll
	for (int h = 0; h < W; ++h) {
		B0[h] = cp[w0 + h];
		B1[h] = cp[J.nwords_pad + w0 + h];
	}
	LaneConst<W> K;
	{
		const size_t wl = lane > 0 ? w0 - W : w0;                /* the first lane's D is never used */
		K.D0 = B0[0] ^ cp[wl];
		K.D1 = B1[0] ^ cp[J.nwords_pad + wl];
#pragma unroll
		for (int h = 0; h < W; ++h) {
			K.E0[h] = B0[0] ^ B0[h];
			K.E1[h] = B1[0] ^ B1[h];
		}
	}
	const uint32_t b00 = __builtin_amdgcn_readfirstlane(B0[0]), b10 = __builtin_amdgcn_readfirstlane(B1[0]);
	const bool feeds = sw + 1 < WAVES && s + 1 < J.nstrips;              /* the next wave of this workgroup reads my ring */
	const bool publishes = WORK && sw + 1 == WAVES && s + 1 < J.nstrips;  /* the next chunk reads my granules */
	const bool from_left_chunk = WORK && sw == 0 && chunk > 0;
	uint4 *ck = reinterpret_cast<uint4 *>(arena + J.ckpt);
	uint2 *hand = reinterpret_cast<uint2 *>(arena + J.hand);
	/* granules [chunk boundary][block][3] */
	unsigned long long *xout = reinterpret_cast<unsigned long long *>(arena + J.xhand) + (size_t)chunk * nb * 3;
	const unsigned long long *xin = reinterpret_cast<const unsigned long long *>(arena + J.xhand) + (size_t)(chunk > 0 ? chunk - 1 : 0) * nb * 3;

	BitState<W> S;
	fresh_state<W>(S);
	const uint32_t *ip = lane == 0 ? &inject[wv][0] : &konst[4];
	/* the constants are waited for HERE: left to the compiler the wait sits at their first use inside the block loop, where
	 * it is s_waitcnt vmcnt(0) -- and drains the checkpoint stores of the block before, every block */
	asm volatile("" : "+v"(K.D0), "+v"(K.D1));
#pragma unroll
	for (int h = 0; h < W; ++h) asm volatile("" : "+v"(K.E0[h]), "+v"(K.E1[h]));

	/* chunk hand-off: lanes 0..2 hold one granule each.  gA / gB: the producer's blocks b+1 and b+2, pre: block b+3, requested a
	 * block before it is needed */
	uint32_t gA[3] = {0, 0, 0}, gB[3] = {0, 0, 0};
	unsigned long long pre = 0;
	auto fetch_granules = [&](int blk, unsigned long long v, uint32_t (&g)[3]) -> bool {
		/* v: what this lane (0..2) has loaded for block blk, or garbage for blk >= nb */
		if (blk >= nb) {
			g[0] = g[1] = g[2] = 0;
			return true;
		}
		const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
		int spins = 0;
		for (;;) {
			const bool ok = lane >= 3 || (uint32_t)(v >> 32) == epoch;
			if (__all(ok)) break;
			__builtin_amdgcn_s_sleep(1);
			if ((++spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) return false;
			if (!ok) v = __hip_atomic_load(&xin[(size_t)blk * 3 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		g[0] = __builtin_amdgcn_readlane((uint32_t)v, 0);
		g[1] = __builtin_amdgcn_readlane((uint32_t)v, 1);
		g[2] = __builtin_amdgcn_readlane((uint32_t)v, 2);
		return true;
	};
	auto request = [&](int blk) -> unsigned long long {
		return (lane < 3 && blk < nb) ? __hip_atomic_load(&xin[(size_t)blk * 3 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
	};
	if (from_left_chunk) {
		const unsigned long long v1 = request(1), v2 = request(2);
		pre = request(3);
		if (!fetch_granules(1, v1, gA) || !fetch_granules(2, v2, gB)) { if (lane == 0) atomicExch(abort_word, 1); return; }
	}

	/* one block of 32 steps; the two ramp blocks of a strip and the steady ones are separate loops: inside ONE loop the two
	 * forms of the step met at a join, and the compiler paid for it with 32 register moves per block (the per-block work is
	 * 8-12 % of a block of two words, 15-20 % of a block of one) */
	auto block_of = [&](int b, auto ramp_tag) -> bool {
		constexpr bool RAMP = decltype(ramp_tag)::value;
		/* what enters lane 0 during this block: lane t prepares step t.  Carries: the producer's steps 32 b + t + 63 (its lane
		 * 63 works on row 32 b + t then), i.e. the last step of its block b + 1 and the first 31 of block b + 2 */
		const int t = lane & 31;
		uint32_t z2 = 0, z1 = 0, z0 = 0;
		if (sw > 0) {
			const int need = (b + 3 < nb) ? b + 3 : nb;
			if (!wait_at_least<TIGHT>(&made[wv - 1], need)) { if (lane == 0) atomicExch(abort_word, 1); return false; }
			uint4 A = make_uint4(0, 0, 0, 0), Bv = make_uint4(0, 0, 0, 0);
			if (b + 1 < nb) A = *reinterpret_cast<const uint4 *>(ring[wv - 1][(b + 1) & (kRing - 1)]);
			if (b + 2 < nb) Bv = *reinterpret_cast<const uint4 *>(ring[wv - 1][(b + 2) & (kRing - 1)]);
			z2 = carry_bit(A.x, Bv.x, t);
			z1 = carry_bit(A.y, Bv.y, t);
			z0 = carry_bit(A.z, Bv.z, t);
			if (lane == 0) __hip_atomic_store(&taken[wv], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else if (from_left_chunk) {
			z2 = carry_bit(gA[0], gB[0], t);
			z1 = carry_bit(gA[1], gB[1], t);
			z0 = carry_bit(gA[2], gB[2], t);
		}
		const uint32_t a0 = a0n, a1 = a1n;
		if (b + 1 < nb) {
			a0n = rp[b + 1];
			a1n = rp[J.rowwords + b + 1];
		}
		if (lane < kBitBlock) {
			const uint32_t r0 = 0u - ((a0 >> t) & 1u), r1 = 0u - ((a1 >> t) & 1u);
			*reinterpret_cast<uint4 *>(&inject[wv][t * kInjWords]) = make_uint4(b00 ^ r0, b10 ^ r1, z2, z1);
			inject[wv][t * kInjWords + INJ_Z0] = z0;
		}
		if constexpr (RAMP) bits_block<W, true, OUT_NONE, PF>(S, K, ip, b * kBitBlock, lane);
		else if constexpr (LONE && W == 1) bits_block_lone(S, K, ip);
		else bits_block<W, false, OUT_NONE, PF>(S, K, ip, b * kBitBlock, lane);
		save_state<W>(ck, hand, (size_t)s * nb + b, lane, S);
		if (feeds) {
			/* the ring slot of this block last held block b - kRing, which the consumer fetches while preparing its blocks
			 * b - kRing - 2 and b - kRing - 1 */
			if (!wait_at_least<TIGHT>(&taken[wv + 1], b - kRing)) { if (lane == 0) atomicExch(abort_word, 1); return false; }
			if (lane == kLanes - 1) {
				*reinterpret_cast<uint4 *>(ring[wv][b & (kRing - 1)]) = make_uint4(S.acc2, S.acc1, S.acc0, 0u);
				__hip_atomic_store(&made[wv], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		if (publishes && lane == kLanes - 1) {                    /* for another compute unit: tagged, written through */
			__hip_atomic_store(&xout[(size_t)b * 3 + 0], granule(S.acc2, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&xout[(size_t)b * 3 + 1], granule(S.acc1, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&xout[(size_t)b * 3 + 2], granule(S.acc0, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		if (from_left_chunk) {
			gA[0] = gB[0];
			gA[1] = gB[1];
			gA[2] = gB[2];
			if (!fetch_granules(b + 3, pre, gB)) { if (lane == 0) atomicExch(abort_word, 1); return false; }
			pre = request(b + 4);
		}
		return true;
	};
	const int nramp = nb < 2 ? nb : 2;
	for (int b = 0; b < nramp; ++b)
		if (!block_of(b, std::true_type{})) return;
	for (int b = nramp; b < nb; ++b)
		if (!block_of(b, std::false_type{})) return;
}

/* the lane column (64 per strip) of 0-based matrix column c: 32 W columns per lane; floor for negative c (left of the matrix) */
template <int W>
__device__ __forceinline__ int lane_column(int c)
{
	if constexpr (W == 3) return c >= 0 ? c / 96 : -((95 - c) / 96);
	else return c >> (W == 1 ? 5 : W == 2 ? 6 : 7);
}

/*
 * K2c.  Traceback (dynamicprogramming.c:1037-1047): no direction planes exist in HBM.  From the current cell (r, k) -- lane `lane0` of
 * strip s, block btop -- the path goes up through block btop - d around lane lane0 - d / W, so piece d = (block btop - d, the 16 lanes
 * around the planned path there) is replayed with the fill's own step function: lane state from the checkpoint before the block, the
 * carries entering the piece's first lane from the accumulators of the lane to its left (every lane has them: a piece starts at ANY
 * lane).  A wave replays 4 pieces at once (one per DPP row) into LDS tiles of (not-diagonal, left) masks; one wave walks them.
 * Round 4 rebuilt the kernel around three findings on real genome pairs (a gap move
 * every 7-9 cells; unrelated letters: every 3-4): the walk was two thirds of a traceback, one D-run and ONE gap move per LDS round
 * trip (900 cycles at three words per lane); the replay of the next pieces waited for it; and a tile held 16 lanes x W words per
 * step of which the path crosses one or two.
 *   - A tile keeps a WINDOW of kWinWords words (256 columns) around the planned path: 2.3 KB instead of 4-12, so two sets of 16
 *     pieces fit the LDS for every W and the four replaying waves always work one set ahead of the walking one.
 *   - Tile rows run up the matrix (row = 31 - step) and the sets' tiles are contiguous, so the cell one row up is kWinPitch cells
 *     further whatever the piece, and the cell one column left is the next lower bit of the same word: ONE address per lane gives the
 *     walk its own diagonal (cells (r - i, k - i)), the two above (U, UU: two more reads at constant offsets) and the two to the
 *     left (L, LL: the same words shifted).  Fifteen ballots turn them into scalar masks (not-D, is-L, invalid per diagonal), and a
 *     scalar loop follows the path through them: a D-run is a find-first-set, a gap move changes diagonal -- towards the main one
 *     it also consumes a lane -- until the lanes, 64 ops, or the five diagonals are exhausted.  An iteration takes up to 64 ops and
 *     any number of gap moves that stay within two diagonals of where it began.
 * The plan (reference cell, strip, slope towards the matrix' corner) outlives a set while the path stays in its windows, and carries
 * on across a strip boundary (the set replayed while the path nears the strip's first column is laid from the predicted crossing into
 * the strip to the left); a path that leaves its windows (a long gap run) is planned afresh from the current cell: one serial round.
 * Every round takes at least one op and the rounds are counted: a defect ends the walk short (reported by nw_expand_rows / the host's
 * trace application as CSADP_ERR_HIP), it does not hang the device.
 */
#ifndef CSADP_TB_SET_PIECES
#define CSADP_TB_SET_PIECES 16                         /* (12 and 8 -- three / two replaying waves -- measured slower: profiles/r04_ab_traceback_variants.txt) */
#endif
constexpr int kSetPieces = CSADP_TB_SET_PIECES;        /* pieces per tile set: four replaying waves x four DPP rows */
constexpr int kSetRows = kSetPieces * kBitBlock;       /* tile rows per set */
constexpr int kTbThreads = (1 + kSetPieces / 4) * kLanes;

/* the lane column (64 per strip) that owns global word gw >= 0 */
template <int W>
__device__ __forceinline__ int lane_of_word(int gw)
{
	if constexpr (W == 3) return (int)((unsigned)gw / 3u);
	else return gw >> (W == 1 ? 0 : W == 2 ? 1 : 2);
}

struct TbPlan {
	int kref, l0, s, btop, slope;                      /* reference column (1-based), local step of the first piece's top, strip, its block; columns per local step / 1024 */
	int m0;                                            /* local steps below the reference cell the pieces begin (a plan laid across a strip boundary: slack) */
};

/* first word of piece dabs' window: the planned path crosses the middle of the piece's 32 steps in column kpred */
__device__ __forceinline__ int plan_window(const TbPlan &P, int dabs)
{
	const int m = max(0, kBitBlock * dabs - kBitBlock / 2 + (P.l0 & (kBitBlock - 1)) - P.m0);
	const int kpred = P.kref - 1 - ((m * P.slope) >> 10);
	return (kpred >> 5) - kWinWords / 2;
}

/* a tile set: sixteen consecutive pieces (dbase .. dbase + 15) of a plan */
struct TbSet {
	TbPlan plan;
	int dbase;
};

/*
 * The scalar walk of one iteration through the masks of its five diagonals (index 0, 1: two, one column left; 2: the lanes' own; 3, 4:
 * one, two rows up).  p = lane reached, left = ops the iteration may still take (64 at most: one store), dcur = the diagonal it
 * ends on; GAP / LM: which of the ops taken are gap moves / L.  NOTD: where a run of D ends (invalid cells included), ISL: the gap
 * move there is L, STOP: the iteration ends there (outside the matrix or the set's windows, or the move leads to a sixth diagonal).
 * A gap move towards diagonal 2 also passes a lane: cell (r - a - p, k - b - p) with min(a, b) > 0 is the cell one lane further on
 * diagonal (a - 1, b - 1).
 * One state per diagonal, written out in assembly: its three masks sit in fixed registers and a segment -- a run of D and the gap
 * move that ends it -- is ~19 scalar instructions (find-first-set for the run, s_bitcmp1_b64 for the two tests, s_bitset1_b64 for the
 * op masks).  In C++ the compiler either selected each mask through four compare-and-select pairs per use (75 scalar instructions
 * per segment) or, given one labelled block per state, rebuilt the jumps between them around a dispatch variable.
 */
#define CSADP_WALK_STATE(D, PU, PL)                                                                                                  \
	"st" #D "_%=:\n\t"                                                                                                               \
	"s_lshr_b64 %[t], %[n" #D "], %[p]\n\t"                                                                                          \
	"s_ff1_i32_b64 %[q], %[t]\n\t"                                                                                                   \
	"s_min_u32 %[q], %[q], %[left]\n\t"                                                                                              \
	"s_add_i32 %[p], %[p], %[q]\n\t"                                                                                                 \
	"s_sub_i32 %[left], %[left], %[q]\n\t"                                                                                           \
	"s_cmp_lt_i32 %[left], 1\n\t"                                                                                                    \
	"s_cbranch_scc1 end" #D "_%=\n\t"                                                                                                \
	"s_bitcmp1_b64 %[s" #D "], %[p]\n\t"                                                                                             \
	"s_cbranch_scc1 end" #D "_%=\n\t"                                                                                                \
	"s_sub_i32 %[q], 64, %[left]\n\t"                                                                                                \
	"s_bitset1_b64 %[gap], %[q]\n\t"                                                                                                 \
	"s_sub_i32 %[left], %[left], 1\n\t"                                                                                              \
	"s_bitcmp1_b64 %[l" #D "], %[p]\n\t"                                                                                             \
	"s_cbranch_scc1 left" #D "_%=\n\t" PU "s_branch end" #D "_%=\n\t"                                                                \
	"left" #D "_%=:\n\t"                                                                                                             \
	"s_bitset1_b64 %[lm], %[q]\n\t" PL "end" #D "_%=:\n\t"                                                                           \
	"s_mov_b32 %[d], " #D "\n\t"                                                                                                     \
	"s_branch out_%=\n\t"
/* after a gap move from diagonal D to diagonal T: ADD = "s_add_i32 p, p, 1" when the move heads for diagonal 2; then on into state
 * T unless the ops or the lanes are used up (the iteration then ends ON diagonal T) */
#define CSADP_WALK_GO(T, ADD) ADD "s_cmp_lt_i32 %[left], 1\n\ts_cbranch_scc1 end" #T "_%=\n\ts_cmp_gt_i32 %[p], 63\n\ts_cbranch_scc1 end" #T "_%=\n\ts_branch st" #T "_%=\n\t"
#define CSADP_WALK_INC "s_add_i32 %[p], %[p], 1\n\t"
__device__ __forceinline__ void walk_masks(const unsigned long long (&NOTD)[5], const unsigned long long (&ISL)[5], const unsigned long long (&STOP)[5], int &p,
                                           int &left, int &dcur, unsigned long long &GAP, unsigned long long &LM)
{
	unsigned long long t;
	int q;
	asm volatile("s_branch st2_%=\n\t"
	             /* diagonal 0: L is in STOP[0]; diagonal 4: U is in STOP[4] */
	             CSADP_WALK_STATE(0, CSADP_WALK_GO(1, CSADP_WALK_INC), "")
	             CSADP_WALK_STATE(1, CSADP_WALK_GO(2, CSADP_WALK_INC), CSADP_WALK_GO(0, ""))
	             CSADP_WALK_STATE(2, CSADP_WALK_GO(3, ""), CSADP_WALK_GO(1, ""))
	             CSADP_WALK_STATE(3, CSADP_WALK_GO(4, ""), CSADP_WALK_GO(2, CSADP_WALK_INC))
	             CSADP_WALK_STATE(4, "", CSADP_WALK_GO(3, CSADP_WALK_INC))
	             "out_%=:"
	             : [p] "+s"(p), [left] "+s"(left), [d] "=&s"(dcur), [gap] "+s"(GAP), [lm] "+s"(LM), [t] "=&s"(t), [q] "=&s"(q)
	             : [n0] "s"(NOTD[0]), [n1] "s"(NOTD[1]), [n2] "s"(NOTD[2]), [n3] "s"(NOTD[3]), [n4] "s"(NOTD[4]), [l0] "s"(ISL[0]), [l1] "s"(ISL[1]),
	               [l2] "s"(ISL[2]), [l3] "s"(ISL[3]), [l4] "s"(ISL[4]), [s0] "s"(STOP[0]), [s1] "s"(STOP[1]), [s2] "s"(STOP[2]), [s3] "s"(STOP[3]),
	               [s4] "s"(STOP[4])
	             : "scc");
}
#undef CSADP_WALK_STATE
#undef CSADP_WALK_GO
#undef CSADP_WALK_INC

template <int W>
__global__ __launch_bounds__(kTbThreads) void nw_traceback_windows(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs)
{
	__shared__ __attribute__((aligned(16))) uint2 tile[2][(kSetRows + 2) * kWinPitch];   /* + 2 rows: the walk reads U, UU of its last row */
	__shared__ __attribute__((aligned(16))) uint32_t inject[kSetPieces][kBitBlock * kInjWords];
	__shared__ __attribute__((aligned(16))) uint32_t konst[kBitBlock * kInjWords + 4];
	__shared__ int wtab[2][kSetPieces + 2];                /* first word of every piece's window, per set */
	__shared__ int pos[4];

	const BitJob &J = jobs[blockIdx.x];
	uint8_t *ops = arena + J.ops;
	int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
	const uint32_t *cp = reinterpret_cast<const uint32_t *>(arena + J.colplanes);
	const uint32_t *rp = reinterpret_cast<const uint32_t *>(arena + J.rowplanes);
	const uint4 *ck = reinterpret_cast<const uint4 *>(arena + J.ckpt);
	const uint2 *hand = reinterpret_cast<const uint2 *>(arena + J.hand);
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	const int nb = J.steps_pad / kBitBlock;
	int r = J.nrows, k = J.ncols;
	int n = 0;
	for (int i = threadIdx.x; i < kBitBlock * kInjWords + 4; i += blockDim.x) konst[i] = kNoCarry;
	if (threadIdx.x < 4) wtab[threadIdx.x >> 1][kSetPieces + (threadIdx.x & 1)] = 0x40000000;
	__syncthreads();

	/* 32-bit indices off the job's bases: an index computed in 64 bits costs three or four instructions, and the replaying waves spend
	 * as long on the addresses of a piece's inputs as on its 32 steps */
	const unsigned unb = (unsigned)nb, nwp = (unsigned)J.nwords_pad, rww = (unsigned)J.rowwords;
	auto acc_of = [&](unsigned st, int blk, unsigned l, uint32_t (&a)[3]) {
		a[0] = a[1] = a[2] = 0;
		if (blk < 0 || blk >= nb) return;
		const unsigned at = st * unb + (unsigned)blk;
		a[0] = ck[(at * W) * kLanes + l].w;
		const uint2 h = hand[at * kLanes + l];
		a[1] = h.x;
		a[2] = h.y;
	};
	auto row_mask = [&](unsigned plane, int row) -> uint32_t {
		if ((unsigned)row >= (unsigned)J.steps_pad) return 0u;
		return 0u - ((rp[plane * rww + ((unsigned)row >> 5)] >> (row & 31)) & 1u);
	};

	/* the set being walked and the one being replayed (two named structs, not an array indexed by `cur`: a private array indexed by a
	 * variable lives in vector registers, and everything derived from it -- the walk's lane masks -- would no longer count as uniform) */
	TbSet Cs, Ns;
	Cs.plan = Ns.plan = TbPlan{0, 0, 0, 0, 1024, 0};
	Cs.dbase = Ns.dbase = 0;
	/* piece x of set S into tile set `set`: called by the replaying waves, a piece per DPP row (x = 4 (wave - 1) + row) */
	auto replay_piece = [&](const TbSet &St, const int set, const int x) {
		const TbPlan &plan = St.plan;
		const int dabs = St.dbase + x;
		const unsigned s = (unsigned)plan.s;
		const int wlo = plan_window(plan, dabs);
		const int lane_lo = (wlo >= 0 ? lane_of_word<W>(wlo) : -((W - 1 - wlo) / W)) - kLanes * plan.s;
		const int f = min(max(lane_lo - 4, 0), kLanes - 16);      /* the window's lanes are lane_lo .. lane_lo + 7 at most */
		const int b = plan.btop - dabs < 0 ? 0 : plan.btop - dabs;   /* pieces above block 0 replay block 0 and are never read */
		const int j = lane & 15;
		const unsigned sl = (unsigned)(f + j);                   /* lane index in the strip */
		const unsigned w0 = (s * kLanes + sl) * W;
		if (j == 0) wtab[set][x] = wlo;
		uint32_t B0[W], B1[W];
#pragma unroll
		for (int h = 0; h < W; ++h) {
			B0[h] = cp[w0 + h];
			B1[h] = cp[nwp + w0 + h];
		}
		const unsigned wl = sl > 0 ? w0 - W : w0;
		const uint32_t L0 = cp[wl], L1 = cp[nwp + wl];
		BitState<W> S;
		fresh_state<W>(S);
		if (b > 0) {
			const unsigned at = s * unb + (unsigned)(b - 1);
			uint4 ckv[W];
#pragma unroll
			for (int h = 0; h < W; ++h) ckv[h] = ck[(at * W + h) * kLanes + sl];
			const uint2 hv = hand[at * kLanes + sl];
			/* x of the step before the block: the lane worked on row 32 b - 1 - sl then (not yet live: any value) */
			const int row = b * kBitBlock - 1 - (int)sl;
			const uint32_t rm0 = row_mask(0, row), rm1 = row_mask(1, row);
#pragma unroll
			for (int h = 0; h < W; ++h) {
				S.nH0[h] = ckv[h].x;
				S.H1[h] = ckv[h].y;
				S.H2[h] = ckv[h].z;
			}
			S.nO2 = (ckv[0].w & 1u) ? 0u : kNoCarry;             /* what the lane put out in the last step of block b - 1 */
			S.nO1 = (hv.x & 1u) ? 0u : kNoCarry;
			S.nO0 = (hv.y & 1u) ? 0u : kNoCarry;
			S.x0 = B0[0] ^ rm0;
			S.x1 = B1[0] ^ rm1;
		}
		/* what enters the piece's first lane: lane j of the row prepares steps j and j + 16 */
		uint32_t older[3], newer[3];
		if (f > 0) {                                           /* the lane to the left, one step earlier */
			acc_of(s, b - 1, (unsigned)(f - 1), older);
			acc_of(s, b, (unsigned)(f - 1), newer);
		} else if (s > 0) {                                    /* lane 63 of the strip to the left is 63 steps ahead */
			acc_of(s - 1, b + 1, kLanes - 1, older);
			acc_of(s - 1, b + 2, kLanes - 1, newer);
		} else {
			older[0] = older[1] = older[2] = newer[0] = newer[1] = newer[2] = 0;
		}
		const unsigned wf = (s * kLanes + (unsigned)f) * W;
		const uint32_t bf0 = cp[wf], bf1 = cp[nwp + wf];
#pragma unroll
		for (int hh = 0; hh < 2; ++hh) {
			const int t = j + 16 * hh;
			const int row = b * kBitBlock + t - f;                 /* the first lane's row at step t */
			*reinterpret_cast<uint4 *>(&inject[x][t * kInjWords]) =
			    make_uint4(bf0 ^ row_mask(0, row), bf1 ^ row_mask(1, row), carry_bit(older[0], newer[0], t), carry_bit(older[1], newer[1], t));
			inject[x][t * kInjWords + INJ_Z0] = carry_bit(older[2], newer[2], t);
		}
		LaneConst<W> K;
		K.D0 = B0[0] ^ L0;
		K.D1 = B1[0] ^ L1;
#pragma unroll
		for (int h = 0; h < W; ++h) {
			K.E0[h] = B0[0] ^ B0[h];
			K.E1[h] = B1[0] ^ B1[h];
		}
		WinOut<W> win;
		win.any = false;
		uint2 *piece = &tile[set][x * kBitBlock * kWinPitch];
#pragma unroll
		for (int h = 0; h < W; ++h) {
			const int gw = (int)w0 + h;
			const bool in = (unsigned)(gw - wlo) < (unsigned)kWinWords;
			win.cell[h] = piece + (in ? (gw & (kWinWords - 1)) : kWinWords);
			win.any = win.any | in;
		}
		const uint32_t *ip = j == 0 ? &inject[x][0] : &konst[4];
		const bool ramp = __any(b < 2);                        /* wave-uniform: some piece of this wave is in block 0 or 1 */
		if (ramp) bits_block<W, true, OUT_WIN, 1>(S, K, ip, b * kBitBlock, (int)sl, &win);
		else bits_block<W, false, OUT_WIN, 1>(S, K, ip, b * kBitBlock, (int)sl, &win);
	};
	/* columns per local step (/ 1024) along the line from cell (r, k) to the matrix' corner: a row up is 1 + (columns per row) /
	 * (columns per lane) local steps */
	auto slope_to_corner = [](int rr, int kk) -> int {
		const float cr = (float)kk / (float)max(rr, 1);
		return min(max((int)(1024.0f * cr / (1.0f + cr / (float)(32 * W))), 256), 4096);
	};
	constexpr int kStripCols = kLanes * 32 * W;
	constexpr int kEntrySlack = 4 * kBitBlock;                 /* local steps a plan laid across a strip boundary begins below the predicted crossing */

	int cur = 0;
	bool need_plan = true, have_next = false;
#ifdef CSADP_TB_TIMERS
	/* in-kernel clocks (tools/r04/tb_timers.sh): kept in LDS, not in registers -- the walk's fifteen lane masks leave no scalar
	 * registers to spare.  Per wave: [0] work, [1] waiting, [2] mark, [3] rounds, [4] plans, [5] walk iterations */
	__shared__ unsigned tmr[kTbThreads / kLanes][8];
	if (lane < 8) tmr[wv][lane] = lane == 2 ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
#define TBW_LAP(slot) do { if (lane == 0) { const unsigned now_ = (unsigned)__builtin_amdgcn_s_memtime(); tmr[wv][slot] += now_ - tmr[wv][2]; tmr[wv][2] = now_; } } while (0)
#define TBW_COUNT(slot) do { if (lane == 0) ++tmr[wv][slot]; } while (0)
#else
#define TBW_LAP(slot) do { } while (0)
#define TBW_COUNT(slot) do { } while (0)
#endif
	/* every walking round takes at least one op (the first cell after a plan lies in its own piece's window) and every planning round is
	 * followed by a walking one: more rounds than this is a defect, and the walk then stops short -- which nw_expand_rows / the host's
	 * trace application report as CSADP_ERR_HIP -- instead of keeping the device busy for ever */
	int rounds_left = 2 * (J.nrows + J.ncols) + 16;
	while (r > 0 && k > 0 && --rounds_left >= 0) {
		const int x = 4 * (wv - 1) + (lane >> 4);                /* the replaying waves' piece */
		const bool planning = need_plan;                        /* a round that only replays: the first set of a new plan */
		if (planning) {
			TbPlan &P = Cs.plan;
			const int wq = lane_column<W>(k - 1);
			P.s = wq >> 6;
			P.l0 = (r - 1) + (wq & 63);
			P.btop = P.l0 / kBitBlock;
			P.kref = k;
			P.slope = slope_to_corner(r, k);
			P.m0 = 0;
			Cs.dbase = 0;
			need_plan = false;
			TBW_COUNT(4);
		} else {
			/* the set to replay while this one is walked: the next sixteen pieces of the same plan -- or, when the planned line leaves
			 * the strip inside the set being walked, a plan laid from the predicted crossing into the strip to the left (a strip
			 * further left the local steps of a row are 63 higher: other blocks, other lanes; round 3 and the first form of this
			 * kernel planned afresh there, one serial round per strip) */
			const TbPlan &C = Cs.plan;
			TbSet &N = Ns;
			const int kstrip0 = C.s * kStripCols;
			const int m_exit = C.m0 + (int)(((long long)max(C.kref - kstrip0, 0) << 10) / C.slope);   /* local steps below l0 at which the line is in the strip's first column */
			const int d_exit = (m_exit + kBitBlock / 2 - (C.l0 & (kBitBlock - 1)) + kBitBlock - 1) / kBitBlock;
			if (C.s > 0 && d_exit < Cs.dbase + kSetPieces) {
				const int rx = max(C.l0 - m_exit + 1, 1);                      /* the row of the crossing: lane 0 of the strip works on row l + 1 */
				N.plan.s = C.s - 1;
				N.plan.kref = kstrip0;                                        /* the strip's last column, 1-based */
				N.plan.l0 = (rx - 1) + (kLanes - 1) + kEntrySlack;
				N.plan.btop = N.plan.l0 / kBitBlock;
				N.plan.slope = slope_to_corner(rx, kstrip0);
				N.plan.m0 = kEntrySlack;
				N.dbase = 0;
			} else {
				N.plan = C;
				N.dbase = Cs.dbase + kSetPieces;
			}
		}
		TBW_LAP(1);
		const TbSet &Rs = planning ? Cs : Ns;                   /* the set the replaying waves fill this round */
		have_next = Rs.plan.btop - Rs.dbase >= 0;               /* some piece of it lies inside the matrix */
		if (wv > 0) {
			if (have_next) replay_piece(Rs, planning ? cur : 1 - cur, x);
		} else if (!planning) {
			__builtin_amdgcn_s_setprio(3);                       /* the walk is a chain of dependent instructions: ahead of the fills' waves on its SIMD (config 4: +2 %) */
			const TbPlan &plan = Cs.plan;
			const int dbase = Cs.dbase;
			const uint2 *T = tile[cur];
			const int *wt = wtab[cur];
			const int G0 = kBitBlock * (plan.btop - dbase) + kBitBlock - 1;     /* tile row of local step l: G0 - l */
			const int kstrip0 = plan.s * kStripCols;                     /* first column (0-based) of the plan's strip */
			for (;;) {
				/* lane i looks at cell (r - i, k - i) and its neighbours one and two rows up / columns left.  Which lanes' cells lie in
				 * the matrix and the plan's strip is a prefix of the lanes: scalar; so is which of them have one / two rows above them */
				const int nin = min(min(r, k - kstrip0), kLanes);
				const unsigned long long PRE0 = nin >= kLanes ? ~0ull : (1ull << max(nin, 0)) - 1ull;
				const unsigned long long PRE1 = PRE0 & (r - 1 >= kLanes ? ~0ull : (1ull << max(r - 1, 0)) - 1ull);
				const unsigned long long PRE2 = PRE0 & (r - 2 >= kLanes ? ~0ull : (1ull << max(r - 2, 0)) - 1ull);
				const int kc = k - 1 - lane;
				const int gw = max(kc, 0) >> 5;
				const int rel = G0 - r + 1 + lane - (lane_of_word<W>(gw) & 63);
				const unsigned long long INSET = PRE0 & __ballot((unsigned)rel < (unsigned)kSetRows);
				const int relc = (INSET >> lane) & 1ull ? rel : 0;
				const int at = __mul24(relc, kWinPitch) + (gw & (kWinWords - 1));
				const uint2 c0 = T[at], c1 = T[at + kWinPitch], c2 = T[at + 2 * kWinPitch];
				const int w0 = wt[relc >> 5];
				const uint32_t sh = (uint32_t)kc & 31u;
				const uint32_t t31 = (uint32_t)relc & 31u;
				/* valid: in the piece's window; the cells above: in the same piece (the next piece's window may differ: a path that
				 * moves up there ends the iteration and starts the next one on that cell); the cells to the left: in the same word */
				const unsigned long long V2 = INSET & __ballot((unsigned)(gw - w0) < (unsigned)kWinWords);
				const unsigned long long V3 = V2 & PRE1 & __ballot(t31 < 31u), V4 = V2 & PRE2 & __ballot(t31 < 30u);
				const unsigned long long V1 = V2 & __ballot(sh >= 1u), V0 = V2 & __ballot(sh >= 2u);
				/* the cell's bit to the sign: bit sh of a word lands in bit 31, the two columns to its left in bits 30 and 29 */
				const uint32_t up = sh ^ 31u;
				const int a0 = (int)(c0.x << up), b0 = (int)(c0.y << up), a1 = (int)(c1.x << up), b1 = (int)(c1.y << up), a2 = (int)(c2.x << up),
				          b2 = (int)(c2.y << up);
				const unsigned long long ND2 = __ballot(a0 < 0), LF2 = __ballot(b0 < 0), ND3 = __ballot(a1 < 0), LF3 = __ballot(b1 < 0), ND4 = __ballot(a2 < 0),
				                         LF4 = __ballot(b2 < 0), ND1 = __ballot((a0 << 1) < 0), LF1 = __ballot((b0 << 1) < 0), ND0 = __ballot((a0 << 2) < 0),
				                         LF0 = __ballot((b0 << 2) < 0);
				/* per diagonal (0, 1: two, one column left; 2: the lanes' own; 3, 4: one, two rows up): where the run of D ends, which of
				 * those cells say L, and where the iteration must end: outside the set's windows or the matrix, or a gap move that
				 * leads to a sixth diagonal */
				unsigned long long NOTD[5] = {ND0 | ~V0, ND1 | ~V1, ND2 | ~V2, ND3 | ~V3, ND4 | ~V4};
				unsigned long long ISL[5] = {ND0 & LF0 & V0, ND1 & LF1 & V1, ND2 & LF2 & V2, ND3 & LF3 & V3, ND4 & LF4 & V4};
				unsigned long long STOP[5] = {~V0 | ISL[0], ~V1, ~V2, ~V3, NOTD[4] & ~ISL[4]};
				int p = 0, left = kLanes, dcur = 2;
				unsigned long long GAP = 0, LM = 0;
				walk_masks(NOTD, ISL, STOP, p, left, dcur, GAP, LM);
				const int nout = kLanes - left;
				if (lane < nout) ops[n + lane] = (uint8_t)(((GAP >> lane) & 1ull) ? (((LM >> lane) & 1ull) ? DIR_L : DIR_U) : DIR_D);
				n += nout;
				r -= p + (dcur > 2 ? dcur - 2 : 0);
				k -= p + (dcur < 2 ? 2 - dcur : 0);
				TBW_COUNT(5);
				if (nout == 0) break;                              /* border, or outside this set */
			}
			if (lane == 0) {
				pos[0] = r;
				pos[1] = k;
				pos[2] = n;
			}
		}
		TBW_LAP(0);
		TBW_COUNT(3);
		__syncthreads();
		if (planning) continue;                                 /* the walk starts in the next round, while the set after this one is replayed */
		r = pos[0];
		k = pos[1];
		n = pos[2];
		__syncthreads();
		if (r > 0 && k > 0) {
			/* does the walk go on in the set just replayed?  The current cell must lie in one of its windows */
			const TbPlan &Np = Ns.plan;
			const int gw = (k - 1) >> 5;
			const int wi = lane_of_word<W>(gw);
			const int rel = kBitBlock * (Np.btop - Ns.dbase) + kBitBlock - 1 - ((r - 1) + (wi & 63));
			bool go = have_next && (wi >> 6) == Np.s && (unsigned)rel < (unsigned)kSetRows;
			if (go) go = (unsigned)(gw - wtab[1 - cur][rel >> 5]) < (unsigned)kWinWords;
			if (go) {
				cur ^= 1;
				Cs = Ns;
			} else {
				need_plan = true;
			}
		}
	}
#ifdef CSADP_TB_TIMERS
	if ((threadIdx.x == 0 || threadIdx.x == 64) && blockIdx.x == 0)
		printf("windowed traceback timers (wave %d, cycles): rounds %u plans %u walk iterations %u  work %u  waiting %u  ops %d\n", wv, tmr[wv][3], tmr[wv][4],
		       tmr[wv][5], tmr[wv][0], tmr[wv][1], n);
#endif
	if (threadIdx.x == 0) {
		summary[0] = n;
		summary[1] = r;
		summary[2] = k;
		summary[3] = 0;
	}
}

/* function attributes are per device: called by Engine::init with that device current */
hipError_t configure_kernels() { return hipSuccess; }

/* static LDS of a fill workgroup of `waves` strips / of a traceback workgroup (without match masks), for the engine's
 * choice of how much dynamic LDS a fill launch reserves on top */
int fill_bits_lds_bytes(int waves) { return waves * (kRing * 16 + kBitBlock * kInjWords * 4 + 8) + (kBitBlock * kInjWords + 4) * 4; }
int traceback_bits_lds_bytes(int)
{
	return 2 * (kSetRows + 2) * kWinPitch * 8 + kSetPieces * kBitBlock * kInjWords * 4 + (kBitBlock * kInjWords + 4) * 4 + 2 * (kSetPieces + 2) * 4 + 16;
}

namespace {

template <int W, int WAVES>
hipError_t launch_fill_w(bool chunked, uint8_t *arena, const BitJob *jobs, int njobs, int passes, int threads, const TileRef *work, int nwork,
                         uint32_t epoch, int *abort_word, hipStream_t st)
{
	/* chunked: `passes` doubles as nothing else; one workgroup per job: `passes` carries the dynamic LDS to reserve */
	/* (whole jobs, work != nullptr: shared workgroups of four waves -- nwork of them per pass, `threads` passes; launch_fill_bits_shared) */
	if constexpr (WAVES == 4) {
		if (!chunked && work != nullptr) {
			hipLaunchKernelGGL((nw_fill_bits<W, 4, false, 4>), dim3(nwork, threads), dim3(4 * kLanes), passes, st, arena, jobs, njobs, work, epoch, abort_word);
			return hipGetLastError();
		}
	}
	if constexpr (WAVES == 4) {
		if (chunked && threads < 0) {                     /* (launch_fill_bits_wide, shared: nwork workgroups of four table entries each) */
			hipLaunchKernelGGL((nw_fill_bits<W, 4, true, 4>), dim3(nwork, passes), dim3(4 * kLanes), 0, st, arena, jobs, njobs, work, epoch, abort_word);
			return hipGetLastError();
		}
	}
	if (chunked) hipLaunchKernelGGL((nw_fill_bits<W, WAVES, true>), dim3(nwork, passes), dim3(WAVES * kLanes), 0, st, arena, jobs, njobs, work, epoch, abort_word);
	else hipLaunchKernelGGL((nw_fill_bits<W, WAVES, false>), dim3(njobs), dim3(threads), passes, st, arena, jobs, njobs, work, epoch, abort_word);
	return hipGetLastError();
}

template <int W>
hipError_t launch_fill_waves(int waves, bool chunked, uint8_t *arena, const BitJob *jobs, int njobs, int passes, int threads, const TileRef *work,
                             int nwork, uint32_t epoch, int *abort_word, hipStream_t st)
{
	if (waves == 4) return launch_fill_w<W, 4>(chunked, arena, jobs, njobs, passes, threads, work, nwork, epoch, abort_word, st);
	if (waves == 8) return launch_fill_w<W, 8>(chunked, arena, jobs, njobs, passes, threads, work, nwork, epoch, abort_word, st);
	if (waves == 16) return launch_fill_w<W, 16>(chunked, arena, jobs, njobs, passes, threads, work, nwork, epoch, abort_word, st);
	return hipErrorInvalidValue;
}

hipError_t launch_fill_any(int words, int waves, bool chunked, uint8_t *arena, const BitJob *jobs, int njobs, int passes, int threads,
                           const TileRef *work, int nwork, uint32_t epoch, int *abort_word, hipStream_t st)
{
	if (words == 1) return launch_fill_waves<1>(waves, chunked, arena, jobs, njobs, passes, threads, work, nwork, epoch, abort_word, st);
	if (words == 2) return launch_fill_waves<2>(waves, chunked, arena, jobs, njobs, passes, threads, work, nwork, epoch, abort_word, st);
	if (words == 3) return launch_fill_waves<3>(waves, chunked, arena, jobs, njobs, passes, threads, work, nwork, epoch, abort_word, st);
	if (words == 4) return launch_fill_waves<4>(waves, chunked, arena, jobs, njobs, passes, threads, work, nwork, epoch, abort_word, st);
	return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_fill_bits(int words, uint8_t *arena, const BitJob *jobs, int njobs, int maxstrips, int lds_pad, int *abort_word, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	if (maxstrips < 1 || maxstrips > kBitMaxStrips || lds_pad < 0 || lds_pad > 60 * 1024) return hipErrorInvalidValue;
	const int waves = maxstrips <= 4 ? 4 : maxstrips <= 8 ? 8 : 16;
	return launch_fill_any(words, waves, false, arena, jobs, njobs, lds_pad, maxstrips * kLanes, nullptr, 0, 0u, abort_word, st);
}

/* jobs of at most four strips sharing four-wave workgroups: `table` = nwgs x 4 entries {job of the pass, strip} (job < 0: empty), the same for
 * each of the `passes` passes whose job tables follow each other (njobs per pass) */
hipError_t launch_fill_bits_shared(int words, uint8_t *arena, const BitJob *jobs, int njobs, int passes, const TileRef *table, int nwgs, int lds_pad,
                                   int *abort_word, hipStream_t st)
{
	if (njobs <= 0 || nwgs <= 0 || passes <= 0) return hipSuccess;
	if (table == nullptr || lds_pad < 0 || lds_pad > 60 * 1024) return hipErrorInvalidValue;
	return launch_fill_any(words, 4, false, arena, jobs, njobs, lds_pad, passes, table, nwgs, 0u, abort_word, st);
}

hipError_t launch_fill_bits_wide(int words, int waves, uint8_t *arena, const BitJob *jobs, int njobs, int passes, const TileRef *work, int nwork,
                                 uint32_t epoch, int *abort_word, hipStream_t st, bool shared)
{
	if (njobs <= 0 || nwork <= 0 || passes <= 0) return hipSuccess;
	if (epoch == 0) return hipErrorInvalidValue;           /* zeroed granules must never look valid */
	if (shared && waves != 4) return hipErrorInvalidValue; /* `work`: four {job, strip} entries per workgroup, nwork workgroups */
	return launch_fill_any(words, waves, true, arena, jobs, njobs, passes, shared ? -1 : waves * kLanes, work, nwork, epoch, abort_word, st);
}

hipError_t launch_traceback_bits(int words, uint8_t *arena, const BitJob *jobs, int njobs, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	if (words == 1) hipLaunchKernelGGL((nw_traceback_windows<1>), dim3(njobs), dim3(kTbThreads), 0, st, arena, jobs);
	else if (words == 2) hipLaunchKernelGGL((nw_traceback_windows<2>), dim3(njobs), dim3(kTbThreads), 0, st, arena, jobs);
	else if (words == 3) hipLaunchKernelGGL((nw_traceback_windows<3>), dim3(njobs), dim3(kTbThreads), 0, st, arena, jobs);
	else if (words == 4) hipLaunchKernelGGL((nw_traceback_windows<4>), dim3(njobs), dim3(kTbThreads), 0, st, arena, jobs);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

}  // namespace csadp
