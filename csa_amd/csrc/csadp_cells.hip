/*
 * csadp_cells.hip -- the matrix fill of dynamicprogramming.c:990-1029 for ANY progressive step (i >= 1,
 * stale borders included) as a persistent cell-per-lane wavefront, and its direction walk
 * (dynamicprogramming.c:1037-1047).  gfx950, wave64.
 *
 *   nw_fill_cells<WIDE, FETCH>   K1c: a lane owns two adjacent columns, a wave 128, a workgroup kCellWaves strips
 *                                (FETCH: + two helper waves for the hand-off between workgroups, launches of few workgroups)
 *   (the direction walk over K1c's tags: csadp_cells_tb.hip)
 *
 * Why this shape.  The reference's own use of the DP (mode N) is a handful of wide gaps, each a chain
 * of up to 63 strictly sequential profile fills: what counts is the LATENCY of one fill, and a fill's
 * critical path is its nrows + ncols anti-diagonals.  The tiled kernel (csadp_kernels.hip) gives a
 * lane 16 columns x 2 rows per step -- ~200 instructions -- and needs nrows/2 + ncols/16 such steps
 * and a launch per tile anti-diagonal; here a step is ONE row of the lane's two columns (13 VALU
 * instructions in the generated statement, csadp_cells_block.inc) and a matrix takes nrows + ncols / 2 of
 * them, on one wave per SIMD so that nothing else competes for the issue slot.
 * Gain form and tie-break as in csadp_device.h: X = 4*H + 4*i*r, candidates tagged U 0 / L 1 / D 2,
 * one v_max3_i32 yields the reference's H and the reference's direction (D >= L >= U, :1014-1025).
 *
 * Data flow.  At local step l lane L of a strip works on row l - L + 1 of its columns A (even) and B
 * (odd); A's left neighbour and the letter offset of that row come from lane L-1 (its column B; DPP
 * wave_shr:1), which had the row one step earlier, B's left neighbour is A's fresh value.  Lane 0
 * takes the VALUE per block of 32 steps from the border column (first strip of a job), from the words
 * the previous strip's lane 63 left in the LDS ring 63 steps earlier (same workgroup; counted in HALF blocks
 * since round 4: a strip follows its neighbour at 80 steps, not 96 -- RingHalf below), or from `hand`
 * in HBM (previous chunk: 8-byte granules tagged with the launch's epoch, written through by one store
 * each -- no counter, no fence.  FETCH layout: a publisher wave sends them half a block at a time and a
 * fetcher wave of the next workgroup polls them into an LDS ring the chunk's first strip reads like any
 * other; plain layout: the last strip stores them itself and the next chunk's first strip requests them two
 * blocks ahead); the LETTER offsets of a block's rows every strip takes from the row table itself (eight
 * vector registers, reloaded by the statement).  Directions:
 * 16 steps of 2-bit tags per word and lane, one coalesced 256-byte store per wave every 16 steps =
 * 0.25 B/cell, the algorithmic figure of SURVEY 8(d).
 *
 * Waits.  Inside a workgroup all waves are resident, so ring waits always end.  Across workgroups the
 * launch relies on the work list (a job's chunks in ascending order) being dispatched in order -- for
 * speed only: every spin is bounded, a time-out raises the abort word and the host repeats the pass
 * chunk by chunk (csadp_engine.cpp: check_abort), where every producer has finished before its
 * consumer starts.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csadp_config.h"
#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

namespace {

constexpr int kRing = 8;                          /* hand-off blocks buffered per strip boundary */
constexpr int kRingSteps = kRing * kCellBlock;
constexpr unsigned long long kSpinTicks = 50000000ull;    /* bound of every wait: 0.5 s of the 100 MHz s_memrealtime clock (round 2: an iteration count) */

/* Counters in LDS that order LDS data only: the LDS executes one wave's instructions in order and is
 * coherent inside the compute unit, so relaxed accesses suffice -- a workgroup-scope release/acquire
 * would also drain the wave's outstanding direction stores (vmcnt(0)), a microsecond per 32-step block. */
__device__ __forceinline__ bool wait_lds(const int *counter, int need)
{
	if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) return true;
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	int spins = 0;
	while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
		if ((++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) return false;
	}
	return true;
}

/* per lane: two adjacent columns, A (even) and B (odd) */
struct CellColumn {
	uint32_t tab;       /* gain table of the column (bytes 8*sv + 2, or 6-bit counts)                   */
	uint32_t tabf;      /* the same for the hand-scheduled blocks: bytes reduced by leftc               */
	int32_t leftc;      /* 4*(gaps - i) + 1                                                             */
	int32_t hup;        /* X of the cell above (this column, previous row)                              */
	int32_t diag;       /* D = X of the cell above-left + leftc                                         */
	int32_t outv;       /* X of the column's cell in the current row, tag cleared                       */
};
struct CellState {
	CellColumn A, B;
	uint32_t outs;      /* the letter offset of the row this lane has just worked on                    */
};

/* What feeds lane 0 of a strip: the border column (the job's first strip), the ring of the wave to the
 * left (same workgroup), or the previous chunk's hand-off granules in HBM (staged in `inject`).  One loop
 * per role, so that the compiler's wait-count bookkeeping stays exact: with the three merged into one
 * loop body every wave waited for ALL its outstanding global accesses -- the direction stores it had just
 * issued -- once per 32-step block (s_waitcnt vmcnt(0) at the merge points). */
enum { ROLE_FIRST = 0, ROLE_RING = 1, ROLE_CHUNK = 2 };
#ifndef CSADP_GRANULE_AHEAD
#define CSADP_GRANULE_AHEAD 2
#endif
constexpr int kGranuleAhead = CSADP_GRANULE_AHEAD;                 /* hand-off granules are requested this many blocks ahead (= unroll of the block loop);
                                                  * 4 was measured: long matrices -3 %, many-strip matrices +5 % (more lag per chunk) */

/*
 * A strip's 32-step block, hand-scheduled: ONE inline-assembly statement on fixed registers,
 * generated by tools/gen_cells_block.py (csadp_cells_block.inc).  Measured on a wave alone on its SIMD
 * (tools/cellstep_microbench.hip): a VALU instruction costs ~4.2 cycles whatever it is and whatever it depends
 * on, an LDS store 16-27 cycles of the WAVE's issue time (b32 16, 2 x b32 20, b128 27), an LDS load ~6.  So:
 *   - the value from the left and the column's gap term are ONE instruction (v_add_u32_dpp: lf = X[left
 *     lane] + leftc; lane 0 has no source lane and keeps what the destination held: its hand-off value +
 *     leftc, prepared at the head of the block);
 *   - the letter offset runs one step ahead of the values (it only ever moves to the right), the next step's
 *     diagonal candidate dg' = lf + gain'(letter') is finished inside this step: the table bytes are
 *     pre-reduced by leftc (gain' = 8*sv + 2 - leftc), which makes lf + gain' = in + gain;
 *   - only X is handed to the next strip (its lane 0 takes the letters from the row sequence, like every first
 *     strip): lane 63's 32 values stay in registers and leave in 8 x ds_write_b128, 9 x ds_read_b128 bring the
 *     next strip's in -- 17 LDS instructions per block where the plain form has 48;
 *   - a lane works on two adjacent columns per step: B's left neighbour is A's fresh value (a plain v_add), so the
 *     cross-lane move, the letter move and every hand-off serve two cells.
 * 13 VALU per step (two cells) + 2 per step at the block's head (lane-0 presets of X and letter offset).
 * A strip's first two blocks (the ramp: lane l's first row arrives at step l) are the same statement with a live mask that moves to the
 * right with the letters and v_bfi_b32 in place of v_and_b32 (CELLS_RAMP_ASM_*: 14 VALU per step; the generator's header says why nothing
 * else needs the mask).  Rounds 2-4 ran them as plain C++ at 1.5 x the time of a hand-scheduled block -- and a strip cannot start before its
 * left neighbour is through two and a half blocks: every strip boundary of a matrix waited for that (tools/r04/cells_times.py: 4.65 us
 * behind the neighbour at the start where 3.6 us are kept up later on).
 */
#include "csadp_cells_block.inc"

/* Half-block hand-off inside a workgroup (round 4).  A strip's lane 0 needs, at its step t of block b, what the left strip's lane 63
 * put out at that strip's step 32 b + t + 63.  Waiting for whole blocks (round 3) started a strip 96 steps behind its neighbour;
 * now the producer also counts HALF blocks (after its 16th step, inside the generated block) and the consumer reads its window in
 * two halves: words 0..19 before the block (the producer's first half of block b + 2), words 20..35 in front of step 17 behind a
 * bounded poll (its second half): 80 steps.  `made` counts half blocks: 2 b + 1 after half a block, 2 b + 2 after block b. */
struct RingHalf {
	const uint32_t *window_b;     /* the window's words 4.. come from here + 16 q bytes (the ring's start - 16 bytes where the window straddles its end) */
	const int *counter;           /* the producer's half-block counter */
	int need1, need2;             /* its values from which the window's first / second half may be read */
	uint32_t *cslot;              /* where THIS wave counts its own half blocks (lane 63: the counter; other lanes: scrap) */
	uint32_t *tslot;              /* ring strips: where lane 0 says which block of its producer's ring it has taken (other lanes: scrap) */
	int chalf;                    /* 2 b + 1 */
};

/* The statement's operands.  The lane state (outvA .. sh) goes in and comes out in the same registers; the block's letters are eight FIXED
 * vector registers with the same content in every lane (CELLS_LT0..7, register variables of run_strip: the statement asks for the next
 * block's itself). */
#define CELLS_BLOCK_OPERANDS                                                                                                   \
	: [outvA] "+v"(outvA), [outvB] "+v"(outvB), [dgA] "+v"(dgA), [dgB] "+v"(dgB), [sh] "+v"(sh), [w0A] "=&v"(w0A),             \
	  [w1A] "=&v"(w1A), [w0B] "=&v"(w0B), [w1B] "=&v"(w1B), [tmo] "+s"(tmo), [scnt] "=&s"(scnt), [sval] "=&s"(sval),           \
	  [vtmp] "=&v"(vtmp), [l0] "+v"(lt0), [l1] "+v"(lt1), [l2] "+v"(lt2), [l3] "+v"(lt3), [l4] "+v"(lt4), [l5] "+v"(lt5),      \
	  [l6] "+v"(lt6), [l7] "+v"(lt7)                                                                                           \
	: [tabA] "v"(S.A.tabf), [tabB] "v"(S.B.tabf), [leftcA] "v"(S.A.leftc), [leftcB] "v"(S.B.leftc), [c2A] "v"(c2A),           \
	  [c2B] "v"(c2B), [waddr] "v"(waddr), [raddr] "v"(raddr), [x0] "v"(xfirst), [lm] "v"(leftmul), [raddrb] "v"(raddrb),       \
	  [paddr] "v"(paddr), [need1] "s"(R.need1), [need2] "s"(R.need2), [lm0] "v"(lm0), [caddr] "v"(caddr), [chalf] "v"(chalf),  \
	  [taddr] "v"(taddr), [tval] "v"(tval), [bidx] "s"(bidx), [lbase] "s"(lbase), [lvoff] "v"(loff), [young] "s"(young)                             \
	: CELLS_BLOCK_CLOBBERS, "scc"

constexpr int kRingWords = kRingSteps;                     /* (round 3: + a mirror of the first two blocks; the window is read from two addresses now) */
constexpr int kInjectWords = 40;                           /* X of step t at word 3 + t: the same 9 x 16-byte window as the ring's */
constexpr int kScrapWords = 4 * kLanes + 64;               /* lane l: 16 bytes at 16 l (+ 16 q per store of a block) */

/* A hand-off granule requested far ahead.  The request is inline assembly so that the compiler's wait counts do
 * not know it: its own waits (vmcnt(0) at every merge point of the loop below) would otherwise wait for the
 * request issued one block ago, i.e. for a trip through memory per block.  `dst` holds 0 (never a valid epoch)
 * until the data arrives; should the compiler ever copy the register before that, the copy fails validation and
 * the granule is simply read again. */
__device__ __forceinline__ void granule_request(unsigned long long &dst, const unsigned long long *p)
{
	asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(dst) : "v"(p) : "memory");
}
/* the rare second look at a granule that had not arrived: request and wait in one piece */
__device__ __forceinline__ void granule_reload(unsigned long long &dst, const unsigned long long *p)
{
	asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "+v"(dst) : "v"(p) : "memory");
}
/* vmcnt counts in order.  When the request for block b is due (head of block b), the wave has issued behind it, oldest first: the four
 * direction stores of block b - 1 (a block old: long done), the request for block b + 1 (kGranuleAhead == 2: issued in block b - 1), and the
 * two letter loads the last statement asked for at its step 18.  Everything YOUNGER than the due request that should stay in flight: the
 * next request and the two letter loads = 3 (2 when requests go one block ahead).  Round 4 waited for vmcnt(2) / (1), i.e. also for the
 * request issued one block earlier -- less than a trip through memory: the latency the two-ahead request was meant to hide was partly
 * paid (round-4 ADVICE; A/B of the plain layout: profiles/r05_granule_wait_ab.txt). */
#ifndef CSADP_GRANULE_KEEP
#define CSADP_GRANULE_KEEP (kGranuleAhead == 2 ? 3 : 2)
#endif
__device__ __forceinline__ void granule_wait(unsigned long long &dst)
{
	asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dst) : "n"(CSADP_GRANULE_KEEP) : "memory");
}

#ifdef CSADP_CELL_TIMERS
/* probe builds only (tools/build_variant.sh): per strip, at its middle block: 100 MHz time at entry, with the hand-off in hand, at the block's end; shader clock of the
 * same three; the XCC the wave runs on */
__device__ unsigned long long g_cell_times[8192 * 12];   /* + [8] entry of block 0, [9] end of the last block, [10] end of block 2 (100 MHz) */
#endif

/*
 * The fetcher of a chunked job's workgroup (launches of few workgroups: one per compute unit, launch_fill_cells): a FIFTH wave that does
 * nothing but bring the previous chunk's hand-off granules in, half a block (16 granules: publish_halves below) at a time.  It keeps FOUR requests for them in flight, a quarter of a
 * round trip apart, looks at each as it lands and asks again; the first sample in which all 16 carry this launch's epoch AND the half block's number goes into ring 0 of
 * the workgroup, the half-block counter behind it -- the chunk's first strip is then a strip like every other (ROLE_RING: LDS window in two
 * halves, its poll inside the generated block) and sees a block of its producer one trip through memory after that was published.  Without
 * it (ROLE_CHUNK below) the strip asks for its granules itself, a block or two ahead: it cannot ask while it computes, so it settles as far
 * behind its producer as makes every request find its data (measured, tools/r04/cells_times.py: 8.1 us between workgroups where three
 * blocks, 4.5 us, plus a trip are needed).
 * One statement on fixed registers: the compiler must not move a register a load is still in flight to.  Bounded by a count of looks
 * (about a second); `false` = ran out.
 */
__device__ __forceinline__ bool fetch_granules(const unsigned long long *hand_in, uint32_t *ring0, int *made0, const int *taken0, int nb,
                                               uint32_t epoch, int lane)
{
	constexpr int kHalf = kCellBlock / 2;                     /* the unit of a hand-off through memory: the 16 granules of HALF a block (publish_halves) */
	const uint32_t rlane = (uint32_t)(uintptr_t)(ring0 + (lane & (kHalf - 1)));         /* lanes 16..63 repeat lanes 0..15 */
	const uint32_t faddr = (uint32_t)(uintptr_t)made0, taddr = (uint32_t)(uintptr_t)taken0;
	const uint32_t epsh = epoch << 8;
	uint32_t voff = (uint32_t)(lane & (kHalf - 1)) * 8u;
	const int nhalves = 2 * nb;
	uint32_t tmo = 0, k, tag, scnt, st, sv;
	unsigned long long sx;
	/* v[100:107]: the four requests; v108: the block's 32 values; v109: scratch.  A request that lands after its block was delivered carries the
	 * previous block's tag and is simply asked again at the new address -- nothing is ever drained before the last block */
#define CSADP_FETCH_ASK(LO) "global_load_dwordx2 v[" #LO ":" #LO "+1], %[voff], %[base] sc1\n\t"
#define CSADP_FETCH_LOOK(HI, FOUND)                                                                                            \
	"s_waitcnt vmcnt(3)\n\t"                                                                                                   \
	"v_cmp_ne_u32 vcc, %[tag], v" #HI "\n\t"                                                                                   \
	"s_cbranch_vccz " #FOUND "f\n\t"
	/* half block k (block k >> 1) is in v<LO>: wait for its ring slots (they last held block (k >> 1) - kRing, which the strip has in registers
	 * once taken0 >= (k >> 1) - kRing), store the values, then the counter k + 1 from lane 0 alone (the LDS runs a wave's stores in order); next */
#define CSADP_FETCH_DELIVER(LO, L1, L2, BACK)                                                                                  \
	"v_mov_b32 v108, v" #LO "\n\t"                                                                                             \
	"s_lshr_b32 %[st], %[k], 1\n\t"                                                                                            \
	"s_sub_u32 %[st], %[st], %[ring]\n\t"                                                                                      \
	"s_cmp_le_i32 %[st], 0\n\t"                                                                                                \
	"s_cbranch_scc1 " #L2 "f\n\t"                                                                                              \
	#L1 ":\n\t"                                                                                                                \
	"ds_read_b32 v109, %[taddr]\n\t"                                                                                           \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                                 \
	"v_readfirstlane_b32 %[sv], v109\n\t"                                                                                      \
	"s_cmp_ge_i32 %[sv], %[st]\n\t"                                                                                            \
	"s_cbranch_scc1 " #L2 "f\n\t"                                                                                              \
	"s_sleep 2\n\t"                                                                                                            \
	"s_sub_u32 %[scnt], %[scnt], 1\n\t"                                                                                        \
	"s_cmp_lg_u32 %[scnt], 0\n\t"                                                                                              \
	"s_cbranch_scc1 " #L1 "b\n\t"                                                                                              \
	"s_mov_b32 %[tmo], 1\n\t"                                                                                                  \
	"s_branch 9f\n\t"                                                                                                          \
	#L2 ":\n\t"                                                                                                                \
	"s_and_b32 %[st], %[k], %[ringm]\n\t"                                                                                      \
	"s_lshl_b32 %[st], %[st], 6\n\t"                                                                                           \
	"v_add_u32 v109, %[st], %[rlane]\n\t"                                                                                      \
	"ds_write_b32 v109, v108\n\t"                                                                                              \
"s_add_u32 %[st], %[k], 1\n\t"                                                                                             \
	"v_mov_b32 v109, %[st]\n\t"                                                                                                \
	"s_mov_b64 %[sx], exec\n\t"                                                                                                \
	"s_mov_b64 exec, 1\n\t"                                                                                                    \
	"ds_write_b32 %[faddr], v109\n\t"                                                                                          \
	"s_mov_b64 exec, %[sx]\n\t"                                                                                                \
	"s_add_u32 %[k], %[k], 1\n\t"                                                                                              \
	"s_cmp_ge_u32 %[k], %[nb]\n\t"                                                                                             \
	"s_cbranch_scc1 9f\n\t"                                                                                                    \
	"v_add_u32 %[voff], 0x80, %[voff]\n\t"                                                                                     \
	"s_and_b32 %[st], %[k], 0xff\n\t"                                                                                          \
	"s_or_b32 %[tag], %[epsh], %[st]\n\t"                                                                                      \
	"s_mov_b32 %[scnt], 0x80000\n\t"                                                                                           \
	"s_branch " #BACK "b\n\t"
	asm volatile(
	    "s_mov_b32 %[k], 0\n\t"
	    "s_mov_b32 %[tag], %[epsh]\n\t"
	    "s_mov_b32 %[scnt], 0x80000\n\t"
	    CSADP_FETCH_ASK(100) "s_sleep 12\n\t" CSADP_FETCH_ASK(102) "s_sleep 12\n\t" CSADP_FETCH_ASK(104) "s_sleep 12\n\t" CSADP_FETCH_ASK(106)
	    "1:\n\t"
	    CSADP_FETCH_LOOK(101, 20)
	    "21:\n\t" CSADP_FETCH_ASK(100)
	    CSADP_FETCH_LOOK(103, 30)
	    "31:\n\t" CSADP_FETCH_ASK(102)
	    CSADP_FETCH_LOOK(105, 40)
	    "41:\n\t" CSADP_FETCH_ASK(104)
	    CSADP_FETCH_LOOK(107, 50)
	    "51:\n\t" CSADP_FETCH_ASK(106)
	    "s_sub_u32 %[scnt], %[scnt], 1\n\t"
	    "s_cmp_lg_u32 %[scnt], 0\n\t"
	    "s_cbranch_scc1 1b\n\t"
	    "s_mov_b32 %[tmo], 1\n\t"
	    "s_branch 9f\n\t"
	    "20:\n\t" CSADP_FETCH_DELIVER(100, 22, 23, 21)
	    "30:\n\t" CSADP_FETCH_DELIVER(102, 32, 33, 31)
	    "40:\n\t" CSADP_FETCH_DELIVER(104, 42, 43, 41)
	    "50:\n\t" CSADP_FETCH_DELIVER(106, 52, 53, 51)
	    "9:\n\t"
	    "s_waitcnt vmcnt(0)\n\t"                                /* requests still on their way to v100..v107 */
	    : [tmo] "+s"(tmo), [voff] "+v"(voff), [k] "=&s"(k), [tag] "=&s"(tag), [scnt] "=&s"(scnt), [st] "=&s"(st), [sv] "=&s"(sv), [sx] "=&s"(sx)
	    : [base] "s"(hand_in), [epsh] "s"(epsh), [nb] "s"(nhalves), [rlane] "v"(rlane), [faddr] "v"(faddr), [taddr] "v"(taddr), [ring] "n"(kRing),
	      [ringm] "n"(2 * kRing - 1)
	    : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "vcc", "scc", "memory");
#undef CSADP_FETCH_ASK
#undef CSADP_FETCH_LOOK
#undef CSADP_FETCH_DELIVER
	return tmo == 0;
}

/*
 * The publisher of a chunked job's workgroup (the same launches): a SIXTH wave that sends the last strip's hand-off values to the next chunk,
 * HALF a block at a time -- it follows that strip's half-block counter like a fifth strip would and tells it through `taken` which blocks
 * of its ring are out.  The strip itself then publishes nothing (in the plain layout it reads its ring back and stores 32 granules behind
 * every block), and the next chunk's fetcher sees the first half of a block half a block earlier: two and a half blocks plus a trip
 * between workgroups, as inside one.  Granule = {X, epoch << 8 | half-block number & 255}.  It sleeps between looks (it shares a SIMD
 * with a strip: a look is two vector instructions).
 */
__device__ __forceinline__ bool publish_halves(unsigned long long *hand_out, const uint32_t *ring_last, const int *made_last, int *taken_pub, int nb,
                                               uint32_t epoch, int lane, int test_slow)
{
	constexpr int kHalf = kCellBlock / 2;
	for (int h = 0; h < 2 * nb; ++h) {
		/* test seam (CSADP_TEST_SLOW_PUBLISHER): a publisher that falls blocks behind its strip -- what a shared or preempted device does to
		 * it -- so that the strip's wait for ring space is what keeps the result right (tests/test_gpu_parity.py) */
		for (int z = 0; z < test_slow; ++z) __builtin_amdgcn_s_sleep(127);
		if (__hip_atomic_load(made_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < h + 1) {
			const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
			int spins = 0;
			while (__hip_atomic_load(made_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < h + 1) {
				__builtin_amdgcn_s_sleep(3);
				if ((++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) return false;
			}
		}
		asm volatile("" ::: "memory");                          /* the compiler must not read the ring before the counter (the LDS itself runs a wave's reads in order) */
		const uint32_t x = ring_last[(h * kHalf) % kRingSteps + (lane & (kHalf - 1))];
		if (lane < kHalf)
			__hip_atomic_store(hand_out + h * kHalf + lane, (unsigned long long)x | ((unsigned long long)((epoch << 8) | (uint32_t)(h & 255)) << 32),
			                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		/* (the store has the values in registers: the strip may overwrite the block's ring slots) */
		if ((h & 1) && lane == 0) __hip_atomic_store(taken_pub, (h >> 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	return true;
}

struct StripShared {
	uint32_t *ring_mine;         /* ring[wv]       */
	const uint32_t *ring_prev;   /* ring[wv - 1]   */
	uint32_t *inject_mine;       /* inject[wv]     */
	uint32_t *scrap_mine;        /* scrap[wv]      */
	int *made, *taken;
};

template <bool WIDE, int ROLE>
__device__ __forceinline__ bool run_strip(const CellJob &J, const uint8_t *rsh, CellState &S, uint32_t *dirs, size_t dirs_half,
                                          const StripShared &L, unsigned long long *hand_out, const unsigned long long *hand_in, int wv,
                                          int lane, int nb, bool feeds, bool publishes, uint32_t epoch, int strip)
{
	(void)strip;                                             /* probe builds (CSADP_CELL_TIMERS) */
	const int t = lane & 31;
	/* the letter offsets of a block's 32 rows: eight uniform words (scalar loads), requested one block ahead;
	 * the same for every strip -- lane 0 of each takes them from here, the others get them through DPP */
	/* (constant address space: the row table is written by the host only, so these become scalar loads, counted
	 * apart from the vector memory accesses) */
	typedef const __attribute__((address_space(4))) uint32_t *ConstWords;
	ConstWords rw = (ConstWords)(uintptr_t)rsh;
	/* the block's 32 letter offsets: eight registers the statement names itself (it loads the next block's into them) */
	register uint32_t lt0 asm(CELLS_LT0) = rw[0];
	register uint32_t lt1 asm(CELLS_LT1) = rw[1];
	register uint32_t lt2 asm(CELLS_LT2) = rw[2];
	register uint32_t lt3 asm(CELLS_LT3) = rw[3];
	register uint32_t lt4 asm(CELLS_LT4) = rw[4];
	register uint32_t lt5 asm(CELLS_LT5) = rw[5];
	register uint32_t lt6 asm(CELLS_LT6) = rw[6];
	register uint32_t lt7 asm(CELLS_LT7) = rw[7];
	const unsigned long long lbase = (unsigned long long)(uintptr_t)rsh;
	/* ROLE_CHUNK: 8-byte granules {X, epoch << 8}, each written by ONE write-through store and valid exactly
	 * when it carries this launch's epoch -- no counter, no fence, one memory round trip, and that one is
	 * hidden: the granules of block b + 2 are requested while block b is computed and only re-read (bounded)
	 * if they had not arrived.  TWO blocks ahead, because producer and consumer usually sit on different XCDs and
	 * the round trip through memory (~1.3 us) is longer than a block (~0.8 us); the loop below is unrolled by two
	 * so that each of the two requests in flight has its own register and no copy waits for it early. */
	unsigned long long preA = 0, preB = 0;
	/* the job's fields the loop needs, in registers: the block is an asm statement with a "memory" clobber, after which
	 * the compiler read them from the job record again -- a scalar load and its wait in front of every block */
	const int steps_pad = J.steps_pad, leftmul = ROLE == ROLE_FIRST ? J.leftmul : 0;
	const int last_granule = steps_pad - 1;                     /* requests past the end repeat this one; nobody looks at them */
	if (ROLE == ROLE_CHUNK) {
		granule_request(preA, hand_in + min(63 + t, last_granule));
		if (kGranuleAhead == 2) granule_request(preB, hand_in + min(kCellBlock + 63 + t, last_granule));
	}
	/* Order of global accesses.  The compiler's wait before the first use of a loaded value is
	 * s_waitcnt vmcnt(0): it also waits for every store issued since.  So a block first consumes what was
	 * loaded for it, THEN stores the direction words of the previous block and requests the next block's
	 * data: everything a wait can see was issued a whole block (~1 us) earlier and costs nothing. */
	uint32_t words[2][kCellBlock / 16] = {{0, 0}, {0, 0}};        /* [column A / B][half of the block] */
	int known_taken = 0;
	uint32_t pubx = 0;
	auto publish = [&](int b) {
		if (lane < kCellBlock)
			__hip_atomic_store(hand_out + b * kCellBlock + lane, (unsigned long long)pubx | ((unsigned long long)((epoch << 8) | (b & 255)) << 32),
			                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	};
	auto block = [&](int b, unsigned long long &pre) -> bool {
#ifdef CSADP_CELL_TIMERS
		const bool timed = b == nb / 2 && strip < 8192;
		if (b == 0 && strip < 8192 && lane == 0) g_cell_times[12 * strip + 8] = __builtin_amdgcn_s_memrealtime();
		unsigned long long tr0 = 0, tc0 = 0, tr1 = 0, tc1 = 0;
		if (timed) {
			tr0 = __builtin_amdgcn_s_memrealtime();
			tc0 = __builtin_amdgcn_s_memtime();
		}
#endif
		const uint32_t *window = L.inject_mine;                 /* 16-byte aligned; X of step t at word 3 + t */
		RingHalf R;
		R.window_b = window;
		R.counter = &L.made[ROLE == ROLE_RING ? wv - 1 : 0];           /* [-1]: the fetcher's (nw_fill_cells) */
		R.need1 = R.need2 = 0;
		if (ROLE == ROLE_RING) {
			/* `made` counts half blocks.  The hand-scheduled blocks (b >= 2) read their window in two halves: words 0..19 need the
			 * producer's steps up to 32 b + 79 = the first half of its block b + 2; so do the ramp blocks */
			const int need_all = std::min(2 * (b + 3), 2 * nb), need_half = std::min(2 * (b + 2) + 1, 2 * nb);
			R.need1 = need_half;
			R.need2 = need_all;
			const int start = (b * kCellBlock + 60) % kRingSteps;      /* 36 words from here; at 252 they straddle the ring's end */
			window = L.ring_prev + start;
			R.window_b = start + 36 > kRingSteps ? L.ring_prev - 4 : window;
		} else if (ROLE == ROLE_CHUNK) {
			const int ps = b * kCellBlock + 63 + t;               /* the producer's lane 63 is 63 steps ahead */
			granule_wait(pre);
			unsigned long long v = pre;
			int spins = 0;
			unsigned long long t0 = 0;
			for (;;) {
				const bool ok = ps >= steps_pad || (uint32_t)(v >> 40) == epoch;
				if (__all(ok)) break;
				__builtin_amdgcn_s_sleep(2);
				if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
				if ((++spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) return false;
				if (!ok) granule_reload(v, hand_in + ps);
			}
			if (lane < kCellBlock) L.inject_mine[3 + lane] = (uint32_t)v;
		}
#ifdef CSADP_CELL_TIMERS
		if (timed) {
			tr1 = __builtin_amdgcn_s_memrealtime();
			tc1 = __builtin_amdgcn_s_memtime();
		}
#endif
		if (b > 0) {
			uint32_t *d = dirs + (size_t)(b - 1) * (kCellBlock / 16) * kLanes;
			d[0] = words[0][0];
			d[kLanes] = words[0][1];
			d[dirs_half] = words[1][0];
			d[dirs_half + kLanes] = words[1][1];
		}
		if (ROLE == ROLE_CHUNK) {
			pre = 0;
			granule_request(pre, hand_in + min((b + kGranuleAhead) * kCellBlock + 63 + t, last_granule));
		}
		const bool ringer = lane == kLanes - 1 && (feeds || publishes);
		uint32_t *lanebuf = ringer ? L.ring_mine + (b * kCellBlock) % kRingSteps : L.scrap_mine + 4 * lane;
		/* the ring slots of this block last held block b - kRing.  A STRIP reads it during its blocks b - kRing - 2 and b - kRing - 1 and
		 * stores taken = c + 1 behind its block c: taken >= b - kRing covers it.  The PUBLISHER (the consumer of a chunk's last strip in
		 * the helper-wave layout) stores taken = B + 1 when both halves of block B are out: it takes taken >= b - kRing + 1 (round-4
		 * ADVICE: with the strips' threshold a publisher seven blocks behind had its slots overwritten under it, and at b == kRing there
		 * was no wait at all).  The consumer's progress is only looked up when the last value seen does not cover this block. */
		const int need_taken = b - kRing + ((feeds && wv + 1 == kCellWaves) ? 1 : 0);
		if (feeds && need_taken > known_taken) {
			if (!wait_lds(&L.taken[wv + 1], need_taken)) return false;
			known_taken = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&L.taken[wv + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));   /* a scalar: the look-ups in between are a compare */
		}
		const int32_t xfirst0 = leftmul * (b * kCellBlock + 1);    /* border column: X[r][0] = leftmul * r (:967); other strips: unused */
		R.cslot = (feeds && lane == kLanes - 1) ? reinterpret_cast<uint32_t *>(&L.made[wv]) : L.scrap_mine + 4 * lane;
		R.chalf = 2 * b + 1;
		R.tslot = (ROLE == ROLE_RING && lane == 0) ? reinterpret_cast<uint32_t *>(&L.taken[wv]) : L.scrap_mine + 4 * lane + 1;
#ifdef CSADP_CELL_TIMERS
		const unsigned long long ta0 = __builtin_amdgcn_s_memtime();
#endif
		/* blocks 0 and 1 are the ramp (lane l is live from step l on: `lm0`), the statement branches on b */
		{
			int32_t outvA = S.A.outv, outvB = S.B.outv, dgA = S.A.diag, dgB = S.B.diag;
			uint32_t sh = S.outs, w0A, w1A, w0B, w1B;
			const uint32_t waddr = (uint32_t)(uintptr_t)lanebuf;       /* LDS byte address = low half of the generic pointer */
			const uint32_t raddr = (uint32_t)(uintptr_t)window;
			const uint32_t raddrb = (uint32_t)(uintptr_t)R.window_b, paddr = (uint32_t)(uintptr_t)R.counter, caddr = (uint32_t)(uintptr_t)R.cslot,
			               taddr = (uint32_t)(uintptr_t)R.tslot, tval = (uint32_t)(b + 1), chalf = (uint32_t)R.chalf;
			const uint32_t lm0 = lane <= b * kCellBlock ? ~3u : 0u;
			const uint32_t loff = (uint32_t)(b + 1) * kCellBlock;      /* bytes: the next block's letters (the row table is padded) */
			const int bidx = b;
			const int young = __builtin_amdgcn_readfirstlane(4 + (publishes ? 1 : 0) + (ROLE == ROLE_CHUNK ? 1 : 0));   /* vector memory instructions between two statements */
			const int32_t xfirst = xfirst0 + S.A.leftc;
			uint32_t tmo = 0, scnt, sval, vtmp;
			const int32_t c2A = 2 - S.A.leftc, c2B = 2 - S.B.leftc;
			if (WIDE && ROLE == ROLE_FIRST) asm volatile(CELLS_BLOCK_ASM_WIDE_FIRST CELLS_BLOCK_OPERANDS);
			else if (WIDE && ROLE == ROLE_RING) asm volatile(CELLS_BLOCK_ASM_WIDE_RING CELLS_BLOCK_OPERANDS);
			else if (WIDE) asm volatile(CELLS_BLOCK_ASM_WIDE_LDS CELLS_BLOCK_OPERANDS);
			else if (ROLE == ROLE_FIRST) asm volatile(CELLS_BLOCK_ASM_BYTE_FIRST CELLS_BLOCK_OPERANDS);
			else if (ROLE == ROLE_RING) asm volatile(CELLS_BLOCK_ASM_BYTE_RING CELLS_BLOCK_OPERANDS);
			else asm volatile(CELLS_BLOCK_ASM_BYTE_LDS CELLS_BLOCK_OPERANDS);
			S.A.outv = S.A.hup = outvA;
			S.B.outv = S.B.hup = outvB;
			S.A.diag = dgA;
			S.B.diag = dgB;
			S.outs = sh;
			words[0][0] = w0A;
			words[0][1] = w1A;
			words[1][0] = w0B;
			words[1][1] = w1B;
			if (tmo != 0) return false;                             /* a poll inside ran out */
		}
#ifdef CSADP_CELL_TIMERS
		if (timed && lane == 0) g_cell_times[12 * strip + 11] = __builtin_amdgcn_s_memtime() - ta0;     /* the statement alone */
#endif
		/* (`taken` = b + 1 and `made` = 2 b + 2 are stored by the statement itself, behind the block's last ring words: the LDS runs a
		 * wave's stores in order, so whoever sees the counter sees the words) */
		/* the block's 32 values for the next chunk: read back from the ring (in order behind the block's
		 * stores) and sent at once -- the next chunk's first strip is waiting on them, and the LDS round trip
		 * costs less than a block's delay does downstream (16384^2: 1.42 -> 1.40 ms) */
		if (publishes) {
			if (lane < kCellBlock) pubx = L.ring_mine[(b * kCellBlock) % kRingSteps + lane];
			publish(b);      /* (the store held back behind the next block's head work, its LDS wait hidden: 16384^2 1.16 -> 1.18 ms) */
		}
#ifdef CSADP_CELL_TIMERS
		if (timed && lane == 0) {
			unsigned long long *g = g_cell_times + 12 * strip;
			g[0] = tr0;
			g[1] = tr1;
			g[2] = __builtin_amdgcn_s_memrealtime();
			g[3] = tc0;
			g[4] = tc1;
			g[5] = __builtin_amdgcn_s_memtime();
			g[6] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   /* HW_REG_XCC_ID, bits 0..3 */
		}
		if (b == nb / 2 + 16 && strip < 8192 && lane == 0) g_cell_times[12 * strip + 7] = __builtin_amdgcn_s_memtime();   /* 16 periods after g[5] */
		if (strip < 8192 && lane == 0 && (b == nb - 1 || b == 2)) g_cell_times[12 * strip + (b == 2 ? 10 : 9)] = __builtin_amdgcn_s_memrealtime();
#endif
		return true;
	};
	if (kGranuleAhead == 2 && ROLE == ROLE_CHUNK) {           /* two requests in flight, each with its own register; the other roles: one copy of the block (a
	                                                           * strip alone: 94 -> 85 cycles per step -- the loop fits the instruction cache better) */
		for (int b = 0; b < nb; b += 2) {
			if (!block(b, preA)) return false;
			if (b + 1 < nb && !block(b + 1, preB)) return false;
		}
	} else {
		for (int b = 0; b < nb; ++b)
			if (!block(b, preA)) return false;
	}
	{
		uint32_t *d = dirs + (size_t)(nb - 1) * (kCellBlock / 16) * kLanes;
		d[0] = words[0][0];
		d[kLanes] = words[0][1];
		d[dirs_half] = words[1][0];
		d[dirs_half + kLanes] = words[1][1];
	}
	return true;
}

}  // namespace

template <bool WIDE, bool FETCH>
__global__ __launch_bounds__((kCellWaves + (FETCH ? 2 : 0)) * kLanes) void nw_fill_cells(uint8_t *__restrict__ arena, const CellJob *__restrict__ jobs,
                                                                    const TileRef *__restrict__ work, uint32_t epoch_and_seam,
                                                                    int *__restrict__ abort_word)
{
	const uint32_t epoch = epoch_and_seam & 0xffffffu;        /* 24 bits travel in a granule; the top byte: the slow-publisher test seam */
	const int test_slow = (int)(epoch_and_seam >> 24);
	/* four words in front of every ring: a window that straddles the ring's end is read from two addresses, the second of them
	 * "ring start - 16 bytes" + 16 q (q >= 1) -- never below the array */
	/* ring[1 + wv]: strip wv's; ring[0]: what the fetcher brings in from the previous chunk (FETCH) */
	__shared__ __attribute__((aligned(16))) uint32_t ring[kCellWaves + 1][4 + kRingWords];
	__shared__ __attribute__((aligned(16))) uint32_t inject[kCellWaves][kInjectWords];
	__shared__ __attribute__((aligned(16))) uint32_t scrap[kCellWaves][kScrapWords];
	__shared__ int made[kCellWaves + 1], taken[kCellWaves + 1];    /* made[1 + wv]: strip wv's half blocks, made[0]: the fetcher's; taken[wv]: blocks strip wv has
	                                                                * taken from its producer's ring, taken[kCellWaves]: the publisher's from the last strip's */

	const TileRef item = work[blockIdx.x];
	const CellJob &J = jobs[item.job];
	const int chunk = item.a;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	const int s = chunk * kCellWaves + wv;                    /* this wave's strip */
	if (threadIdx.x <= kCellWaves) {
		made[threadIdx.x] = 0;
		taken[threadIdx.x] = 0;
	}
	__syncthreads();
	const int nb = J.steps_pad / kCellBlock;
	if (FETCH && wv == kCellWaves) {
		if (chunk == 0 || chunk * kCellWaves >= J.nstrips) return;
		const unsigned long long *from = reinterpret_cast<const unsigned long long *>(arena + J.hand) + (size_t)(chunk - 1) * J.steps_pad;
		if (!fetch_granules(from, ring[0] + 4, &made[0], &taken[0], nb, epoch, lane) && lane == 0) atomicExch(abort_word, 1);
		return;
	}
	if (FETCH && wv == kCellWaves + 1) {                      /* the publisher: the chunk's last strip hands on to a next chunk */
		if ((chunk + 1) * kCellWaves >= J.nstrips) return;
		unsigned long long *to = reinterpret_cast<unsigned long long *>(arena + J.hand) + (size_t)chunk * J.steps_pad;
		if (!publish_halves(to, ring[kCellWaves] + 4, &made[kCellWaves], &taken[kCellWaves], nb, epoch, lane, test_slow) && lane == 0) atomicExch(abort_word, 1);
		return;
	}
	if (s >= J.nstrips) return;

	const int col = s * kCellStripCols + kCellCols * lane;     /* 0-based column A of this lane; B = col + 1 */
	const uint32_t *coltab = reinterpret_cast<const uint32_t *>(arena + J.coltab);
	const int32_t *leftcs = reinterpret_cast<const int32_t *>(arena + J.leftc);
	const int32_t *top = reinterpret_cast<const int32_t *>(arena + J.top);
	const uint8_t *rsh = arena + J.rowshift;
	/* directions: the columns A of a strip form "virtual strip" 2s, the columns B 2s + 1, each [steps_pad / 16][64 lanes] */
	const size_t dirs_half = (size_t)(J.steps_pad / 16) * kLanes;
	uint32_t *dirs = reinterpret_cast<uint32_t *>(arena + J.dirs) + (size_t)s * kCellCols * dirs_half + lane;
	unsigned long long *hand_out = reinterpret_cast<unsigned long long *>(arena + J.hand) + (size_t)chunk * J.steps_pad;
	const unsigned long long *hand_in =
	    reinterpret_cast<const unsigned long long *>(arena + J.hand) + (size_t)(chunk > 0 ? chunk - 1 : 0) * J.steps_pad;
	/* a wave of this workgroup reads my ring: the next strip, or (FETCH) the publisher behind the last strip */
	const bool feeds = s + 1 < J.nstrips && (wv + 1 < kCellWaves || FETCH);
	const bool publishes = !FETCH && wv + 1 == kCellWaves && s + 1 < J.nstrips;   /* plain layout: the last strip sends its hand-off words itself */

	CellState S;
	S.A.tab = coltab[col];
	S.B.tab = coltab[col + 1];
	S.A.leftc = leftcs[col];
	S.B.leftc = leftcs[col + 1];
	S.A.hup = top[col + 1];
	S.B.hup = top[col + 2];
	S.A.diag = top[col] + S.A.leftc;                          /* the blocks keep D = the cell above-left + leftc */
	S.B.diag = top[col + 1] + S.B.leftc;
	S.A.outv = S.A.hup;
	S.B.outv = S.B.hup;
	S.outs = 0;
	StripShared L;
	L.ring_mine = ring[wv + 1] + 4;
	L.ring_prev = ring[wv] + 4;
	L.inject_mine = inject[wv];
	L.scrap_mine = scrap[wv];
	L.made = made + 1;
	L.taken = taken;
	/* table of the hand-scheduled blocks: gain bytes reduced by leftc (8*sv + 2 - leftc = 8*sv + 4*(i - gaps) + 1 <= 12*i + 1:
	 * the host takes the WIDE form from i = 22 on); WIDE keeps the counts and adds 2 - leftc in the step */
	S.A.tabf = S.A.tab;
	S.B.tabf = S.B.tab;
	if (!WIDE) {
		S.A.tabf = S.B.tabf = 0;
#pragma unroll
		for (int y = 0; y < 4; ++y) {
			S.A.tabf |= ((((S.A.tab >> (8 * y)) & 255u) - (uint32_t)S.A.leftc) & 255u) << (8 * y);
			S.B.tabf |= ((((S.B.tab >> (8 * y)) & 255u) - (uint32_t)S.B.leftc) & 255u) << (8 * y);
		}
	}
	bool ok;
	if (s == 0) ok = run_strip<WIDE, ROLE_FIRST>(J, rsh, S, dirs, dirs_half, L, hand_out, hand_in, wv, lane, nb, feeds, publishes, epoch, s);
	else if (wv > 0 || FETCH) ok = run_strip<WIDE, ROLE_RING>(J, rsh, S, dirs, dirs_half, L, hand_out, hand_in, wv, lane, nb, feeds, publishes, epoch, s);
	else ok = run_strip<WIDE, ROLE_CHUNK>(J, rsh, S, dirs, dirs_half, L, hand_out, hand_in, wv, lane, nb, feeds, publishes, epoch, s);
	if (!ok && lane == 0) atomicExch(abort_word, 1);
}

hipError_t launch_fill_cells(bool wide, bool fetch, uint8_t *arena, const CellJob *jobs, const TileRef *work, int nwork, uint32_t epoch,
                             int *abort_word, hipStream_t st, int test_slow_publisher)
{
	if (nwork <= 0) return hipSuccess;
	epoch = (epoch & 0xffffffu) | ((uint32_t)(test_slow_publisher & 255) << 24);     /* 24 bits travel in a granule */
	const dim3 threads((kCellWaves + (fetch ? 2 : 0)) * kLanes);
	if (wide && fetch) hipLaunchKernelGGL((nw_fill_cells<true, true>), dim3(nwork), threads, 0, st, arena, jobs, work, epoch, abort_word);
	else if (wide) hipLaunchKernelGGL((nw_fill_cells<true, false>), dim3(nwork), threads, 0, st, arena, jobs, work, epoch, abort_word);
	else if (fetch) hipLaunchKernelGGL((nw_fill_cells<false, true>), dim3(nwork), threads, 0, st, arena, jobs, work, epoch, abort_word);
	else hipLaunchKernelGGL((nw_fill_cells<false, false>), dim3(nwork), threads, 0, st, arena, jobs, work, epoch, abort_word);
	return hipGetLastError();
}

}  // namespace csadp

#ifdef CSADP_CELL_TIMERS
extern "C" __attribute__((visibility("default"))) int csadp_debug_cell_times(unsigned long long *out, int nstrips)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(csadp::g_cell_times), sizeof(unsigned long long) * 12 * (size_t)nstrips);
}
#endif
