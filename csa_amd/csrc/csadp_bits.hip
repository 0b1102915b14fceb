/*
 * csadp_bits.hip -- bit-parallel form of the pairwise fill (dynamicprogramming.c:990-1029 for a
 * profile of ONE sequence: i = 1, scores +1 match / -1 mismatch / -1 gap, fresh borders) and its
 * tracebacks (dynamicprogramming.c:1037-1047).  gfx950, wave64.
 *
 *   nw_fill_bits<CKPT>    K1b: one workgroup per matrix, one wave per strip of 2048 columns
 *   nw_fill_bits_wide     K1b for matrices wider than 16 strips: a workgroup per chunk of 16 strips
 *   nw_traceback_replay   K2c: checkpoint mode -- replays the blocks on the path, then walks them
 *   nw_traceback_bits     K2b: direction planes in HBM -- walks them through an LDS window
 *
 * Why it is exact.  Let u = H[r][k-1] - H[r-1][k-1] (vertical step left of the cell), w =
 * H[r-1][k] - H[r-1][k-1] (horizontal step above it), both in {-1,0,1,2}.  The reference's cell
 *     H[r][k] = max(diag + s, left - 1, up - 1),  ties D >= L >= U  (:1014-1025)
 * gives c = H[r][k] - H[r-1][k-1] = 1 on a match and max(u, w, 0) - 1 on a mismatch; the new
 * steps are c - w (vertical) and c - u (horizontal); the direction is D iff match or c = -1,
 * else L iff c - u = -1, else U.  Along a row the vertical step is a 4-state machine driven by
 * (match, w); with thermometer planes (">= 0", ">= 1", ">= 2") its ">= 2" and ">= 1" planes are
 * carry chains (generate / propagate), which an integer addition resolves for 32 columns at once,
 * and the ">= 0" plane needs no chain.  tools/bitproto.py checks these formulas against the plain
 * recurrence, tie-breaks included.
 *
 * Work per lane and step: one row of 32 columns in 32 VALU instructions (1 per cell against 4 in
 * the packed-16 kernel: 23 for the recurrence, written as explicit v_bitop3 truth tables, 9 for
 * the hand-off to the right neighbour).  Output: two direction words per step (2 bit per cell), or
 * -- checkpoint mode, the default -- the lane state every 32 steps and four hand-off words per
 * step, 7 % of that, from which the traceback re-derives the directions it needs.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

namespace {

static_assert(kBitCkptWords == 1, "the carry-mask step holds one word of 32 columns per lane");

constexpr int kRing = 8;                 /* hand-off blocks (of 32 steps) buffered per strip boundary */
constexpr int kSpinMax = 1 << 22;        /* bound of every wait (~0.5 s) */

/* v_bitop3_b32: any boolean function of three words in one instruction; the table is the function
 * applied to these three constants */
constexpr uint32_t LA = 0xF0, LB = 0xCC, LC = 0xAA;
#define BITOP3(a, b, c, expr) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), (unsigned char)((expr) & 0xff)))

/* per lane: the horizontal steps of the row above its 32 columns: NOT ">= 0", ">= 1", ">= 2" */
struct BitState {
	uint32_t nH0, H1, H2;
};

/*
 * What crosses from a lane to its right neighbour is one bit per plane and step -- the vertical step
 * at the word's right edge -- and that bit IS the carry out of the word's addition: the ">= 2" and
 * ">= 1" planes are carry chains resolved by s = propagate + generate + carry_in (generate is a subset
 * of propagate, so the carry out of bit k is generate | propagate & carry, the chain itself), and the
 * ">= 0" plane's (O0 << 1) | bit is O0 + O0 + carry_in with carry out = O0's top bit.  v_addc_co_u32
 * takes the carry-in of every lane from an SGPR pair and leaves the carry-outs in one: three lane masks
 * per wave, moved one lane to the right by the SCALAR unit between two steps.  The round-2 form carried
 * them in a VGPR hand-off word (one DPP move, two v_perm, four v_bfe, two v_add3 and two v_bitop3 for the
 * outgoing planes per step, and an LDS store of that word): 31 VALU instructions per step against 22
 * here, and nothing per step in LDS.
 *
 * Carries: the carry-OUT masks of the step before (bit L = lane L).  Feed: what enters lane 0 and
 * what leaves lane 63 -- in: one word per plane and block, consumed from the top bit (step t of the
 * block takes bit 31 - t); acc: the bits leaving lane 63, shifted in from below.  One chain per plane
 * and step does all of it through SCC:  in += in (top bit -> SCC);  lo = 2 lo + SCC;  hi = 2 hi + carry;
 * acc = 2 acc + carry.  The shift at the head of step t moves the carry-outs of step t - 1, so after
 * the 32 steps of block b an accumulator holds the bits that left lane 63 in steps 32b - 1 .. 32b + 30
 * (first at the top), which is exactly what lane 0 of the strip to the right consumes in ITS block
 * b - 2: its row t is lane 63's row at step t + 63.
 */
struct Carries {
	uint32_t l2, h2, l1, h1, l0, h0;
};
struct Feed {
	uint32_t in2, in1, in0;
	uint32_t acc2, acc1, acc0;
};

__device__ __forceinline__ void shift_carries(Carries &C, Feed &F)
{
	asm("s_add_u32 %[i2], %[i2], %[i2]\n\ts_addc_u32 %[l2], %[l2], %[l2]\n\ts_addc_u32 %[h2], %[h2], %[h2]\n\ts_addc_u32 %[a2], %[a2], %[a2]\n\t"
	    "s_add_u32 %[i1], %[i1], %[i1]\n\ts_addc_u32 %[l1], %[l1], %[l1]\n\ts_addc_u32 %[h1], %[h1], %[h1]\n\ts_addc_u32 %[a1], %[a1], %[a1]\n\t"
	    "s_add_u32 %[i0], %[i0], %[i0]\n\ts_addc_u32 %[l0], %[l0], %[l0]\n\ts_addc_u32 %[h0], %[h0], %[h0]\n\ts_addc_u32 %[a0], %[a0], %[a0]"
	    : [i2] "+s"(F.in2), [l2] "+s"(C.l2), [h2] "+s"(C.h2), [a2] "+s"(F.acc2), [i1] "+s"(F.in1), [l1] "+s"(C.l1), [h1] "+s"(C.h1),
	      [a1] "+s"(F.acc1), [i0] "+s"(F.in0), [l0] "+s"(C.l0), [h0] "+s"(C.h0), [a0] "+s"(F.acc0)
	    :
	    : "scc");
}

/* a + b + carry-in of each lane from the mask {hi, lo}; the carry-outs replace the mask */
__device__ __forceinline__ uint32_t add_carry(uint32_t a, uint32_t b, uint32_t &lo, uint32_t &hi)
{
	const uint64_t cin = ((uint64_t)hi << 32) | lo;
	uint64_t cout;
	uint32_t s;
	asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s), "=s"(cout) : "v"(a), "v"(b), "s"(cin));
	lo = (uint32_t)cout;
	hi = (uint32_t)(cout >> 32);
	return s;
}

/* 32 steps.  win0 / win1: the lane's own window of the two row planes, bit t = the letter bit of the row the
 * lane works on at step t of the block (csadp_bits.hip computes it once per block: no letter travels with the
 * carries).  RAMPIN: lanes whose row index is still negative keep an empty row above.
 * OUT_GLOBAL: out = this lane's slot of the strip's direction planes in HBM (step pitch 64).  OUT_TILE
 * (replay): out = the lane's slot of a [32][16] tile in LDS, written by the lanes with `store` set. */
enum : int { OUT_GLOBAL = 0, OUT_NONE = 1, OUT_TILE = 2 };

template <bool RAMPIN, int OUT, bool MATCHES = false>
__device__ __forceinline__ void bits_block(BitState &S, const uint32_t B0, const uint32_t B1, const uint32_t win0, const uint32_t win1,
                                           Carries &C, Feed &F, uint2 *out, bool store, int l0, int lane, uint32_t *outm = nullptr)
{
	constexpr int ostride = (OUT == OUT_TILE) ? 16 : kLanes;
#pragma unroll
	for (int t = 0; t < kBitBlock; ++t) {
		shift_carries(C, F);
		const uint32_t R0 = (uint32_t)__builtin_amdgcn_sbfe((int)win0, t, 1);
		const uint32_t R1 = (uint32_t)__builtin_amdgcn_sbfe((int)win1, t, 1);
		[[maybe_unused]] const uint32_t live = RAMPIN ? ((l0 + t >= lane) ? ~0u : 0u) : ~0u;
		const uint32_t nH0 = S.nH0, H1 = S.H1, H2 = S.H2;
		const uint32_t x0 = B0 ^ R0;
		const uint32_t nE = BITOP3(x0, B1, R1, LA | (LB ^ LC));          /* 1 = mismatch */

		/* vertical step >= 2: generated by a match over w = -1, carried through mismatches over w = -1 */
		const uint32_t g2 = BITOP3(nE, nH0, nH0, ~LA & LB);
		const uint32_t s2 = add_carry(nH0, g2, C.l2, C.h2);
		const uint32_t G2 = BITOP3(s2, nH0, g2, LA ^ LB ^ LC);             /* incoming: u >= 2 */

		/* >= 1: match over w <= 0, or mismatch over w = 0 with u >= 2; carried over w = -1 */
		const uint32_t t1 = BITOP3(nE, nH0, G2, ~LA | (~LB & LC));
		const uint32_t g1 = BITOP3(t1, H1, H1, LA & ~LB);
		const uint32_t A1 = BITOP3(g1, nE, nH0, LA | (LB & LC));
		const uint32_t s1 = add_carry(A1, g1, C.l1, C.h1);
		const uint32_t G1 = BITOP3(s1, A1, g1, LA ^ LB ^ LC);

		/* >= 0: no chain.  match: w <= 1; mismatch: w = -1, or w = 0 and u >= 1, or w = 1 and u >= 2 */
		const uint32_t v = BITOP3(H1, G2, G1, (LA & LB) | (~LA & LC));
		const uint32_t w = BITOP3(nE, v, H2, ~LC & (~LA | LB));
		const uint32_t O0 = BITOP3(w, nE, nH0, LA | (LB & LC));
		const uint32_t G0 = add_carry(O0, O0, C.l0, C.h0);                 /* (O0 << 1) | the bit from the left */

		/* c = H[r][k] - H[r-1][k-1]: C1 = (c = 1), C0 = (c >= 0); new horizontal steps c - u */
		const uint32_t C1 = BITOP3(nE, G2, H2, ~LA | LB | LC);
		const uint32_t C0 = BITOP3(nE, G1, H1, ~LA | LB | LC);
		uint32_t T2 = BITOP3(C1, G0, G0, LA & ~LB);
		const uint32_t a1 = BITOP3(C1, G1, G1, LA & ~LB);
		uint32_t T1 = BITOP3(G0, a1, C0, (LA & LB) | (~LA & LC));
		const uint32_t b0 = BITOP3(C0, G1, G0, LC & (~LA | LB));
		uint32_t nT0 = BITOP3(b0, C1, G2, LA & (~LB | LC));
		if (OUT != OUT_NONE) {
			const uint32_t notdiag = C0 & nE;
			const uint32_t left = notdiag & nT0;
			if (OUT == OUT_GLOBAL || store) {
				out[t * ostride] = make_uint2(notdiag, left);
				if (MATCHES) outm[t * ostride] = ~nE;           /* match mask: the walk scores its path */
			}
		}
		if (RAMPIN) {
			nT0 |= ~live;
			T1 &= live;
			T2 &= live;
		}
		S.nH0 = nT0;
		S.H1 = T1;
		S.H2 = T2;
	}
}

/* the lane's window of a row plane for the block whose lane-0 rows are word `wb`: lane L starts the block at row
 * 32 b - L, i.e. at bit (-L) & 31 of word b - 1 (L = 1..32) or b - 2 (L = 33..63) */
__device__ __forceinline__ uint32_t row_window(uint32_t wb, uint32_t wp, uint32_t wpp, int lane)
{
	const uint32_t lo = lane == 0 ? wb : (lane <= 32 ? wp : wpp);
	const uint32_t hi = lane <= 32 ? wb : wp;
	return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(-lane) & 31u);
}

/* lane state in the checkpoint array: one uint4 per lane and block: 3 planes + the lane's bits of the three
 * carry-out masks (bit 0 ">= 2", bit 1 ">= 1", bit 2 ">= 0") */
__device__ __forceinline__ void save_state(uint4 *ck, size_t idx, const BitState &S, const Carries &C, int lane)
{
	const uint64_t c2 = ((uint64_t)C.h2 << 32) | C.l2, c1 = ((uint64_t)C.h1 << 32) | C.l1, c0 = ((uint64_t)C.h0 << 32) | C.l0;
	const uint32_t bits = (uint32_t)((c2 >> lane) & 1u) | ((uint32_t)((c1 >> lane) & 1u) << 1) | ((uint32_t)((c0 >> lane) & 1u) << 2);
	ck[idx] = make_uint4(S.nH0, S.H1, S.H2, bits);
}

__device__ __forceinline__ void load_state(const uint4 *ck, size_t idx, BitState &S, Carries &C)
{
	const uint4 v = ck[idx];
	S.nH0 = v.x;
	S.H1 = v.y;
	S.H2 = v.z;
	const uint64_t c2 = __ballot((v.w & 1u) != 0), c1 = __ballot((v.w & 2u) != 0), c0 = __ballot((v.w & 4u) != 0);
	C.l2 = (uint32_t)c2;
	C.h2 = (uint32_t)(c2 >> 32);
	C.l1 = (uint32_t)c1;
	C.h1 = (uint32_t)(c1 >> 32);
	C.l0 = (uint32_t)c0;
	C.h0 = (uint32_t)(c0 >> 32);
}

/* Counters in LDS that order LDS data only: the LDS executes one wave's accesses in the order they were issued
 * and is coherent inside the compute unit, so relaxed accesses suffice (a reader that sees the counter sees the
 * ring words stored before it; a ring word read before `taken` is stored was read before anyone can see
 * `taken`).  The round-1 form -- acquire loads, a workgroup-scope release fence before each counter store --
 * also drained the wave's outstanding checkpoint and mark stores (s_waitcnt vmcnt(0)) twice per 32-step block:
 * a round trip to memory on the path between two strips. */
/* TIGHT: poll without sleeping.  Measured both ways per kernel: the one-workgroup-per-job kernel (+3 % of `value`) and the
 * one-wave-per-SIMD launches poll tightly; the chunked launches with 2 or 4 waves per SIMD sleep between polls (config 5:
 * 38.9 vs 40.1 ms per pass) */
template <bool TIGHT = false>
__device__ __forceinline__ bool wait_at_least(const int *counter, int need)
{
	int spins = 0;
	/* the value is the same in every lane; saying so keeps the callers' control flow -- and with it the carry masks -- scalar */
	while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
		if (!TIGHT) __builtin_amdgcn_s_sleep(2);
		if (++spins > kSpinMax) return false;
	}
	return true;
}

/* the row planes are written before the launch (host, or nw_pack_planes in an earlier kernel) and only read here:
 * through the constant address space they become scalar loads, counted apart from the vector memory accesses */
typedef const __attribute__((address_space(4))) uint32_t *ConstWords;

/* the words a strip's lane 63 hands on, per block: ring slot in LDS, and (checkpoint mode) a 32-byte mark in HBM:
 * u32 {">= 2" bits, tag, ">= 1" bits, tag, ">= 0" bits, tag, 0, 0} -- three 8-byte granules {data, tag} */
constexpr int kMarkWords = 8;

__device__ __forceinline__ void feed_from(Feed &F, uint32_t w2, uint32_t w1, uint32_t w0)
{
	F.in2 = __builtin_amdgcn_readfirstlane(w2);
	F.in1 = __builtin_amdgcn_readfirstlane(w1);
	F.in0 = __builtin_amdgcn_readfirstlane(w0);
}

/* the accumulator of plane (2 - lane) in lanes 0..2.  Written with the three words pinned in vector registers first: a
 * plain `lane == 0 ? F.acc2 : ...` is compiled into ONE load through a selected address, which keeps the whole Feed
 * in scratch memory -- and what is loaded from there no longer counts as wave-uniform */
__device__ __forceinline__ uint32_t acc_of_lane(const Feed &F, int lane)
{
	uint32_t x2 = F.acc2, x1 = F.acc1, x0 = F.acc0;
	asm volatile("" : "+v"(x2), "+v"(x1), "+v"(x0));
	return lane == 0 ? x2 : lane == 1 ? x1 : x0;
}

}  // namespace

/*
 * K1b.  One workgroup per job, one wave per strip.  Strip s consumes, for every row, the three carry bits
 * that leave lane 63 of strip s-1 (the producer's lane 63 works on row r at its step r + 63); they travel
 * as three words per block of 32 steps through an LDS ring (`made` = blocks the producer has finished,
 * `taken` = blocks whose words the consumer has fetched, for back-pressure): block b of the consumer needs
 * the words of the producer's block b + 2.  All waves of a workgroup are resident, so the waits always end;
 * each is bounded all the same and a timeout raises *abort_word.
 */
template <bool CKPT>
__global__ __launch_bounds__(kBitMaxStrips *kLanes) void nw_fill_bits(uint8_t *__restrict__ arena,
                                                                      const BitJob *__restrict__ jobs,
                                                                      int *__restrict__ abort_word)
{
	constexpr int OUT = CKPT ? OUT_NONE : OUT_GLOBAL;
	__shared__ __attribute__((aligned(16))) uint4 ring[kBitMaxStrips][kRing];
	__shared__ int made[kBitMaxStrips], taken[kBitMaxStrips];
	const BitJob &J = jobs[blockIdx.x];
	const int s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	if (threadIdx.x < kBitMaxStrips) {
		made[threadIdx.x] = 0;
		taken[threadIdx.x] = 0;
	}
	__syncthreads();
	if (s >= J.nstrips) return;

	const int nb = J.steps_pad / kBitBlock;
	const uint32_t *cp = reinterpret_cast<const uint32_t *>(arena + J.colplanes);
	uint32_t B0 = cp[s * kLanes + lane], B1 = cp[J.nwords_pad + s * kLanes + lane];
	ConstWords rp = (ConstWords)(uintptr_t)(arena + J.rowplanes);
	uint32_t a0n = rp[0], a1n = rp[J.rowwords];                /* requested one block ahead */
	uint32_t a0p = 0, a1p = 0, a0pp = 0, a1pp = 0;             /* the words of the two blocks before */
	uint2 *dirs = reinterpret_cast<uint2 *>(arena + J.dirs) + (size_t)s * J.steps_pad * kLanes + lane;
	const bool feeds = s + 1 < J.nstrips;
	uint32_t *marks = CKPT ? reinterpret_cast<uint32_t *>(arena + J.hand) + (size_t)s * nb * kMarkWords : nullptr;

	BitState S{~0u, 0u, 0u};
	Carries C{0, 0, 0, 0, 0, 0};
	Feed F{0, 0, 0, 0, 0, 0};
	/* the column planes are waited for HERE: left to the compiler the wait sits at their first use inside the block
	 * loop, where it is s_waitcnt vmcnt(0) -- and drains the checkpoint stores of the block before, every block */
	asm volatile("" : "+v"(B0), "+v"(B1));
	/* one block; false = a bounded wait ran out.  Two loops call it (the first two blocks of a strip keep the lanes
	 * above the matrix idle): one loop with a branch on b < 2 merges the scalar carry masks of both forms in phi
	 * nodes the compiler then places in vector registers ("illegal VGPR to SGPR copy") */
	auto block = [&](int b, auto ramp) -> bool {
		constexpr bool RAMP = decltype(ramp)::value;
		/* the bits entering lane 0 during this block left the producer's lane 63 in its steps 32b + 63 .. 32b + 94 */
		uint4 w = make_uint4(0u, 0u, 0u, 0u);
		if (s > 0) {
			const int need = (b + 3 < nb) ? b + 3 : nb;
			if (!wait_at_least<true>(&made[s - 1], need)) return false;
			if (b + 2 < nb) w = ring[s - 1][(b + 2) % kRing];
			if (lane == 0) __hip_atomic_store(&taken[s], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		feed_from(F, w.x, w.y, w.z);
		const uint32_t a0 = a0n, a1 = a1n;
		if (b + 1 < nb) {
			a0n = rp[b + 1];
			a1n = rp[J.rowwords + b + 1];
		}
		const uint32_t win0 = row_window(a0, a0p, a0pp, lane), win1 = row_window(a1, a1p, a1pp, lane);
		a0pp = a0p;
		a0p = a0;
		a1pp = a1p;
		a1p = a1;
		bits_block<RAMP, OUT>(S, B0, B1, win0, win1, C, F, dirs + (size_t)b * kBitBlock * kLanes, true, b * kBitBlock, lane);
		if (feeds) {
			/* ring slot b % kRing last held block b - kRing, which the consumer fetches for its block b - kRing - 2 */
			if (!wait_at_least<true>(&taken[s + 1], b - kRing - 1)) return false;
			if (lane == 0) {
				ring[s][b % kRing] = make_uint4(F.acc2, F.acc1, F.acc0, 0u);
				__hip_atomic_store(&made[s], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		if (CKPT) {
			const uint32_t d = acc_of_lane(F, lane);
			if (feeds && lane < 3) marks[(size_t)b * kMarkWords + 2 * lane] = d;
			save_state(reinterpret_cast<uint4 *>(arena + J.ckpt), ((size_t)s * nb + b) * kLanes + lane, S, C, lane);
		}
		return true;
	};
	bool ok = true;
	for (int b = 0; ok && b < 2 && b < nb; ++b) ok = block(b, std::true_type());
	for (int b = 2; ok && b < nb; ++b) ok = block(b, std::false_type());
	if (!ok && lane == 0) atomicExch(abort_word, 1);
}

/*
 * K1b for jobs that do not run as ONE workgroup (checkpoint mode only): the same step function, one
 * workgroup per CHUNK of WAVES strips, all chunks of a job in flight at once.  Two uses: jobs wider than
 * 16 strips (WAVES = 16), and small batches of large jobs -- a single 16 kbp pair, the first fill of a
 * whole-genome profile alignment, config 5's 200 kbp pairs -- whose strips are spread over compute
 * units at ONE wave per SIMD (WAVES = 4) or two (WAVES = 8) instead of sharing one unit's SIMDs: a step's
 * latency is what bounds such a launch.
 * The first strip of chunk c takes the words that left the last strip of chunk c-1 from that strip's marks in
 * HBM: three 8-byte granules {bits, tag} per block, each written by one write-through store and valid exactly
 * when its tag is this launch's epoch; the consumer requests a block's granules one block ahead and re-reads
 * (bounded) only what had not arrived -- no counter, no fence (MI355X_MICROARCH.md, data-tagged granules).
 * The marks are zeroed when the batch is laid out and epochs are unique per process and never 0.  The work
 * list puts a job's chunks in ascending order, so the chunk a workgroup waits for was dispatched before it --
 * for speed only: every wait is bounded, a time-out raises the abort word and the host repeats the pass chunk
 * by chunk.  The launch reserves enough LDS for ONE workgroup per compute unit.
 */
template <int WAVES>
__global__ __launch_bounds__(WAVES *kLanes) void nw_fill_bits_wide(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs, int njobs,
                                                                   const TileRef *__restrict__ work, uint32_t epoch,
                                                                   int *__restrict__ abort_word)
{
	__shared__ __attribute__((aligned(16))) uint4 ring[WAVES][kRing];
	__shared__ int made[WAVES], taken[WAVES];
	const TileRef item = work[blockIdx.x];                      /* x: the work list of one pass, y: the pass */
	const BitJob &J = jobs[(size_t)blockIdx.y * njobs + item.job];
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	const int chunk = item.a;
	const int s = chunk * WAVES + wv;                          /* this wave's strip */
	if (threadIdx.x < WAVES) {
		made[threadIdx.x] = 0;
		taken[threadIdx.x] = 0;
	}
	__syncthreads();
	if (s >= J.nstrips) return;

	const int nb = J.steps_pad / kBitBlock;
	const uint32_t *cp = reinterpret_cast<const uint32_t *>(arena + J.colplanes);
	ConstWords rp = (ConstWords)(uintptr_t)(arena + J.rowplanes);
	uint32_t a0n = rp[0], a1n = rp[J.rowwords];                /* requested one block ahead */
	uint32_t a0p = 0, a1p = 0, a0pp = 0, a1pp = 0;
	uint32_t B0 = cp[s * kLanes + lane], B1 = cp[J.nwords_pad + s * kLanes + lane];
	const bool feeds = wv + 1 < WAVES && s + 1 < J.nstrips;          /* a wave of this workgroup reads my ring */
	const bool publishes = wv + 1 == WAVES && s + 1 < J.nstrips;     /* the next chunk reads my marks */
	uint32_t *marks = reinterpret_cast<uint32_t *>(arena + J.hand) + (size_t)s * nb * kMarkWords;
	const bool from_left_chunk = wv == 0 && chunk > 0;
	const uint64_t *left_marks = from_left_chunk ? reinterpret_cast<const uint64_t *>(arena + J.hand) + (size_t)(s - 1) * nb * (kMarkWords / 2) : nullptr;

	BitState S{~0u, 0u, 0u};
	Carries C{0, 0, 0, 0, 0, 0};
	Feed F{0, 0, 0, 0, 0, 0};
	asm volatile("" : "+v"(B0), "+v"(B1));
	/* the previous chunk's granules for block 0 (its block 2), requested now; inside the loop always one block ahead */
	const int gl = lane < 3 ? lane : 0;
	uint64_t pre = 0;
	if (from_left_chunk && 2 < nb) pre = __hip_atomic_load(&left_marks[(size_t)2 * (kMarkWords / 2) + gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	auto block = [&](int b, auto ramp) -> bool {
		constexpr bool RAMP = decltype(ramp)::value;
		uint4 w = make_uint4(0u, 0u, 0u, 0u);              /* lanes 0..2 of the first wave of a chunk: .x = the three words */
		if (wv > 0) {
			const int need = (b + 3 < nb) ? b + 3 : nb;
			if (!wait_at_least<WAVES == 4>(&made[wv - 1], need)) return false;
			if (b + 2 < nb) w = ring[wv - 1][(b + 2) % kRing];
			if (lane == 0) __hip_atomic_store(&taken[wv], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else if (from_left_chunk && b + 2 < nb) {
			uint64_t v = pre;
			int spins = 0;
			for (;;) {
				const bool ok = (uint32_t)(v >> 32) == epoch;
				if (__builtin_amdgcn_readfirstlane((int)(__ballot(!ok) == 0ull))) break;       /* uniform, and known to be */
				__builtin_amdgcn_s_sleep(2);
				if (++spins > kSpinMax) return false;
				/* every lane re-reads (a branch on `ok` would make the loop's exit look divergent to the compiler) */
				v = __hip_atomic_load(&left_marks[(size_t)(b + 2) * (kMarkWords / 2) + gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (b + 3 < nb) pre = __hip_atomic_load(&left_marks[(size_t)(b + 3) * (kMarkWords / 2) + gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t d = (uint32_t)v;
			w.x = (uint32_t)__builtin_amdgcn_readlane((int)d, 0);
			w.y = (uint32_t)__builtin_amdgcn_readlane((int)d, 1);
			w.z = (uint32_t)__builtin_amdgcn_readlane((int)d, 2);
		}
		feed_from(F, w.x, w.y, w.z);
		const uint32_t a0 = a0n, a1 = a1n;
		if (b + 1 < nb) {
			a0n = rp[b + 1];
			a1n = rp[J.rowwords + b + 1];
		}
		const uint32_t win0 = row_window(a0, a0p, a0pp, lane), win1 = row_window(a1, a1p, a1pp, lane);
		a0pp = a0p;
		a0p = a0;
		a1pp = a1p;
		a1p = a1;
		bits_block<RAMP, OUT_NONE>(S, B0, B1, win0, win1, C, F, nullptr, false, b * kBitBlock, lane);
		if (feeds) {
			if (!wait_at_least<WAVES == 4>(&taken[wv + 1], b - kRing - 1)) return false;
			if (lane == 0) {
				ring[wv][b % kRing] = make_uint4(F.acc2, F.acc1, F.acc0, 0u);
				__hip_atomic_store(&made[wv], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		}
		const uint32_t d = acc_of_lane(F, lane);
		if (lane < 3) {
			if (publishes) {                                      /* for another compute unit: tagged, written through */
				__hip_atomic_store(reinterpret_cast<uint64_t *>(marks) + (size_t)b * (kMarkWords / 2) + lane, ((uint64_t)epoch << 32) | d,
				                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			} else if (feeds) {
				marks[(size_t)b * kMarkWords + 2 * lane] = d;
			}
		}
		save_state(reinterpret_cast<uint4 *>(arena + J.ckpt), ((size_t)s * nb + b) * kLanes + lane, S, C, lane);
		return true;
	};
	bool ok = true;
	for (int b = 0; ok && b < 2 && b < nb; ++b) ok = block(b, std::true_type());
	for (int b = 2; ok && b < nb; ++b) ok = block(b, std::false_type());
	if (!ok && lane == 0) atomicExch(abort_word, 1);
}

/*
 * K2c.  Traceback in checkpoint mode: no direction planes exist in HBM.  A round starts at the
 * current cell, in block `btop` (32 steps) of strip s, lane L.  Going up its diagonal the path
 * reaches block btop-d around lane L-d, so piece d = block btop-d of the strip is replayed with the
 * fill's own step function: lane state and carry masks from the checkpoint before the block, the bits
 * entering lane 0 from the marks of the strip to the left, and the directions of the 16-lane group
 * holding lane L-d kept in an LDS tile.  One wave per piece, 16 waves = 16 pieces = ~500 path cells per
 * round; wave 0 then walks inside the 16 tiles (run-batched like K2b) until the path leaves them.
 * Replay work: ~(nrows + ncols) / 31 blocks of a strip, 13 % of the fill's work for square matrices
 * (the round-2 form restarted at 16-lane boundaries from hand-off words the fill stored every step:
 * 3 %, paid for by 9 instructions in every step of the fill).
 */
constexpr int kReplay = 16;          /* waves = pieces per round: 64 KB of LDS tiles */
constexpr int kPieces = kReplay;

__device__ __forceinline__ int piece_group(int lane0, int d)
{
	const int l = lane0 - d;
	return (l < 0 ? 0 : l) >> 4;
}

template <bool SCORE>       /* SCORE: also sum the move scores of the path (score-only callers skip the host walk) */
__global__ __launch_bounds__(kReplay *kLanes) void nw_traceback_replay(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs)
{
	__shared__ __attribute__((aligned(16))) uint2 tile[kPieces][kBitBlock * 16];
	__shared__ uint32_t mtile[SCORE ? kPieces : 1][SCORE ? kBitBlock * 16 : 1];    /* match masks of the same cells */
	__shared__ int pos[4];

	const BitJob &J = jobs[blockIdx.x];
	uint8_t *ops = arena + J.ops;
	int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
	const uint32_t *cp = reinterpret_cast<const uint32_t *>(arena + J.colplanes);
	const uint32_t *rp = reinterpret_cast<const uint32_t *>(arena + J.rowplanes);
	const uint4 *ck = reinterpret_cast<const uint4 *>(arena + J.ckpt);
	const uint32_t *marks = reinterpret_cast<const uint32_t *>(arena + J.hand);
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	const int nb = J.steps_pad / kBitBlock;
	int r = J.nrows, k = J.ncols;
	int n = 0;
	int score = 0;                                             /* sum of the move scores along the path (:993-998 for i = 1) */

	while (r > 0 && k > 0) {
		const int w0 = (k - 1) / 32;                         /* lane column of the current cell */
		const int s = __builtin_amdgcn_readfirstlane(w0 >> 6);
		const int lane0 = w0 & 63;
		const int btop = ((r - 1) + lane0) / kBitBlock;
		{
			/* this wave's piece */
			const int d = wv;
			/* wave-uniform by construction; said so, because the carry masks live in scalar registers */
			const int b = __builtin_amdgcn_readfirstlane(btop - d < 0 ? 0 : btop - d);   /* pieces above block 0 replay block 0 and are never read */
			const int g = piece_group(lane0, d);
			BitState S{~0u, 0u, 0u};
			Carries C{0, 0, 0, 0, 0, 0};
			Feed F{0, 0, 0, 0, 0, 0};
			if (b > 0) load_state(ck, ((size_t)s * nb + (b - 1)) * kLanes + lane, S, C);
			const uint32_t B0 = cp[s * kLanes + lane], B1 = cp[J.nwords_pad + s * kLanes + lane];
			if (s > 0 && b + 2 < nb) {
				const uint32_t *m = marks + ((size_t)(s - 1) * nb + (b + 2)) * kMarkWords;
				feed_from(F, m[0], m[2], m[4]);
			}
			/* the lane's rows of this block start at row 32 b - lane */
			const int base = b * kBitBlock - lane;
			const int q = base >> 5;
			const uint32_t sh = (uint32_t)base & 31u;
			const uint32_t lo0 = q >= 0 ? rp[q] : 0u, hi0 = q + 1 >= 0 ? rp[q + 1] : 0u;
			const uint32_t lo1 = q >= 0 ? rp[J.rowwords + q] : 0u, hi1 = q + 1 >= 0 ? rp[J.rowwords + q + 1] : 0u;
			const uint32_t win0 = __builtin_amdgcn_alignbit(hi0, lo0, sh), win1 = __builtin_amdgcn_alignbit(hi1, lo1, sh);
			const bool store = (lane >> 4) == g;
			/* always the ramp form (3 more instructions per step): one call site, see nw_fill_bits */
			bits_block<true, OUT_TILE, SCORE>(S, B0, B1, win0, win1, C, F, tile[d] + (lane & 15), store, b * kBitBlock, lane, mtile[SCORE ? d : 0] + (lane & 15));
		}
		__syncthreads();
		if (wv == 0) {
			for (;;) {
				const int ri = r - lane, ki = k - lane;
				uint32_t code = 3;                             /* 3 = stop: border or outside the replayed pieces */
				bool match = false;
				if (ri > 0 && ki > 0) {
					const int kc = ki - 1;
					const int wi = kc / 32;
					const int sl = wi & 63;
					const int l = (ri - 1) + sl;
					const int d = btop - l / kBitBlock;
					if ((wi >> 6) == s && d >= 0 && d < kPieces && d <= btop && (sl >> 4) == piece_group(lane0, d)) {
						const int at = (l % kBitBlock) * 16 + (sl & 15);
						const uint2 dd = tile[d][at];
						const uint32_t bit = 1u << (kc & 31);
						code = (dd.x & bit) ? ((dd.y & bit) ? (uint32_t)DIR_L : (uint32_t)DIR_U) : (uint32_t)DIR_D;
						if (SCORE) match = (mtile[d][at] & bit) != 0;
					}
				}
				/* a run of 'D' and the gap move that ends it are taken in ONE iteration */
				const unsigned long long stop = __ballot(code != DIR_D);
				const int run = stop ? __builtin_ctzll(stop) : kLanes;
				const uint32_t c0 = run < kLanes ? (uint32_t)__builtin_amdgcn_readlane((int)code, run) : 3u;
				if (lane < run) ops[n + lane] = (uint8_t)DIR_D;
				if (SCORE) {
					const unsigned long long hits = __ballot(match) & (run == kLanes ? ~0ull : ((1ull << run) - 1));
					score += 2 * __builtin_popcountll(hits) - run;          /* +1 per match, -1 per mismatch */
				}
				n += run;
				r -= run;
				k -= run;
				if (c0 == 3) {
					if (run == 0) break;                            /* border, or outside the replayed pieces */
					continue;
				}
				if (lane == 0) ops[n] = (uint8_t)c0;
				++n;
				--score;                                        /* a gap in either sequence */
				if (c0 == DIR_L) --k; else --r;
			}
			if (lane == 0) {
				pos[0] = r;
				pos[1] = k;
				pos[2] = n;
				pos[3] = score;
			}
		}
		__syncthreads();
		r = pos[0];
		k = pos[1];
		n = pos[2];
		score = pos[3];
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		summary[0] = n;
		summary[1] = r;
		summary[2] = k;
		summary[3] = SCORE ? score - r - k : 0;               /* + the border cell the walk stopped on: H[r][0] = -r, H[0][k] = -k */
	}
}

/*
 * K2b.  The walk of dynamicprogramming.c:1037-1047 over the direction planes of nw_fill_bits;
 * same scheme as nw_traceback (csadp_kernels.hip): lane i looks at the i-th cell of the diagonal
 * through the current cell, a ballot finds the end of the run of 'D', one 'L'/'U' is taken from
 * lane 0.  Storage coordinates of cell (r, k), 1-based: word column q = (k-1) >> 5, time
 * tau = (r-1) + q (a strip's local step is tau - 64*(q >> 6)).  Going d cells up a diagonal tau
 * drops by d + d/32 and q by d/32, so the LDS window of WT tau-steps x 8 word columns x 2 planes
 * (64 KiB) is skewed left by one word column every 33 steps, in multiples of 4 columns.
 */
__global__ __launch_bounds__(64) void nw_traceback_bits(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs)
{
	constexpr int WQ = 8;
	constexpr int WT = 16384 / (2 * WQ);
	__shared__ __attribute__((aligned(16))) uint32_t win[WT * WQ * 2];

	const BitJob &J = jobs[blockIdx.x];
	uint8_t *ops = arena + J.ops;
	int32_t *summary = reinterpret_cast<int32_t *>(arena + J.summary);
	const uint32_t *dirs = reinterpret_cast<const uint32_t *>(arena + J.dirs);
	const int lane = threadIdx.x;
	const int qmax = J.nwords_pad;
	int r = J.nrows, k = J.ncols;
	int n = 0;

	while (r > 0 && k > 0) {
		const int q0 = (k - 1) >> 5;
		const int ttop = (r - 1) + q0;
		const int qbase = (q0 & ~3) - 4;
		/* unit u = (window step i, pair of word columns): 4 words = 16 bytes */
		constexpr int UNITS = WT * WQ / 2;
		constexpr int BATCH = 16;
		for (int b0 = 0; b0 < UNITS / kLanes; b0 += BATCH) {
			uint4 v[BATCH];
#pragma unroll
			for (int b = 0; b < BATCH; ++b) {
				const int u = (b0 + b) * kLanes + lane;
				const int i = u / (WQ / 2);
				const int q = qbase - ((i / 33) & ~3) + 2 * (u % (WQ / 2));
				const int tau = ttop - i;
				const int l = tau - ((q >> 6) << 6);           /* local step of the strip holding q, q+1 */
				v[b] = make_uint4(0, 0, 0, 0);
				if (q >= 0 && q < qmax && l >= 0 && l < J.steps_pad)
					v[b] = *reinterpret_cast<const uint4 *>(dirs + (((size_t)(q >> 6) * J.steps_pad + l) * kLanes + (q & 63)) * 2);
			}
#pragma unroll
			for (int b = 0; b < BATCH; ++b) reinterpret_cast<uint4 *>(win)[(b0 + b) * kLanes + lane] = v[b];
		}
		__syncthreads();
		for (;;) {
			const int ri = r - lane, ki = k - lane;
			uint32_t code = 3;                                 /* 3 = stop: border or outside the window */
			if (ri > 0 && ki > 0) {
				const int kc = ki - 1;
				const int q = kc >> 5;
				const int i = ttop - ((ri - 1) + q);
				if (i >= 0 && i < WT) {
					const int j = q - (qbase - ((i / 33) & ~3));
					if (j >= 0 && j < WQ) {
						/* the neighbour word of a pair belongs to row-1 of the NEXT word column: each
						 * (i, j) slot is the word of column q at time tau, i.e. of row tau - q */
						const uint32_t nd = win[(i * WQ + j) * 2];
						const uint32_t lf = win[(i * WQ + j) * 2 + 1];
						const uint32_t bit = 1u << (kc & 31);
						code = (nd & bit) ? ((lf & bit) ? (uint32_t)DIR_L : (uint32_t)DIR_U) : (uint32_t)DIR_D;
					}
				}
			}
			const unsigned long long stop = __ballot(code != DIR_D);
			const int run = stop ? __builtin_ctzll(stop) : kLanes;
			if (run > 0) {
				if (lane < run) ops[n + lane] = (uint8_t)DIR_D;
				n += run;
				r -= run;
				k -= run;
				continue;
			}
			const uint32_t c0 = __builtin_amdgcn_readfirstlane(code);
			if (c0 == 3) break;
			if (lane == 0) ops[n] = (uint8_t)c0;
			++n;
			if (c0 == DIR_L) --k; else --r;
		}
		__syncthreads();
	}
	if (lane == 0) {
		summary[0] = n;
		summary[1] = r;
		summary[2] = k;
		summary[3] = 0;
	}
}

/* static LDS (8 .. 30 KB) + this = more than half of a compute unit's 160 KB: one nw_fill_bits_wide workgroup per unit */
constexpr size_t kWideReserve = 76 * 1024;

/* function attributes are per device: called by Engine::init with that device current */
hipError_t configure_kernels()
{
	hipError_t e = hipFuncSetAttribute((const void *)nw_fill_bits_wide<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideReserve);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nw_fill_bits_wide<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideReserve);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)nw_fill_bits_wide<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWideReserve);
	return e;
}

hipError_t launch_fill_bits(uint8_t *arena, const BitJob *jobs, int njobs, int maxstrips, bool checkpoints, int *abort_word,
                            hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	if (maxstrips < 1 || maxstrips > kBitMaxStrips) return hipErrorInvalidValue;
	if (checkpoints) hipLaunchKernelGGL(nw_fill_bits<true>, dim3(njobs), dim3(maxstrips * kLanes), 0, st, arena, jobs, abort_word);
	else hipLaunchKernelGGL(nw_fill_bits<false>, dim3(njobs), dim3(maxstrips * kLanes), 0, st, arena, jobs, abort_word);
	return hipGetLastError();
}

hipError_t launch_fill_bits_wide(int waves, uint8_t *arena, const BitJob *jobs, int njobs, int passes, const TileRef *work, int nwork,
                                 uint32_t epoch, int *abort_word, hipStream_t st)
{
	if (njobs <= 0 || nwork <= 0 || passes <= 0) return hipSuccess;
	epoch &= 0x1fffffu;                                /* 21 bits travel in a mark word */
	const dim3 grid(nwork, passes);
	if (waves == 4) hipLaunchKernelGGL(nw_fill_bits_wide<4>, grid, dim3(4 * kLanes), kWideReserve, st, arena, jobs, njobs, work, epoch, abort_word);
	else if (waves == 8) hipLaunchKernelGGL(nw_fill_bits_wide<8>, grid, dim3(8 * kLanes), kWideReserve, st, arena, jobs, njobs, work, epoch, abort_word);
	else if (waves == 16) hipLaunchKernelGGL(nw_fill_bits_wide<16>, grid, dim3(16 * kLanes), kWideReserve, st, arena, jobs, njobs, work, epoch, abort_word);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

hipError_t launch_traceback_bits(uint8_t *arena, const BitJob *jobs, int njobs, bool checkpoints, bool scores, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	if (checkpoints && scores) hipLaunchKernelGGL(nw_traceback_replay<true>, dim3(njobs), dim3(kReplay * kLanes), 0, st, arena, jobs);
	else if (checkpoints) hipLaunchKernelGGL(nw_traceback_replay<false>, dim3(njobs), dim3(kReplay * kLanes), 0, st, arena, jobs);
	else hipLaunchKernelGGL(nw_traceback_bits, dim3(njobs), dim3(kLanes), 0, st, arena, jobs);
	return hipGetLastError();
}

}  // namespace csadp
