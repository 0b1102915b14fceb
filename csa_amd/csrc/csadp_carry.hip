/*
 * csadp_carry.hip -- the bit-parallel pairwise fill (dynamicprogramming.c:990-1029 for i = 1, fresh borders) with the
 * carries between lanes kept in SCALAR lane masks, and its checkpoint-replay traceback (dynamicprogramming.c:1037-1047).
 * gfx950, wave64.  SELECTABLE, not the default: CSADP_BITS_CARRY=1 routes the pipelined launches of checkpoint mode that
 * put two or more waves on every SIMD (one workgroup per job) through these kernels; everything else -- and, by default,
 * those launches too -- runs csadp_bits.hip's vector hand-off form.  Measured: the fill is 10 % (two
 * waves per SIMD) to 23 % (four) faster, the traceback pays it back (a replay restarts only at strip boundaries: 64 lanes per
 * piece instead of 16); sustained +5 % with four fill launches in flight, -4..+2 % over 20 passes, the one-pass-per-batch
 * streaming leg -25 %; a lone wave pays this form's VALU -> SGPR -> SALU -> VALU round trips in full.
 *
 *   nw_fill_carry        K1b': one workgroup per matrix, one wave per strip of 2048 columns
 *   nw_traceback_carry   K2c': replays whole strip blocks on the path (16 waves = 16 pieces per round), then walks them
 *
 * The recurrence, the three thermometer planes and the proof of exactness are csadp_bits.hip's.  What differs is how a
 * lane learns the vertical step at the right edge of its left neighbour's word -- see Carries below.
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
 * per wave, moved one lane to the right by the SCALAR unit between two steps.  csadp_bits.hip carries
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
 * lane works on at step t of the block (computed once per block, row_window below: no letter travels with the
 * carries).  RAMPIN: lanes whose row index is still negative keep an empty row above.
 * OUT_NONE: the fill (checkpoint mode stores no directions).  OUT_TILE (replay): out = the lane's slot of a [32][16]
 * tile in LDS, written by the lanes of `storemask`. */
enum : int { OUT_GLOBAL = 0, OUT_NONE = 1, OUT_TILE = 2 };

template <bool RAMPIN, int OUT, bool MATCHES = false>
__device__ __forceinline__ void bits_block(BitState &S, const uint32_t B0, const uint32_t B1, const uint32_t win0, const uint32_t win1,
                                           Carries &C, Feed &F, [[maybe_unused]] uint2 *out, [[maybe_unused]] uint64_t storemask, int l0, int lane,
                                           [[maybe_unused]] uint32_t *outm = nullptr)
{
	constexpr int ostride = (OUT == OUT_TILE) ? 16 : kLanes;
	/* the row letter masks run one step ahead: two of the next step's preparatory instructions fill the wait state between a
	 * v_addc_co and the first use of its sum (the compiler puts an s_nop there otherwise) */
	uint32_t R0 = (uint32_t)__builtin_amdgcn_sbfe((int)win0, 0, 1);
	uint32_t R1 = (uint32_t)__builtin_amdgcn_sbfe((int)win1, 0, 1);
#pragma unroll
	for (int t = 0; t < kBitBlock; ++t) {
		shift_carries(C, F);
		[[maybe_unused]] const uint32_t live = RAMPIN ? ((l0 + t >= lane) ? ~0u : 0u) : ~0u;
		const uint32_t nH0 = S.nH0, H1 = S.H1, H2 = S.H2;
		/* v_xor as an 8-byte instruction: with the twelve 4-byte scalar instructions of the step the number of 4-byte instructions
		 * per step is even, so every step's vector instructions sit in the same phase of the 8-byte grid (tools/code_phase.py) */
		uint32_t x0;
		asm("v_xor_b32_e64 %0, %1, %2" : "=v"(x0) : "v"(B0), "v"(R0));
		const uint32_t nE = BITOP3(x0, B1, R1, LA | (LB ^ LC));          /* 1 = mismatch */

		/* vertical step >= 2: generated by a match over w = -1, carried through mismatches over w = -1 */
		const uint32_t g2 = BITOP3(nE, nH0, nH0, ~LA & LB);
		const uint32_t s2 = add_carry(nH0, g2, C.l2, C.h2);
		if (t + 1 < kBitBlock) R0 = (uint32_t)__builtin_amdgcn_sbfe((int)win0, t + 1, 1);
		const uint32_t G2 = BITOP3(s2, nH0, g2, LA ^ LB ^ LC);             /* incoming: u >= 2 */

		/* >= 1: match over w <= 0, or mismatch over w = 0 with u >= 2; carried over w = -1 */
		const uint32_t t1 = BITOP3(nE, nH0, G2, ~LA | (~LB & LC));
		const uint32_t g1 = BITOP3(t1, H1, H1, LA & ~LB);
		const uint32_t A1 = BITOP3(g1, nE, nH0, LA | (LB & LC));
		const uint32_t s1 = add_carry(A1, g1, C.l1, C.h1);
		if (t + 1 < kBitBlock) R1 = (uint32_t)__builtin_amdgcn_sbfe((int)win1, t + 1, 1);
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
			if (OUT == OUT_TILE) {
				/* replay: 16 of the 64 lanes keep their directions.  EXEC is set and restored around the store by hand (all
				 * lanes are active here): the compiler's form is a saved EXEC, a branch and a restore per step */
				const uint2 dd = make_uint2(notdiag, left);
				asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"
				             :
				             : "v"((uint32_t)(uintptr_t)out), "v"(dd), "s"(storemask), "n"(t * ostride * 8)
				             : "memory");
				if (MATCHES) {
					asm volatile("s_mov_b64 exec, %2\n\tds_write_b32 %0, %1 offset:%3\n\ts_mov_b64 exec, -1"
					             :
					             : "v"((uint32_t)(uintptr_t)outm), "v"(~nE), "s"(storemask), "n"(t * ostride * 4)
					             : "memory");
				}
			} else {
				out[t * ostride] = make_uint2(notdiag, left);
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
 * K1b'.  One workgroup per job, one wave per strip.  Strip s consumes, for every row, the three carry bits that leave lane 63
 * of strip s-1 (the producer's lane 63 works on row r at its step r + 63); they travel as three words per block of 32 steps
 * through an LDS ring (`made` = blocks the producer has finished, `taken` = blocks whose words the consumer has fetched, for
 * back-pressure): block b of the consumer needs the words of the producer's block b + 2.  Per block a strip stores its lane
 * state with the lanes' bits of the three carry masks (`ckpt`: one uint4 per lane) and, if it feeds a strip, the three words
 * that left it (`hand`: u32 [nstrips][blocks][8], words 0 / 2 / 4) for the replay.  All waves of a workgroup are resident, so
 * the waits always end; each is bounded all the same and a timeout raises *abort_word.
 */
__global__ __launch_bounds__(kBitMaxStrips *kLanes) void nw_fill_carry(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs,
                                                                        int *__restrict__ abort_word)
{
	constexpr int OUT = OUT_NONE;
	constexpr bool CKPT = true;
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
			const int need = (b + 3 < nb) ? b + 3 : nb;               /* the words of the producer's block b + 2 */
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
		bits_block<RAMP, OUT>(S, B0, B1, win0, win1, C, F, nullptr, 0, b * kBitBlock, lane);
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
	/* the steady-state block in the faster phase of the 8-byte grid (tools/code_phase.py) */
	asm volatile("s_nop 0");
	for (int b = 2; ok && b < nb; ++b) ok = block(b, std::false_type());
	if (!ok && lane == 0) atomicExch(abort_word, 1);
}

/*
 * K2c'.  Traceback in checkpoint mode.  A round starts at the current cell, in block `btop` (32 steps) of strip s, lane L.
 * Piece p of a round is one whole block of the strip -- lane state and carry masks from the checkpoint before it, the bits
 * entering lane 0 from the marks of the strip to the left, the row letters from the row planes -- replayed by one wave with the
 * fill's own step function; of its 64 lanes the 16-lane group the path is expected in keeps its directions in an LDS tile.
 * The fill stores nothing that would let a replay start in the middle of a strip (that is what its step saves: the vector
 * form computes the outgoing planes' top bits in every lane and stores them per step), so a piece costs 64 lanes where
 * csadp_bits.hip's costs 16: ~(nrows + ncols) / 30 blocks = 16 % of the fill's work for square matrices.
 * Rounds are planned along the diagonal through the current cell as in csadp_bits.hip: the block in which the path changes
 * its 16-lane group is replayed twice, once for each group.  16 waves = 16 pieces = ~480 path cells per round; wave 0 then
 * walks inside the 16 tiles until the path leaves them.
 */
constexpr int kReplayWaves = 16;     /* waves = pieces per round: 64 KB of LDS tiles */

struct RoundPlan {
	int ghi;              /* 16-lane group of the current cell */
	int dc;               /* blocks below btop at which the diagonal enters group ghi - 1 (huge: not in this strip / matrix) */
};

__device__ __forceinline__ RoundPlan plan_round(int r, int k, int lane0, int btop)
{
	RoundPlan P;
	P.ghi = lane0 >> 4;
	P.dc = 1 << 20;
	const int cx = ((k - 1) % (16 * 32)) + 1;              /* cells up the diagonal to the first cell of the group below */
	const int r2 = r - cx, k2 = k - cx;
	if (P.ghi > 0 && r2 > 0 && k2 > 0) {
		const int lane2 = ((k2 - 1) / 32) & 63;
		P.dc = btop - ((r2 - 1) + lane2) / kBitBlock;
	}
	return P;
}

template <bool SCORE>       /* SCORE: also sum the move scores of the path (score-only callers skip the host walk) */
__global__ __launch_bounds__(kReplayWaves *kLanes) void nw_traceback_carry(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs)
{
	constexpr int kPieces = kReplayWaves;
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
		const RoundPlan P = plan_round(r, k, lane0, btop);
		{
			/* this wave's piece.  Wave-uniform by construction; said so, because the carry masks live in scalar registers */
			const int d = wv;
			const bool hi = d <= P.dc;
			const int delta = hi ? d : d - 1;
			const int g = __builtin_amdgcn_readfirstlane(hi ? P.ghi : P.ghi - 1);
			const int b = __builtin_amdgcn_readfirstlane(btop - delta < 0 ? 0 : btop - delta);   /* pieces above block 0 replay block 0 and are never read */
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
			const uint64_t store = 0xffffull << (16 * g);
			/* two call sites are fine HERE: nothing of the scalar state is used after the block (see nw_fill_carry) */
			if (b < 2) bits_block<true, OUT_TILE, SCORE>(S, B0, B1, win0, win1, C, F, tile[d] + (lane & 15), store, b * kBitBlock, lane, mtile[SCORE ? d : 0] + (lane & 15));
			else bits_block<false, OUT_TILE, SCORE>(S, B0, B1, win0, win1, C, F, tile[d] + (lane & 15), store, b * kBitBlock, lane, mtile[SCORE ? d : 0] + (lane & 15));
			/* the tile stores are inline assembly: the compiler's wait counts do not know them */
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
					const int delta = btop - l / kBitBlock;        /* <= btop: l >= 0 */
					const int grp = sl >> 4;
					const bool second = delta > P.dc || (delta == P.dc && grp != P.ghi);
					const int d = delta + (second ? 1 : 0);
					if ((wi >> 6) == s && delta >= 0 && d < kPieces && grp == (second ? P.ghi - 1 : P.ghi)) {
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

hipError_t launch_fill_carry(uint8_t *arena, const BitJob *jobs, int njobs, int maxstrips, int *abort_word, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	if (maxstrips < 1 || maxstrips > kBitMaxStrips) return hipErrorInvalidValue;
	hipLaunchKernelGGL(nw_fill_carry, dim3(njobs), dim3(maxstrips * kLanes), 0, st, arena, jobs, abort_word);
	return hipGetLastError();
}

hipError_t launch_traceback_carry(uint8_t *arena, const BitJob *jobs, int njobs, bool scores, hipStream_t st)
{
	if (njobs <= 0) return hipSuccess;
	if (scores) hipLaunchKernelGGL(nw_traceback_carry<true>, dim3(njobs), dim3(kReplayWaves * kLanes), 0, st, arena, jobs);
	else hipLaunchKernelGGL(nw_traceback_carry<false>, dim3(njobs), dim3(kReplayWaves * kLanes), 0, st, arena, jobs);
	return hipGetLastError();
}

}  // namespace csadp
