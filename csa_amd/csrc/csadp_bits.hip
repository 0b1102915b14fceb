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

#include "csadp_device.h"
#include "csadp_kernels.h"

namespace csadp {

namespace {

constexpr int kRing = 8;                 /* hand-off blocks (of 32 steps) buffered per strip boundary */
constexpr int kRingSteps = kRing * kBitBlock;
constexpr int kSpinMax = 1 << 22;        /* bound of every wait (~0.5 s) */

/* v_bitop3_b32: any boolean function of three words in one instruction; the table is the function
 * applied to these three constants */
constexpr uint32_t LA = 0xF0, LB = 0xCC, LC = 0xAA;
#define BITOP3(a, b, c, expr) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), (unsigned char)((expr) & 0xff)))

/* value of lane-1: over the whole wave (lane 0 keeps `old`), or inside each row of 16 lanes (the
 * first lane of every row keeps `old`) */
template <bool ROWS>
__device__ __forceinline__ uint32_t from_left(uint32_t old, uint32_t src)
{
	/* written as the instruction itself so that the register holding `old` IS the destination
	 * (the builtin costs a v_mov plus two wait states per step) */
	if (ROWS) asm("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
	else asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
	return old;
}

/*
 * The word a lane hands to its right neighbour after every step (one DPP move per step):
 *   bit 31 / 23 / 15  top bit of the outgoing vertical-step planes ">= 0" / ">= 1" / ">= 2"
 *   bits 1..0          letter of the row both lanes are working on (the right lane is one step behind)
 */
template <int W>            /* W words of 32 columns per lane */
struct BitState {
	uint32_t nH0[W], H1[W], H2[W];   /* horizontal steps of the row above: NOT ">= 0", ">= 1", ">= 2" */
	uint32_t PP;                     /* hand-off word of the previous step */
};

/* words per lane in checkpoint mode (csadp_device.h); always 1 with direction planes in HBM */
constexpr int kCkptWords = kBitCkptWords;

/* 32 steps.  inject[t] is the hand-off word entering lane 0 at step t; ringout (lane 63 of a strip
 * that has a right neighbour) receives the word leaving the strip.  RAMPIN: lanes whose row index
 * is still negative keep an empty row above.
 * FEEDS: every lane stores its hand-off word of step t to lanebuf[t] (LDS; a per-lane pointer
 * that is the ring / the marks buffer for the lanes that matter and a shared scrap row for the
 * rest -- no EXEC juggling in the step).
 * OUT_GLOBAL: dirs = this strip's direction planes in HBM.  OUT_TILE (replay): the wave is four
 * independent 16-lane pieces of strips (DPP stays inside a row, `inject` is per lane and only
 * the first lane of a row uses it, `lane` is the lane's index in its strip) and dirs = the
 * row's [32][16] tile. */
enum : int { OUT_GLOBAL = 0, OUT_NONE = 1, OUT_TILE = 2 };

template <bool RAMPIN, bool FEEDS, int OUT, int W, bool MATCHES = false, bool AHEAD = false>
__device__ __forceinline__ void bits_block(BitState<W> &S, const uint32_t (&B0)[W], const uint32_t (&B1)[W], const uint32_t *inject,
                                           uint32_t *lanebuf, uint2 *dirs, int l0, int lane, uint32_t *matches = nullptr)
{
	/* MATCHES (replay with scoring): `matches` is an LDS tile like dirs and receives the match masks.
	 * It must stay an LDS-typed pointer: merged with nullptr it would become a flat pointer, and
	 * flat accesses do not reach LDS beyond 64 KB. */
	[[maybe_unused]] uint32_t *outm = MATCHES ? matches + (lane & 15) * W : nullptr;
	static_assert(OUT != OUT_GLOBAL || W == 1, "direction planes in HBM are laid out for one word per lane");
	constexpr bool ROWS = (OUT == OUT_TILE);
	constexpr int ostride = ROWS ? 16 : kLanes;
	uint2 *out = (OUT == OUT_GLOBAL) ? dirs + (size_t)l0 * kLanes + lane : dirs + (lane & 15) * W;
	/* The words entering the row's first lane come from LDS (address kept in a VGPR: broadcast reads), each
	 * into the register the DPP move of its step then completes.  AHEAD = false: fetched one step ahead -- with
	 * several waves per SIMD the round trip disappears behind the other waves.  AHEAD = true (the launches that
	 * run ONE wave per SIMD): all 32 up front; a lone wave otherwise waits out an LDS round trip in every step,
	 * 235 cycles per step where its 31 VALU instructions take 130 (tools/cellstep_microbench.hip).  Up-front
	 * reads in the many-wave kernel cost 12 % of its throughput (measured), hence the switch. */
	/* One wave per SIMD (AHEAD): LDS instructions cost a lone wave 16-27 cycles of its issue time each, so both directions
	 * move four steps at a time -- 8 ds_read_b128 up front, 8 ds_write_b128 -- instead of 32 + 32 (a 16 kbp pair 1.49 -> 1.44 ms,
	 * a 200 kbp pair 17.9 -> 17.4).  With several waves per SIMD the same batching LOSES 2 % sustained and 10 % at four
	 * passes per launch (measured), although the bare step says otherwise (tools/carrystep_probe.hip: 113 cycles at four
	 * waves per SIMD, 144 with a write and a read per step, 124 with one of each per four steps): word by word there. */
	uint32_t ioff = 0;
	asm volatile("" : "+v"(ioff));
	const uint32_t *ip = reinterpret_cast<const uint32_t *>(__builtin_assume_aligned(inject, 16)) + ioff;
	const uint4 *ip4 = reinterpret_cast<const uint4 *>(inject) + ioff;
	uint4 inj4[AHEAD ? kBitBlock / 4 : 1];
	uint32_t cur = AHEAD ? 0u : ip[0];
	if (AHEAD) {
#pragma unroll
		for (int j = 0; j < kBitBlock / 4; ++j) inj4[AHEAD ? j : 0] = ip4[j];
	}
	[[maybe_unused]] uint32_t pp[4];
#pragma unroll
	for (int t = 0; t < kBitBlock; ++t) {
		uint32_t in;
		if (AHEAD) {
			const uint4 g4 = inj4[AHEAD ? t / 4 : 0];
			in = from_left<ROWS>((t & 3) == 0 ? g4.x : (t & 3) == 1 ? g4.y : (t & 3) == 2 ? g4.z : g4.w, S.PP);
		} else {
			const uint32_t nxt = ip[t + 1 < kBitBlock ? t + 1 : t];
			in = from_left<ROWS>(cur, S.PP);
			cur = nxt;
		}
		const uint32_t R0 = (uint32_t)__builtin_amdgcn_sbfe((int)in, 0, 1);
		const uint32_t R1 = (uint32_t)__builtin_amdgcn_sbfe((int)in, 1, 1);
		uint32_t c2 = __builtin_amdgcn_ubfe(in, 15, 1);
		uint32_t c1 = __builtin_amdgcn_ubfe(in, 23, 1);
		uint32_t below = in;                                  /* its bit 31 enters the ">= 0" plane */
		uint32_t O0 = 0, O1 = 0, O2 = 0;
		[[maybe_unused]] const uint32_t live = RAMPIN ? ((l0 + t >= lane) ? ~0u : 0u) : ~0u;
#pragma unroll
		for (int h = 0; h < W; ++h) {
			const uint32_t nH0 = S.nH0[h], H1 = S.H1[h], H2 = S.H2[h];
			const uint32_t x0 = B0[h] ^ R0;
			const uint32_t nE = BITOP3(x0, B1[h], R1, LA | (LB ^ LC));          /* 1 = mismatch */

			/* vertical step >= 2: generated by a match over w = -1, carried through mismatches over w = -1 */
			const uint32_t g2 = BITOP3(nE, nH0, nH0, ~LA & LB);
			const uint32_t s2 = nH0 + g2 + c2;
			const uint32_t G2 = BITOP3(s2, nH0, g2, LA ^ LB ^ LC);             /* incoming: u >= 2 */
			O2 = BITOP3(g2, nH0, G2, LA | (LB & LC));                         /* outgoing */

			/* >= 1: match over w <= 0, or mismatch over w = 0 with u >= 2; carried over w = -1 */
			const uint32_t t1 = BITOP3(nE, nH0, G2, ~LA | (~LB & LC));
			const uint32_t g1 = BITOP3(t1, H1, H1, LA & ~LB);
			const uint32_t A1 = BITOP3(g1, nE, nH0, LA | (LB & LC));
			const uint32_t s1 = A1 + g1 + c1;
			const uint32_t G1 = BITOP3(s1, A1, g1, LA ^ LB ^ LC);
			O1 = BITOP3(g1, A1, G1, LA | (LB & LC));

			/* >= 0: no chain.  match: w <= 1; mismatch: w = -1, or w = 0 and u >= 1, or w = 1 and u >= 2 */
			const uint32_t v = BITOP3(H1, G2, G1, (LA & LB) | (~LA & LC));
			const uint32_t w = BITOP3(nE, v, H2, ~LC & (~LA | LB));
			O0 = BITOP3(w, nE, nH0, LA | (LB & LC));
			const uint32_t G0 = __builtin_amdgcn_alignbit(O0, below, 31);        /* (O0 << 1) | carry */
			if (h == W - 1) {
				/* hand-off word for the right neighbour (top bits of the last word's outgoing planes),
				 * early: the rest of the step lies between this write and the next step's DPP read */
				const uint32_t q = __builtin_amdgcn_perm(O1, O2, 0x0c07030cu);      /* byte 2 <- O1 byte 3, byte 1 <- O2 byte 3 */
				const uint32_t pq = __builtin_amdgcn_perm(O0, q, 0x0702010cu);      /* byte 3 <- O0 byte 3 */
				S.PP = BITOP3(pq, in, 0xffu, LA | (LB & LC));
				if (FEEDS && AHEAD) {                               /* LDS: lane 63 -> ring, lanes 15/31/47 -> marks, all others -> a scrap slot */
					pp[t & 3] = S.PP;
					if ((t & 3) == 3) *reinterpret_cast<uint4 *>(lanebuf + t - 3) = make_uint4(pp[0], pp[1], pp[2], pp[3]);
				} else if (FEEDS) {
					lanebuf[t] = S.PP;
				}
			}

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
				out[t * ostride * W + h] = make_uint2(notdiag, left);
				if (MATCHES) outm[t * ostride * W + h] = ~nE;       /* match mask: the walk scores its path */
			}
			if (RAMPIN) {
				nT0 |= ~live;
				T1 &= live;
				T2 &= live;
			}
			S.nH0[h] = nT0;
			S.H1[h] = T1;
			S.H2[h] = T2;
			if (h + 1 < W) {                                       /* carries into the lane's next word */
				c2 = O2 >> 31;
				c1 = O1 >> 31;
				below = O0;
			}
		}
	}
}

/* lane state in the checkpoint array: W = 1: one uint4 (3 planes + hand-off word); W = 2: two */
template <int W>
__device__ __forceinline__ void save_state(uint4 *ck, size_t idx, const BitState<W> &S)
{
	if (W == 1) {
		ck[idx] = make_uint4(S.nH0[0], S.H1[0], S.H2[0], S.PP);
	} else {
		ck[idx * 2] = make_uint4(S.nH0[0], S.H1[0], S.H2[0], S.PP);
		ck[idx * 2 + 1] = make_uint4(S.nH0[W - 1], S.H1[W - 1], S.H2[W - 1], 0u);
	}
}

template <int W>
__device__ __forceinline__ void load_state(const uint4 *ck, size_t idx, BitState<W> &S)
{
	const uint4 v = ck[idx * W];
	S.nH0[0] = v.x;
	S.H1[0] = v.y;
	S.H2[0] = v.z;
	S.PP = v.w;
	if (W > 1) {
		const uint4 u = ck[idx * W + 1];
		S.nH0[W - 1] = u.x;
		S.H1[W - 1] = u.y;
		S.H2[W - 1] = u.z;
	}
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
	while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
		if (!TIGHT) __builtin_amdgcn_s_sleep(2);
		if (++spins > kSpinMax) return false;
	}
	return true;
}

/* the row planes are written before the launch (host, or nw_pack_planes in an earlier kernel) and only read here:
 * through the constant address space they become scalar loads, counted apart from the vector memory accesses */
typedef const __attribute__((address_space(4))) uint32_t *ConstWords;

}  // namespace

/*
 * K1b.  One workgroup per job, one wave per strip.  Strip s consumes, for every row, the hand-off
 * word that leaves lane 63 of strip s-1 (the producer's lane 63 works on row r at its step r + 63);
 * the words travel through an LDS ring of kRingSteps steps, synchronised per block of 32 steps
 * (`made` = blocks the producer has finished, `taken` = blocks whose inputs the consumer has
 * fetched, for back-pressure).  All waves of a workgroup are resident, so the waits always end;
 * each is bounded all the same and a timeout raises *abort_word.
 */
template <bool CKPT>
__global__ __launch_bounds__(kBitMaxStrips *kLanes) void nw_fill_bits(uint8_t *__restrict__ arena,
                                                                      const BitJob *__restrict__ jobs,
                                                                      int *__restrict__ abort_word)
{
	constexpr int OUT = CKPT ? OUT_NONE : OUT_GLOBAL;
	constexpr int W = CKPT ? kCkptWords : 1;
	__shared__ __attribute__((aligned(16))) uint32_t ring[kBitMaxStrips][kRingSteps];
	__shared__ __attribute__((aligned(16))) uint32_t inject[kBitMaxStrips][kBitBlock];
	__shared__ __attribute__((aligned(16))) uint32_t mbuf[kBitMaxStrips][3][kBitBlock];
	__shared__ uint32_t scrap[kBitMaxStrips][kLanes + kBitBlock];   /* lane l, step t -> word l + t: 64 different banks per store */
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
	uint32_t B0[W], B1[W];
#pragma unroll
	for (int h = 0; h < W; ++h) {
		B0[h] = cp[(s * kLanes + lane) * W + h];
		B1[h] = cp[J.nwords_pad + (s * kLanes + lane) * W + h];
	}
	ConstWords rp = (ConstWords)(uintptr_t)(arena + J.rowplanes);
	uint32_t a0n = rp[0], a1n = rp[J.rowwords];                /* requested one block ahead */
	uint2 *dirs = reinterpret_cast<uint2 *>(arena + J.dirs) + (size_t)s * J.steps_pad * kLanes;   /* wave-uniform */
	const bool feeds = s + 1 < J.nstrips;
	/* checkpoint mode: the words leaving lanes 15, 31, 47 (and 63: the ring) are kept per block in
	 * LDS and copied out once per block as four streams [4][steps_pad] per strip */
	uint32_t *marks = CKPT ? reinterpret_cast<uint32_t *>(arena + J.hand) + (size_t)s * 4 * J.steps_pad : nullptr;
	const bool writes = (lane == kLanes - 1) ? feeds : (CKPT && (lane & 15) == 15);

	BitState<W> S;
#pragma unroll
	for (int h = 0; h < W; ++h) {
		S.nH0[h] = ~0u;
		S.H1[h] = S.H2[h] = 0;
		/* the column planes are waited for HERE: left to the compiler the wait sits at their first use inside the block
		 * loop, where it is s_waitcnt vmcnt(0) -- and drains the checkpoint stores of the block before, every block */
		asm volatile("" : "+v"(B0[h]), "+v"(B1[h]));
	}
	S.PP = 0;
	for (int b = 0; b < nb; ++b) {
		/* hand-off words entering lane 0 during this block: lane t prepares step t.  Carries from
		 * the producer's step 32b + t + 63, row letter of row 32b + t */
		uint32_t word = 0;
		if (s > 0) {
			const int need = (b + 3 < nb) ? b + 3 : nb;         /* producer steps up to 32b + 94 */
			if (!wait_at_least<true>(&made[s - 1], need)) { if (lane == 0) atomicExch(abort_word, 1); return; }
			const int ps = b * kBitBlock + 63 + (lane & 31);
			if (ps < J.steps_pad) word = ring[s - 1][ps % kRingSteps] & 0xffffff00u;
			if (lane == 0) __hip_atomic_store(&taken[s], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		const uint32_t a0 = a0n, a1 = a1n;
		if (b + 1 < nb) {
			a0n = rp[b + 1];
			a1n = rp[J.rowwords + b + 1];
		}
		word |= ((a0 >> (lane & 31)) & 1u) | (((a1 >> (lane & 31)) & 1u) << 1);
		if (lane < kBitBlock) inject[s][lane] = word;
		uint32_t *lanebuf = !writes ? &scrap[s][lane] : (lane == kLanes - 1) ? &ring[s][(b * kBitBlock) % kRingSteps] : &mbuf[s][lane >> 4][0];
		if (feeds) {
			/* the ring slots of this block last held block b - kRing, whose words the consumer
			 * fetches while preparing its blocks b - kRing - 2 and b - kRing - 1 */
			if (!wait_at_least<true>(&taken[s + 1], b - kRing)) { if (lane == 0) atomicExch(abort_word, 1); return; }
		}
		if (feeds || CKPT) {
			if (b < 2) bits_block<true, true, OUT, W>(S, B0, B1, inject[s], lanebuf, dirs, b * kBitBlock, lane);
			else bits_block<false, true, OUT, W>(S, B0, B1, inject[s], lanebuf, dirs, b * kBitBlock, lane);
		} else {
			if (b < 2) bits_block<true, false, OUT, W>(S, B0, B1, inject[s], nullptr, dirs, b * kBitBlock, lane);
			else bits_block<false, false, OUT, W>(S, B0, B1, inject[s], nullptr, dirs, b * kBitBlock, lane);
		}
		if (CKPT) {
			/* streams 0..2: lanes 15/31/47 from mbuf, stream 3: lane 63 from the ring (strips that feed) */
			const int g = lane >> 4 >> 1, t = lane & 31;          /* lanes 0..31 -> stream 0, 32..63 -> stream 1 */
			marks[(size_t)g * J.steps_pad + b * kBitBlock + t] = mbuf[s][g][t];
			marks[(size_t)(g + 2) * J.steps_pad + b * kBitBlock + t] =
			    (g == 0) ? mbuf[s][2][t] : (feeds ? ring[s][(b * kBitBlock + t) % kRingSteps] : 0u);
			save_state<W>(reinterpret_cast<uint4 *>(arena + J.ckpt), ((size_t)s * nb + b) * kLanes + lane, S);
		}
		if (feeds && lane == kLanes - 1) __hip_atomic_store(&made[s], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

/*
 * K1b for jobs that do not run as ONE workgroup (checkpoint mode only): the same step function, one
 * workgroup per CHUNK of WAVES strips, all chunks of a job in flight at once.  Two uses: jobs wider than
 * 16 strips (WAVES = 16), and small batches of large jobs -- a single 16 kbp pair, the first fill of a
 * whole-genome profile alignment, config 5's 200 kbp pairs -- whose strips are spread over compute
 * units at ONE wave per SIMD (WAVES = 4) or two (WAVES = 8) instead of sharing one unit's SIMDs: a step's
 * latency is what bounds such a launch.
 * The first strip of chunk c takes the hand-off words that left the last strip of chunk c-1 from stream 3 of
 * that strip's marks in HBM.  Only bits 31 / 23 / 15 of those words are ever used (the outgoing planes'
 * top bits; the row letter comes from the row planes), so the other bits of bytes 1..3 carry the launch's
 * epoch: a word is valid exactly when it holds this launch's epoch, each is written by one write-through
 * store, and the consumer requests a block's 32 words one block ahead and re-reads (bounded) only what
 * had not arrived -- no counter, no fence (MI355X_MICROARCH.md, data-tagged granules; the round-1 form
 * published a counter behind an agent-scope release per block, which stalls a lone wave for microseconds).
 * The marks are zeroed when the batch is laid out and epochs are unique per process.  The work list puts
 * a job's chunks in ascending order, so the chunk a workgroup waits for was dispatched before it -- for
 * speed only: every wait is bounded, a time-out raises the abort word and the host repeats the pass chunk
 * by chunk.  The launch reserves enough LDS for ONE workgroup per compute unit.
 */
constexpr uint32_t kMarkPayload = 0x80808000u;            /* bits 31, 23, 15 */
__device__ __forceinline__ uint32_t mark_tag(uint32_t epoch)   /* 21 bits of epoch into bits 30..24, 22..16, 14..8 */
{
	return ((epoch & 0x7fu) << 8) | (((epoch >> 7) & 0x7fu) << 16) | (((epoch >> 14) & 0x7fu) << 24);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES *kLanes) void nw_fill_bits_wide(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs, int njobs,
                                                                   const TileRef *__restrict__ work, uint32_t epoch,
                                                                   int *__restrict__ abort_word)
{
	constexpr int OUT = OUT_NONE;
	constexpr int W = kCkptWords;
	__shared__ __attribute__((aligned(16))) uint32_t ring[WAVES][kRingSteps];
	__shared__ __attribute__((aligned(16))) uint32_t inject[WAVES][kBitBlock];
	__shared__ __attribute__((aligned(16))) uint32_t mbuf[WAVES][3][kBitBlock];
	/* word l + t per lane and step (64 different banks per store); one wave per SIMD: 16 bytes per lane and four steps at word 4 l + t */
	__shared__ __attribute__((aligned(16))) uint32_t scrap[WAVES][(WAVES == 4 ? 4 : 1) * kLanes + kBitBlock];
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
	uint32_t B0[W], B1[W];
#pragma unroll
	for (int h = 0; h < W; ++h) {
		B0[h] = cp[(s * kLanes + lane) * W + h];
		B1[h] = cp[J.nwords_pad + (s * kLanes + lane) * W + h];
	}
	const bool feeds = wv + 1 < WAVES && s + 1 < J.nstrips;          /* a wave of this workgroup reads my ring */
	const bool publishes = wv + 1 == WAVES && s + 1 < J.nstrips;     /* the next chunk reads my marks */
	uint32_t *marks = reinterpret_cast<uint32_t *>(arena + J.hand) + (size_t)s * 4 * J.steps_pad;
	const bool from_left_chunk = wv == 0 && chunk > 0;
	const uint32_t *left_marks = from_left_chunk ? reinterpret_cast<const uint32_t *>(arena + J.hand) + ((size_t)(s - 1) * 4 + 3) * J.steps_pad : nullptr;
	const bool writes = (lane == kLanes - 1) ? (feeds || publishes) : ((lane & 15) == 15);
	const uint32_t tag = mark_tag(epoch);
	constexpr uint32_t kTagMask = ~kMarkPayload & 0xffffff00u;

	BitState<W> S;
#pragma unroll
	for (int h = 0; h < W; ++h) {
		S.nH0[h] = ~0u;
		S.H1[h] = S.H2[h] = 0;
		/* the column planes are waited for HERE: left to the compiler the wait sits at their first use inside the block
		 * loop, where it is s_waitcnt vmcnt(0) -- and drains the checkpoint stores of the block before, every block */
		asm volatile("" : "+v"(B0[h]), "+v"(B1[h]));
	}
	S.PP = 0;
	/* the previous chunk's words for block 0, requested now; inside the loop always one block ahead */
	uint32_t pre = 0;
	if (from_left_chunk && 63 + (lane & 31) < J.steps_pad) pre = __hip_atomic_load(&left_marks[63 + (lane & 31)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	for (int b = 0; b < nb; ++b) {
		/* hand-off words entering lane 0 during this block: lane t prepares step t.  Carries from
		 * the producer's step 32b + t + 63, row letter of row 32b + t */
		uint32_t word = 0;
		const int ps = b * kBitBlock + 63 + (lane & 31);
		const int need = (b + 3 < nb) ? b + 3 : nb;             /* producer steps up to 32b + 94 */
		if (wv > 0) {
			if (!wait_at_least<WAVES == 4>(&made[wv - 1], need)) { if (lane == 0) atomicExch(abort_word, 1); return; }
			if (ps < J.steps_pad) word = ring[wv - 1][ps % kRingSteps] & 0xffffff00u;
			if (lane == 0) __hip_atomic_store(&taken[wv], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		} else if (from_left_chunk) {
			uint32_t v = pre;
			int spins = 0;
			for (;;) {
				const bool ok = ps >= J.steps_pad || (v & kTagMask) == tag;
				if (__all(ok)) break;
				__builtin_amdgcn_s_sleep(2);
				if (++spins > kSpinMax) { if (lane == 0) atomicExch(abort_word, 1); return; }
				if (!ok) v = __hip_atomic_load(&left_marks[ps], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (ps + kBitBlock < J.steps_pad) pre = __hip_atomic_load(&left_marks[ps + kBitBlock], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (ps < J.steps_pad) word = v & kMarkPayload;
		}
		const uint32_t a0 = a0n, a1 = a1n;
		if (b + 1 < nb) {
			a0n = rp[b + 1];
			a1n = rp[J.rowwords + b + 1];
		}
		word |= ((a0 >> (lane & 31)) & 1u) | (((a1 >> (lane & 31)) & 1u) << 1);
		if (lane < kBitBlock) inject[wv][lane] = word;
		uint32_t *lanebuf = !writes ? &scrap[wv][(WAVES == 4 ? 4 : 1) * lane] : (lane == kLanes - 1) ? &ring[wv][(b * kBitBlock) % kRingSteps] : &mbuf[wv][lane >> 4][0];
		if (feeds) {
			/* the ring slots of this block last held block b - kRing, whose words the consumer
			 * fetches while preparing its blocks b - kRing - 2 and b - kRing - 1 */
			if (!wait_at_least<WAVES == 4>(&taken[wv + 1], b - kRing)) { if (lane == 0) atomicExch(abort_word, 1); return; }
		}
		if (b < 2) bits_block<true, true, OUT, W, false, WAVES == 4>(S, B0, B1, inject[wv], lanebuf, nullptr, b * kBitBlock, lane);
		else bits_block<false, true, OUT, W, false, WAVES == 4>(S, B0, B1, inject[wv], lanebuf, nullptr, b * kBitBlock, lane);
		{
			/* streams 0..2: lanes 15/31/47 from mbuf, stream 3: lane 63 from the ring */
			const int g = lane >> 5, t = lane & 31;            /* lanes 0..31 -> streams 0 and 2, 32..63 -> 1 and 3 */
			marks[(size_t)g * J.steps_pad + b * kBitBlock + t] = mbuf[wv][g][t];
			if (g == 0) {
				marks[(size_t)2 * J.steps_pad + b * kBitBlock + t] = mbuf[wv][2][t];
			} else if (publishes) {                             /* for another compute unit: tagged, written through */
				const uint32_t v = (ring[wv][(b * kBitBlock + t) % kRingSteps] & kMarkPayload) | tag;
				__hip_atomic_store(&marks[(size_t)3 * J.steps_pad + b * kBitBlock + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			} else {
				marks[(size_t)3 * J.steps_pad + b * kBitBlock + t] = feeds ? ring[wv][(b * kBitBlock + t) % kRingSteps] : 0u;
			}
			save_state<W>(reinterpret_cast<uint4 *>(arena + J.ckpt), ((size_t)s * nb + b) * kLanes + lane, S);
		}
		if (feeds && lane == kLanes - 1) __hip_atomic_store(&made[wv], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

/*
 * K2c.  Traceback in checkpoint mode: no direction planes exist in HBM.  A round starts at the
 * current cell, in block `btop` (32 steps) of strip s, lane L.  Going up its diagonal the path
 * reaches block btop-d around lane L-d, so piece d = (block btop-d, the 16-lane group holding lane
 * L-d) is replayed with the fill's own step function: lane state from the checkpoint before the
 * block, the words entering the group's first lane from the fill's marks (or, for group 0, from the
 * strip to the left).  A wave replays 4 pieces at once (one per DPP row), 4 waves = 16 pieces =
 * ~480 path cells per round; wave 0 then walks inside the 16 LDS tiles (run-batched like K2b)
 * until the path leaves them.  Replay work: ~(nrows + ncols) / 31 pieces of 16 lanes x 32 steps,
 * 3 % of the fill's work for square matrices.
 * Which 16-lane group a piece holds: the path crosses a group boundary every 16 lanes = every 16.5 blocks,
 * and it does so in the MIDDLE of a block -- the cells of that block lie in two groups.  The round-2 form gave
 * every block one group ((L - d) / 16) and so ended a round at each crossing, usually twice (in-kernel timers on a
 * 16 kbp pair: 64 rounds where 34 + 8 strip crossings would do; a round costs 15 k cycles: checkpoint loads 1.8 k,
 * replay 5.4 k, walk 7.5 k).  Now a round is planned from the diagonal through the current cell: `dc` = the block
 * (counted down from btop) holding the first cell of the next group down; blocks before it are replayed for the
 * current group, blocks after it for the next one, and block dc for BOTH (piece dc and piece dc + 1).
 */
constexpr int kReplay = kBitCkptWords == 1 ? 4 : 2;   /* waves: 64 KB of LDS tiles either way */
constexpr int kPieces = 4 * kReplay;

struct RoundPlan {
	int ghi;              /* 16-lane group of the current cell */
	int dc;               /* blocks below btop at which the diagonal enters group ghi - 1 (huge: not in this strip / matrix) */
};

__device__ __forceinline__ RoundPlan plan_round(int r, int k, int lane0, int btop)
{
	constexpr int CL = 32 * kCkptWords;                    /* columns per lane */
	RoundPlan P;
	P.ghi = lane0 >> 4;
	P.dc = 1 << 20;
	const int cx = ((k - 1) % (16 * CL)) + 1;              /* cells up the diagonal to the first cell of the group below */
	const int r2 = r - cx, k2 = k - cx;
	if (P.ghi > 0 && r2 > 0 && k2 > 0) {
		const int lane2 = ((k2 - 1) / CL) & 63;
		P.dc = btop - ((r2 - 1) + lane2) / kBitBlock;
	}
	return P;
}

/* piece p of a round: how many blocks below btop, and which group */
__device__ __forceinline__ void piece_of(const RoundPlan &P, int p, int &delta, int &g)
{
	const bool hi = p <= P.dc;
	delta = hi ? p : p - 1;
	g = hi ? P.ghi : P.ghi - 1;
}

template <bool SCORE>       /* SCORE: also sum the move scores of the path (score-only callers skip the host walk) */
__global__ __launch_bounds__(kReplay *kLanes) void nw_traceback_replay(uint8_t *__restrict__ arena, const BitJob *__restrict__ jobs)
{
	constexpr int W = kCkptWords;
	__shared__ __attribute__((aligned(16))) uint2 tile[kPieces][kBitBlock * 16 * W];
	__shared__ uint32_t mtile[SCORE ? kPieces : 1][SCORE ? kBitBlock * 16 * W : 1];    /* match masks of the same cells */
	__shared__ __attribute__((aligned(16))) uint32_t inject[kPieces][kBitBlock];
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
		const int w0 = (k - 1) / (32 * W);                   /* lane column of the current cell */
		const int s = w0 >> 6;
		const int lane0 = w0 & 63;
		const int btop = ((r - 1) + lane0) / kBitBlock;
		const RoundPlan P = plan_round(r, k, lane0, btop);
		{
			/* this lane's piece */
			const int d = 4 * wv + (lane >> 4);
			int delta, g;
			piece_of(P, d, delta, g);
			const int b = btop - delta < 0 ? 0 : btop - delta;   /* pieces above block 0 replay block 0 and are never read */
			const int sl = 16 * g + (lane & 15);               /* lane index in the strip */
			BitState<W> S;
			if (b > 0) {
				load_state<W>(ck, ((size_t)s * nb + (b - 1)) * kLanes + sl, S);
			} else {
#pragma unroll
				for (int h = 0; h < W; ++h) {
					S.nH0[h] = ~0u;
					S.H1[h] = S.H2[h] = 0;
				}
				S.PP = 0;
			}
			uint32_t B0[W], B1[W];
#pragma unroll
			for (int h = 0; h < W; ++h) {
				B0[h] = cp[(s * kLanes + sl) * W + h];
				B1[h] = cp[J.nwords_pad + (s * kLanes + sl) * W + h];
			}
			/* words entering the piece's first lane: lane j of the row prepares steps j and j + 16 */
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int t = (lane & 15) + 16 * h;
				uint32_t word = 0;
				if (g == 0) {
					const int ps = b * kBitBlock + 63 + t;     /* lane 63 of the strip to the left is 63 steps ahead */
					if (s > 0 && ps < J.steps_pad) word = marks[((size_t)(s - 1) * 4 + 3) * J.steps_pad + ps] & 0xffffff00u;
					const uint32_t a0 = rp[b], a1 = rp[J.rowwords + b];
					word |= ((a0 >> t) & 1u) | (((a1 >> t) & 1u) << 1);
				} else {
					const int ps = b * kBitBlock + t - 1;      /* lane 16g-1 after the previous step */
					if (ps >= 0) word = marks[((size_t)s * 4 + (g - 1)) * J.steps_pad + ps];
				}
				inject[d][t] = word;
			}
			const bool ramp = btop - 4 * wv - 3 < 2;            /* wave-uniform: some piece of this wave is in block 0 or 1 */
			if (ramp) bits_block<true, false, OUT_TILE, W, SCORE>(S, B0, B1, inject[d], nullptr, tile[d], b * kBitBlock, sl, mtile[SCORE ? d : 0]);
			else bits_block<false, false, OUT_TILE, W, SCORE>(S, B0, B1, inject[d], nullptr, tile[d], b * kBitBlock, sl, mtile[SCORE ? d : 0]);
		}
		__syncthreads();
		if (wv == 0) {
			for (;;) {
				const int ri = r - lane, ki = k - lane;
				uint32_t code = 3;                             /* 3 = stop: border or outside the replayed pieces */
				bool match = false;
				if (ri > 0 && ki > 0) {
					const int kc = ki - 1;
					const int wi = kc / (32 * W);
					const int sl = wi & 63;
					const int l = (ri - 1) + sl;
					const int delta = btop - l / kBitBlock;        /* <= btop: l >= 0 */
					const int grp = sl >> 4;
					const bool second = delta > P.dc || (delta == P.dc && grp != P.ghi);
					const int d = delta + (second ? 1 : 0);
					if ((wi >> 6) == s && delta >= 0 && d < kPieces && grp == (second ? P.ghi - 1 : P.ghi)) {
						const int at = ((l % kBitBlock) * 16 + (sl & 15)) * W + ((kc >> 5) % W);
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
