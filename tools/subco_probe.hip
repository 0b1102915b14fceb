// Prices and checks the round-3 form of the bit-parallel step: the three carries a lane takes from its left
// neighbour arrive as a borrow -- v_sub_co_u32_dpp on the neighbour's complemented outgoing plane sets VCC to
// "top bit was set" for every lane at once -- and enter the chain through v_addc_co_u32; the row letters travel
// as x = B ^ R chained by v_xor_b32_dpp.  No hand-off word: no v_perm, v_bfe, v_alignbit, v_add3, no LDS store
// per step.  Each lane keeps the history of its outgoing carries in three accumulators (acc = 2 acc + carry).
//   part 1: what the DPP forms do in the first lane of a wave / row (zero fill, borrow, destination kept)
//   part 2: the new step against the round-2 step on one wave, 256 steps, same inputs -> same planes
//   part 3: cycles per wave-step per SIMD at 1 / 2 / 4 / 8 waves per SIMD, W = 1 and 2 words per lane -- the C++ form as
//           shipped, and the same block as ONE generated inline-assembly statement (tools/gen_bits_block.py): with several waves per
//           SIMD 4-5 % faster here and not faster inside nw_fill_bits (profiles/r03_ab_asm_block.txt); alone on its SIMD 7 % faster at
//           0 mod 8 ("at 0 mod 8", w1 column) and 10 % in the kernel: shipped for the one-wave-per-SIMD launches
// Build: python tools/gen_bits_block.py build/csadp_bits_block.inc --probe; hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/subco_probe.hip -o build/subco_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr uint32_t LA = 0xF0, LB = 0xCC, LC = 0xAA;
#define BITOP3(a, b, c, expr) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), (unsigned char)((expr) & 0xff)))

/* ---- part 1 -------------------------------------------------------------------------------------------- */
__global__ void k_sem(uint32_t *out, const uint32_t *in)
{
	const int lane = threadIdx.x;
	const uint32_t x = in[lane], z = in[64 + lane];
	uint32_t junk = 0xdeadbeefu, lo, hi;
	asm volatile("v_sub_co_u32_dpp %0, vcc, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	             "s_mov_b32 %1, vcc_lo\n\ts_mov_b32 %2, vcc_hi"
	             : "+v"(junk), "=s"(lo), "=s"(hi) : "v"(x), "v"(z) : "vcc");
	out[lane] = junk;
	if (lane == 0) { out[64] = lo; out[65] = hi; }
	uint32_t junk2 = 0xdeadbeefu, lo2, hi2;
	asm volatile("v_sub_co_u32_dpp %0, vcc, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	             "s_mov_b32 %1, vcc_lo\n\ts_mov_b32 %2, vcc_hi"
	             : "+v"(junk2), "=s"(lo2), "=s"(hi2) : "v"(x), "v"(z) : "vcc");
	if (lane == 0) { out[66] = lo2; out[67] = hi2; }
	uint32_t keep = 0x11110000u + lane;
	asm volatile("v_xor_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(keep) : "v"(x), "v"(z));
	out[128 + lane] = keep;
	uint32_t keep2 = 0x22220000u + lane;
	asm volatile("v_xor_b32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(keep2) : "v"(x), "v"(z));
	out[192 + lane] = keep2;
}

/* ---- the two steps ------------------------------------------------------------------------------------- */
template <int W>
struct StN {
	uint32_t nH0[W], H1[W], H2[W];
	uint32_t x0, x1;                 /* word 0's letter planes xor the row letter: what the right neighbour chains from */
	uint32_t nO2, nO1, nO0;          /* complements of the last word's outgoing planes */
	uint32_t acc2, acc1, acc0;       /* history of the lane's outgoing carries */
};

/* one step; pre0 / pre1 / z2 / z1 / z0: the values the first lane uses (its x0, x1; 1 = carry in), every other lane
 * must hold 0x80000000 in z2 / z1 / z0 */
template <int W, bool ROWS>
__device__ __forceinline__ void new_step(StN<W> &S, const uint32_t (&D)[2], const uint32_t (&E)[2], uint32_t pre0, uint32_t pre1, uint32_t z2, uint32_t z1,
                                         uint32_t z0)
{
	if (ROWS) {
		asm("v_xor_b32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pre0) : "v"(S.x0), "v"(D[0]));
		asm("v_xor_b32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pre1) : "v"(S.x1), "v"(D[1]));
	} else {
		asm("v_xor_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pre0) : "v"(S.x0), "v"(D[0]));
		asm("v_xor_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pre1) : "v"(S.x1), "v"(D[1]));
	}
	S.x0 = pre0;
	S.x1 = pre1;
	uint32_t nE[W], g2[W], s2[W], G2[W], g1[W], A1[W], s1[W], G1[W], O0[W], G0[W];
#pragma unroll
	for (int h = 0; h < W; ++h) {
		const uint32_t a0 = h == 0 ? pre0 : pre0 ^ E[0], a1 = h == 0 ? pre1 : pre1 ^ E[1];
		nE[h] = a0 | a1;
		g2[h] = BITOP3(nE[h], S.nH0[h], S.nH0[h], ~LA & LB);
	}
	uint32_t junk;
	if (W == 1) {
		if (ROWS)
			asm("v_sub_co_u32_dpp %0, vcc, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %5, %6, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc"
			    : "=&v"(junk), "=&v"(s2[0]), "+v"(S.acc2) : "v"(S.nO2), "v"(z2), "v"(S.nH0[0]), "v"(g2[0]) : "vcc");
		else
			asm("v_sub_co_u32_dpp %0, vcc, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %5, %6, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc"
			    : "=&v"(junk), "=&v"(s2[0]), "+v"(S.acc2) : "v"(S.nO2), "v"(z2), "v"(S.nH0[0]), "v"(g2[0]) : "vcc");
	} else {
		asm("v_sub_co_u32_dpp %0, vcc, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %6, %7, vcc\n\tv_addc_co_u32 %2, vcc, %8, %9, vcc\n\t"
		    "v_addc_co_u32 %3, vcc, %3, %3, vcc"
		    : "=&v"(junk), "=&v"(s2[0]), "=&v"(s2[W - 1]), "+v"(S.acc2)
		    : "v"(S.nO2), "v"(z2), "v"(S.nH0[0]), "v"(g2[0]), "v"(S.nH0[W - 1]), "v"(g2[W - 1]) : "vcc");
	}
#pragma unroll
	for (int h = 0; h < W; ++h) {
		G2[h] = BITOP3(s2[h], S.nH0[h], g2[h], LA ^ LB ^ LC);
		if (h == W - 1) S.nO2 = BITOP3(g2[h], S.nH0[h], G2[h], ~(LA | (LB & LC)));
		const uint32_t t1 = BITOP3(nE[h], S.nH0[h], G2[h], ~LA | (~LB & LC));
		g1[h] = BITOP3(t1, S.H1[h], S.H1[h], LA & ~LB);
		A1[h] = BITOP3(g1[h], nE[h], S.nH0[h], LA | (LB & LC));
	}
	if (W == 1) {
		if (ROWS)
			asm("v_sub_co_u32_dpp %0, vcc, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %5, %6, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc"
			    : "=&v"(junk), "=&v"(s1[0]), "+v"(S.acc1) : "v"(S.nO1), "v"(z1), "v"(A1[0]), "v"(g1[0]) : "vcc");
		else
			asm("v_sub_co_u32_dpp %0, vcc, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %5, %6, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc"
			    : "=&v"(junk), "=&v"(s1[0]), "+v"(S.acc1) : "v"(S.nO1), "v"(z1), "v"(A1[0]), "v"(g1[0]) : "vcc");
	} else {
		asm("v_sub_co_u32_dpp %0, vcc, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %6, %7, vcc\n\tv_addc_co_u32 %2, vcc, %8, %9, vcc\n\t"
		    "v_addc_co_u32 %3, vcc, %3, %3, vcc"
		    : "=&v"(junk), "=&v"(s1[0]), "=&v"(s1[W - 1]), "+v"(S.acc1)
		    : "v"(S.nO1), "v"(z1), "v"(A1[0]), "v"(g1[0]), "v"(A1[W - 1]), "v"(g1[W - 1]) : "vcc");
	}
#pragma unroll
	for (int h = 0; h < W; ++h) {
		G1[h] = BITOP3(s1[h], A1[h], g1[h], LA ^ LB ^ LC);
		if (h == W - 1) S.nO1 = BITOP3(g1[h], A1[h], G1[h], ~(LA | (LB & LC)));
		const uint32_t v = BITOP3(S.H1[h], G2[h], G1[h], (LA & LB) | (~LA & LC));
		const uint32_t w = BITOP3(nE[h], v, S.H2[h], ~LC & (~LA | LB));
		O0[h] = BITOP3(w, nE[h], S.nH0[h], LA | (LB & LC));
	}
	const uint32_t nO0last = BITOP3(O0[W - 1], O0[W - 1], O0[W - 1], ~LA);
	if (W == 1) {
		if (ROWS)
			asm("v_sub_co_u32_dpp %0, vcc, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %5, %5, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc"
			    : "=&v"(junk), "=&v"(G0[0]), "+v"(S.acc0) : "v"(S.nO0), "v"(z0), "v"(O0[0]) : "vcc");
		else
			asm("v_sub_co_u32_dpp %0, vcc, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %5, %5, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc"
			    : "=&v"(junk), "=&v"(G0[0]), "+v"(S.acc0) : "v"(S.nO0), "v"(z0), "v"(O0[0]) : "vcc");
	} else {
		asm("v_sub_co_u32_dpp %0, vcc, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %6, %6, vcc\n\tv_addc_co_u32 %2, vcc, %7, %7, vcc\n\t"
		    "v_addc_co_u32 %3, vcc, %3, %3, vcc"
		    : "=&v"(junk), "=&v"(G0[0]), "=&v"(G0[W - 1]), "+v"(S.acc0)
		    : "v"(S.nO0), "v"(z0), "v"(O0[0]), "v"(O0[W - 1]) : "vcc");
	}
	S.nO0 = nO0last;
#pragma unroll
	for (int h = 0; h < W; ++h) {
		const uint32_t C1 = BITOP3(nE[h], G2[h], S.H2[h], ~LA | LB | LC);
		const uint32_t C0 = BITOP3(nE[h], G1[h], S.H1[h], ~LA | LB | LC);
		const uint32_t T2 = BITOP3(C1, G0[h], G0[h], LA & ~LB);
		const uint32_t a1 = BITOP3(C1, G1[h], G1[h], LA & ~LB);
		const uint32_t T1 = BITOP3(G0[h], a1, C0, (LA & LB) | (~LA & LC));
		const uint32_t b0 = BITOP3(C0, G1[h], G0[h], LC & (~LA | LB));
		S.nH0[h] = BITOP3(b0, C1, G2[h], LA & (~LB | LC));
		S.H1[h] = T1;
		S.H2[h] = T2;
	}
}

struct StO {
	uint32_t nH0, H1, H2, PP;
};
__device__ __forceinline__ void old_step(StO &S, uint32_t B0, uint32_t B1, uint32_t cur)
{
	uint32_t inw = cur;
	asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(inw) : "v"(S.PP));
	const uint32_t R0 = (uint32_t)__builtin_amdgcn_sbfe((int)inw, 0, 1);
	const uint32_t R1 = (uint32_t)__builtin_amdgcn_sbfe((int)inw, 1, 1);
	const uint32_t c2 = __builtin_amdgcn_ubfe(inw, 15, 1);
	const uint32_t c1 = __builtin_amdgcn_ubfe(inw, 23, 1);
	const uint32_t nH0 = S.nH0, H1 = S.H1, H2 = S.H2;
	const uint32_t x0 = B0 ^ R0;
	const uint32_t nE = BITOP3(x0, B1, R1, LA | (LB ^ LC));
	const uint32_t g2 = BITOP3(nE, nH0, nH0, ~LA & LB);
	const uint32_t s2 = nH0 + g2 + c2;
	const uint32_t G2 = BITOP3(s2, nH0, g2, LA ^ LB ^ LC);
	const uint32_t O2 = BITOP3(g2, nH0, G2, LA | (LB & LC));
	const uint32_t t1 = BITOP3(nE, nH0, G2, ~LA | (~LB & LC));
	const uint32_t g1 = BITOP3(t1, H1, H1, LA & ~LB);
	const uint32_t A1 = BITOP3(g1, nE, nH0, LA | (LB & LC));
	const uint32_t s1 = A1 + g1 + c1;
	const uint32_t G1 = BITOP3(s1, A1, g1, LA ^ LB ^ LC);
	const uint32_t O1 = BITOP3(g1, A1, G1, LA | (LB & LC));
	const uint32_t v = BITOP3(H1, G2, G1, (LA & LB) | (~LA & LC));
	const uint32_t w = BITOP3(nE, v, H2, ~LC & (~LA | LB));
	const uint32_t O0 = BITOP3(w, nE, nH0, LA | (LB & LC));
	const uint32_t G0 = __builtin_amdgcn_alignbit(O0, inw, 31);
	const uint32_t q = __builtin_amdgcn_perm(O1, O2, 0x0c07030cu);
	const uint32_t pq = __builtin_amdgcn_perm(O0, q, 0x0702010cu);
	S.PP = BITOP3(pq, inw, 0xffu, LA | (LB & LC));
	const uint32_t C1 = BITOP3(nE, G2, H2, ~LA | LB | LC);
	const uint32_t C0 = BITOP3(nE, G1, H1, ~LA | LB | LC);
	S.H2 = BITOP3(C1, G0, G0, LA & ~LB);
	const uint32_t a1 = BITOP3(C1, G1, G1, LA & ~LB);
	S.H1 = BITOP3(G0, a1, C0, (LA & LB) | (~LA & LC));
	const uint32_t b0 = BITOP3(C0, G1, G0, LC & (~LA | LB));
	S.nH0 = BITOP3(b0, C1, G2, LA & (~LB | LC));
}

/* ---- part 2: one wave, `steps` steps, the same column letters, row letters and carries into lane 0 ----- */
/* in: [0..63] B0, [64..127] B1, [128 + t] bits 0,1 = row letter of step t, bits 2,3,4 = carries >= 2, >= 1, >= 0 into lane 0 */
template <int W>
__global__ void k_verify_new(uint32_t *out, const uint32_t *in, int steps)
{
	const int lane = threadIdx.x;
	uint32_t B0[W], B1[W];
	for (int h = 0; h < W; ++h) { B0[h] = in[lane * W + h]; B1[h] = in[64 * W + lane * W + h]; }
	/* chain constants: this lane's word 0 against the left lane's word 0 */
	uint32_t D[2], E[2] = {B0[0] ^ B0[W - 1], B1[0] ^ B1[W - 1]};
	const int left = lane > 0 ? lane - 1 : 0;
	D[0] = B0[0] ^ in[left * W];
	D[1] = B1[0] ^ in[64 * W + left * W];
	StN<W> S;
	for (int h = 0; h < W; ++h) { S.nH0[h] = ~0u; S.H1[h] = S.H2[h] = 0; }
	S.x0 = S.x1 = 0;
	S.nO2 = S.nO1 = S.nO0 = ~0u;
	S.acc2 = S.acc1 = S.acc0 = 0;
	for (int t = 0; t < steps; ++t) {
		const uint32_t u = in[128 * W + t];
		const uint32_t R0 = (u & 1) ? ~0u : 0u, R1 = (u & 2) ? ~0u : 0u;
		const uint32_t zc = lane == 0 ? 0u : 0x80000000u;
		new_step<W, false>(S, D, E, B0[0] ^ R0, B1[0] ^ R1, lane == 0 ? (u >> 2) & 1 : zc, lane == 0 ? (u >> 3) & 1 : zc, lane == 0 ? (u >> 4) & 1 : zc);
		if ((t & 31) == 31) {
			out[(t >> 5) * 64 * 3 + lane * 3 + 0] = S.acc2;
			out[(t >> 5) * 64 * 3 + lane * 3 + 1] = S.acc1;
			out[(t >> 5) * 64 * 3 + lane * 3 + 2] = S.acc0;
		}
	}
	uint32_t *fin = out + 65536;
	for (int h = 0; h < W; ++h) {
		fin[(lane * W + h) * 3 + 0] = S.nH0[h];
		fin[(lane * W + h) * 3 + 1] = S.H1[h];
		fin[(lane * W + h) * 3 + 2] = S.H2[h];
	}
}

/* the round-2 step on 64 * W lanes' worth of words, run as W waves' worth by ONE wave per word column is not possible (the
 * carries cross every word), so the reference runs words one per lane over 64 * W virtual lanes: host side (plain C below) */

/* ---- part 3: timing ------------------------------------------------------------------------------------ */
template <int W, int LDS>
__global__ void k_time_new(uint32_t *out, const uint32_t *in, int nblocks)
{
	__shared__ __attribute__((aligned(16))) uint32_t inj[16][32 * 8];
	__shared__ __attribute__((aligned(16))) uint32_t konst[32 * 8];
	const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 15;
	for (int i = threadIdx.x; i < 32 * 8; i += blockDim.x) konst[i] = 0x80000000u;
	for (int i = lane; i < 32 * 8; i += 64) inj[wv][i] = in[(i * 7) & 1023] & 1u;
	__syncthreads();
	const uint32_t *src = lane == 0 ? &inj[wv][0] : &konst[0];
	StN<W> S;
	for (int h = 0; h < W; ++h) { S.nH0[h] = in[lane + 64 * h]; S.H1[h] = in[128 + lane + 64 * h]; S.H2[h] = in[256 + lane + 64 * h]; }
	S.x0 = in[384 + lane]; S.x1 = in[448 + lane];
	S.nO2 = in[512 + lane]; S.nO1 = in[576 + lane]; S.nO0 = in[640 + lane];
	S.acc2 = S.acc1 = S.acc0 = 0;
	const uint32_t D[2] = {in[704 + lane], in[768 + lane]}, E[2] = {in[832 + lane], in[896 + lane]};
	uint32_t sink = 0;
	uint32_t zc = lane == 0 ? 0u : 0x80000000u;
	uint32_t q0 = in[960 + lane], q1 = in[1000 + lane];
	for (int b = 0; b < nblocks; ++b) {
		uint4 nx = *reinterpret_cast<const uint4 *>(src);
		uint32_t nx4 = src[4];
#pragma unroll
		for (int t = 0; t < 32; ++t) {
			if (LDS) {
				const uint4 c = nx;
				const uint32_t c4 = nx4;
				nx = *reinterpret_cast<const uint4 *>(src + ((t + 1) & 31) * 8);
				nx4 = src[((t + 1) & 31) * 8 + 4];
				new_step<W, false>(S, D, E, c.x, c.y, c.z, c.w, c4);
			} else {
				const uint32_t o0 = S.x0, o1 = S.x1;
				new_step<W, false>(S, D, E, q0, q1, zc, zc, zc);
				q0 = o0;
				q1 = o1;
			}
		}
		sink ^= S.acc2 ^ S.acc1 ^ S.acc0;
		asm volatile("" : "+v"(zc));
	}
	uint32_t r = sink ^ S.x0 ^ S.x1 ^ S.nO2 ^ S.nO1 ^ S.nO0;
	for (int h = 0; h < W; ++h) r ^= S.nH0[h] ^ S.H1[h] ^ S.H2[h];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int LDSMODE>
__global__ void k_time_old(uint32_t *out, const uint32_t *in, int nblocks)
{
	__shared__ __attribute__((aligned(16))) uint32_t lbuf[16][64 + 64];
	const int lane = threadIdx.x & 63;
	uint32_t *mine = &lbuf[(threadIdx.x >> 6) & 15][lane];
	const uint32_t *inj = &lbuf[(threadIdx.x >> 6) & 15][0];
	StO S{in[lane], in[64 + lane], in[128 + lane], in[320 + lane]};
	const uint32_t B0 = in[192 + lane], B1 = in[256 + lane];
	const uint32_t *uni = in + 512;
	for (int b = 0; b < nblocks; ++b) {
		uint32_t cur = uni[(b & 15) * 5];
#pragma unroll
		for (int t = 0; t < 32; ++t) {
			old_step(S, B0, B1, cur);
			if (LDSMODE == 1) {
				mine[t] = S.PP;
				cur ^= inj[(t + 1) & 31];
			}
			cur = cur * 5 + 1;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = S.nH0 ^ S.H1 ^ S.H2 ^ S.PP;
}


/* ---- the same block as ONE generated inline-assembly statement (tools/gen_bits_block.py) --------------- */
#include "../build/csadp_bits_block.inc"       /* python tools/gen_bits_block.py build/csadp_bits_block.inc --probe */

template <int W, int VAR>      /* 0: as shipped (4-byte instructions paired, 8-byte ones at 4 mod 8), 1: unpaired, 2: early borrows, 3: at 0 mod 8 */
__device__ __forceinline__ void asm_block(StN<W> &S, const uint32_t (&D)[2], const uint32_t (&E)[2], const uint32_t *ip)
{
	const uint32_t ipa = (uint32_t)(uintptr_t)ip;           /* LDS byte address = low half of the generic pointer */
	if constexpr (W == 1) {
#define OPS1 : [nh0_0] "+v"(S.nH0[0]), [h1_0] "+v"(S.H1[0]), [h2_0] "+v"(S.H2[0]), [x0] "+v"(S.x0), [x1] "+v"(S.x1), [no2] "+v"(S.nO2), \
	[no1] "+v"(S.nO1), [no0] "+v"(S.nO0), [a2] "+v"(S.acc2), [a1] "+v"(S.acc1), [a0] "+v"(S.acc0) \
	: [d0] "v"(D[0]), [d1] "v"(D[1]), [ip] "v"(ipa) : BITS_BLOCK_CLOBBERS_W1
		if (VAR == 1) asm volatile(BITS_BLOCK_ASM_W1_PLAIN OPS1);
		else if (VAR == 2) asm volatile(BITS_BLOCK_ASM_W1_EARLY OPS1);
		else if (VAR == 3) asm volatile(BITS_BLOCK_ASM_W1_AT0 OPS1);
		else asm volatile(BITS_BLOCK_ASM_W1 OPS1);
#undef OPS1
	} else {
#define OPS2 : [nh0_0] "+v"(S.nH0[0]), [h1_0] "+v"(S.H1[0]), [h2_0] "+v"(S.H2[0]), [nh0_1] "+v"(S.nH0[1]), [h1_1] "+v"(S.H1[1]), [h2_1] "+v"(S.H2[1]), \
	[x0] "+v"(S.x0), [x1] "+v"(S.x1), [no2] "+v"(S.nO2), [no1] "+v"(S.nO1), [no0] "+v"(S.nO0), [a2] "+v"(S.acc2), [a1] "+v"(S.acc1), [a0] "+v"(S.acc0) \
	: [d0] "v"(D[0]), [d1] "v"(D[1]), [e0_1] "v"(E[0]), [e1_1] "v"(E[1]), [ip] "v"(ipa) : BITS_BLOCK_CLOBBERS_W2
		if (VAR == 1) asm volatile(BITS_BLOCK_ASM_W2_PLAIN OPS2);
		else if (VAR == 2) asm volatile(BITS_BLOCK_ASM_W2_EARLY OPS2);
		else if (VAR == 3) asm volatile(BITS_BLOCK_ASM_W2_AT0 OPS2);
		else asm volatile(BITS_BLOCK_ASM_W2 OPS2);
#undef OPS2
	}
}

/* part 2 with the generated block: the rows lane 0 reads are written to LDS per block, everybody else reads the constant rows */
template <int W>
__global__ void k_verify_asm(uint32_t *out, const uint32_t *in, int steps)
{
	__shared__ __attribute__((aligned(16))) uint32_t inj[32 * 8];
	__shared__ __attribute__((aligned(16))) uint32_t konst[32 * 8 + 4];
	const int lane = threadIdx.x;
	for (int i = lane; i < 32 * 8 + 4; i += 64) konst[i] = 0x80000000u;
	uint32_t B0[W], B1[W];
	for (int h = 0; h < W; ++h) { B0[h] = in[lane * W + h]; B1[h] = in[64 * W + lane * W + h]; }
	uint32_t D[2], E[2] = {B0[0] ^ B0[W - 1], B1[0] ^ B1[W - 1]};
	const int left = lane > 0 ? lane - 1 : 0;
	D[0] = B0[0] ^ in[left * W];
	D[1] = B1[0] ^ in[64 * W + left * W];
	const uint32_t b00 = __builtin_amdgcn_readfirstlane(B0[0]), b10 = __builtin_amdgcn_readfirstlane(B1[0]);
	StN<W> S;
	for (int h = 0; h < W; ++h) { S.nH0[h] = ~0u; S.H1[h] = S.H2[h] = 0; }
	S.x0 = S.x1 = 0;
	S.nO2 = S.nO1 = S.nO0 = 0x80000000u;
	S.acc2 = S.acc1 = S.acc0 = 0;
	const uint32_t *ip = lane == 0 ? &inj[0] : &konst[4];
	for (int b = 0; b < steps / 32; ++b) {
		if (lane < 32) {
			const uint32_t u = in[128 * W + b * 32 + lane];
			inj[lane * 8 + 0] = b00 ^ ((u & 1) ? ~0u : 0u);
			inj[lane * 8 + 1] = b10 ^ ((u & 2) ? ~0u : 0u);
			inj[lane * 8 + 2] = (u >> 2) & 1;
			inj[lane * 8 + 3] = (u >> 3) & 1;
			inj[lane * 8 + 4] = (u >> 4) & 1;
		}
		asm_block<W, 0>(S, D, E, ip);
		out[b * 64 * 3 + lane * 3 + 0] = S.acc2;
		out[b * 64 * 3 + lane * 3 + 1] = S.acc1;
		out[b * 64 * 3 + lane * 3 + 2] = S.acc0;
	}
	uint32_t *fin = out + 65536;
	for (int h = 0; h < W; ++h) {
		fin[(lane * W + h) * 3 + 0] = S.nH0[h];
		fin[(lane * W + h) * 3 + 1] = S.H1[h];
		fin[(lane * W + h) * 3 + 2] = S.H2[h];
	}
}

template <int W, int VAR, int PHASE>
__global__ void k_time_asm(uint32_t *out, const uint32_t *in, int nblocks)
{
	__shared__ __attribute__((aligned(16))) uint32_t inj[16][32 * 8];
	__shared__ __attribute__((aligned(16))) uint32_t konst[32 * 8 + 4];
	const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 15;
	for (int i = threadIdx.x; i < 32 * 8 + 4; i += blockDim.x) konst[i] = 0x80000000u;
	for (int i = lane; i < 32 * 8; i += 64) inj[wv][i] = in[(i * 7) & 1023] & 1u;
	__syncthreads();
	const uint32_t *ip = lane == 0 ? &inj[wv][0] : &konst[4];
	StN<W> S;
	for (int h = 0; h < W; ++h) { S.nH0[h] = in[lane + 64 * h]; S.H1[h] = in[128 + lane + 64 * h]; S.H2[h] = in[256 + lane + 64 * h]; }
	S.x0 = in[384 + lane]; S.x1 = in[448 + lane];
	S.nO2 = in[512 + lane]; S.nO1 = in[576 + lane]; S.nO0 = in[640 + lane];
	S.acc2 = S.acc1 = S.acc0 = 0;
	const uint32_t D[2] = {in[704 + lane], in[768 + lane]}, E[2] = {in[832 + lane], in[896 + lane]};
	uint32_t sink = 0;
	for (int b = 0; b < nblocks; ++b) {
		if (PHASE == 1) asm volatile("s_nop 0");
		asm_block<W, VAR>(S, D, E, ip);
		sink ^= S.acc2 ^ S.acc1 ^ S.acc0;
	}
	uint32_t r = sink ^ S.x0 ^ S.x1 ^ S.nO2 ^ S.nO1 ^ S.nO0;
	for (int h = 0; h < W; ++h) r ^= S.nH0[h] ^ S.H1[h] ^ S.H2[h];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K>
static void run(const char *name, K kernel, uint32_t *out, const uint32_t *in, int cells_per_step)
{
	const int nblocks = 512;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int shapes[4][2] = {{256, 256}, {256, 512}, {256, 1024}, {512, 1024}};
	printf("%-44s", name);
	for (int s = 0; s < 4; ++s) {
		hipLaunchKernelGGL(kernel, dim3(shapes[s][0]), dim3(shapes[s][1]), 0, 0, out, in, 4);
		CHECK(hipDeviceSynchronize());
		float best = 1e9f;
		for (int r = 0; r < 3; ++r) {
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(kernel, dim3(shapes[s][0]), dim3(shapes[s][1]), 0, 0, out, in, nblocks);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		const int wps = 1 << s;
		const double cyc = best * 1e-3 * 2.4e9 / (nblocks * 32.0) / wps;   /* SIMD cycles per wave-step at a nominal 2.4 GHz */
		printf("  w%d %6.1f (%.2f/cell)", wps, cyc, cyc / cells_per_step);
	}
	printf("\n");
}

/* host reference of part 2: the plain word recurrence, lane by lane (W words each), skewed like the wave: lane L works on
 * row t - L at step t; a lane's first word takes the carries its left neighbour's last word put out one step earlier, the
 * other words the carries of the word before them in the same step */
static void host_ref(int W, const uint32_t *B0, const uint32_t *B1, const uint32_t *rows, int steps, uint32_t *fin, uint32_t *accs)
{
	const int nwords = 64 * W;
	uint32_t *nH0 = (uint32_t *)malloc(nwords * 4), *H1 = (uint32_t *)calloc(nwords, 4), *H2 = (uint32_t *)calloc(nwords, 4);
	uint32_t o2[64] = {0}, o1[64] = {0}, o0[64] = {0}, a2[64] = {0}, a1_[64] = {0}, a0[64] = {0};
	for (int j = 0; j < nwords; ++j) nH0[j] = ~0u;
	for (int t = 0; t < steps; ++t) {
		for (int L = 63; L >= 0; --L) {                       /* right to left: lane L reads lane L-1's outputs of the PREVIOUS step */
			const int row = t - L;
			uint32_t c2, c1, c0, xa, xb;                      /* xa / xb: word 0's letter planes xor the row letter */
			if (L == 0) {
				const uint32_t u = rows[t];
				xa = B0[0] ^ ((u & 1) ? ~0u : 0u);
				xb = B1[0] ^ ((u & 2) ? ~0u : 0u);
				c2 = (u >> 2) & 1; c1 = (u >> 3) & 1; c0 = (u >> 4) & 1;
			} else {
				c2 = o2[L - 1] >> 31; c1 = o1[L - 1] >> 31; c0 = o0[L - 1] >> 31;
				if (row >= 0) {
					const uint32_t u = rows[row];
					xa = B0[L * W] ^ ((u & 1) ? ~0u : 0u);
					xb = B1[L * W] ^ ((u & 2) ? ~0u : 0u);
				} else {                                      /* the chain has not arrived: it started from lane L-t-1's initial 0 */
					xa = B0[L * W] ^ B0[(L - t - 1) * W];
					xb = B1[L * W] ^ B1[(L - t - 1) * W];
				}
			}
			for (int hh = 0; hh < W; ++hh) {
				const int j = L * W + hh;
				const uint32_t x0 = xa ^ B0[L * W] ^ B0[j], x1 = xb ^ B1[L * W] ^ B1[j];
				const uint32_t nE = x0 | x1;
				const uint32_t g2 = ~nE & nH0[j];
				const uint32_t s2 = nH0[j] + g2 + c2;
				const uint32_t G2 = s2 ^ nH0[j] ^ g2;
				const uint32_t O2 = g2 | (nH0[j] & G2);
				const uint32_t t1 = ~nE | (~nH0[j] & G2);
				const uint32_t g1 = t1 & ~H1[j];
				const uint32_t A1 = g1 | (nE & nH0[j]);
				const uint32_t s1 = A1 + g1 + c1;
				const uint32_t G1 = s1 ^ A1 ^ g1;
				const uint32_t O1 = g1 | (A1 & G1);
				const uint32_t v = (H1[j] & G2) | (~H1[j] & G1);
				const uint32_t w = ~H2[j] & (~nE | v);
				const uint32_t O0 = w | (nE & nH0[j]);
				const uint32_t G0 = (O0 << 1) | c0;
				const uint32_t C1 = ~nE | G2 | H2[j];
				const uint32_t C0 = ~nE | G1 | H1[j];
				const uint32_t T2 = C1 & ~G0;
				const uint32_t aa = C1 & ~G1;
				const uint32_t T1 = (G0 & aa) | (~G0 & C0);
				const uint32_t b0 = G0 & (~C0 | G1);
				const uint32_t nT0 = b0 & (~C1 | G2);
				nH0[j] = nT0; H1[j] = T1; H2[j] = T2;
				c2 = O2 >> 31; c1 = O1 >> 31; c0 = O0 >> 31;
				if (hh == W - 1) { o2[L] = O2; o1[L] = O1; o0[L] = O0; }
			}
			a2[L] = (a2[L] << 1) | c2; a1_[L] = (a1_[L] << 1) | c1; a0[L] = (a0[L] << 1) | c0;
		}
		if ((t & 31) == 31)
			for (int L = 0; L < 64; ++L) {
				accs[((t >> 5) * 64 + L) * 3 + 0] = a2[L];
				accs[((t >> 5) * 64 + L) * 3 + 1] = a1_[L];
				accs[((t >> 5) * 64 + L) * 3 + 2] = a0[L];
			}
	}
	for (int j = 0; j < nwords; ++j) { fin[j * 3] = nH0[j]; fin[j * 3 + 1] = H1[j]; fin[j * 3 + 2] = H2[j]; }
}

int main()
{
	uint32_t *in, *out;
	CHECK(hipMalloc(&in, 8192 * 4));
	CHECK(hipMalloc(&out, 1024 * 1024 * 4));
	static uint32_t h[8192], o[1024 * 1024 / 4];
	uint32_t x = 12345;
	for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }

	/* part 1 */
	for (int i = 0; i < 64; ++i) { h[i] = (i & 1) ? 0x80000000u | i : (uint32_t)i; h[64 + i] = (i == 0 || i == 16) ? 1u : (i == 32 ? 0u : 0x80000000u); }
	CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, out, in);
	CHECK(hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost));
	/* expected, wave_shr: lane 0: 0 - 1 borrows -> 1; lane i >= 1: x[i-1] < z[i] unsigned */
	unsigned long long exp_w = 0, exp_r = 0;
	for (int i = 0; i < 64; ++i) {
		const uint32_t s_w = i ? h[i - 1] : 0u, s_r = (i & 15) ? h[i - 1] : 0u;
		if (s_w < h[64 + i]) exp_w |= 1ull << i;
		if (s_r < h[64 + i]) exp_r |= 1ull << i;
	}
	printf("sub_co dpp wave_shr: vcc %08x%08x expected %016llx  lane0 dst %08x\n", o[65], o[64], exp_w, o[0]);
	printf("sub_co dpp row_shr : vcc %08x%08x expected %016llx\n", o[67], o[66], exp_r);
	printf("xor dpp wave_shr keeps lane 0: %08x (0x11110000 expected)  lane 1: %08x (expected %08x)\n", o[128], o[129], h[0] ^ h[65]);
	printf("xor dpp row_shr  keeps lane 16: %08x (0x22220010 expected) lane 17: %08x (expected %08x)\n", o[192 + 16], o[192 + 17], h[16] ^ h[64 + 17]);

	/* part 2 */
	for (int W = 1; W <= 2; ++W) {
		for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
		const int steps = 256, nwords = 64 * W;
		/* correlated letters so that all planes get exercised: rows mostly equal to column letters */
		for (int t = 0; t < steps; ++t) h[128 * W + t] &= 31u;
		CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
		CHECK(hipMemset(out, 0, 1024 * 1024 * 4));
		if (W == 1) hipLaunchKernelGGL(k_verify_new<1>, dim3(1), dim3(64), 0, 0, out, in, steps);
		else hipLaunchKernelGGL(k_verify_new<2>, dim3(1), dim3(64), 0, 0, out, in, steps);
		CHECK(hipDeviceSynchronize());
		static uint32_t dv[65536 + 1024];
		CHECK(hipMemcpy(dv, out, sizeof dv, hipMemcpyDeviceToHost));
		static uint32_t fin[128 * 3], accs[8 * 128 * 3];
		host_ref(W, h, h + 64 * W, h + 128 * W, steps, fin, accs);
		int bad = 0;
		for (int j = 0; j < nwords * 3; ++j)
			if (fin[j] != dv[65536 + j]) { if (bad < 5) printf("  W=%d plane mismatch word %d/%d: %08x vs %08x\n", W, j / 3, j % 3, dv[65536 + j], fin[j]); ++bad; }
		/* accumulators: the device keeps one set per LANE = of its last word */
		for (int b = 0; b < steps / 32; ++b)
			for (int l = 0; l < 64; ++l)
				for (int p = 0; p < 3; ++p)
					if (dv[b * 64 * 3 + l * 3 + p] != accs[(b * 64 + l) * 3 + p]) { if (bad < 5) printf("  W=%d acc mismatch block %d lane %d plane %d: %08x vs %08x\n", W, b, l, p, dv[b * 64 * 3 + l * 3 + p], accs[(b * 64 + l) * 3 + p]); ++bad; }
		printf("W=%d: new step vs plain word recurrence over %d steps: %s (%d mismatches)\n", W, steps, bad ? "DIFFERENT" : "identical", bad);
		/* the generated block on the same inputs */
		CHECK(hipMemset(out, 0, 1024 * 1024 * 4));
		if (W == 1) hipLaunchKernelGGL(k_verify_asm<1>, dim3(1), dim3(64), 0, 0, out, in, steps);
		else hipLaunchKernelGGL(k_verify_asm<2>, dim3(1), dim3(64), 0, 0, out, in, steps);
		CHECK(hipDeviceSynchronize());
		CHECK(hipMemcpy(dv, out, sizeof dv, hipMemcpyDeviceToHost));
		bad = 0;
		for (int j = 0; j < nwords * 3; ++j)
			if (fin[j] != dv[65536 + j]) { if (bad < 5) printf("  W=%d asm plane mismatch word %d/%d: %08x vs %08x\n", W, j / 3, j % 3, dv[65536 + j], fin[j]); ++bad; }
		for (int b = 0; b < steps / 32; ++b)
			for (int l = 0; l < 64; ++l)
				for (int p = 0; p < 3; ++p)
					if (dv[b * 64 * 3 + l * 3 + p] != accs[(b * 64 + l) * 3 + p]) { if (bad < 5) printf("  W=%d asm acc mismatch block %d lane %d plane %d: %08x vs %08x\n", W, b, l, p, dv[b * 64 * 3 + l * 3 + p], accs[(b * 64 + l) * 3 + p]); ++bad; }
		printf("W=%d: generated assembly block vs plain word recurrence: %s (%d mismatches)\n", W, bad ? "DIFFERENT" : "identical", bad);
	}

	/* part 3 */
	for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
	CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
	printf("cycles per wave-step per SIMD at a nominal 2.4 GHz (cycles per cell)\n");
	run("round-2 step (31 VALU), no LDS", k_time_old<0>, out, in, 32);
	run("round-2 step + ds_write_b32 + ds_read_b32", k_time_old<1>, out, in, 32);
	run("borrow step W=1 (31 VALU), no LDS", (k_time_new<1, 0>), out, in, 32);
	run("borrow step W=1 + ds_read_b128 + b32", (k_time_new<1, 1>), out, in, 32);
	run("borrow step W=2 (53 VALU), no LDS", (k_time_new<2, 0>), out, in, 64);
	run("borrow step W=2 + ds_read_b128 + b32", (k_time_new<2, 1>), out, in, 64);
	run("generated block W=1 as shipped", (k_time_asm<1, 0, 0>), out, in, 32);
	run("generated block W=1, 4-byte ones unpaired", (k_time_asm<1, 1, 0>), out, in, 32);
	run("generated block W=1, at 0 mod 8", (k_time_asm<1, 3, 0>), out, in, 32);
	run("generated block W=2 as shipped", (k_time_asm<2, 0, 0>), out, in, 64);
	run("generated block W=2, 4-byte ones unpaired", (k_time_asm<2, 1, 0>), out, in, 64);
	run("generated block W=2, early borrows", (k_time_asm<2, 2, 0>), out, in, 64);
	run("generated block W=2, at 0 mod 8", (k_time_asm<2, 3, 0>), out, in, 64);
	return 0;
}
