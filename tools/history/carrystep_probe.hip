// Prices the bit-parallel step with the carries kept in SGPR lane masks (v_addc_co_u32 with an SGPR-pair carry
// in / carry out, the shift to the right neighbour done by the scalar unit) against the round-2 step (carries
// in a hand-off word: DPP move, 4 v_bfe, 2 v_perm, v_add3).  The question it answers: does the scalar work
// (up to 18 SALU instructions per step) hide behind the 22 VALU instructions when several waves share a SIMD?
//   MODE 0: VALU part only (masks held constant)          MODE 1: + scalar shifts/injects written in C++
//   MODE 2: + scalar part as s_bitcmp1 / s_addc chains     MODE 3: the round-2 step (31 VALU) for reference
// Build: hipcc --offload-arch=gfx950 -O3 tools/carrystep_probe.hip -o build/carrystep_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr uint32_t LA = 0xF0, LB = 0xCC, LC = 0xAA;
#define BITOP3(a, b, c, expr) ((uint32_t)__builtin_amdgcn_bitop3_b32((a), (b), (c), (unsigned char)((expr) & 0xff)))

__device__ __forceinline__ uint32_t addc(uint32_t a, uint32_t b, uint64_t cin, uint64_t &cout)
{
	uint32_t s;
	asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(s), "=s"(cout) : "v"(a), "v"(b), "s"(cin));
	return s;
}
__device__ __forceinline__ uint32_t sel(uint32_t f, uint32_t t, uint64_t m)
{
	uint32_t d;
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(f), "v"(t), "s"(m));
	return d;
}

struct St {
	uint32_t nH0, H1, H2;
};

__device__ __forceinline__ void valu_step(St &S, uint32_t N0, uint32_t N1, uint32_t N2, uint32_t N3, uint64_t m0, uint64_t m1, uint64_t ci2,
                                          uint64_t ci1, uint64_t ci0, uint64_t &co2, uint64_t &co1, uint64_t &co0)
{
	const uint32_t nH0 = S.nH0, H1 = S.H1, H2 = S.H2;
	const uint32_t ta = sel(N0, N1, m0), tb = sel(N2, N3, m0);
	const uint32_t nE = sel(ta, tb, m1);
	const uint32_t g2 = BITOP3(nE, nH0, nH0, ~LA & LB);
	const uint32_t s2 = addc(nH0, g2, ci2, co2);
	const uint32_t G2 = BITOP3(s2, nH0, g2, LA ^ LB ^ LC);
	const uint32_t t1 = BITOP3(nE, nH0, G2, ~LA | (~LB & LC));
	const uint32_t g1 = BITOP3(t1, H1, H1, LA & ~LB);
	const uint32_t A1 = BITOP3(g1, nE, nH0, LA | (LB & LC));
	const uint32_t s1 = addc(A1, g1, ci1, co1);
	const uint32_t G1 = BITOP3(s1, A1, g1, LA ^ LB ^ LC);
	const uint32_t v = BITOP3(H1, G2, G1, (LA & LB) | (~LA & LC));
	const uint32_t w = BITOP3(nE, v, H2, ~LC & (~LA | LB));
	const uint32_t O0 = BITOP3(w, nE, nH0, LA | (LB & LC));
	const uint32_t G0 = addc(O0, O0, ci0, co0);
	const uint32_t C1 = BITOP3(nE, G2, H2, ~LA | LB | LC);
	const uint32_t C0 = BITOP3(nE, G1, H1, ~LA | LB | LC);
	S.H2 = BITOP3(C1, G0, G0, LA & ~LB);
	const uint32_t a1 = BITOP3(C1, G1, G1, LA & ~LB);
	S.H1 = BITOP3(G0, a1, C0, (LA & LB) | (~LA & LC));
	const uint32_t b0 = BITOP3(C0, G1, G0, LC & (~LA | LB));
	S.nH0 = BITOP3(b0, C1, G2, LA & (~LB | LC));
}

template <int MODE>
__global__ void k_new(uint32_t *out, const uint32_t *in, int nblocks)
{
	const int lane = threadIdx.x & 63;
	St S{in[lane], in[64 + lane], in[128 + lane]};
	const uint32_t N0 = in[192 + lane], N1 = in[256 + lane], N2 = in[320 + lane], N3 = in[384 + lane];
	uint64_t co2 = 0, co1 = 0, co0 = 0, m0 = 0, m1 = 0;
	uint32_t acc2 = 0, acc1 = 0, acc0 = 0, sink = 0;
	const uint32_t *uni = in + 512;
	for (int b = 0; b < nblocks; ++b) {
		const uint32_t I2 = __builtin_amdgcn_readfirstlane(uni[(b & 15) * 5 + 0]);
		const uint32_t I1 = __builtin_amdgcn_readfirstlane(uni[(b & 15) * 5 + 1]);
		const uint32_t I0 = __builtin_amdgcn_readfirstlane(uni[(b & 15) * 5 + 2]);
		const uint32_t a0 = __builtin_amdgcn_readfirstlane(uni[(b & 15) * 5 + 3]);
		const uint32_t a1 = __builtin_amdgcn_readfirstlane(uni[(b & 15) * 5 + 4]);
		if (MODE == 0) {
			m0 = ((uint64_t)a0 << 32) | I0;
			m1 = ((uint64_t)a1 << 32) | I1;
		}
#pragma unroll
		for (int t = 0; t < 32; ++t) {
			uint64_t ci2, ci1, ci0;
			if (MODE == 0) {
				ci2 = co2;
				ci1 = co1;
				ci0 = co0;
			} else if (MODE == 1) {
				acc2 = (acc2 << 1) | (uint32_t)(co2 >> 63);
				acc1 = (acc1 << 1) | (uint32_t)(co1 >> 63);
				acc0 = (acc0 << 1) | (uint32_t)(co0 >> 63);
				ci2 = (co2 << 1) | ((I2 >> (31 - t)) & 1u);
				ci1 = (co1 << 1) | ((I1 >> (31 - t)) & 1u);
				ci0 = (co0 << 1) | ((I0 >> (31 - t)) & 1u);
				m0 = (m0 << 1) | ((a0 >> t) & 1u);
				m1 = (m1 << 1) | ((a1 >> t) & 1u);
			} else {
				/* SCC chains: inject bit -> SCC, lo = 2 lo + SCC, hi = 2 hi + carry, acc = 2 acc + bit 63 */
				uint32_t l2 = (uint32_t)co2, h2 = (uint32_t)(co2 >> 32), l1 = (uint32_t)co1, h1 = (uint32_t)(co1 >> 32), l0 = (uint32_t)co0,
				         h0 = (uint32_t)(co0 >> 32);
				uint32_t ml0 = (uint32_t)m0, mh0 = (uint32_t)(m0 >> 32), ml1 = (uint32_t)m1, mh1 = (uint32_t)(m1 >> 32);
				asm volatile(
				    "s_bitcmp1_b32 %[I2], %[bi]\n s_addc_u32 %[l2], %[l2], %[l2]\n s_addc_u32 %[h2], %[h2], %[h2]\n s_addc_u32 %[c2], %[c2], %[c2]\n"
				    "s_bitcmp1_b32 %[I1], %[bi]\n s_addc_u32 %[l1], %[l1], %[l1]\n s_addc_u32 %[h1], %[h1], %[h1]\n s_addc_u32 %[c1], %[c1], %[c1]\n"
				    "s_bitcmp1_b32 %[I0], %[bi]\n s_addc_u32 %[l0], %[l0], %[l0]\n s_addc_u32 %[h0], %[h0], %[h0]\n s_addc_u32 %[c0], %[c0], %[c0]\n"
				    "s_bitcmp1_b32 %[a0], %[bt]\n s_addc_u32 %[ml0], %[ml0], %[ml0]\n s_addc_u32 %[mh0], %[mh0], %[mh0]\n"
				    "s_bitcmp1_b32 %[a1], %[bt]\n s_addc_u32 %[ml1], %[ml1], %[ml1]\n s_addc_u32 %[mh1], %[mh1], %[mh1]\n"
				    : [l2] "+s"(l2), [h2] "+s"(h2), [l1] "+s"(l1), [h1] "+s"(h1), [l0] "+s"(l0), [h0] "+s"(h0), [c2] "+s"(acc2), [c1] "+s"(acc1),
				      [c0] "+s"(acc0), [ml0] "+s"(ml0), [mh0] "+s"(mh0), [ml1] "+s"(ml1), [mh1] "+s"(mh1)
				    : [I2] "s"(I2), [I1] "s"(I1), [I0] "s"(I0), [a0] "s"(a0), [a1] "s"(a1), [bi] "n"(31 - t), [bt] "n"(t)
				    : "scc");
				ci2 = ((uint64_t)h2 << 32) | l2;
				ci1 = ((uint64_t)h1 << 32) | l1;
				ci0 = ((uint64_t)h0 << 32) | l0;
				m0 = ((uint64_t)mh0 << 32) | ml0;
				m1 = ((uint64_t)mh1 << 32) | ml1;
			}
			valu_step(S, N0, N1, N2, N3, m0, m1, ci2, ci1, ci0, co2, co1, co0);
		}
		sink ^= acc2 ^ acc1 ^ acc0;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = S.nH0 ^ S.H1 ^ S.H2 ^ sink ^ (uint32_t)co2 ^ (uint32_t)co1 ^ (uint32_t)co0;
}

/* the round-2 step, as in csadp_bits.hip; LDSMODE 0: without the LDS traffic, 1: one ds_write_b32 of the hand-off word and one
 * broadcast ds_read_b32 of the next inject word per step (as shipped), 2: one ds_write_b128 and one ds_read_b128 per four steps */
template <int LDSMODE, int PHASE = -1>
__global__ void k_old(uint32_t *out, const uint32_t *in, int nblocks)
{
	__shared__ __attribute__((aligned(16))) uint32_t lbuf[16][64 * 4 + 64];
	const int lane = threadIdx.x & 63;
	uint32_t *mine = &lbuf[(threadIdx.x >> 6) & 15][LDSMODE == 2 ? lane * 4 : lane];
	const uint32_t *inj = &lbuf[(threadIdx.x >> 6) & 15][0];
	uint32_t pp4[4] = {0, 0, 0, 0};
	uint4 in4 = make_uint4(0, 0, 0, 0);
	St S{in[lane], in[64 + lane], in[128 + lane]};
	const uint32_t B0 = in[192 + lane], B1 = in[256 + lane];
	uint32_t PP = in[320 + lane];
	const uint32_t *uni = in + 512;
	for (int b = 0; b < nblocks; ++b) {
		uint32_t cur = uni[(b & 15) * 5];
		if (PHASE == 1) asm volatile("s_nop 0");
		if (PHASE == 2) asm volatile("s_nop 0\n\ts_nop 0");
#pragma unroll
		for (int t = 0; t < 32; ++t) {
			uint32_t inw = cur;
			asm("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(inw) : "v"(PP));
			const uint32_t R0 = (uint32_t)__builtin_amdgcn_sbfe((int)inw, 0, 1);
			const uint32_t R1 = (uint32_t)__builtin_amdgcn_sbfe((int)inw, 1, 1);
			const uint32_t c2 = __builtin_amdgcn_ubfe(inw, 15, 1);
			const uint32_t c1 = __builtin_amdgcn_ubfe(inw, 23, 1);
			const uint32_t nH0 = S.nH0, H1 = S.H1, H2 = S.H2;
			uint32_t x0;
			if (PHASE >= 0) asm("v_xor_b32_e64 %0, %1, %2" : "=v"(x0) : "v"(B0), "v"(R0));   /* 8 bytes, like everything else in the step */
			else x0 = B0 ^ R0;
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
			PP = BITOP3(pq, inw, 0xffu, LA | (LB & LC));
			if (LDSMODE == 1) {
				mine[t] = PP;
				cur ^= inj[(t + 1) & 31];
			} else if (LDSMODE == 2) {
				pp4[t & 3] = PP;
				if ((t & 3) == 3) {
					*reinterpret_cast<uint4 *>(mine) = make_uint4(pp4[0], pp4[1], pp4[2], pp4[3]);
					in4 = *reinterpret_cast<const uint4 *>(inj + ((t + 1) & 28));
				}
				cur ^= (t & 3) == 0 ? in4.x : (t & 3) == 1 ? in4.y : (t & 3) == 2 ? in4.z : in4.w;
			}
			const uint32_t C1 = BITOP3(nE, G2, H2, ~LA | LB | LC);
			const uint32_t C0 = BITOP3(nE, G1, H1, ~LA | LB | LC);
			S.H2 = BITOP3(C1, G0, G0, LA & ~LB);
			const uint32_t a1 = BITOP3(C1, G1, G1, LA & ~LB);
			S.H1 = BITOP3(G0, a1, C0, (LA & LB) | (~LA & LC));
			const uint32_t b0 = BITOP3(C0, G1, G0, LC & (~LA | LB));
			S.nH0 = BITOP3(b0, C1, G2, LA & (~LB | LC));
			if (PHASE >= 0) asm("v_mad_u32_u24 %0, %0, 5, 1" : "+v"(cur)); else cur = cur * 5 + 1;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = S.nH0 ^ S.H1 ^ S.H2 ^ PP;
}

template <typename K>
static void run(const char *name, K kernel, uint32_t *out, const uint32_t *in)
{
	const int nblocks = 512;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	/* 1, 2, 4, 8 waves per SIMD on every compute unit; then 2 waves per SIMD on 128 and on 64 workgroups only (half / a
	 * quarter of the chip busy: does a workgroup run faster when its neighbours are idle?) */
	const int shapes[6][2] = {{256, 256}, {256, 512}, {256, 1024}, {512, 1024}, {128, 512}, {64, 512}};
	printf("%-34s", name);
	for (int s = 0; s < 6; ++s) {
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
		const int wps = s < 4 ? 1 << s : 2;
		printf("  %s%d %6.1f", s < 4 ? "w" : (s == 4 ? "half-chip w" : "quarter-chip w"), wps, best * 1e-3 * 2.4e9 / (nblocks * 32.0) / wps);
	}
	printf("\n");
}

int main()
{
	uint32_t *in, *out;
	CHECK(hipMalloc(&in, 4096 * 4));
	CHECK(hipMalloc(&out, 512 * 1024 * 4));
	uint32_t h[4096];
	uint32_t x = 12345;
	for (int i = 0; i < 4096; ++i) {
		x = x * 1664525u + 1013904223u;
		h[i] = x;
	}
	CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
	run("round-2 step (31 VALU)", k_old<0>, out, in);
	run(" + ds_write_b32, ds_read_b32 / step", k_old<1>, out, in);
	run(" + b128 write and read / 4 steps", k_old<2>, out, in);
	run("all-8-byte step, block at +0", (k_old<0, 0>), out, in);
	run("all-8-byte step, block at +4", (k_old<0, 1>), out, in);
	run("all-8-byte step, block at +8", (k_old<0, 2>), out, in);
	run("carry masks, VALU only (22)", k_new<0>, out, in);
	run("carry masks + scalar part in C++", k_new<1>, out, in);
	run("carry masks + SCC chains (18 SALU)", k_new<2>, out, in);
	return 0;
}
