/*
 * accfree_probe.hip -- TEST / MEASUREMENT ONLY.  Can the W = 2 step of nw_fill_bits do without its three accumulator instructions
 * (v_addc_co_u32 acc, acc, acc: half rate, 13 of the step's 158 priced cycles, there only so that the traceback can restart a replay
 * at any lane)?  The carry a chain leaves is a lane mask; the last addc of the chain can write it to an SGPR pair instead of VCC
 * (VOP3 form) and a scalar store can put it in memory: no vector instruction at all.
 *   mode 0  the shipped step (accumulators in VGPRs)
 *   mode 1  no record of the carries at all (the floor)
 *   mode 2  carry masks to SGPR pairs, one s_store_dwordx2 per plane and step, s_dcache_wb at the end
 *   mode 3  mode 2 + lane 63's bit shifted into a scalar accumulator per plane (what the next strip's hand-off needs)
 * Part 1 checks on the device that mode 2's stored masks are the shipped step's accumulators, bit for bit.
 *   hipcc --offload-arch=gfx950 -O3 -o accfree_probe tools/r04/accfree_probe.hip && ./accfree_probe
 */
#define main subco_probe_main
#include "../subco_probe.hip"
#undef main

template <int MODE>
__device__ __forceinline__ void step2(StN<2> &S, const uint32_t (&D)[2], const uint32_t (&E)[2], uint32_t pre0, uint32_t pre1, uint32_t z2, uint32_t z1, uint32_t z0,
                                      unsigned long long *mem, int t, uint32_t (&sacc)[3])
{
	constexpr int W = 2;
	asm("v_xor_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pre0) : "v"(S.x0), "v"(D[0]));
	asm("v_xor_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(pre1) : "v"(S.x1), "v"(D[1]));
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
	unsigned long long m2 = 0, m1 = 0, m0 = 0;
#define CHAIN(sA, sB, acc, nO, z, a0_, b0_, a1_, b1_, mask)                                                                                      \
	if (MODE == 0)                                                                                                                                \
		asm("v_sub_co_u32_dpp %0, vcc, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %6, %7, vcc\n\t"        \
		    "v_addc_co_u32 %2, vcc, %8, %9, vcc\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc"                                                           \
		    : "=&v"(junk), "=&v"(sA), "=&v"(sB), "+v"(acc) : "v"(nO), "v"(z), "v"(a0_), "v"(b0_), "v"(a1_), "v"(b1_) : "vcc");                  \
	else if (MODE == 1)                                                                                                                           \
		asm("v_sub_co_u32_dpp %0, vcc, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %5, %6, vcc\n\t"        \
		    "v_addc_co_u32 %2, vcc, %7, %8, vcc"                                                                                                  \
		    : "=&v"(junk), "=&v"(sA), "=&v"(sB) : "v"(nO), "v"(z), "v"(a0_), "v"(b0_), "v"(a1_), "v"(b1_) : "vcc");                             \
	else                                                                                                                                          \
		asm("v_sub_co_u32_dpp %0, vcc, %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_addc_co_u32 %1, vcc, %6, %7, vcc\n\t"        \
		    "v_addc_co_u32_e64 %2, %3, %8, %9, vcc"                                                                                               \
		    : "=&v"(junk), "=&v"(sA), "=&v"(sB), "=&s"(mask) : "v"(nO), "v"(z), "v"(a0_), "v"(b0_), "v"(a1_), "v"(b1_) : "vcc")
	CHAIN(s2[0], s2[1], S.acc2, S.nO2, z2, S.nH0[0], g2[0], S.nH0[1], g2[1], m2);
#pragma unroll
	for (int h = 0; h < W; ++h) {
		G2[h] = BITOP3(s2[h], S.nH0[h], g2[h], LA ^ LB ^ LC);
		if (h == W - 1) S.nO2 = BITOP3(g2[h], S.nH0[h], G2[h], ~(LA | (LB & LC)));
		const uint32_t t1 = BITOP3(nE[h], S.nH0[h], G2[h], ~LA | (~LB & LC));
		g1[h] = BITOP3(t1, S.H1[h], S.H1[h], LA & ~LB);
		A1[h] = BITOP3(g1[h], nE[h], S.nH0[h], LA | (LB & LC));
	}
	CHAIN(s1[0], s1[1], S.acc1, S.nO1, z1, A1[0], g1[0], A1[1], g1[1], m1);
#pragma unroll
	for (int h = 0; h < W; ++h) {
		G1[h] = BITOP3(s1[h], A1[h], g1[h], LA ^ LB ^ LC);
		if (h == W - 1) S.nO1 = BITOP3(g1[h], A1[h], G1[h], ~(LA | (LB & LC)));
		const uint32_t v = BITOP3(S.H1[h], G2[h], G1[h], (LA & LB) | (~LA & LC));
		const uint32_t w = BITOP3(nE[h], v, S.H2[h], ~LC & (~LA | LB));
		O0[h] = BITOP3(w, nE[h], S.nH0[h], LA | (LB & LC));
	}
	const uint32_t nO0last = BITOP3(O0[W - 1], O0[W - 1], O0[W - 1], ~LA);
	CHAIN(G0[0], G0[1], S.acc0, S.nO0, z0, O0[0], O0[0], O0[1], O0[1], m0);
#undef CHAIN
	S.nO0 = nO0last;
	if (MODE >= 2) {
		/* one 8-byte scalar store per plane: [step][plane] */
		asm volatile("s_store_dwordx2 %0, %3, %4\n\ts_store_dwordx2 %1, %3, %5\n\ts_store_dwordx2 %2, %3, %6"
		             :: "s"(m2), "s"(m1), "s"(m0), "s"(mem), "n"(0), "n"(8), "n"(16), "s"(t) : "memory");
	}
	if (MODE == 3) {
		sacc[0] = (sacc[0] << 1) + (uint32_t)(m2 >> 63);
		sacc[1] = (sacc[1] << 1) + (uint32_t)(m1 >> 63);
		sacc[2] = (sacc[2] << 1) + (uint32_t)(m0 >> 63);
	}
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

/* timing: as k_time_new<2, 1> (the first lane's inputs from LDS every step) */
template <int MODE>
__global__ void k_time_acc(uint32_t *out, const uint32_t *in, int nblocks)
{
	__shared__ __attribute__((aligned(16))) uint32_t inj[16][32 * 8];
	__shared__ __attribute__((aligned(16))) uint32_t konst[32 * 8];
	const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 15;
	for (int i = threadIdx.x; i < 32 * 8; i += blockDim.x) konst[i] = 0x80000000u;
	for (int i = lane; i < 32 * 8; i += 64) inj[wv][i] = in[(i * 7) & 1023] & 1u;
	__syncthreads();
	const uint32_t *src = lane == 0 ? &inj[wv][0] : &konst[0];
	StN<2> S;
	for (int h = 0; h < 2; ++h) { S.nH0[h] = in[lane + 64 * h]; S.H1[h] = in[128 + lane + 64 * h]; S.H2[h] = in[256 + lane + 64 * h]; }
	S.x0 = in[384 + lane]; S.x1 = in[448 + lane];
	S.nO2 = in[512 + lane]; S.nO1 = in[576 + lane]; S.nO0 = in[640 + lane];
	S.acc2 = S.acc1 = S.acc0 = 0;
	const uint32_t D[2] = {in[704 + lane], in[768 + lane]}, E[2] = {in[832 + lane], in[896 + lane]};
	uint32_t sink = 0, sacc[3] = {0, 0, 0};
	/* per wave 32 steps x 3 planes x 8 bytes, well away from the result words */
	const int gw = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	unsigned long long *mem = reinterpret_cast<unsigned long long *>(out + (1 << 20)) + (size_t)gw * 96;
	for (int b = 0; b < nblocks; ++b) {
		uint4 nx = *reinterpret_cast<const uint4 *>(src);
		uint32_t nx4 = src[4];
#pragma unroll
		for (int t = 0; t < 32; ++t) {
			const uint4 c = nx;
			const uint32_t c4 = nx4;
			nx = *reinterpret_cast<const uint4 *>(src + ((t + 1) & 31) * 8);
			nx4 = src[((t + 1) & 31) * 8 + 4];
			step2<MODE>(S, D, E, c.x, c.y, c.z, c.w, c4, mem + t * 3, t, sacc);
		}
		sink ^= S.acc2 ^ S.acc1 ^ S.acc0 ^ sacc[0] ^ sacc[1] ^ sacc[2];
	}
	if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
	uint32_t r = sink ^ S.x0 ^ S.x1 ^ S.nO2 ^ S.nO1 ^ S.nO0;
	for (int h = 0; h < 2; ++h) r ^= S.nH0[h] ^ S.H1[h] ^ S.H2[h];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

/* correctness: one wave, 64 steps; mode 0's accumulators against mode 2's stored masks */
template <int MODE>
__global__ void k_check(uint32_t *out, const uint32_t *in)
{
	const int lane = threadIdx.x;
	StN<2> S;
	for (int h = 0; h < 2; ++h) { S.nH0[h] = in[lane + 64 * h]; S.H1[h] = in[128 + lane + 64 * h] & S.nH0[h] ? 0 : 0; S.H2[h] = 0; }
	for (int h = 0; h < 2; ++h) { S.nH0[h] = ~0u; }
	S.x0 = S.x1 = 0;
	S.nO2 = S.nO1 = S.nO0 = 0x80000000u;
	S.acc2 = S.acc1 = S.acc0 = 0;
	const uint32_t D[2] = {in[704 + lane], in[768 + lane]}, E[2] = {in[832 + lane], in[896 + lane]};
	uint32_t sacc[3] = {0, 0, 0};
	unsigned long long *mem = reinterpret_cast<unsigned long long *>(out + 4096);
	for (int b = 0; b < 2; ++b) {
#pragma unroll
		for (int t = 0; t < 32; ++t) {
			const uint32_t u = in[1024 + b * 32 + t];
			const uint32_t zc = lane == 0 ? 0u : 0x80000000u;
			step2<MODE>(S, D, E, in[lane] ^ (0u - (u & 1u)), in[64 + lane] ^ (0u - ((u >> 1) & 1u)), lane == 0 ? (u >> 2) & 1 : zc, lane == 0 ? (u >> 3) & 1 : zc,
			            lane == 0 ? (u >> 4) & 1 : zc, mem + (b * 32 + t) * 3, t, sacc);
		}
		out[b * 192 + lane * 3 + 0] = S.acc2;
		out[b * 192 + lane * 3 + 1] = S.acc1;
		out[b * 192 + lane * 3 + 2] = S.acc0;
	}
	if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
	if (MODE == 3 && lane == 0) { out[3000] = sacc[0]; out[3001] = sacc[1]; out[3002] = sacc[2]; }
}

int main()
{
	uint32_t *in, *out;
	CHECK(hipMalloc(&in, 8192 * 4));
	const size_t out_words = (1u << 20) + (size_t)8192 * 96 * 2 + 4096;
	CHECK(hipMalloc(&out, out_words * 4));
	static uint32_t h[8192];
	uint32_t x = 4711;
	for (int i = 0; i < 8192; ++i) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
	for (int t = 0; t < 64; ++t) h[1024 + t] &= 31u;
	CHECK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));

	/* part 1: mode 2 / 3 record what mode 0 accumulates */
	static uint32_t a0[8192], a2[8192 + 1024];
	CHECK(hipMemset(out, 0, 64 * 1024));
	hipLaunchKernelGGL(k_check<0>, dim3(1), dim3(64), 0, 0, out, in);
	CHECK(hipDeviceSynchronize());
	CHECK(hipMemcpy(a0, out, 4096 * 4, hipMemcpyDeviceToHost));
	for (int mode = 2; mode <= 3; ++mode) {
		CHECK(hipMemset(out, 0, 64 * 1024));
		if (mode == 2) hipLaunchKernelGGL(k_check<2>, dim3(1), dim3(64), 0, 0, out, in);
		else hipLaunchKernelGGL(k_check<3>, dim3(1), dim3(64), 0, 0, out, in);
		CHECK(hipDeviceSynchronize());
		CHECK(hipMemcpy(a2, out, (4096 + 64 * 3 * 2) * 4, hipMemcpyDeviceToHost));
		const unsigned long long *masks = reinterpret_cast<const unsigned long long *>(a2 + 4096);
		int bad = 0;
		for (int b = 0; b < 2; ++b)
			for (int t = 0; t < 32; ++t)
				for (int p = 0; p < 3; ++p)
					for (int l = 0; l < 64; ++l) {
						const unsigned want = (a0[b * 192 + l * 3 + p] >> (31 - t)) & 1u;      /* first step of a block in bit 31 */
						const unsigned got = (unsigned)((masks[(b * 32 + t) * 3 + p] >> l) & 1ull);
						if (want != got) { if (bad < 4) printf("  mode %d: block %d step %d plane %d lane %d: stored %u accumulated %u\n", mode, b, t, p, l, got, want); ++bad; }
					}
		if (mode == 3)
			for (int p = 0; p < 3; ++p)
				if (a2[3000 + p] != a0[192 + 63 * 3 + p]) { printf("  mode 3: scalar accumulator of plane %d %08x, lane 63's %08x\n", p, a2[3000 + p], a0[192 + 63 * 3 + p]); ++bad; }
		printf("mode %d: carry masks through SGPRs and scalar stores vs the accumulators of the shipped step: %s (%d mismatches)\n", mode, bad ? "DIFFERENT" : "identical", bad);
	}

	/* part 2: timing */
	printf("cycles per wave-step per SIMD at a nominal 2.4 GHz (cycles per cell), W = 2, first lane's inputs from LDS\n");
	run("mode 0: accumulators (shipped)", k_time_acc<0>, out, in, 64);
	run("mode 1: no record of the carries", k_time_acc<1>, out, in, 64);
	run("mode 2: SGPR masks + 3 s_store_dwordx2", k_time_acc<2>, out, in, 64);
	run("mode 3: mode 2 + lane 63 into scalars", k_time_acc<3>, out, in, 64);
	run("mode 0 again", k_time_acc<0>, out, in, 64);
	return 0;
}
