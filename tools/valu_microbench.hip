// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the
// integer ops the DP fill kernel is made of, at 1/2/4/8 waves per SIMD, independent (8
// accumulators) and dependent (1 accumulator) chains.  Build: hipcc --offload-arch=gfx950
// -O3 tools/valu_microbench.hip -o build/valu_microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

#define DEF_KERNEL(NAME, ASM_INDEP, ASM_DEP)                                                          \
__global__ void k_##NAME(int *out, int seed, int dep) {                                               \
	extern __shared__ int pad[];                                                                       \
	int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,   \
	    a6 = a0 * 17, a7 = a0 * 19;                                                                    \
	int b = seed * 31 + 6, c = seed | 3;                                                               \
	if (dep) {                                                                                         \
		for (int i = 0; i < ITERS; ++i) {                                                              \
			_Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { asm volatile(ASM_DEP : "+v"(a0) : "v"(b), "v"(c)); } \
		}                                                                                              \
	} else {                                                                                           \
		for (int i = 0; i < ITERS; ++i) {                                                              \
			asm volatile(ASM_INDEP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
		}                                                                                              \
	}                                                                                                  \
	if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x7fffffff) out[0] = pad[0];                          \
}

#define OP8(fmt) fmt("%0") "\n" fmt("%1") "\n" fmt("%2") "\n" fmt("%3") "\n" fmt("%4") "\n" fmt("%5") "\n" fmt("%6") "\n" fmt("%7")

#define F_ADD(r) "v_add_u32 " r ", %8, " r
#define F_AND(r) "v_and_b32 " r ", %8, " r
#define F_MIN3(r) "v_min3_i32 " r ", " r ", %8, %9"
#define F_BFE(r) "v_bfe_u32 " r ", " r ", %8, 6"
#define F_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 3, %8"
#define F_ALIGN(r) "v_alignbit_b32 " r ", " r ", %8, 2"
#define F_FMA(r) "v_fma_f32 " r ", " r ", %8, %9"
#define F_PKADD(r) "v_pk_add_i16 " r ", " r ", %8"
#define F_PKMIN(r) "v_pk_min_i16 " r ", " r ", %8"
#define F_PKFMA(r) "v_pk_add_u16 " r ", " r ", %8"
#define F_DPP(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define F_MIN(r) "v_min_i32 " r ", %8, " r
#define F_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9"
#define F_MAX3(r) "v_max3_i32 " r ", " r ", %8, %9"
#define F_SUBB(r) "v_sub_u32 " r ", " r ", %8"

DEF_KERNEL(add, OP8(F_ADD), "v_add_u32 %0, %1, %0")
DEF_KERNEL(and, OP8(F_AND), "v_and_b32 %0, %1, %0")
DEF_KERNEL(min3, OP8(F_MIN3), "v_min3_i32 %0, %0, %1, %2")
DEF_KERNEL(bfe, OP8(F_BFE), "v_bfe_u32 %0, %0, %1, 6")
DEF_KERNEL(lshladd, OP8(F_LSHLADD), "v_lshl_add_u32 %0, %0, 3, %1")
DEF_KERNEL(alignbit, OP8(F_ALIGN), "v_alignbit_b32 %0, %0, %1, 2")
DEF_KERNEL(fma, OP8(F_FMA), "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(pkadd16, OP8(F_PKADD), "v_pk_add_i16 %0, %0, %1")
DEF_KERNEL(pkmin16, OP8(F_PKMIN), "v_pk_min_i16 %0, %0, %1")
DEF_KERNEL(min, OP8(F_MIN), "v_min_i32 %0, %1, %0")
DEF_KERNEL(add3, OP8(F_ADD3), "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(dpp, OP8(F_DPP), "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
#define F_SDWA(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
#define F_CNDM(r) "v_cndmask_b32_e64 " r ", " r ", %8, s[20:21]"
#define F_CNDV(r) "v_cndmask_b32_e32 " r ", " r ", %8, vcc"
#define F_PERM(r) "v_perm_b32 " r ", " r ", %8, %9"
#define F_LSHL(r) "v_lshlrev_b32 " r ", 1, " r
#define F_LSHR(r) "v_lshrrev_b32 " r ", %8, " r
#define F_OR(r) "v_or_b32 " r ", %8, " r
#define F_XOR(r) "v_xor_b32 " r ", %8, " r
#define F_SUB(r) "v_sub_u32 " r ", " r ", %8"
#define F_MINF(r) "v_min_f32 " r ", %8, " r
#define F_MIN3F(r) "v_min3_f32 " r ", " r ", %8, %9"
#define F_ADDF(r) "v_add_f32 " r ", %8, " r
#define F_CMP(r) "v_cmp_lt_i32 vcc, %8, " r
#define F_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9"
#define F_BFI(r) "v_bfi_b32 " r ", %8, " r ", %9"
#define F_MAD24(r) "v_mad_i32_i24 " r ", " r ", %8, %9"
#define F_MUL24(r) "v_mul_u32_u24 " r ", %8, " r
#define F_MINU16(r) "v_min_u16 " r ", %8, " r
#define F_MOV(r) "v_mov_b32 " r ", %8"
#define F_ADDC(r) "v_addc_co_u32 " r ", vcc, " r ", %8, vcc"
#define F_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 2, %8"
#define F_MED3(r) "v_med3_i32 " r ", " r ", %8, %9"
#define F_ADDSDWAW(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
DEF_KERNEL(add_sdwa, OP8(F_SDWA), "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF_KERNEL(add_sdwaw, OP8(F_ADDSDWAW), "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
DEF_KERNEL(cndmask64, OP8(F_CNDM), "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEF_KERNEL(cndmask32, OP8(F_CNDV), "v_cndmask_b32_e32 %0, %0, %1, vcc")
DEF_KERNEL(perm, OP8(F_PERM), "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(lshl, OP8(F_LSHL), "v_lshlrev_b32 %0, 1, %0")
DEF_KERNEL(lshr, OP8(F_LSHR), "v_lshrrev_b32 %0, %1, %0")
DEF_KERNEL(or, OP8(F_OR), "v_or_b32 %0, %1, %0")
DEF_KERNEL(xor, OP8(F_XOR), "v_xor_b32 %0, %1, %0")
DEF_KERNEL(sub, OP8(F_SUB), "v_sub_u32 %0, %0, %1")
DEF_KERNEL(minf, OP8(F_MINF), "v_min_f32 %0, %1, %0")
DEF_KERNEL(min3f, OP8(F_MIN3F), "v_min3_f32 %0, %0, %1, %2")
DEF_KERNEL(addf, OP8(F_ADDF), "v_add_f32 %0, %1, %0")
DEF_KERNEL(cmp, OP8(F_CMP), "v_cmp_lt_i32 vcc, %1, %0")
DEF_KERNEL(andor, OP8(F_ANDOR), "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(bfi, OP8(F_BFI), "v_bfi_b32 %0, %1, %0, %2")
DEF_KERNEL(mad24, OP8(F_MAD24), "v_mad_i32_i24 %0, %0, %1, %2")
DEF_KERNEL(mul24, OP8(F_MUL24), "v_mul_u32_u24 %0, %1, %0")
DEF_KERNEL(minu16, OP8(F_MINU16), "v_min_u16 %0, %1, %0")
DEF_KERNEL(mov, OP8(F_MOV), "v_mov_b32 %0, %1")
DEF_KERNEL(addc, OP8(F_ADDC), "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEF_KERNEL(lshlor, OP8(F_LSHLOR), "v_lshl_or_b32 %0, %0, 2, %1")
DEF_KERNEL(med3, OP8(F_MED3), "v_med3_i32 %0, %0, %1, %2")
#define F_BITOP3(r) "v_bitop3_b32 " r ", " r ", %8, %9 bitop3:0x96"
#define F_OR3(r) "v_or3_b32 " r ", " r ", %8, %9"
#define F_NOT(r) "v_not_b32 " r ", " r
DEF_KERNEL(bitop3, OP8(F_BITOP3), "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF_KERNEL(or3, OP8(F_OR3), "v_or3_b32 %0, %0, %1, %2")
DEF_KERNEL(not, OP8(F_NOT), "v_not_b32 %0, %0")
/* mixed streams around v_bitop3 (the bit-parallel kernel's mix): does a 4-cycle instruction slow its
 * 2.6-cycle neighbours down? */
#define MIX8(a, b) F_BITOP3("%0") "\n" F_BITOP3("%1") "\n" F_BITOP3("%2") "\n" a("%3") "\n" F_BITOP3("%4") "\n" F_BITOP3("%5") "\n" F_BITOP3("%6") "\n" b("%7")
DEF_KERNEL(mix_b3_add3, MIX8(F_ADD3, F_ADD3), "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF_KERNEL(mix_b3_add, MIX8(F_ADD, F_ADD), "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF_KERNEL(mix_b3_bfe, MIX8(F_BFE, F_PERM), "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF_KERNEL(mix_b3_dpp, MIX8(F_DPP, F_BITOP3), "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF_KERNEL(mix_b3_and, MIX8(F_AND, F_LSHR), "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")

// in-kernel clock under an all-CU integer VALU load: shader cycles (s_memtime) per 100 MHz
// reference tick (s_memrealtime), MI355X_MICROARCH.md 'DVFS give-back' item 6
__global__ void k_clockprobe(unsigned long long *out, int seed, int iters) {
	int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed * 31 + 6, c = seed | 3;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
	for (int i = 0; i < iters; ++i) {
		asm volatile("v_min3_i32 %0, %0, %4, %5\n v_add_u32 %1, %4, %1\n v_bfe_u32 %2, %2, %4, 6\n v_alignbit_b32 %3, %3, %4, 2\n"
		             "v_min3_i32 %0, %0, %4, %5\n v_add_u32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_lshl_add_u32 %3, %3, 3, %4"
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
	if (a0 + a1 + a2 + a3 == 0x7fffffff) out[0] = 0;
}

// the fill kernel's cell recurrence (gain form, R = 2 rows x C = 16 columns per step) with
// registers only: how many issue cycles per 64 cells does the instruction MIX sustain?
__global__ void k_cellmix(int *out, const unsigned *in, int iters) {
	unsigned tab[16]; int leftc[16], hup[16];
	for (int c = 0; c < 16; ++c) { tab[c] = in[threadIdx.x * 16 + c]; leftc[c] = (int)(in[c] & 31) - 40; hup[c] = c * 4; }
	int last0 = 0, last1 = 4, diag_in = 0;
	unsigned acc0 = 0, acc1 = 0, sh0 = (in[3] & 3) * 8, sh1 = (in[5] & 3) * 8;
	for (int t = 0; t < iters; ++t) {
		int cd0 = diag_in, cd1 = last0, cl0 = last0 + t, cl1 = last1 + t;
		diag_in = cl1;
#pragma unroll
		for (int i = 0; i < 17; ++i) {
#pragma unroll
			for (int q = 0; q < 2; ++q) {
				const int c = i - q;
				if (c < 0 || c >= 16) continue;
				int &cd = q ? cd1 : cd0; int &cl = q ? cl1 : cl0; unsigned &acc = q ? acc1 : acc0;
				const int dg = cd + (int)__builtin_amdgcn_ubfe(tab[c], q ? sh1 : sh0, 8);
				const int lf = cl + leftc[c];
				int h = max(max(dg, hup[c]), lf);
				acc = __builtin_amdgcn_alignbit((unsigned)h, acc, 2);
				cd = hup[c];
				h &= ~3;
				hup[c] = h; cl = h;
			}
		}
		last0 = cl0; last1 = cl1;
		sh0 = (sh0 + (acc0 & 8)) & 24; sh1 = (sh1 + (acc1 & 8)) & 24;
	}
	int sum = last0 + last1 + (int)acc0 + (int)acc1;
	for (int c = 0; c < 16; ++c) sum += hup[c];
	out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// does a full-rate op keep its 2-cycle issue when it sits between half-rate ops?
// 12 independent ops per iteration on 12 accumulators, in different orders
#define ORDER_KERNEL(NAME, BODY)                                                                      \
__global__ void k_order_##NAME(int *out, int seed, int iters) {                                       \
	int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,      \
	    a6 = a0 * 17, a7 = a0 * 19, a8 = a0 * 23, a9 = a0 * 29, a10 = a0 * 31, a11 = a0 * 37;           \
	int b = seed * 31 + 6, c = seed | 3;                                                               \
	for (int t = 0; t < iters; ++t) {                                                                  \
		asm volatile(BODY BODY BODY BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),  \
		             "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11) : "v"(b), "v"(c));                  \
	}                                                                                                  \
	if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 == 0x7fffffff) out[0] = a0;        \
}
#define H1(r) "v_max3_i32 " r ", " r ", %12, %13\n"
#define H2(r) "v_bfe_u32 " r ", " r ", %12, 8\n"
#define H3(r) "v_alignbit_b32 " r ", " r ", %12, 2\n"
#define F1(r) "v_add_u32 " r ", %12, " r "\n"
#define F2(r) "v_and_b32 " r ", %12, " r "\n"
ORDER_KERNEL(alt, H2("%0") F1("%1") H1("%2") F1("%3") H3("%4") F2("%5") H2("%6") F1("%7") H1("%8") F1("%9") H3("%10") F2("%11"))
ORDER_KERNEL(grp, H2("%0") H1("%2") H3("%4") H2("%6") H1("%8") H3("%10") F1("%1") F1("%3") F2("%5") F1("%7") F1("%9") F2("%11"))
ORDER_KERNEL(pair, H2("%0") H1("%2") F1("%1") F1("%3") H3("%4") H2("%6") F2("%5") F1("%7") H1("%8") H3("%10") F1("%9") F2("%11"))
ORDER_KERNEL(allf, F1("%0") F1("%1") F2("%2") F1("%3") F1("%4") F2("%5") F1("%6") F1("%7") F2("%8") F1("%9") F1("%10") F2("%11"))
ORDER_KERNEL(allh, H2("%0") H1("%1") H3("%2") H2("%3") H1("%4") H3("%5") H2("%6") H1("%7") H3("%8") H2("%9") H1("%10") H3("%11"))

// packed-16 variant of the cell recurrence: two independent matrices in the lo/hi halves of
// every register (v_perm_b32 gain lookup, v_pk_add_i16 x2, v_pk_max_i16 x2, tag and/accumulate,
// clean) = 8 ops per 2 cells; 2 chains (rows) per wave as in the real kernel
__global__ void k_cellmix16(int *out, const unsigned *in, int iters) {
	unsigned tabA[16], tabB[16], hup[16]; unsigned leftc = 0xfffdfffd;
	for (int c = 0; c < 16; ++c) { tabA[c] = in[threadIdx.x * 16 + c]; tabB[c] = in[threadIdx.x * 16 + c + 7]; hup[c] = c * 4; }
	unsigned last0 = 0, last1 = 4, diag_in = 0, acc0 = 0, acc1 = 0;
	unsigned sel0 = 0x0c020c00u | (in[3] & 1), sel1 = 0x0c030c01u ^ (in[5] & 1);
	for (int t = 0; t < iters; ++t) {
		unsigned cd0 = diag_in, cd1 = last0, cl0 = last0 + t, cl1 = last1 + t;
		diag_in = cl1;
#pragma unroll
		for (int i = 0; i < 17; ++i) {
#pragma unroll
			for (int q = 0; q < 2; ++q) {
				const int c = i - q;
				if (c < 0 || c >= 16) continue;
				unsigned &cd = q ? cd1 : cd0; unsigned &cl = q ? cl1 : cl0; unsigned &acc = q ? acc1 : acc0;
				unsigned g, dg, lf, m, h, tg;
				asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(g) : "v"(tabA[c]), "v"(tabB[c]), "v"(q ? sel1 : sel0));
				asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(dg) : "v"(cd), "v"(g));
				asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(lf) : "v"(cl), "v"(leftc));
				asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(m) : "v"(dg), "v"(hup[c]));
				asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(h) : "v"(m), "v"(lf));
				tg = h & 0x00030003u;
				acc = (acc << 2) + tg;
				cd = hup[c];
				h &= 0xfffcfffcu;
				hup[c] = h; cl = h;
			}
		}
		last0 = cl0; last1 = cl1;
		sel0 ^= (acc0 & 1); sel1 ^= (acc1 & 1);
	}
	unsigned sum = last0 + last1 + acc0 + acc1;
	for (int c = 0; c < 16; ++c) sum += hup[c];
	out[blockIdx.x * blockDim.x + threadIdx.x] = (int)sum;
}

typedef void (*kern_t)(int *, int, int);

static void run(const char *name, kern_t k, int ncu, double ghz_hint)
{
	int *out;
	CHECK(hipMalloc(&out, 64));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int dep = 0; dep <= 1; ++dep) {
		printf("%-9s %s:", name, dep ? "dep  " : "indep");
		for (int wps = 1; wps <= 8; wps *= 2) {
			const int threads = 64 * 4 * wps;                  // wps waves on each of the 4 SIMDs
			if (threads > 1024) {                              // 2 blocks of 1024 per CU
				// 8 waves/SIMD: two 1024-thread blocks per CU, 32 KB LDS each
			}
			const int blocks = (threads > 1024) ? ncu * 2 : ncu;
			const int tpb = (threads > 1024) ? 1024 : threads;
			const size_t lds = (threads > 1024) ? 70 * 1024 : 100 * 1024;   // pins blocks per CU
			hipLaunchKernelGGL(k, dim3(blocks), dim3(tpb), lds, 0, out, 1, dep);   // warm-up
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(k, dim3(blocks), dim3(tpb), lds, 0, out, 1, dep);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			const double instr_per_simd = (double)ITERS * UNROLL * wps;    // wave-instructions per SIMD
			const double ns_per = ms * 1e6 / instr_per_simd;
			printf("  w%d %.3f ns (%.2f cyc@%.1fGHz)", wps, ns_per, ns_per * ghz_hint, ghz_hint);
		}
		printf("\n");
	}
	CHECK(hipFree(out));
}

int main()
{
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int ncu = p.multiProcessorCount;
	const double ghz = p.clockRate / 1e6;
	printf("device %s CUs %d clock %.2f GHz; numbers = time per wave64 instruction per SIMD\n", p.gcnArchName, ncu, ghz);
	{
		// cell-mix: blocks of 64*4*wps threads, one block per CU (LDS-pinned)
		int *o; unsigned *inp;
		CHECK(hipMalloc(&o, 256 * 2048 * sizeof(int)));
		CHECK(hipMalloc(&inp, 2048 * 16 * sizeof(unsigned)));
		CHECK(hipMemset(inp, 0x5a, 2048 * 16 * sizeof(unsigned)));
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		CHECK(hipFuncSetAttribute((const void *)k_cellmix, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
		for (int wps = 1; wps <= 4; wps *= 2) {
			const int iters = 20000;
			const int tpb = 64 * 4 * wps;
			hipLaunchKernelGGL(k_cellmix, dim3(ncu), dim3(tpb), 100 * 1024, 0, o, inp, iters);
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(k_cellmix, dim3(ncu), dim3(tpb), 100 * 1024, 0, o, inp, iters);
			CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			const double cellwaves_per_simd = (double)iters * 32 * wps;    // 32 cells per step per wave
			printf("cellmix (6 ops/cell, 2 chains) %d waves/SIMD: %.2f ns per 64 cells per SIMD = %.1f cyc@2.4GHz -> chip %.2f TCUPS\n",
			       wps, ms * 1e6 / cellwaves_per_simd, ms * 1e6 / cellwaves_per_simd * 2.4,
			       (double)iters * 32 * 64 * wps * 4 * ncu / (ms * 1e-3) / 1e12);
		}
		CHECK(hipFuncSetAttribute((const void *)k_cellmix16, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
		for (int wps = 1; wps <= 4; wps *= 2) {
			const int iters = 20000;
			const int tpb = 64 * 4 * wps;
			hipLaunchKernelGGL(k_cellmix16, dim3(ncu), dim3(tpb), 100 * 1024, 0, o, inp, iters);
			CHECK(hipDeviceSynchronize());
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(k_cellmix16, dim3(ncu), dim3(tpb), 100 * 1024, 0, o, inp, iters);
			CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			const double cellwaves_per_simd = (double)iters * 64 * wps;    // 2 matrices x 32 cells per step per wave
			printf("cellmix16 (packed, 8 ops / 2 cells) %d waves/SIMD: %.2f ns per 64 cells per SIMD = %.1f cyc@2.4GHz -> chip %.2f TCUPS\n",
			       wps, ms * 1e6 / cellwaves_per_simd, ms * 1e6 / cellwaves_per_simd * 2.4,
			       (double)iters * 64 * 64 * wps * 4 * ncu / (ms * 1e-3) / 1e12);
		}
		CHECK(hipFree(o)); CHECK(hipFree(inp));
	}
	{
		int *o; CHECK(hipMalloc(&o, 64));
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		struct { const char *name; void (*k)(int *, int, int); } orders[] = {
			{"alternating H F H F..", k_order_alt}, {"grouped 6H then 6F   ", k_order_grp}, {"pairs HH FF HH FF    ", k_order_pair},
			{"12 full-rate         ", k_order_allf}, {"12 half-rate         ", k_order_allh}};
		for (auto &od : orders) {
			CHECK(hipFuncSetAttribute((const void *)od.k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
			printf("order %s:", od.name);
			for (int wps = 1; wps <= 4; wps *= 2) {
				const int iters = 50000;
				hipLaunchKernelGGL(od.k, dim3(ncu), dim3(64 * 4 * wps), 100 * 1024, 0, o, 1, iters);
				CHECK(hipDeviceSynchronize());
				CHECK(hipEventRecord(e0));
				hipLaunchKernelGGL(od.k, dim3(ncu), dim3(64 * 4 * wps), 100 * 1024, 0, o, 1, iters);
				CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
				float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
				printf("  w%d %.2f cyc/instr", wps, ms * 1e6 / ((double)iters * 48 * wps) * 2.4);
			}
			printf("   (6 half-rate + 6 full-rate per 12; ideal 3.0)\n");
		}
		CHECK(hipFree(o));
	}
	{
		unsigned long long *d, h[2 * 2048];
		CHECK(hipMalloc(&d, sizeof(h)));
		for (int wps = 2; wps <= 8; wps *= 2) {
			const int blocks = ncu * wps / 2;                       // 512-thread blocks: 2 waves per SIMD each
			for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_clockprobe, dim3(blocks), dim3(512), 0, 0, d, 1, 400000);
			CHECK(hipDeviceSynchronize());
			CHECK(hipMemcpy(h, d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
			double lo = 1e9, hi = 0, sum = 0;
			for (int b = 0; b < blocks; ++b) {
				const double g = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;   // GHz
				lo = g < lo ? g : lo; hi = g > hi ? g : hi; sum += g;
			}
			printf("clockprobe int-VALU mix, %d waves/SIMD: in-kernel clock mean %.3f GHz (min %.3f max %.3f), %.1f ms per launch\n",
			       wps, sum / blocks, lo, hi, (double)h[1] / 1e5);
		}
		CHECK(hipFree(d));
	}
	CHECK(hipFuncSetAttribute((const void *)k_add, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
#define RUN(n) CHECK(hipFuncSetAttribute((const void *)k_##n, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)); run(#n, k_##n, ncu, ghz)
	RUN(fma); RUN(add); RUN(and); RUN(min); RUN(min3); RUN(add3); RUN(bfe); RUN(lshladd); RUN(alignbit);
	RUN(pkadd16); RUN(pkmin16); RUN(dpp);
	RUN(add_sdwa); RUN(add_sdwaw); RUN(cndmask64); RUN(cndmask32); RUN(perm); RUN(lshl); RUN(lshr); RUN(or); RUN(xor);
	RUN(sub); RUN(minf); RUN(min3f); RUN(addf); RUN(cmp); RUN(andor); RUN(bfi); RUN(mad24); RUN(mul24); RUN(minu16);
	RUN(mov); RUN(addc); RUN(lshlor); RUN(med3); RUN(bitop3); RUN(or3); RUN(not); RUN(mix_b3_add3); RUN(mix_b3_add); RUN(mix_b3_bfe); RUN(mix_b3_dpp); RUN(mix_b3_and);
	return 0;
}
