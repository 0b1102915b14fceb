// What one step of nw_fill_cells costs on a LONE wave (one wave per SIMD is how that kernel runs): variants of
// the step with and without its LDS hand-off store, hipEvent-timed over many steps.
// Build: hipcc --offload-arch=gfx950 -O3 tools/cellstep_microbench.hip -o build/cellstep_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 1 << 16;      // x 4 steps each

#define VALU7                                                                     \
	"v_add_u32_dpp %[lf], %[outv], %[leftc] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_mov_b32_dpp %[shn], %[sh] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"        \
	"v_max3_i32 %[h], %[dg], %[outv], %[lf]\n\t"                                   \
	"v_bfe_u32 %[g], %[tab], %[shn], 8\n\t"                                        \
	"v_and_b32 %[outv], -4, %[h]\n\t"                                              \
	"v_alignbit_b32 %[acc], %[h], %[acc], 2\n\t"                                   \
	"v_add_u32 %[dg], %[lf], %[g]\n\t"
#define SDWA6                                                                     \
	"v_add_u32_dpp %[lf], %[outv], %[leftc] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_lshrrev_b32_sdwa %[g], %[sh], %[tab] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
	"v_max3_i32 %[h], %[dg], %[outv], %[lf]\n\t"                                   \
	"v_and_b32 %[outv], -4, %[h]\n\t"                                              \
	"v_alignbit_b32 %[acc], %[h], %[acc], 2\n\t"                                   \
	"v_add_u32_sdwa %[dg], %[g], %[lf] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\t"
#define CHAIN3                                                                    \
	"v_add_u32_dpp %[lf], %[outv], %[leftc] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_max3_i32 %[h], %[dg], %[outv], %[lf]\n\t"                                   \
	"v_and_b32 %[outv], -4, %[h]\n\t"
#define VALU9 VALU7 "v_bfe_u32 %[shn], %[tab], 8, 8\n\t" "v_add_u32 %[lf], %[lf], %[leftc]\n\t"
#define RB128 "ds_read_b128 %[quad], %[addr]\n\t"
#define NOLDS ""
#define W2B32 "ds_write2_b32 %[addr], %[outv], %[sh] offset0:0 offset1:1\n\t"
#define WB32 "ds_write_b32 %[addr], %[outv]\n\t"
#define WB128 "ds_write_b128 %[addr], %[quad]\n\t"

#define OPERANDS                                                                                                   \
	: [lf] "+v"(lf), [shn] "+v"(shn), [outv] "+v"(outv), [acc] "+v"(acc), [dg] "+v"(dg), [h] "=&v"(h), [g] "=&v"(g) \
	  , [quad] "+v"(quad)                                                                                           \
	: [sh] "v"(sh), [tab] "v"(tab), [leftc] "v"(leftc), [addr] "v"(addr)                                            \
	: "memory"

#define DEF(NAME, STEP, S0, S1, S2, S3)                                                          \
__global__ void k_##NAME(int *out, int seed, int onelane) {                                      \
	__shared__ int lds[4096];                                                                     \
	int lf = seed, outv = seed * 3 + threadIdx.x, dg = seed * 5, h, g;                            \
	unsigned shn = threadIdx.x & 24, sh = (threadIdx.x * 8) & 24, acc = 0, tab = 0x0a020a12u;     \
	int leftc = -7;                                                                               \
	unsigned addr = (unsigned)(size_t)&lds[0] + threadIdx.x * 16;                                 \
	u4 quad = {(unsigned)seed, (unsigned)seed + 1, (unsigned)seed + 2, (unsigned)seed + 3};       \
	if (onelane && threadIdx.x != 63) addr = (unsigned)(size_t)&lds[0];                           \
	for (int i = 0; i < ITERS; ++i) {                                                             \
		asm volatile(STEP S0 STEP S1 STEP S2 STEP S3 OPERANDS);                                    \
	}                                                                                             \
	if (lf + outv + dg + (int)acc + (int)shn + (int)quad.x == 0x7fffffff) out[0] = lds[seed & 4095]; \
}

DEF(valu7, VALU7, NOLDS, NOLDS, NOLDS, NOLDS)
DEF(valu7_w2b32, VALU7, W2B32, W2B32, W2B32, W2B32)
DEF(valu7_wb32, VALU7, WB32, WB32, WB32, WB32)
DEF(valu7_wb128_4, VALU7, NOLDS, NOLDS, NOLDS, WB128)
DEF(valu7_w2b32_2, VALU7, NOLDS, W2B32, NOLDS, W2B32)
DEF(sdwa6, SDWA6, NOLDS, NOLDS, NOLDS, NOLDS)
DEF(sdwa6_wb128_4, SDWA6, NOLDS, NOLDS, NOLDS, WB128)
DEF(sdwa6_wb32, SDWA6, WB32, WB32, WB32, WB32)
DEF(valu9_d2, VALU9, NOLDS, NOLDS, NOLDS, WB128 RB128)
DEF(valu9, VALU9, NOLDS, NOLDS, NOLDS, NOLDS)
DEF(chain3, CHAIN3, NOLDS, NOLDS, NOLDS, NOLDS)
DEF(chain3_wb32, CHAIN3, WB32, WB32, WB32, WB32)

// the same store issued by ONE active lane (EXEC = lane 63 only around the store)
#define W2B32_X "s_mov_b64 exec, %[one]\n\tds_write2_b32 %[addr], %[outv], %[sh] offset0:0 offset1:1\n\ts_mov_b64 exec, -1\n\t"
__global__ void k_valu7_w2b32_exec(int *out, int seed, int) {
	__shared__ int lds[4096];
	int lf = seed, outv = seed * 3 + threadIdx.x, dg = seed * 5, h, g;
	unsigned shn = threadIdx.x & 24, sh = (threadIdx.x * 8) & 24, acc = 0, tab = 0x0a020a12u;
	int leftc = -7;
	unsigned addr = (unsigned)(size_t)&lds[0] + threadIdx.x * 16;
	u4 quad = {(unsigned)seed, (unsigned)seed + 1, (unsigned)seed + 2, (unsigned)seed + 3};
	const unsigned long long one = 1ull << 63;
	for (int i = 0; i < ITERS; ++i) {
		asm volatile(VALU7 W2B32_X VALU7 W2B32_X VALU7 W2B32_X VALU7 W2B32_X
		             : [lf] "+v"(lf), [shn] "+v"(shn), [outv] "+v"(outv), [acc] "+v"(acc), [dg] "+v"(dg), [h] "=&v"(h), [g] "=&v"(g)
		               , [quad] "+v"(quad)
		             : [sh] "v"(sh), [tab] "v"(tab), [leftc] "v"(leftc), [addr] "v"(addr), [one] "s"(one)
		             : "memory");
	}
	if (lf + outv + dg + (int)acc + (int)shn == 0x7fffffff) out[0] = lds[seed & 4095];
}

template <typename K>
static void run(const char *name, K kernel, int *d, int onelane)
{
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int waves = 0; waves <= 2; ++waves) {
		// one workgroup per CU, `waves` x 4 waves in it: 1 or 2 waves per SIMD
		hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, 0, d, 1, onelane);
		CHECK(hipDeviceSynchronize());
		float best = 1e9f;
		for (int r = 0; r < 3; ++r) {
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(kernel, dim3(waves == 2 ? 256 : 1), dim3(waves == 0 ? 256 : waves == 1 ? 64 : 512), 0, 0, d, 1, onelane);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		printf("%-22s %s: %.1f cycles per step @2.4GHz\n", name, waves == 0 ? "4 waves on one CU   " : waves == 1 ? "one wave alone      " : "2 waves/SIMD, 256 CU", best * 1e-3 * 2.4e9 / (ITERS * 4.0));
	}
}

int main()
{
	int *d;
	CHECK(hipMalloc(&d, 4096));
	run("valu7", k_valu7, d, 0);
	run("valu7+ds_write2_b32", k_valu7_w2b32, d, 0);
	run("valu7+w2b32 (1 addr)", k_valu7_w2b32, d, 1);
	run("valu7+w2b32 exec=1lane", k_valu7_w2b32_exec, d, 0);
	run("valu7+ds_write_b32", k_valu7_wb32, d, 0);
	run("valu7+w2b32 /2 steps", k_valu7_w2b32_2, d, 0);
	run("valu7+b128 /4 steps", k_valu7_wb128_4, d, 0);
	run("sdwa6", k_sdwa6, d, 0);
	run("sdwa6+b32", k_sdwa6_wb32, d, 0);
	run("sdwa6+b128 /4 steps", k_sdwa6_wb128_4, d, 0);
	run("valu9", k_valu9, d, 0);
	run("valu9+b128 w+r /4 steps", k_valu9_d2, d, 0);
	run("chain3", k_chain3, d, 0);
	run("chain3+b32", k_chain3_wb32, d, 0);
	return 0;
}
