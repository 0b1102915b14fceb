/* csadp_kernels.h -- host-callable launchers of the HIP kernels (csadp_bits.hip, csadp_cells.hip, csadp_pairio.hip, csadp_kernels.hip). */
#ifndef CSADP_KERNELS_H
#define CSADP_KERNELS_H

#include <hip/hip_runtime_api.h>

#include "csadp_device.h"

namespace csadp {

/* csadp_bits.hip: bit-parallel first fills and their traceback; words = 32-column words per lane (1 .. 4: BitJob::wpl
 * of every job of the table).  One workgroup per job (at most 16 strips each) ... */
hipError_t launch_fill_bits(int words, uint8_t *arena, const BitJob *jobs, int njobs, int maxstrips, int lds_pad, int *abort_word, hipStream_t st);
hipError_t launch_fill_bits_shared(int words, uint8_t *arena, const BitJob *jobs, int njobs, int passes, const TileRef *table, int nwgs, int lds_pad,
                                   int *abort_word, hipStream_t st);
/* ... or chunked: one workgroup per work item = (job, chunk of `waves` strips; 4, 8 or 16); `work` lists the items of
 * ONE pass, `passes` consecutive passes (job tables of njobs entries each) share a launch; epoch = a non-zero value
 * no earlier launch on this memory has used: it tags the hand-off granules between chunks */
hipError_t launch_fill_bits_wide(int words, int waves, uint8_t *arena, const BitJob *jobs, int njobs, int passes, const TileRef *work, int nwork,
                                 uint32_t epoch, int *abort_word, hipStream_t st, bool shared = false);
int fill_bits_lds_bytes(int waves);          /* static LDS of a fill workgroup */
int traceback_bits_lds_bytes(int words);    /* ... of a traceback workgroup */
/* scores: the replay traceback also sums the move scores of its path into summary[3] */
/* overlap: one word per lane, no scores, few jobs and nothing else on the chip: walk and replay side by side (156 KB of LDS per workgroup) */
hipError_t launch_traceback_bits(int words, uint8_t *arena, const BitJob *jobs, int njobs, hipStream_t st);

/* csadp_cells.hip: any fill as a persistent cell-per-lane wavefront; work = (job, chunk) items */
/* epoch: a value no earlier launch on this memory has used (24 bits): it tags the hand-off granules between chunks */
/* fetch: the workgroups carry a fetcher wave (few workgroups: every chain on compute units of its own; csadp_cells.hip) */
/* test_slow_publisher: test seam, units of ~0.1 ms the publisher wave sleeps per half block (0 in production) */
hipError_t launch_fill_cells(bool wide, bool fetch, uint8_t *arena, const CellJob *jobs, const TileRef *work, int nwork, uint32_t epoch,
                             int *abort_word, hipStream_t st, int test_slow_publisher = 0);
/* csadp_cells_tb.hip: the direction walk.  max_bands = 0: every matrix by one serial walk (CellJob::banded all 0);
 * else the most bands / scout groups of any banded job of the batch (scout, resolve, emit, gather) */
hipError_t configure_traceback_cells();
hipError_t launch_traceback_cells(uint8_t *arena, const CellJob *jobs, int njobs, int max_bands, int max_groups, hipStream_t st);
/* csadp_pairio.hip: 2-sequence tasks whose letters live in the arena (BitJob::text): bit planes from
 * the raw circular texts, and the two aligned rows + the DP score from the traceback's op list */
hipError_t launch_pack_planes(uint8_t *arena, const BitJob *jobs, int njobs, hipStream_t st);
hipError_t launch_expand_rows(uint8_t *arena, const BitJob *jobs, int njobs, hipStream_t st);
/* once per device (function attributes), with that device current */
hipError_t configure_kernels();
/* Column statistics (tools.c:259-281): out[0] gaps, out[1] conserved columns, out[2] SP score;
 * chars = nseq x length bytes, sequence-major; out must be zeroed. */
hipError_t launch_sp_columns(const uint8_t *chars, int nseq, int length, long long *out, hipStream_t st);
/* bytes from pinned host memory to the device, read by a kernel (small latency-bound uploads) */
hipError_t launch_pull_pinned(void *dst, const void *pinned_src, size_t bytes, hipStream_t st);

}  // namespace csadp

#endif
