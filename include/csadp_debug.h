/*
 * csadp_debug.h -- test seam of libcsadp.so (NOT part of the drop-in surface).
 *
 * csadp_debug_align_with_filler runs the HOST logic of one task (ordering, profile
 * seeding, border-refresh rule, traceback application, DeleteGappedColumns -- i.e.
 * everything of dynamicprogramming.c:906-1171 that stays on the CPU) and asks the caller
 * for every matrix fill + direction walk.  The CPU test-suite passes a filler built on
 * the oracle so the host logic can be checked without a GPU; the library itself never
 * supplies a filler: the product path fills matrices on the GPU only.
 */
#ifndef CSADP_DEBUG_H
#define CSADP_DEBUG_H

#include "csadp.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * One fill in the reference's own terms (dynamicprogramming.c:957-1049):
 *   sv[(ncols+1)*5]   profile counts (column 0 unused), nprev = loop variable i
 *   rowcodes[nrows]   0..3
 *   top[ncols+1]      border row 0 as last refreshed, left_i: H[j][0] = -left_i*j
 * outputs: ops[] = direction codes (2 'D', 1 'L', 0 'U') of the walk from (nrows,ncols)
 * until a border is hit, *nops, rows/columns left (*remj,*remk), *score = H[nrows][ncols].
 */
typedef int (*csadp_debug_fill_fn)(void *user, int nrows, int ncols, int nprev, const int *sv,
                                   const signed char *rowcodes, const int *top, int left_i,
                                   unsigned char *ops, int *nops, int *remj, int *remk, int *score);

CSADP_API int csadp_debug_align_with_filler(const csadp_task *task, csadp_debug_fill_fn fill, void *user,
                                  csadp_result *result);

/* The same for a batch of tasks, through the round driver of csadp_align_batch itself -- lock-step rounds, the tasks dealt
 * over round groups (CSADP_ROUND_GROUPS, default 2), one host thread per group, per-task host work on the pool -- with the
 * caller's filler where the product runs its device batch.  `fill` is called from several threads at once. */
CSADP_API int csadp_debug_align_batch_with_filler(const csadp_task *tasks, int ntasks, csadp_debug_fill_fn fill, void *user,
                                        csadp_result *results);

/* The library reads its environment switches ONCE per process (csa_amd/csrc/csadp_config.h; INTEGRATION.md lists them).  This
 * reads them again: the test-suite flips switches between calls inside one process. */
CSADP_API void csadp_debug_reload_config(void);

/* Runs one parallel region of the library's persistent host thread pool (the one that spreads per-task
 * host work: table packing, traceback application, result strings) over `items` work items, each adding
 * its index to an atomic; *sum receives items*(items-1)/2.  For the thread-sanitizer build: callable from
 * several threads at once. */
CSADP_API int csadp_debug_pool_selftest(int items, long long *sum);

/* The hand-off granules between the workgroups of a chunked fill are valid when they carry their launch's epoch, a
 * process-wide counter of 24 bits that never takes the value 0 (freshly zeroed granules must never look valid).
 * Sets the counter, so that a test can put launches either side of its wrap; returns the previous value. */
CSADP_API unsigned csadp_debug_set_epoch(unsigned next);

#ifdef __cplusplus
}
#endif
#endif
