/*
 * oracle/csa_dp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's progressive profile DP
 * (/root/reference/source/dynamicprogramming.c: ProgressiveDP :906-1171,
 * SortSequencesForDP :276-308, DeleteGappedColumns :643-899,
 * CharCodeFromSeq :57-71, GetCharCode :74-85; CharAt alignment.c:16-20).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (libcsadp.so)
 * never links, loads or calls it.
 *
 * Parity status: PINNED -- checked against oracle/_ref/libcsa_ref.so (the
 * unmodified reference sources compiled by oracle/Makefile) and against the
 * golden vectors in tests/golden/ (generated from that library by
 * tests/golden/make_golden.py).
 */
#ifndef CSA_DP_ORACLE_H
#define CSA_DP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ODP_OK            0
#define ODP_ERR_ARG      -1
#define ODP_ERR_ALPHABET -2   /* non-ACGT letter: reference behaviour is UB (survey Q4) */
#define ODP_ERR_NOMEM    -3

typedef struct odp_stats {
	long long cells;          /* sum of nrows*ncols over all fills              */
	int fills;                /* number of matrix fills                         */
	int stale_border_fills;   /* fills that ran on un-refreshed borders (Q1)    */
	int last_score;           /* dpmatrix[nrows][ncols] of the last fill        */
	int consensus;            /* final consensus size (common string length)    */
	double fill_seconds;      /* wall time spent inside the fill loops          */
} odp_stats;

/*
 * Restatement of ProgressiveDP on one region (rotated coords, end exclusive).
 * out[s] receives malloc'd NUL-terminated strings in ORIGINAL index order, or
 * all NULL when every region is empty (reference early return, :916).
 * Returns the consensus size (>=0) or a negative ODP_ERR_*.
 */
int odp_progressive_dp(int nseq, const char *const *texts, const int *textsizes,
                       const int *rotations, const int *starts, const int *ends,
                       char **out, odp_stats *stats);

/*
 * One matrix fill exactly as dynamicprogramming.c:957-1029 does it, for testing
 * the HIP fill kernel in isolation.
 *   rowcodes[0..nrows)      codes 0..3 of the row sequence
 *   sv[(ncols+1)*5]         profile counts, column 0 unused (:931)
 *   nprev                   the loop variable i (number of already aligned seqs)
 *   top[0..ncols], left_i   border row 0 values and the i used for column 0
 *                           (H[j][0] = -left_i*j); pass top=NULL to have the
 *                           fresh borders of :963-973 computed from sv/nprev.
 *   H[(nrows+1)*(ncols+1)], dirs[(nrows+1)*(ncols+1)]   outputs ('D','L','U')
 */
int odp_fill(int nrows, int ncols, const signed char *rowcodes, const int *sv, int nprev,
             const int *top, int left_i, int *H, char *dirs);

/*
 * Linear-space score of a 2-sequence task: dpmatrix[nrows][ncols] of the single fill
 * ProgressiveDP runs for numberofseqs == 2 (:990-1029 with i = 1 on the fresh borders of
 * :963-973), computed with two rows and no direction matrix, so that pairs far beyond what
 * the reference's 5 B/cell matrices allow (200 kbp x 200 kbp) can still be checked for
 * OPTIMALITY: an aligned pair is optimal iff its SP score (tools.c:274-280) equals this.
 * Regions in rotated coordinates as in odp_progressive_dp.  Returns ODP_OK / ODP_ERR_*.
 */
int odp_pair_score_linear(const char *const *texts, const int *textsizes, const int *rotations,
                          const int *starts, const int *ends, long long *score);

/* The four statistics of tools.c:194-293 (CalculateSumOfPairsScore, the reference's mode S):
 * consensus size, total '-' count (the tool prints total / nseq), conserved columns, SP score. */
int odp_sp_stats(int nseq, const char *const *aligned, int *consensus, long long *gaps, int *conserved,
                 long long *sp);

/* Sum-of-pairs score of aligned strings (rule of tools.c:274-280). */
long long odp_sp_score(int nseq, const char *const *aligned);

/* FNV-1a-32 over aligned[0], aligned[1], ... (the digest used in SURVEY.md 8c). */
unsigned odp_fnv1a(int nseq, const char *const *aligned);

void odp_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
