/*
 * csadp.h -- C-ABI of libcsadp.so: MI355X (gfx950) implementation of CSA's
 * dynamic-programming alignment hot path.
 *
 * The library replaces the reference translation unit
 * /root/reference/source/dynamicprogramming.c (public surface:
 * `void ProgressiveDP(struct _alignmapsegment *segment);`,
 * dynamicprogramming.h:3, sole live caller RunAlignment, alignment.c:201).
 * Plain C types only: pointers, ints, sizes.  Every entry point returns
 * CSADP_OK (0) or a negative CSADP_ERR_* code and never calls exit().
 *
 * There is NO CPU fallback: if no gfx950 device / HIP runtime is usable the
 * calls fail with CSADP_ERR_NO_DEVICE.
 */
#ifndef CSADP_H
#define CSADP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CSADP_VERSION 500

/* only the C-ABI below is exported from libcsadp.so */
#define CSADP_API __attribute__((visibility("default")))

#define CSADP_OK              0
#define CSADP_ERR_ARG        -1   /* NULL / out-of-range argument                         */
#define CSADP_ERR_ALPHABET   -2   /* region holds a letter other than A,C,G,T (see below)  */
#define CSADP_ERR_NOMEM      -3   /* host allocation failed                                */
#define CSADP_ERR_NO_DEVICE  -4   /* no usable HIP device / library not initialised        */
#define CSADP_ERR_HIP        -5   /* a HIP runtime call or kernel failed                   */
#define CSADP_ERR_RANGE      -6   /* task too large for 32-bit scores or device memory     */
#define CSADP_ERR_STATE      -7   /* call sequence error (e.g. fetch before run)           */

#define CSADP_MAX_SEQS 64         /* MAXNUMBEROFSEQS, csamsa.c:23 */
#define CSADP_MAX_DEVICES 16      /* GPUs one host process may drive (csadp_align_batch_multi) */

/* ---- library lifetime ------------------------------------------------------------ */

typedef struct csadp_config {
	int device;        /* HIP device ordinal of the PRIMARY device; -1 = take LOCAL_RANK / 0.  An   */
	                   /* ordinal that does not exist is CSADP_ERR_NO_DEVICE (CSADP_SHARE_DEVICE=1 */
	                   /* maps it onto the visible devices instead: multi-rank rehearsals only)    */
	int tile_rows;     /* ignored since round 3 (the tiled kernels it tuned are gone); kept for ABI    */
	int verbose;       /* 1 = print the reference's progress tokens in the drop-in adapter */
} csadp_config;

/* Selects the primary device (the one the entry points without a device argument use) and brings
 * it up; idempotent, NULL = defaults.  Every entry point makes its device current on the calling
 * thread (hipSetDevice is a per-thread setting), so calls may come from any host thread. */
CSADP_API int csadp_init(const csadp_config *cfg);
/* Optional.  Pays now what a process' first batch would otherwise pay on top of its work: the kernels' code objects (loaded
 * at the first launch out of each), the host pool's threads, the arenas and pinned staging of csadp_align_batch (1.25 GB of
 * HBM reserved).  Runs a small synthetic batch through every kernel family.  Thread-safe like every entry point; a caller
 * with host work in front of its first batch runs it on a helper thread meanwhile (csadp_dropin.c does: the reference program
 * builds its suffix tree first).  Measured on the reference program relinked with the drop-in: first batch 60-70 -> 12-15 ms
 * (profiles/r05_dropin_*.json). */
CSADP_API int csadp_warmup(void);
CSADP_API void csadp_shutdown(void);
CSADP_API int csadp_version(void);
CSADP_API const char *csadp_strerror(int code);
/* name of the primary device, number of compute units; CSADP_ERR_NO_DEVICE before init */
CSADP_API int csadp_device_info(char *name, int namelen, int *compute_units);
/* HIP devices visible to this process */
CSADP_API int csadp_device_count(int *count);

/* ---- one alignment task = one ProgressiveDP call ---------------------------------- */

/*
 * Replaces the implicit inputs of ProgressiveDP (globals csamsa.h:8-12 and the
 * alignmapsegment fields alignmentmap.h:3-10):
 *   nseq          numberofseqs (2..64)
 *   texts[s]      texts[s]: circular sequence, uppercase, textsizes[s] letters
 *   rotations[s]  rotations[s]
 *   starts[s]     segment->positions[s] + segment->size       (rotated coordinates)
 *   ends[s]       segment->next->positions[s]                 (exclusive)
 * A letter at rotated position p is texts[s][(rotations[s]+p) wrapped once]
 * (CharAt, alignment.c:16-20).
 *
 * Letters other than A,C,G,T inside a region make the reference index
 * scorevector[][-1] (dynamicprogramming.c:941-942,:991-993; undefined
 * behaviour).  This library rejects such a task with CSADP_ERR_ALPHABET.
 */
typedef struct csadp_task {
	int nseq;
	const char *const *texts;
	const int *textsizes;
	const int *rotations;
	const int *starts;
	const int *ends;
} csadp_task;

/*
 * Replaces the outputs of ProgressiveDP:
 *   aligned       segment->alignedstrings (dynamicprogramming.c:1160): calloc'd array of
 *                 nseq malloc'd NUL-terminated strings of equal length, ORIGINAL index
 *                 order; NULL when every region is empty (early return, :916).
 *                 Ownership passes to the caller: release with free() exactly as
 *                 DeleteAlignmentMap does (alignmentmap.c:174-179) or csadp_free_result.
 *   consensus     final consensus size (the value printed at :1159)
 *   score         dpmatrix[nrows][ncols] of the last fill (not observable in the
 *                 reference, which frees the matrix at :1161-1165)
 *   cells         sum of nrows*ncols over the fills of this task
 */
typedef struct csadp_result {
	int status;
	int score;
	int consensus;
	int fills;
	long long cells;
	char **aligned;
	char *progress;    /* the reference's stdout tokens between "[(min-max)" and "->": one '.' per fill  */
	                   /* (:1156), each followed by one '!' per all-gap column DeleteGappedColumns met   */
	                   /* (:689); malloc'd, NUL-terminated, released by csadp_free_result / free()       */
} csadp_result;

/* Align ntasks independent tasks (any nseq each) on the device: upload, fill, traceback,
 * progressive profile update.  results[t].status carries per-task errors. */
CSADP_API int csadp_align_batch(const csadp_task *tasks, int ntasks, csadp_result *results);
CSADP_API void csadp_free_result(csadp_result *r, int nseq);
/* csadp_free_result on count results of tasks with nseq sequences each (a batch of pairs: nseq = 2); returns how
 * many of them carried a status other than CSADP_OK -- one call per batch for callers that stream batches. */
CSADP_API int csadp_free_results(csadp_result *results, int count, int nseq);

/* ---- more than one GPU (independent tasks, no collective inside a matrix) -------- */

/* The same on an explicitly named HIP device (brought up on first use).  Calls naming different
 * devices may run concurrently from different host threads. */
/* Where the time of the calling process' LAST csadp_align_batch[_on] went (host clocks; summed over the lock-step rounds of all round
 * groups, which run side by side: the phases can add up to more than wall_ms).  device_ms = upload + kernels + download as the host waits
 * for them; tables / apply / refine_* = the host's part of ProgressiveDP between two fills (dynamicprogramming.c:957-987, :1050-1155,
 * :643-899).  What bench.py's profile_batch leg reports. */
typedef struct csadp_batch_phases {
	int tasks, rounds, round_groups;
	double wall_ms, seed_ms, layout_ms, tables_ms, device_ms, apply_ms, refine_speculate_ms, refine_commit_ms, results_ms;
} csadp_batch_phases;
CSADP_API int csadp_last_batch_phases(csadp_batch_phases *out);
/* passes of the primary device's batches that were repeated chunk by chunk after a bounded cross-workgroup wait of a chunked fill ran out,
 * since the library was initialised (csadp_timing.recoveries is the per-batch figure of the pair batches); 0 in any healthy run */
CSADP_API long csadp_recoveries(void);
CSADP_API int csadp_align_batch_on(int device, const csadp_task *tasks, int ntasks, csadp_result *results);

/* Work estimate of a task in DP cells (sum over its fills of rows x columns, the consensus taken
 * as the longest region so far; exact for nseq = 2): the cost csadp_align_batch_multi balances. */
CSADP_API long long csadp_task_cost(const csadp_task *task);

typedef struct csadp_multi_stats {
	int ndevices;
	int tasks[CSADP_MAX_DEVICES];        /* tasks given to each device                      */
	long long cost[CSADP_MAX_DEVICES];   /* their summed csadp_task_cost                    */
	double ms[CSADP_MAX_DEVICES];        /* wall time of each device's host thread          */
	long long total_cost, max_cost;      /* imbalance = max_cost * ndevices / total_cost    */
	double wall_ms;
} csadp_multi_stats;

/* One batch over ndevices GPUs of this node from ONE host process: tasks are partitioned by
 * longest-processing-time-first over their csadp_task_cost (csadp_partition_lpt), one host thread
 * per GPU aligns its part (csadp_align_batch_on), results land in results[] in task order -- the
 * gather is host memory.  devices = NULL means ordinals 0..ndevices-1.  stats may be NULL.
 * Every results[] entry is defined on return, also when the call fails: the tasks of a device whose
 * thread failed carry that error in .status with NULL strings, the others are complete and the
 * caller's to free (csadp_free_results over the whole array is always safe). */
CSADP_API int csadp_align_batch_multi(const csadp_task *tasks, int ntasks, csadp_result *results,
                                      const int *devices, int ndevices, csadp_multi_stats *stats);

/* ---- device-resident pair batches (the benchmarked path) --------------------------- */

/*
 * A batch of 2-sequence tasks whose inputs live in HBM.  create() checks the arguments, copies the
 * letters of every distinct text into pinned memory and STARTS one H2D copy (it does not wait and does
 * not reference the caller's texts afterwards); run() requests one pass and returns immediately;
 * flush() enqueues the requested passes, sync() also waits; fetch() downloads the LAST pass' results
 * in one D2H copy.  A pass = nw_pack_planes (CharAt + letter codes + alphabet check of both regions,
 * from the raw circular texts) -> fill -> traceback -> nw_expand_rows (the two aligned rows and the DP
 * score), all on the device: the host never touches a letter between create and fetch.  run() may be
 * called repeatedly before a sync() (benchmark steps, streaming use): consecutive passes are merged
 * into launches that rotate over independent result/scratch slots on separate HIP streams.  A
 * streaming caller keeps several batches in flight: create + run + flush of batch n+1 before fetch
 * of batch n.  A pass that is flushed ALONE while the device holds nothing else of the batch takes a
 * shape of its own where one pass does not fill the chip (one word of 32 columns per lane, a matrix'
 * strips spread over all compute units: 128 pairs of 16 kbp in 1.9 instead of 2.6 ms).  After a fetch
 * the batch may run and be fetched again, as long as every task has a matrix (no empty region).
 * (CSADP_DEVICE_IO=0 or CSADP_BITS=0: the host packs tables and applies traces, as for N-sequence
 * tasks; such a batch is fetched once.)
 */
typedef struct csadp_pairbatch csadp_pairbatch;

CSADP_API int csadp_pairs_create(const csadp_task *tasks, int ntasks, csadp_pairbatch **out);
CSADP_API int csadp_pairs_create_on(int device, const csadp_task *tasks, int ntasks, csadp_pairbatch **out);
CSADP_API int csadp_pairs_run(csadp_pairbatch *b);
CSADP_API int csadp_pairs_flush(csadp_pairbatch *b);    /* enqueue every requested pass, do not wait */
CSADP_API int csadp_pairs_sync(csadp_pairbatch *b);
CSADP_API int csadp_pairs_fetch(csadp_pairbatch *b, csadp_result *results);
CSADP_API void csadp_pairs_destroy(csadp_pairbatch *b);

/* Score-only path for 2-sequence tasks (all-vs-all distance matrices, guide trees): fill and
 * traceback run on the device exactly as for csadp_pairs_*, the host derives
 * scores[t] = dpmatrix[nrows][ncols] from the traced path and skips building the aligned
 * strings.  status[t] (may be NULL) receives per-task errors. */
CSADP_API int csadp_score_pairs(const csadp_task *tasks, int ntasks, int *scores, int *status);

typedef struct csadp_timing {
	long long cells;        /* DP cells of one run()                                      */
	int fill_launches;      /* fill-kernel launches of the last run()                     */
	long long fill_tiles;   /* tile workgroups of the last run()                          */
	float fill_ms;          /* HIP-event time from first to last fill launch, last run()  */
	float traceback_ms;     /* HIP-event time of the traceback kernel, last run()         */
	float total_ms;         /* fill + traceback, HIP events on the library stream         */
	long long dir_bytes;    /* direction bytes written to HBM by one run() (0 in          */
	                        /* checkpoint mode, where the traceback replays the path)      */
	long long border_bytes; /* tile hand-off bytes written + read by one run(); checkpoint */
	                        /* mode: lane-state checkpoints + strip hand-off words written   */
	int launch_passes;      /* passes carried by the launch that held the last run(): the   */
	                        /* bit-parallel path merges consecutive run() calls into one     */
	                        /* launch; fill_ms / traceback_ms / total_ms are that launch's    */
	int bit_parallel;       /* 0 = 32-bit kernels, 2 = bit-parallel kernels (nw_fill_bits: lane   */
	                        /* checkpoints, replay traceback)                                     */
	int merge_group;        /* passes a full launch of this batch carries (bit-parallel path)     */
	int recoveries;         /* passes repeated on the wait-free path after a bounded wait of the  */
	                        /* chunked fill ran out (0 in any healthy run)                        */
	int device_io;          /* 1 = a pass starts from the raw letters in HBM and ends with the    */
	                        /* aligned rows in HBM (nw_pack_planes / nw_expand_rows in the pass)  */
	int words_per_lane;     /* bit-parallel path: 32-column words a lane owns (1 .. 4)           */
	int streams;            /* bit-parallel path: fill launches kept in flight                    */
} csadp_timing;

CSADP_API int csadp_pairs_timing(csadp_pairbatch *b, csadp_timing *t);

/* ---- alignment statistics on the device (tools.c:194-293, CalculateSumOfPairsScore) ---- */

typedef struct csadp_sp_stats {
	int consensus;            /* common length of the aligned strings ("Consensus size")        */
	long long total_gaps;     /* '-' characters over all sequences (the tool prints gaps/nseq)   */
	int conserved_columns;    /* columns whose characters are identical in every sequence        */
	long long sp_score;       /* sum over columns and sequence pairs: gap/gap 0, equal +1, else -1 */
} csadp_sp_stats;

/* aligned[0..nseq) are NUL-terminated strings of equal length (else CSADP_ERR_ARG). */
CSADP_API int csadp_sp_score(const char *const *aligned, int nseq, csadp_sp_stats *out);

/* ---- rotation finder (csamsa.c:69-308 analyzeTree / getRotations) ----------------------- */

typedef struct csadp_rotation_info {
	int blocks;               /* unique maximal common blocks found (the reference's "nodes left")   */
	int chain_size;           /* summed block length of the winning chain                            */
	int chain_span;           /* its length including the intervals between blocks                   */
	int first_block_depth;    /* length of the block whose positions are the rotations               */
} csadp_rotation_info;

/*
 * rotations[s] = position in texts[s] of the first block of the heaviest chain of blocks common
 * to all sequences, as the reference computes it from its generalized cyclic suffix tree.  Host
 * computation (the reference spends < 3 % of its time here).  CSADP_ERR_RANGE: no unique common
 * block exists, or the reference's own chain bookkeeping would not terminate on this input.
 * Blocks as long as the shortest sequence are not considered.
 */
CSADP_API int csadp_find_rotations(int nseq, const char *const *texts, const int *sizes, int *rotations,
                                   csadp_rotation_info *info);

/* ---- anchor stage (alignment.c:69-86 PrepareTreeForAlignment, :163-214 RunAlignment) ------ */

/*
 * The alignment map the reference builds before and around its ProgressiveDP calls: a list of
 * segments from the fake first one (size 1 at position -1, alignment.c:57-64) to the fake last
 * one (size 0 at the sequence ends).  Segment k fixes size[k] letters of every sequence s starting
 * at rotated position positions[k*nseq+s] as an anchor column block; dp[k] = 1 marks the gap
 * between segment k and k+1 as one ProgressiveDP task (starts = positions+size, ends = the next
 * segment's positions).  A gap with dp[k] = 0 is one the reference skips because some sequence has
 * nothing in it (alignment.c:180-183) -- its letters are then absent from the output, as there.
 */
typedef struct csadp_anchor_map {
	int nseq;
	int nsegs;
	int border_nodes;     /* border nodes found before the loop (the reference's list length) */
	int *size;            /* [nsegs]       */
	int *dp;              /* [nsegs]       */
	int *positions;       /* [nsegs*nseq]  */
} csadp_anchor_map;

/*
 * Host computation.  texts are the un-rotated circular sequences; letters other than A,C,G,T
 * compare equal to each other (gencycsuffixtrees.c:320).  CSADP_ERR_RANGE: a proper suffix of one
 * rotated sequence is a whole rotation of another one -- the reference's walk then leaves the
 * sequence (morenodeslinkedlists.c:590-617) and has no defined result.
 */
CSADP_API int csadp_build_anchor_map(int nseq, const char *const *texts, const int *sizes, const int *rotations,
                                     csadp_anchor_map *out);
CSADP_API void csadp_free_anchor_map(csadp_anchor_map *m);

/* ---- the whole alignment stage (csamsa.c:606-623, mode N / mode A) ------------------------- */

typedef struct csadp_msa_stats {
	int nseq;
	int border_nodes;
	int segments;             /* alignment map segments including the two fake ones ("alignment segments") */
	int dp_gaps;              /* gaps aligned by DP = ProgressiveDP calls of the reference                */
	int fills;                /* DP matrix fills over all gaps                                             */
	int alignment_length;     /* length of row 0 ("Alignment size")                                        */
	long long cells;
	double rotations_ms, anchors_ms, dp_ms, rows_ms;
	int recoveries;           /* (version 500) passes of the DP batch repeated chunk by chunk after a bounded cross-workgroup wait ran out; 0 in any healthy run */
} csadp_msa_stats;

/*
 * rotations_in NULL: the rotations are found as in mode N (csadp_find_rotations); otherwise they
 * are used as given (all zero = the reference's mode A).  rows_out receives nseq malloc'd rows in
 * input order, exactly the sequence lines of the reference's "-Aligned.fasta"; free with
 * csadp_free_rows.  All gaps are aligned in one device batch.
 */
CSADP_API int csadp_msa(int nseq, const char *const *texts, const int *sizes, const int *rotations_in, int *rotations_out,
                        char ***rows_out, csadp_msa_stats *stats);
CSADP_API void csadp_free_rows(char **rows, int nseq);
/* SaveAlignment's file format (alignment.c:97-105): ">desc @ rot" (">desc" when rotations is NULL), the row. */
CSADP_API int csadp_write_aligned_fasta(const char *path, const char *const *descs, const int *rotations,
                                        const char *const *rows, int nseq);

/* ---- host helpers ------------------------------------------------------------------ */

/* Longest-processing-time partition of n task costs over nparts devices.
 * assign[i] receives the part of task i; returns the maximum part load via *maxload. */
CSADP_API int csadp_partition_lpt(const long long *cost, int n, int nparts, int *assign, long long *maxload);

/* FNV-1a-32 over strings[0..n): the digest used for golden values and for the
 * fixed-size result records that ranks exchange */
CSADP_API unsigned csadp_fnv1a(const char *const *strings, int n);

/* FASTA loader following the reference's rules (csamsa.c:433-519): skips \n \r NUL '-'
 * and space, uppercases, admits IUPAC letters, drops a record holding any other byte,
 * at most CSADP_MAX_SEQS records.  texts/descs/sizes are malloc'd arrays of *nseq entries. */
CSADP_API int csadp_load_fasta(const char *path, char ***texts, char ***descs, int **sizes, int *nseq);
CSADP_API void csadp_free_fasta(char **texts, char **descs, int *sizes, int nseq);

/* "<base>-Rotated.fasta" wire format of saveRotatedSequences (csamsa.c:416-431): one record per
 * sequence, header ">desc @ rot", the text rotated left by rot on ONE line.  The reader returns
 * the rotation offsets of such a file (how a caller feeds the reference's mode-R output to the
 * DP); at most nmax records, *nread receives the count. */
CSADP_API int csadp_write_rotated_fasta(const char *path, const char *const *descs, const char *const *texts,
                                        const int *sizes, const int *rotations, int nseq);
CSADP_API int csadp_read_rotations(const char *path, int *rotations, int nmax, int *nread);

#ifdef __cplusplus
}
#endif
#endif /* CSADP_H */
