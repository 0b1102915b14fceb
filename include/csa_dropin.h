/*
 * csa_dropin.h -- the reference's own call surface for the DP hot path, as re-exported by
 * csa_amd/csrc/csadp_dropin.c.  A CSA build replaces source/dynamicprogramming.c by
 * csadp_dropin.c (or links csadp_dropin.o) and adds -lcsadp; nothing else changes.
 *
 *   replaces                                     reference declaration
 *   void ProgressiveDP(alignmapsegment *)        dynamicprogramming.h:3
 *
 * The adapter reads the reference's header-defined globals (csamsa.h:8-12) and the segment
 * fields (alignmentmap.h:3-10), exactly the implicit inputs of the original.
 */
#ifndef CSA_DROPIN_H
#define CSA_DROPIN_H

#ifdef __cplusplus
extern "C" {
#endif

/* alignmentmap.h:3-10 -- field order and types must match the reference's struct */
typedef struct _alignmapsegment {
	int *positions;                 /* per sequence: start of the anchor (rotated coordinates) */
	int size;                       /* length of the anchor block                              */
	int mingapsize;
	int maxgapsize;
	char **alignedstrings;          /* OUT: malloc'd strings, owned by the segment            */
	struct _alignmapsegment *next;
} alignmapsegment;

/* csamsa.h:8-12 -- defined by the CSA program (tentative definitions in its headers) */
extern int numberofseqs;
extern char **texts;
extern int *textsizes;
extern int *rotations;

void ProgressiveDP(struct _alignmapsegment *segment);

/*
 * Deferred mode (round 5).  The reference calls ProgressiveDP once per un-anchored gap, one after the other
 * (RunAlignment, alignment.c:179-206); nothing reads a gap's strings before SaveAlignment (alignment.c:134-156).  In
 * deferred mode a call only RECORDS its gap (segment, region bounds) and returns; csadp_dropin_finish() submits every
 * recorded gap as ONE csadp_align_batch, assigns segment->alignedstrings and prints the gaps' log lines in call order.
 * Two ways to switch it on:
 *   link flag only   add -Wl,--wrap=SaveAlignment to the program's link line: __wrap_SaveAlignment (csadp_dropin.c)
 *                    finishes and then calls the real SaveAlignment.  Files and stdout stay byte-identical
 *                    (RunAlignment prints nothing between two gaps).  CSADP_DROPIN_DEFER=0 in the environment keeps
 *                    such a binary synchronous.
 *   one source line  csadp_dropin_defer(1) before RunAlignment(), csadp_dropin_finish() behind it.
 * The default (neither) is the synchronous mode: every call is finished when it returns.
 * csadp_dropin_finish returns the number of gaps it finished (0: nothing pending).
 *
 * Errors.  ProgressiveDP is void in the reference and checks nothing (unchecked malloc, dynamicprogramming.c:964-981), so
 * the adapter has no way to report a failed gap to its caller: on any library error (no gfx950 device, letters outside
 * A/C/G/T inside a region -- undefined behaviour in the reference, SURVEY quirk Q4 --, out of device memory) it prints
 * csadp_strerror() to stderr and calls exit(2), in both modes.  Programs that want to handle errors use csadp.h directly.
 *
 * CSADP_DROPIN_STATS=<file>: at exit the adapter appends one JSON line with its own clock readings (calls, seconds inside
 * ProgressiveDP, the first call's, csadp_init's, the finish) -- what bench.py's `dropin` leg reads.
 */
void csadp_dropin_defer(int on);
int csadp_dropin_finish(void);

#ifdef __cplusplus
}
#endif
#endif
