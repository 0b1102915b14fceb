/*
 * csadp_dropin.c -- source-compatible replacement for the reference's
 * `void ProgressiveDP(alignmapsegment *segment)` (dynamicprogramming.c:906,
 * declared dynamicprogramming.h:3, called from RunAlignment alignment.c:201).
 *
 * Packs the reference-shaped globals into one csadp_task, runs it on the GPU through
 * libcsadp.so and hands the malloc'd strings to the segment (freed later by
 * DeleteAlignmentMap, alignmentmap.c:174-179).  Prints the reference's progress tokens
 * (:917, :1156, :689, :1159) so logs diff cleanly.  On a library error the process cannot
 * continue meaningfully (the reference has no error path here either): the adapter prints
 * the reason and exits non-zero -- it never falls back to a CPU computation.
 */
#include <stdio.h>
#include <stdlib.h>

#include "csa_dropin.h"
#include "csadp.h"

void ProgressiveDP(struct _alignmapsegment *segment)
{
	csadp_task task;
	csadp_result res;
	int *starts, *ends;
	int s, rc;

	if (segment->maxgapsize == 0) return;                               /* :916 */
	printf("[(%-4d-%4d)", segment->mingapsize, segment->maxgapsize);    /* :917 */
	fflush(stdout);
	starts = (int *)malloc((size_t)numberofseqs * sizeof(int));
	ends = (int *)malloc((size_t)numberofseqs * sizeof(int));
	if (!starts || !ends) { fprintf(stderr, "csadp drop-in: out of memory\n"); exit(2); }
	for (s = 0; s < numberofseqs; s++) {
		starts[s] = segment->positions[s] + segment->size;              /* :288, :936 */
		ends[s] = segment->next->positions[s];                          /* :1069 */
	}
	task.nseq = numberofseqs;
	task.texts = (const char *const *)texts;
	task.textsizes = textsizes;
	task.rotations = rotations;
	task.starts = starts;
	task.ends = ends;
	rc = csadp_align_batch(&task, 1, &res);
	if (rc == CSADP_OK) rc = res.status;
	if (rc != CSADP_OK) {
		fprintf(stderr, "\ncsadp drop-in: ProgressiveDP failed: %s\n", csadp_strerror(rc));
		exit(2);
	}
	fputs(res.progress ? res.progress : "", stdout);                    /* :1156 '.' per fill, :689 '!' per all-gap column */
	free(res.progress);
	printf("->%4d]\n", res.consensus);                                  /* :1159 */
	fflush(stdout);
	segment->alignedstrings = res.aligned;                              /* :1160 */
	free(starts);
	free(ends);
}
