/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin driver that is compiled TOGETHER WITH the unmodified reference sources
 * where they lie (/root/reference/source/*.c, see oracle/Makefile target
 * "_ref") into oracle/_ref/libcsa_ref.so.  It drives the reference's own
 * ProgressiveDP() (dynamicprogramming.c:906) the way the reference's commented
 * whole-sequence route does (alignment.c:173-178 + alignment.c:57-65):
 * build a first/last alignmapsegment pair, fill the header-defined globals
 * (csamsa.h:8-12), call ProgressiveDP, hand back segment->alignedstrings.
 *
 * No reference source text is copied here: only its public declarations are
 * #included at build time from /root/reference/source.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <time.h>

#include "csamsa.h"            /* numberofseqs, texts, textsizes, descs, rotations */
#include "alignment.h"         /* CharAt */
#include "alignmentmap.h"      /* alignmapsegment, NewAlignmentMapSegment, UpdateSegmentGapSizes */
#include "dynamicprogramming.h"/* ProgressiveDP */

/* file-level globals of dynamicprogramming.c (non-static there, :28-33).  They
 * are sized by the FIRST call's numberofseqs (:284-285, survey quirk Q5), so a
 * driver that changes numberofseqs between calls must drop them. */
extern int *orderedseqs;
extern int *seqlengths;
extern int **scorevector;

static int quiet_begin(void)
{
	int saved;
	int devnull;
	fflush(stdout);
	saved = dup(1);
	devnull = open("/dev/null", O_WRONLY);
	if (devnull >= 0) { dup2(devnull, 1); close(devnull); }
	return saved;
}

static void quiet_end(int saved)
{
	fflush(stdout);
	if (saved >= 0) { dup2(saved, 1); close(saved); }
}

/*
 * Run the reference ProgressiveDP on one region.
 *   nseq               2..64 sequences
 *   txts[s], sizes[s]  full (un-rotated) circular texts, uppercase ACGT
 *   rots[s]            rotation offsets (csamsa.h:12)
 *   starts[s], ends[s] region in ROTATED coordinates, end exclusive
 *   out[s]             receives a malloc'd NUL-terminated aligned string
 *                      (original-index order) or NULL when the reference
 *                      returns early (maxgapsize==0, dynamicprogramming.c:916)
 *   seconds            wall time of the ProgressiveDP call alone
 * returns the common aligned length (consensus size), or -1.
 */
int csa_ref_progressive_dp(int nseq, const char **txts, const int *sizes, const int *rots,
                           const int *starts, const int *ends, char **out, double *seconds)
{
	alignmapsegment *first, *last;
	struct timespec t0, t1;
	int s, saved, cons = -1;

	if (orderedseqs) { free(orderedseqs); orderedseqs = NULL; }
	if (seqlengths) { free(seqlengths); seqlengths = NULL; }

	numberofseqs = nseq;
	texts = (char **)calloc((size_t)nseq, sizeof(char *));
	textsizes = (int *)calloc((size_t)nseq, sizeof(int));
	rotations = (int *)calloc((size_t)nseq, sizeof(int));
	for (s = 0; s < nseq; s++) {
		texts[s] = (char *)txts[s];
		textsizes[s] = sizes[s];
		rotations[s] = rots[s];
	}
	first = NewAlignmentMapSegment(NULL);
	last = NewAlignmentMapSegment(NULL);
	first->next = last;
	first->size = 0;
	for (s = 0; s < nseq; s++) {
		first->positions[s] = starts[s];
		last->positions[s] = ends[s];
	}
	UpdateSegmentGapSizes(first);

	saved = quiet_begin();
	clock_gettime(CLOCK_MONOTONIC, &t0);
	ProgressiveDP(first);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	quiet_end(saved);
	if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);

	for (s = 0; s < nseq; s++) out[s] = NULL;
	if (first->alignedstrings != NULL) {
		for (s = 0; s < nseq; s++) {
			out[s] = first->alignedstrings[s];
			if (out[s] != NULL && cons < 0) cons = (int)strlen(out[s]);
		}
		free(first->alignedstrings);
		first->alignedstrings = NULL;
	} else {
		cons = 0;
	}
	free(first->positions); free(first);
	free(last->positions); free(last);
	free(texts); texts = NULL;
	free(textsizes); textsizes = NULL;
	free(rotations); rotations = NULL;
	return cons;
}

void csa_ref_free(void *p) { free(p); }

/* ---- rotation finder (csamsa.c:analyzeTree) ------------------------------------------- */

#include "gencycsuffixtrees.h"   /* treenode, buildGeneralizedTree */
#include "nodeslinkedlists.h"    /* linkedblock */

extern struct _linkedblock *blockslist;      /* csamsa.c:36 */
extern int maxinterval, minblocksize, maxblocksize;
void analyzeTree(void);                      /* csamsa.c:271 */

/*
 * Build the reference's generalized cyclic suffix tree over the given sequences and run its
 * analysis (csamsa.c:271-308).  rot[s] receives rotations[s] (csamsa.c:260-267).  dump (may
 * be NULL, capacity dumpcap ints) receives the analysed block list in list order, per block:
 *   depth, size, totalsize, index of nextblock in this list (-1), positions[0..nseq)
 * *nblocks receives the number of blocks.  The texts are copied (the tree code frees the
 * text of a discarded identical rotation).  NOTE: the reference calls exit() when no common
 * unique block exists; callers must feed related sequences.
 */
int csa_ref_rotations(int nseq, const char **txts, const int *sizes, int *rot, int *dump, int dumpcap, int *nblocks)
{
	int s, saved, n = 0, used = 0;
	struct _linkedblock *b, *c;

	numberofseqs = nseq;
	texts = (char **)calloc(64, sizeof(char *));
	descs = (char **)calloc(64, sizeof(char *));
	textsizes = (int *)calloc(64, sizeof(int));
	for (s = 0; s < nseq; s++) {
		texts[s] = strdup(txts[s]);
		descs[s] = strdup("seq");
		textsizes[s] = sizes[s];
	}
	minblocksize = 10;
	maxblocksize = 0x7fffffff;
	maxinterval = 0x7fffffff;
	rotations = NULL;
	saved = quiet_begin();
	buildGeneralizedTree();
	analyzeTree();
	quiet_end(saved);
	if (numberofseqs != nseq || rotations == NULL) return -1;     /* a sequence was discarded */
	for (s = 0; s < nseq; s++) rot[s] = rotations[s];
	for (b = blockslist; b != NULL; b = b->next) {
		if (dump != NULL && used + 4 + nseq <= dumpcap) {
			int idx = -1, j = 0;
			for (c = blockslist; c != NULL; c = c->next, j++)
				if (c == b->nextblock) { idx = j; break; }
			dump[used++] = b->item->depth;
			dump[used++] = b->size;
			dump[used++] = b->totalsize;
			dump[used++] = idx;
			for (s = 0; s < nseq; s++) dump[used++] = b->positions ? b->positions[s] : -1;
		}
		n++;
	}
	if (nblocks) *nblocks = n;
	return 0;
}

/* ---- anchor map (alignment.c:RunAlignment) -------------------------------------------- */

#include "morenodeslinkedlists.h"  /* bordernode, linkedpos, firstbordernode */

void PrepareTreeForAlignment(void);          /* alignment.c:69 */
void RunAlignment(void);                     /* alignment.c:163 */
void SaveAlignment(char *outputfilename);    /* alignment.c:88 */

/*
 * Run the reference's N-mode alignment stage on the given sequences: build the cyclic tree, take
 * the rotations from analyzeTree (given_rot == NULL) or use the given ones, prepare the border
 * nodes (alignment.c:69-86), run the anchor loop (alignment.c:163-214, including the reference's
 * own ProgressiveDP for the gaps) and report
 *   border (capacity bcap ints): the border-node list right after PrepareTreeForAlignment, per
 *          node: size, then for every sequence: count, positions...
 *   segs   (capacity scap ints): the final alignment map from firstsegment to lastsegment, per
 *          segment: size, 1 if the gap after it was filled by DP else 0, positions[0..nseq)
 * savepath (may be NULL): SaveAlignment output file.
 */
int csa_ref_alignment_map(int nseq, const char **txts, const int *sizes, const int *given_rot, int *rot_out,
                          int *border, int bcap, int *nborder, int *segs, int scap, int *nsegs, const char *savepath)
{
	int s, saved, used, n;
	bordernode *b;
	linkedpos *p;
	alignmapsegment *g;

	if (orderedseqs) { free(orderedseqs); orderedseqs = NULL; }
	if (seqlengths) { free(seqlengths); seqlengths = NULL; }
	numberofseqs = nseq;
	texts = (char **)calloc(64, sizeof(char *));
	descs = (char **)calloc(64, sizeof(char *));
	textsizes = (int *)calloc(64, sizeof(int));
	for (s = 0; s < nseq; s++) {
		char name[32];
		snprintf(name, sizeof(name), "seq%d", s);
		texts[s] = strdup(txts[s]);
		descs[s] = strdup(name);
		textsizes[s] = sizes[s];
	}
	minblocksize = 10;
	maxblocksize = 0x7fffffff;
	maxinterval = 0x7fffffff;
	rotations = NULL;
	saved = quiet_begin();
	buildGeneralizedTree();
	if (numberofseqs != nseq) { quiet_end(saved); return -1; }
	if (given_rot == NULL) {
		analyzeTree();
	} else {
		rotations = (int *)calloc((size_t)nseq, sizeof(int));
		for (s = 0; s < nseq; s++) rotations[s] = given_rot[s];
	}
	if (rotations == NULL) { quiet_end(saved); return -1; }
	for (s = 0; s < nseq; s++) rot_out[s] = rotations[s];
	PrepareTreeForAlignment();
	used = 0; n = 0;
	for (b = firstbordernode->next; b != NULL; b = b->next) {
		int need = 1 + nseq;
		for (s = 0; s < nseq; s++) for (p = b->positions[s]; p != NULL; p = p->next) need++;
		if (border != NULL && used + need <= bcap) {
			border[used++] = b->size;
			for (s = 0; s < nseq; s++) {
				int at = used++, c = 0;
				for (p = b->positions[s]; p != NULL; p = p->next) { border[used++] = p->k; c++; }
				border[at] = c;
			}
		}
		n++;
	}
	if (nborder) *nborder = n;
	RunAlignment();
	used = 0; n = 0;
	for (g = firstsegment; g != NULL; g = g->next) {
		if (segs != NULL && used + 2 + nseq <= scap) {
			segs[used++] = g->size;
			segs[used++] = (g->alignedstrings != NULL) ? 1 : 0;
			for (s = 0; s < nseq; s++) segs[used++] = g->positions[s];
		}
		n++;
	}
	if (nsegs) *nsegs = n;
	if (savepath != NULL) SaveAlignment((char *)savepath);
	quiet_end(saved);
	return 0;
}
