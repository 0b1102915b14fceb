/*
 * dropin_driver.c -- test program for the adapter's explicit deferred mode (include/csa_dropin.h:
 * csadp_dropin_defer / csadp_dropin_finish), for callers who prefer a source line to the link flag.
 * Plays the part of the CSA program: defines the globals the adapter reads (csamsa.h:8-12), builds a
 * chain of alignment-map segments over the sequences of a FASTA-like input and calls ProgressiveDP on
 * every gap, as RunAlignment does (alignment.c:179-206).  Built and run by tests/test_gpu_dropin.py.
 *
 *   dropin_driver <deferred 0|1> <nseq> <ngaps> <seq 1> ... <seq nseq>
 * The gaps split every sequence into ngaps consecutive regions of (nearly) equal length.
 * Prints one line per gap and sequence: "<gap> <seq> <aligned string>".
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "csa_dropin.h"

int numberofseqs;
char **texts;
int *textsizes;
int *rotations;

int main(int argc, char **argv)
{
	int deferred, ngaps, g, s;
	alignmapsegment *segs;
	if (argc < 5) return 64;
	deferred = atoi(argv[1]);
	numberofseqs = atoi(argv[2]);
	ngaps = atoi(argv[3]);
	if (argc != 4 + numberofseqs) return 64;
	texts = argv + 4;
	textsizes = (int *)calloc((size_t)numberofseqs, sizeof(int));
	rotations = (int *)calloc((size_t)numberofseqs, sizeof(int));
	for (s = 0; s < numberofseqs; s++) {
		textsizes[s] = (int)strlen(texts[s]);
		rotations[s] = (7 * s) % textsizes[s];
	}
	/* segment g = an "anchor" of size 0 at the boundary in front of region g; the last one closes the chain */
	segs = (alignmapsegment *)calloc((size_t)ngaps + 1, sizeof(alignmapsegment));
	for (g = 0; g <= ngaps; g++) {
		segs[g].positions = (int *)calloc((size_t)numberofseqs, sizeof(int));
		for (s = 0; s < numberofseqs; s++) segs[g].positions[s] = (int)((long long)textsizes[s] * g / (ngaps > 0 ? ngaps : 1));
		segs[g].next = g < ngaps ? &segs[g + 1] : NULL;
	}
	for (g = 0; g < ngaps; g++) {
		int mn = 1 << 30, mx = 0;
		for (s = 0; s < numberofseqs; s++) {
			const int len = segs[g + 1].positions[s] - segs[g].positions[s];
			if (len < mn) mn = len;
			if (len > mx) mx = len;
		}
		segs[g].mingapsize = mn;
		segs[g].maxgapsize = mx;
	}
	if (deferred) csadp_dropin_defer(1);
	for (g = 0; g < ngaps; g++) ProgressiveDP(&segs[g]);
	if (deferred) {
		for (g = 0; g < ngaps; g++)
			if (segs[g].alignedstrings != NULL) return 3;       /* nothing is computed before the finish */
		if (csadp_dropin_finish() != ngaps) return 4;
		if (csadp_dropin_finish() != 0) return 5;
	}
	for (g = 0; g < ngaps; g++)
		for (s = 0; s < numberofseqs; s++)
			printf("%d %d %s\n", g, s, segs[g].alignedstrings ? segs[g].alignedstrings[s] : "(null)");
	return 0;
}
