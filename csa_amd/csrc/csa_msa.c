/*
 * csa_msa.c -- command-line front end over libcsadp.so for the reference's alignment modes:
 *
 *   csa_msa N <input.fasta>    rotate + align   (csamsa.c:606-623, mode N)
 *   csa_msa A <input.fasta>    align as given   (mode A: rotations all zero)
 *
 * Writes "<input without extension>-Rotated.fasta" (mode N) and "...-Aligned.fasta" beside the
 * input, named as the reference names them (csamsa.c:44-58), with the same bytes.  Everything
 * between the FASTA reader and the writers happens inside the C-ABI (include/csadp.h).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "csadp.h"

static char *out_name(const char *input, const char *suffix)
{
	size_t n = strlen(input), cut = n, i;
	char *name;
	for (i = n; i-- > 1;)
		if (input[i] == '.') { cut = i; break; }
	name = (char *)calloc(cut + strlen(suffix) + 1, 1);
	if (!name) return NULL;
	memcpy(name, input, cut);
	strcat(name, suffix);
	return name;
}

static double now_ms(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

int main(int argc, char **argv)
{
	char **texts = NULL, **descs = NULL, **rows = NULL, *rotated = NULL, *aligned = NULL;
	int *sizes = NULL, *rot = NULL, nseq = 0, rc, s;
	csadp_msa_stats st;
	double t0;

	if (argc != 3 || (strcmp(argv[1], "N") != 0 && strcmp(argv[1], "A") != 0)) {
		fprintf(stderr, "usage: %s N|A <input.fasta>\n", argv[0]);
		return 2;
	}
	rc = csadp_load_fasta(argv[2], &texts, &descs, &sizes, &nseq);
	if (rc != CSADP_OK) { fprintf(stderr, "csa_msa: cannot load %s: %s\n", argv[2], csadp_strerror(rc)); return 1; }
	rot = (int *)calloc((size_t)nseq, sizeof(int));
	rotated = out_name(argv[2], "-Rotated.fasta");
	aligned = out_name(argv[2], "-Aligned.fasta");
	if (!rot || !rotated || !aligned) return 1;

	t0 = now_ms();
	rc = csadp_msa(nseq, (const char *const *)texts, sizes, argv[1][0] == 'A' ? rot : NULL, rot, &rows, &st);
	if (rc != CSADP_OK) { fprintf(stderr, "csa_msa: %s\n", csadp_strerror(rc)); return 1; }
	t0 = now_ms() - t0;
	if (argv[1][0] == 'N')
		rc = csadp_write_rotated_fasta(rotated, (const char *const *)descs, (const char *const *)texts, sizes, rot, nseq);
	if (rc == CSADP_OK)
		rc = csadp_write_aligned_fasta(aligned, (const char *const *)descs, rot, (const char *const *)rows, nseq);
	if (rc != CSADP_OK) { fprintf(stderr, "csa_msa: cannot write outputs: %s\n", csadp_strerror(rc)); return 1; }

	printf("> %d sequences, rotations:", nseq);
	for (s = 0; s < nseq; s++) printf(" %d", rot[s]);
	printf("\n> Alignment size: %d (%d alignment segments)\n", st.alignment_length, st.segments - 1);
	printf("> %d border nodes, %d gaps by DP, %d fills, %lld cells\n", st.border_nodes, st.dp_gaps, st.fills, st.cells);
	printf("> ms: rotations %.1f  anchors %.1f  dp %.1f  rows %.1f  total %.1f\n", st.rotations_ms, st.anchors_ms, st.dp_ms,
	       st.rows_ms, t0);
	printf("> Done!\n");
	csadp_free_rows(rows, nseq);
	csadp_free_fasta(texts, descs, sizes, nseq);
	free(rot);
	free(rotated);
	free(aligned);
	csadp_shutdown();
	return 0;
}
