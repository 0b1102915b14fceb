/*
 * csa_pairs.c -- C harness over the C-ABI: reads a
 * multi-FASTA with the reference's loader rules, takes rotation offsets (from a list or from
 * the headers of the reference's own "<base>-Rotated.fasta"), enumerates sequence pairs and
 * aligns every pair as a whole-sequence 2-sequence ProgressiveDP task (the route of
 * alignment.c:173-178) on the GPU.  Prints, per pair: alignment length, sum-of-pairs score
 * (tools.c:274-280) and the FNV-1a digest of the two aligned strings.
 *
 *   csa_pairs <input.fasta> [--rot r0,r1,...] [--rotated <base>-Rotated.fasta] [--find-rotations]
 *             [--pair a,b] [--write-rotated out.fasta] [--gpus N]
 *
 * --gpus N (N > 1): the pairs are one batch over N GPUs of this node (csadp_align_batch_multi:
 * longest-processing-time partition, one host thread per GPU, results gathered in host memory).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "csadp.h"

static double now_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static unsigned fnv1a2(const char *a, const char *b)
{
	unsigned h = 0x811c9dc5u;
	const unsigned char *p;
	for (p = (const unsigned char *)a; *p; p++) { h ^= *p; h *= 0x01000193u; }
	for (p = (const unsigned char *)b; *p; p++) { h ^= *p; h *= 0x01000193u; }
	return h;
}

static long long sp2(const char *a, const char *b)
{
	long long s = 0;
	for (; *a; a++, b++) {
		if (*a == '-' && *b == '-') continue;
		s += (*a == *b) ? 1 : -1;
	}
	return s;
}

static void die(const char *what, int rc)
{
	fprintf(stderr, "csa_pairs: %s: %s\n", what, csadp_strerror(rc));
	exit(2);
}

int main(int argc, char **argv)
{
	char **texts, **descs;
	int *sizes, *rot;
	int nseq = 0, i, a, b, rc, only_a = -1, only_b = -1, npairs = 0, p;
	const char *rotated = NULL, *rotlist = NULL, *write_rot = NULL;
	int find_rot = 0, gpus = 1;
	csadp_task *tasks;
	csadp_result *res;
	csadp_pairbatch *batch;
	csadp_timing tm;
	int *zero2, *pa, *pb;
	double t0, t1;
	long long cells = 0;

	if (argc < 2) {
		fprintf(stderr, "usage: csa_pairs <input.fasta> [--rot r0,r1,...] [--rotated file] [--find-rotations] [--pair a,b] [--write-rotated out] [--gpus N]\n");
		return 1;
	}
	for (i = 2; i < argc; i++) {
		if (!strcmp(argv[i], "--rot") && i + 1 < argc) rotlist = argv[++i];
		else if (!strcmp(argv[i], "--rotated") && i + 1 < argc) rotated = argv[++i];
		else if (!strcmp(argv[i], "--write-rotated") && i + 1 < argc) write_rot = argv[++i];
		else if (!strcmp(argv[i], "--find-rotations")) find_rot = 1;
		else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--pair") && i + 1 < argc) { if (sscanf(argv[++i], "%d,%d", &only_a, &only_b) != 2) return 1; }
		else { fprintf(stderr, "csa_pairs: unknown argument %s\n", argv[i]); return 1; }
	}
	if ((rc = csadp_load_fasta(argv[1], &texts, &descs, &sizes, &nseq)) != CSADP_OK) die("load_fasta", rc);
	rot = (int *)calloc((size_t)nseq, sizeof(int));
	if (rotated) {
		int nread = 0;
		if ((rc = csadp_read_rotations(rotated, rot, nseq, &nread)) != CSADP_OK || nread != nseq) die("read_rotations", rc ? rc : CSADP_ERR_ARG);
	} else if (find_rot) {       /* the reference's mode R, computed natively (csadp_find_rotations) */
		csadp_rotation_info ri;
		double tr0 = now_s();
		if ((rc = csadp_find_rotations(nseq, (const char *const *)texts, sizes, rot, &ri)) != CSADP_OK) die("find_rotations", rc);
		printf("> rotation finder: %d blocks, heaviest chain %d (span %d), %.1f ms\n", ri.blocks, ri.chain_size, ri.chain_span,
		       (now_s() - tr0) * 1e3);
	} else if (rotlist) {
		const char *q = rotlist;
		for (i = 0; i < nseq && *q; i++) {
			rot[i] = atoi(q);
			while (*q && *q != ',') q++;
			if (*q == ',') q++;
		}
	}
	if (write_rot && (rc = csadp_write_rotated_fasta(write_rot, (const char *const *)descs, (const char *const *)texts, sizes, rot, nseq)) != CSADP_OK)
		die("write_rotated_fasta", rc);
	printf("> %d sequences, rotations:", nseq);
	for (i = 0; i < nseq; i++) printf(" %d", rot[i]);
	printf("\n");

	for (a = 0; a < nseq; a++)
		for (b = a + 1; b < nseq; b++)
			if (only_a < 0 || (a == only_a && b == only_b)) npairs++;
	if (npairs == 0) { fprintf(stderr, "csa_pairs: no such pair\n"); return 1; }
	tasks = (csadp_task *)calloc((size_t)npairs, sizeof(*tasks));
	res = (csadp_result *)calloc((size_t)npairs, sizeof(*res));
	pa = (int *)calloc((size_t)npairs, sizeof(int));
	pb = (int *)calloc((size_t)npairs, sizeof(int));
	zero2 = (int *)calloc(2, sizeof(int));
	p = 0;
	for (a = 0; a < nseq; a++) {
		for (b = a + 1; b < nseq; b++) {
			const char **tx;
			int *sz, *rt, *en;
			if (!(only_a < 0 || (a == only_a && b == only_b))) continue;
			tx = (const char **)calloc(2, sizeof(char *));
			sz = (int *)calloc(2, sizeof(int));
			rt = (int *)calloc(2, sizeof(int));
			en = (int *)calloc(2, sizeof(int));
			tx[0] = texts[a]; tx[1] = texts[b];
			sz[0] = en[0] = sizes[a]; sz[1] = en[1] = sizes[b];
			rt[0] = rot[a]; rt[1] = rot[b];
			tasks[p].nseq = 2; tasks[p].texts = tx; tasks[p].textsizes = sz; tasks[p].rotations = rt;
			tasks[p].starts = zero2; tasks[p].ends = en;
			pa[p] = a; pb[p] = b;
			cells += (long long)sizes[a] * sizes[b];
			p++;
		}
	}
	if (gpus > 1) {
		csadp_multi_stats ms;
		t0 = now_s();
		if ((rc = csadp_align_batch_multi(tasks, npairs, res, NULL, gpus, &ms)) != CSADP_OK) die("align_batch_multi", rc);
		t1 = now_s();
		for (p = 0; p < npairs; p++) {
			if (res[p].status != CSADP_OK) die("pair", res[p].status);
			printf("pair %d %d len %d SP %lld score %d fnv1a %08x\n", pa[p], pb[p], res[p].consensus,
			       sp2(res[p].aligned[0], res[p].aligned[1]), res[p].score, fnv1a2(res[p].aligned[0], res[p].aligned[1]));
			csadp_free_result(&res[p], 2);
		}
		for (i = 0; i < ms.ndevices; i++)
			printf("> gpu %d: %d pairs, %lld cells, %.1f ms\n", i, ms.tasks[i], ms.cost[i], ms.ms[i]);
		printf("> %d pairs, %lld cells over %d GPUs: imbalance %.3f, host-to-host %.1f ms = %.1f GCUPS\n", npairs, cells, gpus,
		       ms.total_cost ? (double)ms.max_cost * ms.ndevices / (double)ms.total_cost : 1.0, (t1 - t0) * 1e3,
		       (double)cells / ((t1 - t0) * 1e9));
		csadp_shutdown();
		csadp_free_fasta(texts, descs, sizes, nseq);
		return 0;
	}
	setenv("CSADP_SLOTS", "1", 0);      /* one pass only: no need for pipelined scratch sets */
	if ((rc = csadp_init(NULL)) != CSADP_OK) die("init", rc);
	t0 = now_s();
	if ((rc = csadp_pairs_create(tasks, npairs, &batch)) != CSADP_OK) die("pairs_create", rc);
	if ((rc = csadp_pairs_run(batch)) != CSADP_OK) die("pairs_run", rc);
	if ((rc = csadp_pairs_sync(batch)) != CSADP_OK) die("pairs_sync", rc);
	if ((rc = csadp_pairs_timing(batch, &tm)) != CSADP_OK) die("pairs_timing", rc);
	if ((rc = csadp_pairs_fetch(batch, res)) != CSADP_OK) die("pairs_fetch", rc);
	t1 = now_s();
	for (p = 0; p < npairs; p++) {
		if (res[p].status != CSADP_OK) die("pair", res[p].status);
		printf("pair %d %d len %d SP %lld score %d fnv1a %08x\n", pa[p], pb[p], res[p].consensus,
		       sp2(res[p].aligned[0], res[p].aligned[1]), res[p].score, fnv1a2(res[p].aligned[0], res[p].aligned[1]));
		csadp_free_result(&res[p], 2);
	}
	printf("> %d pairs, %lld cells: device fill %.3f ms + traceback %.3f ms = %.1f GCUPS; host-to-host %.1f ms = %.1f GCUPS\n",
	       npairs, cells, tm.fill_ms, tm.traceback_ms, (double)cells / (tm.total_ms * 1e6), (t1 - t0) * 1e3,
	       (double)cells / ((t1 - t0) * 1e9));
	csadp_pairs_destroy(batch);
	csadp_shutdown();
	csadp_free_fasta(texts, descs, sizes, nseq);
	return 0;
}
