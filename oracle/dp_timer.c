/*
 * dp_timer.c -- TEST / BENCH INFRASTRUCTURE ONLY.  A clock around the reference's own ProgressiveDP
 * (dynamicprogramming.c:906), linked into the UNMODIFIED reference program with -Wl,--wrap=ProgressiveDP
 * (oracle/Makefile: CSA_ref_timed).  CSADP_DROPIN_STATS=<file>: one JSON line at exit, the same fields the
 * csadp drop-in writes (csa_amd/csrc/csadp_dropin.c), so bench.py reads both with one parser.
 */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "alignmentmap.h"      /* the reference's own header (-I$(REF)): the wrapper only looks at maxgapsize, to count the calls that compute */

void __real_ProgressiveDP(struct _alignmapsegment *segment);

static int calls, registered;
static double in_calls_s, first_call_s;

static void write_stats(void)
{
	const char *path = getenv("CSADP_DROPIN_STATS");
	FILE *f;
	if (!path || !*path) return;
	f = fopen(path, "a");
	if (!f) return;
	fprintf(f, "{\"mode\": \"reference\", \"calls\": %d, \"batches\": %d, \"init_s\": 0.0, \"early_thread_s\": 0.0, \"in_calls_s\": %.6f, "
	           "\"first_call_s\": %.6f, \"finish_s\": 0.0, \"dp_s\": %.6f}\n", calls, calls, in_calls_s, first_call_s, in_calls_s);
	fclose(f);
}

void __wrap_ProgressiveDP(struct _alignmapsegment *segment)
{
	struct timespec a, b;
	double d;
	if (!registered) { registered = 1; atexit(write_stats); }
	clock_gettime(CLOCK_MONOTONIC, &a);
	__real_ProgressiveDP(segment);
	clock_gettime(CLOCK_MONOTONIC, &b);
	d = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
	in_calls_s += d;
	if (segment->maxgapsize == 0) return;           /* dynamicprogramming.c:916: nothing was done */
	if (calls++ == 0) first_call_s = d;
}
