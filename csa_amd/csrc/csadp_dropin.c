/*
 * csadp_dropin.c -- source-compatible replacement for the reference's
 * `void ProgressiveDP(alignmapsegment *segment)` (dynamicprogramming.c:906,
 * declared dynamicprogramming.h:3, called from RunAlignment alignment.c:201).
 *
 * Packs the reference-shaped globals into csadp_tasks, runs them on the GPU through
 * libcsadp.so and hands the malloc'd strings to the segments (freed later by
 * DeleteAlignmentMap, alignmentmap.c:174-179).  Prints the reference's progress tokens
 * (:917, :1156, :689, :1159) so logs diff cleanly.
 *
 * Two modes (include/csa_dropin.h):
 *   synchronous  every call is one one-task batch, finished when the call returns -- what
 *                the reference's caller sees today; the default.
 *   deferred     a call records its gap and returns; csadp_dropin_finish() submits ALL
 *                recorded gaps as ONE csadp_align_batch.  Nothing in RunAlignment
 *                (alignment.c:169-214) reads a gap's strings; their first reader is
 *                SaveAlignment (alignment.c:134-156), so one finish in front of it keeps
 *                the program's files and -- RunAlignment prints nothing of its own -- its
 *                stdout byte for byte.  Linking the program with
 *                -Wl,--wrap=SaveAlignment switches this mode on without a source change:
 *                __wrap_SaveAlignment below finishes, then calls the real one.
 *
 * The service thread.  Everything that touches the library runs on ONE thread the adapter owns; the program's
 * thread posts a request and waits.  Two reasons, both measured on the reference program (profiles/r05_dropin_*):
 *   start-up   the program spends its first half second on the host (LoadSequences, the suffix tree, the rotations:
 *              csamsa.c:592-613) before RunAlignment reaches its first gap; HIP's start-up, the load of the kernels'
 *              code objects and the first allocations take 0.3 s.  A constructor starts the thread when the program
 *              starts and it runs csadp_init + csadp_warmup at once (CSADP_DROPIN_EARLY_INIT=0: at the first call).
 *   the heap   by the time of its first gap the program's malloc arena is 150 MB of suffix-tree nodes, 70 MB of them
 *              freed: a malloc of 64 bytes on that thread took 1 ms at some gaps, one of 2 KB 35 ms at the first
 *              (glibc consolidates the free lists of the whole arena).  A batch allocates thousands of strings and
 *              vectors; on the service thread they come from a fresh arena of its own.  The result strings are
 *              libc-malloc'd all the same and the program free()s them as before.  For the same reason a recorded
 *              gap costs no malloc on the program's thread (its bounds live in blocks the adapter maps itself).
 * A failure of the start-up is reported only if a gap is ever computed (mode R never calls ProgressiveDP:
 * csamsa.c:607-613).
 *
 * Error policy: the reference's ProgressiveDP is void and checks nothing (unchecked malloc,
 * dynamicprogramming.c:964-981); a failed task has no representation its caller could act
 * on.  The adapter prints csadp_strerror() to stderr and exit(2)s.  It never falls back to
 * a CPU computation.
 */
#define _GNU_SOURCE                  /* clock_gettime, MAP_ANONYMOUS under -std=c11 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

#include "csa_dropin.h"
#include "csadp.h"

/* resolved by the linker only under -Wl,--wrap=SaveAlignment; NULL otherwise */
extern void __real_SaveAlignment(char *outputfilename) __attribute__((weak));

typedef struct {
	struct _alignmapsegment *segment;
	int *starts, *ends;
	int mingap, maxgap;
} pending_gap;

static pending_gap *pending;
static int npending, cappending;
static int defer_mode = -1;              /* -1: not decided yet */

/* CSADP_DROPIN_STATS=<file>: one JSON line about the adapter's own time, appended at exit */
static struct {
	int calls, batches, registered;
	double in_calls_s, first_call_s, finish_s, init_wait_s, init_thread_s;
} stats;

static double now_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* ---- memory of the recorded gaps: mapped blocks, never the program's malloc arena ---------------------------- */
static char *block;
static size_t block_left;

static void *block_alloc(size_t bytes)
{
	void *p;
	bytes = (bytes + 15) & ~(size_t)15;
	if (bytes > block_left) {
		size_t want = bytes > ((size_t)1 << 20) ? bytes : (size_t)1 << 20;
		block = (char *)mmap(NULL, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (block == (char *)MAP_FAILED) { fprintf(stderr, "csadp drop-in: out of memory\n"); exit(2); }
		block_left = want;            /* blocks live as long as the process: 128 bytes per gap of 16 sequences */
	}
	p = block;
	block += bytes;
	block_left -= bytes;
	return p;
}

/* ---- the service thread ---------------------------------------------------------------------------------------- */
static pthread_t service;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t cv_req = PTHREAD_COND_INITIALIZER, cv_done = PTHREAD_COND_INITIALIZER;
static int service_started, service_quit, init_done, init_rc;
static struct { const csadp_task *tasks; int n; csadp_result *res; int rc, posted, done; } req;

static void *service_main(void *arg)
{
	const double t0 = now_s();
	int rc;
	(void)arg;
	rc = csadp_init(NULL);
	if (rc == CSADP_OK) rc = csadp_warmup();
	pthread_mutex_lock(&mu);
	init_rc = rc;
	init_done = 1;
	stats.init_thread_s = now_s() - t0;
	pthread_cond_broadcast(&cv_done);
	for (;;) {
		while (!req.posted && !service_quit) pthread_cond_wait(&cv_req, &mu);
		if (!req.posted) break;
		req.posted = 0;
		pthread_mutex_unlock(&mu);
		rc = csadp_align_batch(req.tasks, req.n, req.res);
		pthread_mutex_lock(&mu);
		req.rc = rc;
		req.done = 1;
		pthread_cond_broadcast(&cv_done);
	}
	pthread_mutex_unlock(&mu);
	return NULL;
}

static void service_stop(void)
{
	if (!service_started) return;
	pthread_mutex_lock(&mu);
	service_quit = 1;
	pthread_cond_broadcast(&cv_req);
	pthread_mutex_unlock(&mu);
	pthread_join(service, NULL);      /* never leave main() while the runtime is still coming up on another thread */
	service_started = 0;
}

static void service_start(void)
{
	if (service_started) return;
	if (pthread_create(&service, NULL, service_main, NULL) != 0) { fprintf(stderr, "csadp drop-in: cannot start a thread\n"); exit(2); }
	service_started = 1;
	atexit(service_stop);
}

__attribute__((constructor)) static void early_start(void)
{
	const char *e = getenv("CSADP_DROPIN_EARLY_INIT");
	if (e && *e && atoi(e) == 0) return;
	service_start();
}

static void write_stats(void)
{
	const char *path = getenv("CSADP_DROPIN_STATS");
	FILE *f;
	if (!path || !*path) return;
	f = fopen(path, "a");
	if (!f) return;
	fprintf(f, "{\"mode\": \"%s\", \"calls\": %d, \"batches\": %d, \"init_s\": %.6f, \"early_thread_s\": %.6f, \"in_calls_s\": %.6f, "
	           "\"first_call_s\": %.6f, \"finish_s\": %.6f, \"dp_s\": %.6f}\n",
	        defer_mode == 1 ? "deferred" : "synchronous", stats.calls, stats.batches, stats.init_wait_s, stats.init_thread_s, stats.in_calls_s,
	        stats.first_call_s, stats.finish_s, stats.in_calls_s + stats.finish_s);
	fclose(f);
}

static void die(int rc, const char *what)
{
	fprintf(stderr, "\ncsadp drop-in: %s failed: %s\n", what, csadp_strerror(rc));
	exit(2);
}

static void decide_mode(void)
{
	const char *e;
	if (defer_mode >= 0) return;
	defer_mode = __real_SaveAlignment != NULL;         /* the link flag alone switches it on */
	e = getenv("CSADP_DROPIN_DEFER");
	if (e && *e) defer_mode = atoi(e) != 0 && __real_SaveAlignment != NULL;   /* the variable can only switch it OFF: without
	                                                                              the wrap nobody would call finish */
}

/* init_wait_s = what the first gap WAITS for: the rest of the service thread's start-up (usually nothing: it had the
 * suffix tree's half second), or the whole of it under CSADP_DROPIN_EARLY_INIT=0 */
static void first_use(void)
{
	double t0;
	if (stats.registered) return;
	stats.registered = 1;
	atexit(write_stats);
	t0 = now_s();
	service_start();
	pthread_mutex_lock(&mu);
	while (!init_done) pthread_cond_wait(&cv_done, &mu);
	pthread_mutex_unlock(&mu);
	stats.init_wait_s = now_s() - t0;
	if (init_rc != CSADP_OK) die(init_rc, "csadp_init");
}

static int run_batch(const csadp_task *tasks, int n, csadp_result *res)
{
	int rc;
	pthread_mutex_lock(&mu);
	req.tasks = tasks;
	req.n = n;
	req.res = res;
	req.done = 0;
	req.posted = 1;
	pthread_cond_signal(&cv_req);
	while (!req.done) pthread_cond_wait(&cv_done, &mu);
	rc = req.rc;
	pthread_mutex_unlock(&mu);
	stats.batches++;
	return rc;
}

void csadp_dropin_defer(int on)
{
	if (!on && npending) csadp_dropin_finish();
	defer_mode = on != 0;
}

static void fill_task(csadp_task *task, int *starts, int *ends)
{
	task->nseq = numberofseqs;
	task->texts = (const char *const *)texts;
	task->textsizes = textsizes;
	task->rotations = rotations;
	task->starts = starts;
	task->ends = ends;
}

static void publish(struct _alignmapsegment *segment, csadp_result *res)
{
	fputs(res->progress ? res->progress : "", stdout);                  /* :1156 '.' per fill, :689 '!' per all-gap column */
	free(res->progress);
	printf("->%4d]\n", res->consensus);                                 /* :1159 */
	segment->alignedstrings = res->aligned;                             /* :1160 */
}

int csadp_dropin_finish(void)
{
	csadp_task *tasks;
	csadp_result *res;
	double t0;
	int g, rc, n = npending;

	if (n == 0) return 0;
	t0 = now_s();
	tasks = (csadp_task *)block_alloc((size_t)n * sizeof(csadp_task));
	res = (csadp_result *)block_alloc((size_t)n * sizeof(csadp_result));
	for (g = 0; g < n; g++) fill_task(&tasks[g], pending[g].starts, pending[g].ends);
	rc = run_batch(tasks, n, res);
	for (g = 0; g < n && rc == CSADP_OK; g++) rc = res[g].status;
	if (rc != CSADP_OK) die(rc, "ProgressiveDP (deferred batch)");
	for (g = 0; g < n; g++) {                                           /* in call order: the log reads as the reference's */
		printf("[(%-4d-%4d)", pending[g].mingap, pending[g].maxgap);    /* :917 */
		publish(pending[g].segment, &res[g]);
	}
	fflush(stdout);
	npending = 0;
	stats.finish_s += now_s() - t0;
	return n;
}

void __wrap_SaveAlignment(char *outputfilename)
{
	csadp_dropin_finish();
	__real_SaveAlignment(outputfilename);
}

void ProgressiveDP(struct _alignmapsegment *segment)
{
	csadp_task task;
	csadp_result res;
	int *starts, *ends;
	double t0;
	int s, rc;

	if (segment->maxgapsize == 0) return;                               /* :916 */
	decide_mode();
	first_use();
	t0 = now_s();
	starts = (int *)block_alloc(2 * (size_t)numberofseqs * sizeof(int));
	ends = starts + numberofseqs;
	for (s = 0; s < numberofseqs; s++) {
		starts[s] = segment->positions[s] + segment->size;              /* :288, :936 */
		ends[s] = segment->next->positions[s];                          /* :1069 */
	}
	if (defer_mode == 1) {
		if (npending == cappending) {
			const int cap = cappending ? 2 * cappending : 1024;
			pending_gap *grown = (pending_gap *)block_alloc((size_t)cap * sizeof(pending_gap));
			if (npending) memcpy(grown, pending, (size_t)npending * sizeof(pending_gap));
			pending = grown;
			cappending = cap;
		}
		pending[npending].segment = segment;
		pending[npending].starts = starts;
		pending[npending].ends = ends;
		pending[npending].mingap = segment->mingapsize;
		pending[npending].maxgap = segment->maxgapsize;
		npending++;
	} else {
		printf("[(%-4d-%4d)", segment->mingapsize, segment->maxgapsize);/* :917: shown while the gap is being worked on */
		fflush(stdout);
		fill_task(&task, starts, ends);
		rc = run_batch(&task, 1, &res);
		if (rc == CSADP_OK) rc = res.status;
		if (rc != CSADP_OK) die(rc, "ProgressiveDP");
		publish(segment, &res);
		fflush(stdout);
	}
	stats.calls++;
	t0 = now_s() - t0;
	stats.in_calls_s += t0;
	if (stats.calls == 1) stats.first_call_s = t0;
	if (getenv("CSADP_DROPIN_TRACE")) fprintf(stderr, "csadp drop-in: call %d  %.3f ms\n", stats.calls, t0 * 1e3);
}
