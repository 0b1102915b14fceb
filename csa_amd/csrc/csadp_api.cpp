/*
 * csadp_api.cpp -- the C-ABI of libcsadp.so (include/csadp.h).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "csadp.h"
#include "csadp_debug.h"
#include <hip/hip_runtime_api.h>

#include "csadp_config.h"
#include "csadp_engine.h"
#include "csadp_hostpar.h"
#include "csadp_kernels.h"
#include "csadp_progressive.h"

using csadp::config;
using csadp::Engine;
using csadp::FillBatch;
using csadp::Progressive;

namespace {

/* Per-task host work (validation, table packing, traceback application, string building) is
 * independent across tasks: spread it over the process' persistent pool of host threads (csadp_hostpar.h),
 * at most 16 of them here: more only adds allocator contention (measured); n small -> run inline. */
constexpr int kNoDeviceIo = 1;    /* pairs_create_io: use the host-I/O path instead (never leaves this file) */

template <class F>
void parallel_for(int n, F &&fn)
{
	if (n < 4) {
		for (int i = 0; i < n; ++i) fn(i);
		return;
	}
	csadp::host_parallel_for(n, fn, 16);
}

}  // namespace

extern "C" {

int csadp_version(void) { return CSADP_VERSION; }

const char *csadp_strerror(int code)
{
	switch (code) {
	case CSADP_OK: return "ok";
	case CSADP_ERR_ARG: return "invalid argument";
	case CSADP_ERR_ALPHABET: return "region holds a letter other than A, C, G, T";
	case CSADP_ERR_NOMEM: return "out of host memory";
	case CSADP_ERR_NO_DEVICE: return "no usable gfx950 HIP device (there is no CPU fallback)";
	case CSADP_ERR_HIP: return "HIP runtime or kernel error";
	case CSADP_ERR_RANGE: return "task exceeds the 32-bit score range or the device memory";
	case CSADP_ERR_STATE: return "call sequence error";
	default: return "unknown error";
	}
}

int csadp_init(const csadp_config *cfg)
{
	int rc = CSADP_OK;
	Engine::primary(cfg, &rc);
	return rc;
}

void csadp_shutdown(void) { Engine::shutdown_all(); }

int csadp_device_info(char *name, int namelen, int *compute_units)
{
	Engine *E = Engine::primary_if_ready();
	if (!E) return CSADP_ERR_NO_DEVICE;
	if (name && namelen > 0) snprintf(name, (size_t)namelen, "%s", E->name());
	if (compute_units) *compute_units = E->compute_units();
	return CSADP_OK;
}

int csadp_device_count(int *count)
{
	if (!count) return CSADP_ERR_ARG;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { *count = 0; return CSADP_ERR_NO_DEVICE; }
	*count = n;
	return CSADP_OK;
}

void csadp_free_result(csadp_result *r, int nseq)
{
	if (!r) return;
	if (!r->aligned) { free(r->progress); r->progress = NULL; return; }
	for (int s = 0; s < nseq; ++s) free(r->aligned[s]);
	free(r->aligned);
	r->aligned = NULL;
	free(r->progress);
	r->progress = NULL;
}

int csadp_free_results(csadp_result *results, int count, int nseq)
{
	if (!results) return 0;
	int failed = 0;
	for (int t = 0; t < count; ++t) {
		failed += results[t].status != CSADP_OK;
		csadp_free_result(&results[t], nseq);
	}
	return failed;
}

}  // extern "C"

namespace {

/* Bring every task to its next pending fill; fills with an empty profile (ncols == 0) need
 * no matrix and are completed on the host (the walk of :1037 never starts, :1115-1127 runs). */
bool advance(Progressive &p)
{
	while (p.next_fill()) {
		if (p.ncols() > 0) return true;
		if (p.apply_trace(nullptr, 0, p.nrows(), 0) != CSADP_OK) return false;   /* score = dpmatrix[nrows][0], :967 */
	}
	return false;
}

/* What the fills of one round leave for the host: per active task its walk (dynamicprogramming.c:1037-1047) */
struct RoundTrace {
	const uint8_t *ops = nullptr;
	int nops = 0, remj = 0, remk = 0;
	const int32_t *score = nullptr;          /* H[nrows][ncols] where the filler knows it */
};
/* the fills + walks of one round: the device (a FillBatch) in the product, the caller's filler in the CPU test seam */
typedef std::function<int(std::vector<Progressive> &, const std::vector<int> &, std::vector<RoundTrace> &, double *ms)> RoundFills;

/* the product's: tables -> one device batch -> op lists */
int device_fills(FillBatch &fb, std::vector<Progressive> &tasks, const std::vector<int> &active, std::vector<RoundTrace> &out, double *ms)
{
	auto tick = std::chrono::steady_clock::now();
	int phase = 0;
	auto lap = [&]() {
		const auto now = std::chrono::steady_clock::now();
		ms[phase++] = std::chrono::duration<double, std::milli>(now - tick).count();
		tick = now;
	};
	fb.clear();
	bool unit = true;
	for (int t : active) {
		fb.add(tasks[t].nrows(), tasks[t].ncols(), tasks[t].nprev(), tasks[t].border_i());
		unit = unit && tasks[t].unit_borders();
	}
	fb.allow_bits(unit);
	int rc = fb.layout();
	if (rc != CSADP_OK) return rc;
	lap();
	parallel_for((int)active.size(), [&](int j) {
		Progressive &p = tasks[active[(size_t)j]];
		if (fb.bits()) p.write_tables_bits(fb.bit_cols(j), fb.bit_nwords(j), fb.bit_rows(j), fb.bit_rowwords(j));
		else p.write_tables(fb.coltab(j), fb.leftc(j), fb.ncols_pad(j), fb.rowshift(j), fb.top(j), fb.wide());
	});
	lap();
	if ((rc = fb.upload()) != CSADP_OK) return rc;
	if ((rc = fb.run()) != CSADP_OK) return rc;
	if ((rc = fb.download()) != CSADP_OK) return rc;
	lap();
	for (size_t j = 0; j < active.size(); ++j) {
		const int32_t *sm = fb.summary((int)j);
		out[j].ops = fb.ops((int)j);
		out[j].nops = sm[0];
		out[j].remj = sm[1];
		out[j].remk = sm[2];
	}
	return CSADP_OK;
}

/* csadp_last_batch_phases: accumulated by the rounds of the last batch (any group's thread) */
std::mutex g_phases_mutex;
csadp_batch_phases g_phases;

/* one lock-step round: every task with a pending fill contributes one job */
int run_round(std::vector<Progressive> &tasks, const std::vector<int> &active, const RoundFills &fills, std::vector<int> &status)
{
	const bool trace = config().trace_host;
	auto tick = std::chrono::steady_clock::now();
	double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int phase = 0;
	auto lap = [&]() {
		const auto now = std::chrono::steady_clock::now();
		ms[phase++] = std::chrono::duration<double, std::milli>(now - tick).count();
		tick = now;
	};
	std::vector<RoundTrace> traces(active.size());
	{
		const int rc = fills(tasks, active, traces, ms);
		if (rc != CSADP_OK) return rc;
	}
	phase = 3;
	tick = std::chrono::steady_clock::now();
	parallel_for((int)active.size(), [&](int j) {
		const RoundTrace &T = traces[(size_t)j];
		const int a = tasks[active[(size_t)j]].apply_trace(T.ops, T.nops, T.remj, T.remk, T.score, true);
		if (a != CSADP_OK) status[active[(size_t)j]] = a;
	});
	/* DeleteGappedColumns of all tasks: the scoring of the candidate columns -- nearly all of its time -- as ONE flat
	 * list of (task, chunk of columns) items over all host threads, then the reference's sequential pass per task, which
	 * only re-scores what a slide has touched (csadp_progressive.cpp).  Per task it used to be one thread's work, and the
	 * largest gap of a run kept every round waiting for it. */
	lap();
	{
		/* items: (task, chunk >= 0) = speculate that chunk of a LARGE task; (task, -1) = the whole refinement of a small
		 * one, plain.  The passes of the large tasks follow -- there are a handful at most, so that loop usually runs on
		 * this thread without waking the pool a third time. */
		/* ... unless the round holds several tasks per host thread: the tasks themselves then fill the pool, and the plain pass of a task
		 * does less work than speculation + commit (a fifth of the candidates are scored twice) and needs no second region
		 * (N-sequence throughput batches, profiles/r05_profile_batch_sweep.txt) */
		const bool many_tasks = (int)active.size() >= 2 * std::min(csadp::HostPool::get().size(), 16);
		const int kLargeColumns = many_tasks ? 2000000000 : 1024;
		std::vector<std::pair<int, int>> items;
		std::vector<int> large;
		for (size_t j = 0; j < active.size(); ++j) {
			const int t = active[j];
			if (status[t] != CSADP_OK) continue;
			if (tasks[t].ncols() < kLargeColumns) { items.emplace_back(t, -1); continue; }
			const int chunks = tasks[t].refine_prepare();
			for (int c = 0; c < chunks; ++c) items.emplace_back(t, c);
			if (chunks > 0) large.push_back(t);
		}
		parallel_for((int)items.size(), [&](int i) {
			Progressive &p = tasks[items[(size_t)i].first];
			if (items[(size_t)i].second >= 0) p.refine_speculate(items[(size_t)i].second);
			else p.refine_commit();
		});
		lap();
		parallel_for((int)large.size(), [&](int i) { tasks[large[(size_t)i]].refine_commit(); });
	}
	lap();
	{
		std::lock_guard<std::mutex> lock(g_phases_mutex);
		++g_phases.rounds;
		g_phases.layout_ms += ms[0];
		g_phases.tables_ms += ms[1];
		g_phases.device_ms += ms[2];
		g_phases.apply_ms += ms[3];
		g_phases.refine_speculate_ms += ms[4];
		g_phases.refine_commit_ms += ms[5];
	}
	if (trace)
		fprintf(stderr, "csadp round: %3d jobs  layout %.2f  tables %.2f  device %.2f  apply %.2f  refine: speculate %.2f  commit %.2f ms\n",
		        (int)active.size(), ms[0], ms[1], ms[2], ms[3], ms[4], ms[5]);
	return CSADP_OK;
}

}  // namespace

/* geometry of a 2-sequence task on the device-I/O path: which sequence is the column one */
struct PairGeom {
	int col = 0;                 /* index (0/1) of the sequence that seeds the profile = the columns (:290-307) */
	int nrows = 0, ncols = 0;
};

struct csadp_pairbatch {
	explicit csadp_pairbatch(Engine *e) : fb(e) {}
	std::vector<Progressive> tasks;
	std::vector<int> status;
	std::vector<int> active;     /* tasks that own a job of the batch */
	FillBatch fb;
	bool ran = false, fetched = false;
	/* device-I/O path (csadp_pairio.hip): the letters are copied into the batch at create(), nothing of the
	 * caller's memory is referenced afterwards; tasks with an empty region never reach the device and are
	 * finished by the host logic (`tasks`, indexed like the batch) */
	bool device_io = false;
	std::vector<PairGeom> geom;
};

extern "C" {
static int pairs_create_io(csadp_pairbatch *b, const csadp_task *tasks, int ntasks, bool pipelined, bool strings);
}

namespace {

/* (task groups whose rounds run side by side: Config::round_groups, default 2 -- measured: DESIGN.md section 4) */

/* N independent ProgressiveDP calls as lock-step rounds over round groups; `fills_of(g)` = group g's fills, `prepare(groups)` readies
 * them, `enter()` runs first on every extra group's thread (the product binds its device there). */
int align_batch_rounds(const csadp_task *tasks, int ntasks, csadp_result *results, int max_groups, const std::function<int(int)> &prepare,
                       const std::function<RoundFills(int)> &fills_of, const std::function<int()> &enter)
{
	const bool trace = config().trace_host;
	const auto t_begin = std::chrono::steady_clock::now();
	auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
	{
		std::lock_guard<std::mutex> lock(g_phases_mutex);
		memset(&g_phases, 0, sizeof(g_phases));
		g_phases.tasks = ntasks;
	}
	std::vector<Progressive> prog((size_t)ntasks);
	std::vector<int> status((size_t)ntasks, CSADP_OK);
	parallel_for(ntasks, [&](int t) {
		memset(&results[t], 0, sizeof(results[t]));
		status[(size_t)t] = prog[(size_t)t].init(tasks[t]);
	});
	const double ms_init = since();
	/* Lock-step rounds: round i = step i of every task that has one.  A round is [tables -> device -> trace application,
	 * refinement], and the device idles through the host's part.  Tasks are independent, so the list is dealt (longest
	 * first) over up to config().round_groups groups, each driven through its own rounds by its own host thread on its own arena
	 * and stream: one group's host part runs under the others' device part (the pool serves one parallel region at a
	 * time; the groups' kernels are small enough to share the chip).  One task, or CSADP_ROUND_GROUPS=1: the plain loop. */
	auto drive = [&](const std::vector<int> &mine, const RoundFills &fills) -> int {
		for (;;) {
			std::vector<int> active;
			for (int t : mine)
				if (status[(size_t)t] == CSADP_OK && advance(prog[(size_t)t])) active.push_back(t);
			if (active.empty()) return CSADP_OK;
			const int rc = run_round(prog, active, fills, status);
			if (rc != CSADP_OK) return rc;
		}
	};
	std::vector<int> live;
	for (int t = 0; t < ntasks; ++t)
		if (status[(size_t)t] == CSADP_OK) live.push_back(t);
	int groups = 1;
	{
		/* two groups for the reference's own sets (a large gap alone beside the many small ones: DESIGN.md section 4); FOUR for batches of
		 * many tasks, whose rounds are device-bound -- the fill of one group then runs under the walks, the trace application and the
		 * refinement of three others (profiles/r05_profile_batch_sweep_colinear.txt: 512 families of 8 x 4 kbp 62-69 -> 48-52 ms, 16 of
		 * 16 x 16 kbp 50 -> 47; the example sets the same within their spread) */
		const int want = (!config().round_groups_forced && live.size() >= 128) ? 4 : config().round_groups;      /* 64 tasks: 11.2 ms with two groups, 12.0 with four */
		groups = std::max(1, std::min(std::min(want, max_groups), (int)live.size()));
	}
	{
		const int prc = prepare(groups);
		if (prc != CSADP_OK) return prc;
	}
	if (groups <= 1) {
		const int rc = drive(live, fills_of(0));
		if (rc != CSADP_OK) return rc;
	} else {
		std::vector<long long> cost(live.size());
		for (size_t i = 0; i < live.size(); ++i) cost[i] = std::max(1LL, csadp_task_cost(&tasks[live[i]]));
		std::vector<int> part(live.size());
		long long maxload = 0;
		int prc = csadp_partition_lpt(cost.data(), (int)live.size(), groups, part.data(), &maxload);
		if (prc != CSADP_OK) return prc;
		std::vector<std::vector<int>> mine((size_t)groups);
		for (size_t i = 0; i < live.size(); ++i) mine[(size_t)part[i]].push_back(live[i]);
		for (auto &m : mine) std::sort(m.begin(), m.end());
		std::vector<RoundFills> fills;
		for (int g = 0; g < groups; ++g) fills.push_back(fills_of(g));
		std::vector<int> grc((size_t)groups, CSADP_OK);
		std::vector<std::thread> threads;
		for (int g = 1; g < groups; ++g)
			threads.emplace_back([&, g] { grc[(size_t)g] = enter() == CSADP_OK ? drive(mine[(size_t)g], fills[(size_t)g]) : CSADP_ERR_HIP; });
		grc[0] = drive(mine[0], fills[0]);
		for (std::thread &th : threads) th.join();
		for (int g = 0; g < groups; ++g)
			if (grc[(size_t)g] != CSADP_OK) return grc[(size_t)g];
	}
	const double ms_rounds = since();
	parallel_for(ntasks, [&](int t) {
		if (status[(size_t)t] == CSADP_OK) status[(size_t)t] = prog[(size_t)t].finish(&results[t]);
		results[t].status = status[(size_t)t];
	});
	{
		std::lock_guard<std::mutex> lock(g_phases_mutex);
		g_phases.round_groups = groups;
		g_phases.seed_ms = ms_init;
		g_phases.results_ms = since() - ms_rounds;
		g_phases.wall_ms = since();
	}
	if (trace)
		fprintf(stderr, "csadp_align_batch: %d tasks in %d round group(s): seed %.2f  rounds %.2f  results %.2f ms\n", ntasks, groups, ms_init,
		        ms_rounds - ms_init, since() - ms_rounds);
	return CSADP_OK;
}

/* csadp_align_batch on one engine: lock-step rounds over the engine's cached arena(s) */
int align_batch_on(Engine *E, const csadp_task *tasks, int ntasks, csadp_result *results)
{
	/* a batch of 2-sequence tasks takes the device-I/O pair path: letters up, rows down */
	bool all_pairs = ntasks > 0 && config().bits && config().device_io;
	for (int t = 0; t < ntasks && all_pairs; ++t) all_pairs = tasks[t].nseq == 2;
	if (all_pairs) {
		csadp_pairbatch b(E);
		int rc = pairs_create_io(&b, tasks, ntasks, false, true);
		if (rc == CSADP_OK) rc = csadp_pairs_run(&b);
		if (rc == CSADP_OK) rc = csadp_pairs_fetch(&b, results);
		if (rc != kNoDeviceIo) return rc;
	}
	/* one FillBatch (arena, streams) per round group: the engine's cached one and its spares */
	std::lock_guard<std::mutex> lock(E->batch_mutex);
	auto prepare = [&](int groups) -> int {
		E->cells_sharers.store(groups, std::memory_order_relaxed);
		if (!E->cached_batch) E->cached_batch = new (std::nothrow) FillBatch(E);
		if (!E->cached_batch) return CSADP_ERR_NOMEM;
		if ((int)E->extra_batches.size() < groups - 1) E->extra_batches.resize((size_t)groups - 1, nullptr);
		for (int g = 1; g < groups; ++g) {
			if (!E->extra_batches[(size_t)g - 1]) E->extra_batches[(size_t)g - 1] = new (std::nothrow) FillBatch(E);
			if (!E->extra_batches[(size_t)g - 1]) return CSADP_ERR_NOMEM;
		}
		return CSADP_OK;
	};
	auto fills_of = [&](int g) -> RoundFills {
		FillBatch *fb = g == 0 ? E->cached_batch : E->extra_batches[(size_t)g - 1];
		fb->set_stream_base(g);
		return [fb](std::vector<Progressive> &tasks_, const std::vector<int> &active, std::vector<RoundTrace> &out, double *ms) {
			return device_fills(*fb, tasks_, active, out, ms);
		};
	};
	const int rc = align_batch_rounds(tasks, ntasks, results, E->main_streams(), prepare, fills_of, [E]() { return E->bind(); });
	E->cells_sharers.store(1, std::memory_order_relaxed);
	return rc;
}

}  // namespace

extern "C" {

int csadp_align_batch(const csadp_task *tasks, int ntasks, csadp_result *results)
{
	if (!tasks || !results || ntasks < 0) return CSADP_ERR_ARG;
	int rc = CSADP_OK;
	Engine *E = Engine::primary(NULL, &rc);
	if (!E) return rc;
	return align_batch_on(E, tasks, ntasks, results);
}

int csadp_align_batch_on(int device, const csadp_task *tasks, int ntasks, csadp_result *results)
{
	if (!tasks || !results || ntasks < 0) return CSADP_ERR_ARG;
	int rc = CSADP_OK;
	Engine *E = Engine::open(device, NULL, &rc);
	if (!E) return rc;
	return align_batch_on(E, tasks, ntasks, results);
}

/* What a process' FIRST batch pays once and no later one does (measured on the reference program relinked with the drop-in,
 * profiles/r05_dropin_cold_batch_*: 50-60 ms on top of 10 ms of work): the code objects of the kernels (HIP loads a module at the
 * first launch out of it: 16-24 ms), the first copy of either direction and size class on a stream (7-9 ms each: four of them),
 * the host pool's threads (5 ms), the arenas and pinned staging of the round groups and their growth from round to round (4-5 ms
 * per hipMalloc).  csadp_warmup pays all of it NOW -- a caller with something else to do first (the reference program builds
 * its suffix tree for half a second before its first gap) calls it from a helper thread, as csadp_dropin.c does. */
int csadp_warmup(void)
{
	int rc = CSADP_OK;
	Engine *E = Engine::primary(NULL, &rc);
	if (!E) return rc;
	/* a small batch through every kernel family: four 3..5-sequence tasks (bit-parallel first fills, cell-per-lane profile
	 * fills, the serial and -- 640 rows -- the band-parallel traceback, two round groups), then two 2-sequence tasks
	 * (device-I/O pair path: pack, fill, windowed traceback, row expansion) */
	uint64_t x = 0x9E3779B97F4A7C15ull;
	auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
	auto family = [&](int n, int len, std::vector<std::string> &out) {
		std::string base((size_t)len, 'A');
		for (char &c : base) c = "ACGT"[next() & 3];
		for (int s = 0; s < n; ++s) {
			std::string t;
			for (char c : base) {
				const unsigned u = (unsigned)(next() % 100);
				if (u < 3) continue;
				if (u < 6) t.push_back("ACGT"[next() & 3]);
				t.push_back(u < 16 ? "ACGT"[next() & 3] : c);
			}
			out.push_back(t);
		}
	};
	struct Fam { std::vector<std::string> seqs; std::vector<const char *> ptr; std::vector<int> size, zero, end; };
	auto run = [&](std::vector<Fam> &fams) -> int {
		std::vector<csadp_task> tasks(fams.size());
		std::vector<csadp_result> res(fams.size());
		for (size_t f = 0; f < fams.size(); ++f) {
			Fam &F = fams[f];
			for (const std::string &t : F.seqs) { F.ptr.push_back(t.c_str()); F.size.push_back((int)t.size()); F.zero.push_back(0); }
			F.end = F.size;
			tasks[f] = csadp_task{(int)F.seqs.size(), F.ptr.data(), F.size.data(), F.zero.data(), F.zero.data(), F.end.data()};
		}
		int r = align_batch_on(E, tasks.data(), (int)tasks.size(), res.data());
		for (size_t f = 0; f < fams.size(); ++f) {
			if (r == CSADP_OK && res[f].status != CSADP_OK) r = res[f].status;
			csadp_free_result(&res[f], (int)fams[f].seqs.size());
		}
		return r;
	};
	std::vector<Fam> profile(4), pairs(2);
	family(4, 640, profile[0].seqs);
	family(3, 200, profile[1].seqs);
	family(5, 90, profile[2].seqs);
	family(3, 40, profile[3].seqs);
	family(2, 700, pairs[0].seqs);
	family(2, 300, pairs[1].seqs);
	rc = E->warm_copy_paths();
	if (rc == CSADP_OK) rc = run(profile);
	if (rc == CSADP_OK) rc = run(pairs);
	if (rc != CSADP_OK) return rc;
	/* room for real batches: a whole-genome gap of 19 mitochondrial sequences (the reference's Set3: fills of 17 k x 21 k cells,
	 * 90 MB of directions each) lays out 0.5 GB; of 288 GB */
	std::lock_guard<std::mutex> lock(E->batch_mutex);
	if (E->cached_batch) rc = E->cached_batch->reserve((size_t)1 << 30, (size_t)8 << 20, (size_t)8 << 20);
	for (FillBatch *fb : E->extra_batches)
		if (fb && rc == CSADP_OK) rc = fb->reserve((size_t)256 << 20, (size_t)8 << 20, (size_t)8 << 20);
	return rc;
}

long csadp_recoveries(void) { return csadp::primary_engine_recoveries(); }

int csadp_last_batch_phases(csadp_batch_phases *out)
{
	if (!out) return CSADP_ERR_ARG;
	std::lock_guard<std::mutex> lock(g_phases_mutex);
	*out = g_phases;
	return CSADP_OK;
}

long long csadp_task_cost(const csadp_task *task)
{
	if (!task || task->nseq < 2 || task->nseq > CSADP_MAX_SEQS || !task->starts || !task->ends) return -1;
	std::vector<long long> len((size_t)task->nseq);
	for (int s = 0; s < task->nseq; ++s) {
		len[(size_t)s] = (long long)task->ends[s] - task->starts[s];
		if (len[(size_t)s] < 0) return -1;
	}
	std::sort(len.begin(), len.end());                    /* SortSequencesForDP, :276-308 */
	long long cost = 0, consensus = len[0];
	for (int i = 1; i < task->nseq; ++i) {               /* the consensus never shrinks below the longest row so far */
		cost += len[(size_t)i] * consensus;
		consensus = std::max(consensus, len[(size_t)i]);
	}
	return cost;
}

int csadp_align_batch_multi(const csadp_task *tasks, int ntasks, csadp_result *results, const int *devices, int ndevices,
                            csadp_multi_stats *stats)
{
	if (!tasks || !results || ntasks < 0 || ndevices < 1 || ndevices > CSADP_MAX_DEVICES) return CSADP_ERR_ARG;
	/* every entry is defined whatever happens below: the caller may hand the whole array to csadp_free_result(s) */
	memset(results, 0, sizeof(csadp_result) * (size_t)ntasks);
	std::vector<long long> cost((size_t)ntasks);
	for (int t = 0; t < ntasks; ++t) {
		/* a task no device could align (nseq < 2, bad bounds) costs nothing here and gets its error per task from
		 * csadp_align_batch_on, like in csadp_align_batch */
		cost[(size_t)t] = std::max(0LL, csadp_task_cost(&tasks[t]));
	}
	std::vector<int> part((size_t)ntasks);
	long long maxload = 0;
	int rc = csadp_partition_lpt(cost.data(), ntasks, ndevices, part.data(), &maxload);
	if (rc != CSADP_OK) return rc;
	/* engines first, on this thread: a missing device fails the call before any work starts */
	std::vector<Engine *> eng((size_t)ndevices);
	for (int d = 0; d < ndevices; ++d) {
		eng[(size_t)d] = Engine::open(devices ? devices[d] : d, NULL, &rc);
		if (!eng[(size_t)d]) return rc;
	}
	std::vector<std::vector<int>> mine((size_t)ndevices);
	for (int t = 0; t < ntasks; ++t) mine[(size_t)part[(size_t)t]].push_back(t);
	std::vector<int> drc((size_t)ndevices, CSADP_OK);
	std::vector<double> dms((size_t)ndevices, 0.0);
	const auto t0 = std::chrono::steady_clock::now();
	/* one host thread per GPU: gathers its tasks, aligns them on its device, scatters the results back */
	auto work = [&](int d) {
		const std::vector<int> &idx = mine[(size_t)d];
		if (idx.empty()) return;
		const auto td = std::chrono::steady_clock::now();
		std::vector<csadp_task> sub(idx.size());
		std::vector<csadp_result> res(idx.size());
		for (size_t i = 0; i < idx.size(); ++i) sub[i] = tasks[idx[i]];
		int r = eng[(size_t)d]->bind();
		if (r == CSADP_OK) r = align_batch_on(eng[(size_t)d], sub.data(), (int)sub.size(), res.data());
		drc[(size_t)d] = r;
		if (r == CSADP_OK)
			for (size_t i = 0; i < idx.size(); ++i) results[idx[i]] = res[i];
		else
			for (size_t i = 0; i < idx.size(); ++i) results[idx[i]].status = r;      /* aligned / progress stay NULL */
		dms[(size_t)d] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
	};
	std::vector<std::thread> threads;
	for (int d = 1; d < ndevices; ++d) threads.emplace_back(work, d);
	work(0);
	for (std::thread &th : threads) th.join();
	(void)eng[0]->bind();
	if (stats) {
		memset(stats, 0, sizeof(*stats));
		stats->ndevices = ndevices;
		stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		stats->max_cost = maxload;
		for (int t = 0; t < ntasks; ++t) {
			stats->cost[part[(size_t)t]] += cost[(size_t)t];
			stats->tasks[part[(size_t)t]]++;
			stats->total_cost += cost[(size_t)t];
		}
		for (int d = 0; d < ndevices; ++d) stats->ms[d] = dms[(size_t)d];
	}
	for (int d = 0; d < ndevices; ++d)
		if (drc[(size_t)d] != CSADP_OK) return drc[(size_t)d];
	return CSADP_OK;
}

static int pairs_create(Engine *E, const csadp_task *tasks, int ntasks, csadp_pairbatch **out, bool scores_only);

int csadp_pairs_create(const csadp_task *tasks, int ntasks, csadp_pairbatch **out)
{
	if (!tasks || !out || ntasks <= 0) return CSADP_ERR_ARG;
	int rc = CSADP_OK;
	Engine *E = Engine::primary(NULL, &rc);
	if (!E) return rc;
	return pairs_create(E, tasks, ntasks, out, false);
}

int csadp_pairs_create_on(int device, const csadp_task *tasks, int ntasks, csadp_pairbatch **out)
{
	if (!tasks || !out || ntasks <= 0) return CSADP_ERR_ARG;
	int rc = CSADP_OK;
	Engine *E = Engine::open(device, NULL, &rc);
	if (!E) return rc;
	return pairs_create(E, tasks, ntasks, out, false);
}

namespace {

/* the argument checks of Progressive::init for one sequence of a task, without reading its letters */
int check_region(const csadp_task &t, int s)
{
	if (!t.texts[s] || t.textsizes[s] < 0) return CSADP_ERR_ARG;
	if (t.starts[s] < 0 || t.ends[s] < t.starts[s] || t.ends[s] > t.textsizes[s]) return CSADP_ERR_ARG;
	if (t.ends[s] > t.starts[s] && (t.rotations[s] < 0 || t.rotations[s] >= t.textsizes[s])) return CSADP_ERR_ARG;
	return CSADP_OK;
}

}  // namespace

/*
 * Device-I/O form of a pair batch: raw letters up, aligned rows down.  The host only checks the
 * arguments, decides which sequence is the column one (SortSequencesForDP: the first strict minimum,
 * :290-307), copies every distinct text into the pinned staging area once, and starts ONE H2D copy.
 * CharAt / CharCodeFromSeq / the alphabet check run in nw_pack_planes, the traceback application of
 * :1066-1138 in nw_expand_rows.  Nothing waits here: run()/flush() may follow at once.
 */
static int pairs_create_io(csadp_pairbatch *b, const csadp_task *tasks, int ntasks, bool pipelined, bool strings)
{
	b->device_io = true;
	b->geom.assign((size_t)ntasks, PairGeom());
	b->tasks = std::vector<Progressive>((size_t)ntasks);
	b->status.assign((size_t)ntasks, CSADP_OK);
	FillBatch &fb = b->fb;
	fb.set_pipelined(pipelined);
	fb.allow_bits(true);
	fb.want_strings(strings);
	for (int t = 0; t < ntasks; ++t) {
		const csadp_task &T = tasks[t];
		if (!T.texts || !T.textsizes || !T.rotations || !T.starts || !T.ends) { b->status[(size_t)t] = CSADP_ERR_ARG; continue; }
		int rc = check_region(T, 0);
		if (rc == CSADP_OK) rc = check_region(T, 1);
		if (rc != CSADP_OK) { b->status[(size_t)t] = rc; continue; }
		const int len0 = T.ends[0] - T.starts[0], len1 = T.ends[1] - T.starts[1];
		if (len0 == 0 || len1 == 0) {
			/* no matrix: the host logic finishes the task (:916 early return, :950-956, or a walk that never starts) */
			rc = b->tasks[(size_t)t].init(T);
			if (rc == CSADP_OK) (void)advance(b->tasks[(size_t)t]);
			b->status[(size_t)t] = rc;
			b->geom[(size_t)t].nrows = -1;
			continue;
		}
		PairGeom &G = b->geom[(size_t)t];
		G.col = (len1 < len0) ? 1 : 0;
		G.ncols = G.col ? len1 : len0;
		G.nrows = G.col ? len0 : len1;
		const int j = fb.add(G.nrows, G.ncols, 1, 1);
		int tid[2], first[2];
		for (int w = 0; w < 2; ++w) {
			const int sq = w == 0 ? G.col : 1 - G.col;
			tid[w] = fb.add_text(T.texts[sq], T.textsizes[sq]);
			int f = T.rotations[sq] + T.starts[sq];          /* CharAt, alignment.c:16-20 */
			if (f >= T.textsizes[sq]) f -= T.textsizes[sq];
			first[w] = f;
		}
		fb.set_pair_io(j, tid[0], first[0], tid[1], first[1]);
		b->active.push_back(t);
	}
	if (b->active.empty()) return CSADP_OK;
	if (config().bits && fb.lone_pairs_take_cells()) return kNoDeviceIo;   /* a few large pairs alone: the cell-per-lane path, host I/O */
	int rc = fb.layout();
	if (rc != CSADP_OK) return rc;
	if (!fb.device_io()) return kNoDeviceIo;      /* the batch did not qualify for the bit-parallel kernels */
	parallel_for(fb.ntexts(), [&](int i) { memcpy(fb.text_staging(i), fb.text_source(i), (size_t)fb.text_size(i)); });
	return fb.upload_async();
}

/* scores_only: the caller only wants DP scores: device-I/O batches then skip the aligned rows (nw_expand_rows sums the path) */
static int pairs_create(Engine *E, const csadp_task *tasks, int ntasks, csadp_pairbatch **out, bool scores_only)
{
	std::unique_ptr<csadp_pairbatch> b(new (std::nothrow) csadp_pairbatch(E));
	if (!b) return CSADP_ERR_NOMEM;
	for (int t = 0; t < ntasks; ++t)
		if (tasks[t].nseq != 2) return CSADP_ERR_ARG;
	if (config().bits && config().device_io) {
		const int rc = pairs_create_io(b.get(), tasks, ntasks, true, !scores_only);
		if (rc == CSADP_OK) {
			*out = b.release();
			return CSADP_OK;
		}
		if (rc != kNoDeviceIo) return rc;
		b.reset(new (std::nothrow) csadp_pairbatch(E));   /* e.g. direction planes in HBM and a job wider than one workgroup */
		if (!b) return CSADP_ERR_NOMEM;
	}
	b->tasks = std::vector<Progressive>((size_t)ntasks);
	b->status.assign((size_t)ntasks, CSADP_OK);
	for (int t = 0; t < ntasks; ++t)
		if (tasks[t].nseq != 2) return CSADP_ERR_ARG;
	std::vector<char> pending((size_t)ntasks, 0);
	csadp_pairbatch *bp = b.get();
	const bool trace = config().trace_host;
	auto tick = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!trace) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "csadp_pairs_create: %-14s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
		tick = now;
	};
	parallel_for(ntasks, [&](int t) {
		bp->status[(size_t)t] = bp->tasks[(size_t)t].init(tasks[t]);
		pending[(size_t)t] = (bp->status[(size_t)t] == CSADP_OK && advance(bp->tasks[(size_t)t])) ? 1 : 0;
	});
	lap("task init");
	for (int t = 0; t < ntasks; ++t)
		if (pending[(size_t)t]) b->active.push_back(t);
	if (!b->active.empty()) {
		b->fb.set_pipelined(true);
		bool unit = true;
		for (int t : b->active) {
			b->fb.add(b->tasks[t].nrows(), b->tasks[t].ncols(), b->tasks[t].nprev(), b->tasks[t].border_i());
			unit = unit && b->tasks[t].unit_borders();
		}
		b->fb.allow_bits(unit);
		int rc = b->fb.layout();
		if (rc != CSADP_OK) return rc;
		lap("layout");
		parallel_for((int)b->active.size(), [&](int j) {
			Progressive &p = bp->tasks[(size_t)bp->active[(size_t)j]];
			FillBatch &fb = bp->fb;
			if (fb.bits()) p.write_tables_bits(fb.bit_cols(j), fb.bit_nwords(j), fb.bit_rows(j), fb.bit_rowwords(j));
				else p.write_tables(fb.coltab(j), fb.leftc(j), fb.ncols_pad(j), fb.rowshift(j), fb.top(j), fb.wide());
		});
		lap("tables");
		if ((rc = b->fb.upload()) != CSADP_OK) return rc;
		if ((rc = b->fb.sync()) != CSADP_OK) return rc;
		lap("upload");
	}
	*out = b.release();
	return CSADP_OK;
}

int csadp_pairs_run(csadp_pairbatch *b)
{
	if (!b) return CSADP_ERR_ARG;
	if (b->fetched) return CSADP_ERR_STATE;
	b->ran = true;
	if (b->active.empty()) return CSADP_OK;
	return b->fb.run();
}

int csadp_pairs_flush(csadp_pairbatch *b)
{
	if (!b) return CSADP_ERR_ARG;
	return b->active.empty() ? CSADP_OK : b->fb.flush();
}

int csadp_pairs_sync(csadp_pairbatch *b)
{
	if (!b) return CSADP_ERR_ARG;
	return b->fb.sync();
}

int csadp_pairs_timing(csadp_pairbatch *b, csadp_timing *t)
{
	if (!b || !t) return CSADP_ERR_ARG;
	if (!b->ran || b->active.empty()) return CSADP_ERR_STATE;
	return b->fb.timing(t);
}

int csadp_pairs_fetch(csadp_pairbatch *b, csadp_result *results)
{
	if (!b || !results) return CSADP_ERR_ARG;
	if (!b->ran || b->fetched) return CSADP_ERR_STATE;
	const bool trace = config().trace_host;
	auto tick = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!trace) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "csadp_pairs_fetch:  %-14s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
		tick = now;
	};
	if (b->device_io) {
		if (!b->active.empty()) {
			const int rc = b->fb.download();
			if (rc != CSADP_OK) return rc;
		}
		lap("download");
		std::vector<int> job_of((size_t)b->tasks.size(), -1);
		for (size_t j = 0; j < b->active.size(); ++j) job_of[(size_t)b->active[j]] = (int)j;
		parallel_for((int)b->tasks.size(), [&](int t) {
			csadp_result &R = results[t];
			memset(&R, 0, sizeof(R));
			int st = b->status[(size_t)t];
			const int j = job_of[(size_t)t];
			if (st == CSADP_OK && j < 0) st = b->tasks[(size_t)t].finish(&R);      /* a task without a matrix */
			if (st == CSADP_OK && j >= 0) {
				const PairGeom &G = b->geom[(size_t)t];
				const int32_t *sm = b->fb.summary(j);
				if (sm[4] & 1) st = CSADP_ERR_ALPHABET;
				else if (sm[4] != 0) st = CSADP_ERR_HIP;
				else {
					R.consensus = sm[0] + sm[1] + sm[2];
					R.score = sm[3];
					R.fills = 1;
					R.cells = (long long)G.nrows * G.ncols;
					R.aligned = (char **)calloc(2, sizeof(char *));
					R.progress = strdup(".");
					char *a = (char *)malloc((size_t)R.consensus + 1), *c = (char *)malloc((size_t)R.consensus + 1);
					if (!R.aligned || !R.progress || !a || !c) {
						free(a); free(c); free(R.aligned); free(R.progress);
						R.aligned = NULL; R.progress = NULL;
						st = CSADP_ERR_NOMEM;
					} else {
						memcpy(a, b->fb.out_row(j, 0), (size_t)R.consensus + 1);
						memcpy(c, b->fb.out_row(j, 1), (size_t)R.consensus + 1);
						R.aligned[G.col] = a;                /* original index order (:1160) */
						R.aligned[1 - G.col] = c;
					}
				}
			}
			b->status[(size_t)t] = st;
			R.status = st;
		});
		lap("result strings");
		/* a device-I/O batch keeps nothing on the host that a fetch uses up: it may run and be fetched again (tasks without a
		 * matrix are finished by the host logic, once) */
		b->fetched = b->active.size() != b->tasks.size();
		return CSADP_OK;
	}
	if (!b->active.empty()) {
		const int rc = b->fb.download();
		if (rc != CSADP_OK) return rc;
		lap("download");
		parallel_for((int)b->active.size(), [&](int j) {
			const int32_t *sm = b->fb.summary(j);
			const size_t t = (size_t)b->active[(size_t)j];
			const int a = b->tasks[t].apply_trace(b->fb.ops(j), sm[0], sm[1], sm[2]);
			if (a != CSADP_OK) b->status[t] = a;
		});
	}
	lap("apply trace");
	parallel_for((int)b->tasks.size(), [&](int t) {
		memset(&results[t], 0, sizeof(results[t]));
		if (b->status[(size_t)t] == CSADP_OK) b->status[(size_t)t] = b->tasks[(size_t)t].finish(&results[t]);
		results[t].status = b->status[(size_t)t];
	});
	lap("result strings");
	b->fetched = true;
	return CSADP_OK;
}

void csadp_pairs_destroy(csadp_pairbatch *b) { delete b; }

int csadp_score_pairs(const csadp_task *tasks, int ntasks, int *scores, int *status)
{
	if (!tasks || !scores || ntasks <= 0) return CSADP_ERR_ARG;
	csadp_pairbatch *b = nullptr;
	int rc = CSADP_OK;
	Engine *E = Engine::primary(NULL, &rc);
	if (!E) return rc;
	rc = pairs_create(E, tasks, ntasks, &b, true);
	if (rc != CSADP_OK) return rc;
	if ((rc = csadp_pairs_run(b)) == CSADP_OK && !b->active.empty()) rc = b->fb.download();
	if (rc != CSADP_OK) { delete b; return rc; }
	for (int t = 0; t < ntasks; ++t) scores[t] = 0;
	parallel_for((int)b->active.size(), [&](int j) {
		const int32_t *sm = b->fb.summary(j);
		const size_t t = (size_t)b->active[(size_t)j];
		if (b->device_io) {                                /* nw_expand_rows summed the path */
			if (sm[4] & 1) b->status[t] = CSADP_ERR_ALPHABET;
			else if (sm[4] != 0) b->status[t] = CSADP_ERR_HIP;
			else scores[t] = sm[3];
			return;
		}
		const int a = b->tasks[t].score_from_trace(b->fb.ops(j), sm[0], sm[1], sm[2], &scores[t]);
		if (a != CSADP_OK) b->status[t] = a;
	});
	/* tasks without a matrix (an empty region): the score is the border cell the host already knows */
	for (int t = 0; t < ntasks; ++t) {
		const bool no_matrix = b->device_io ? b->geom[(size_t)t].nrows < 0 : !b->tasks[(size_t)t].next_fill();
		if (b->status[(size_t)t] == CSADP_OK && no_matrix) {
			csadp_result r;
			memset(&r, 0, sizeof(r));
			if (b->tasks[(size_t)t].finish(&r) == CSADP_OK) { scores[t] = r.score; csadp_free_result(&r, 2); }
		}
		if (status) status[t] = b->status[(size_t)t];
	}
	delete b;
	return CSADP_OK;
}

int csadp_sp_score(const char *const *aligned, int nseq, csadp_sp_stats *out)
{
	if (!aligned || !out || nseq < 2 || !aligned[0]) return CSADP_ERR_ARG;
	int erc = CSADP_OK;
	Engine *E = Engine::primary(NULL, &erc);
	if (!E) return erc;
	const size_t len = strlen(aligned[0]);
	for (int s = 1; s < nseq; ++s)
		if (!aligned[s] || strlen(aligned[s]) != len) return CSADP_ERR_ARG;   /* tools.c:248-254 */
	memset(out, 0, sizeof(*out));
	out->consensus = (int)len;
	if (len == 0) return CSADP_OK;
	std::vector<char> host((size_t)nseq * len);
	for (int s = 0; s < nseq; ++s) memcpy(&host[(size_t)s * len], aligned[s], len);
	hipStream_t st = E->stream(0);
	uint8_t *d_chars = nullptr;
	long long *d_out = nullptr;
	long long h_out[3] = {0, 0, 0};
	int rc = CSADP_ERR_HIP;
	if (hipMalloc((void **)&d_chars, host.size()) == hipSuccess && hipMalloc((void **)&d_out, sizeof(h_out)) == hipSuccess &&
	    hipMemcpyAsync(d_chars, host.data(), host.size(), hipMemcpyHostToDevice, st) == hipSuccess &&
	    hipMemsetAsync(d_out, 0, sizeof(h_out), st) == hipSuccess &&
	    csadp::launch_sp_columns(d_chars, nseq, (int)len, d_out, st) == hipSuccess &&
	    hipMemcpyAsync(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost, st) == hipSuccess &&
	    hipStreamSynchronize(st) == hipSuccess) {
		out->total_gaps = h_out[0];
		out->conserved_columns = (int)h_out[1];
		out->sp_score = h_out[2];
		rc = CSADP_OK;
	}
	if (d_chars) (void)hipFree(d_chars);
	if (d_out) (void)hipFree(d_out);
	return rc;
}

unsigned csadp_debug_set_epoch(unsigned next) { return Engine::set_epoch_counter(next); }

int csadp_debug_pool_selftest(int items, long long *sum)
{
	if (items < 0 || !sum) return CSADP_ERR_ARG;
	std::atomic<long long> acc(0);
	parallel_for(items, [&](int i) { acc.fetch_add(i, std::memory_order_relaxed); });
	*sum = acc.load();
	return CSADP_OK;
}

/* ---- test seam: run the HOST logic of one task with a caller-supplied matrix filler ------- */

int csadp_debug_align_with_filler(const csadp_task *task, csadp_debug_fill_fn fill, void *user, csadp_result *result)
{
	if (!task || !fill || !result) return CSADP_ERR_ARG;
	Progressive p;
	memset(result, 0, sizeof(*result));
	int rc = p.init(*task);
	if (rc != CSADP_OK) { result->status = rc; return rc; }
	std::vector<unsigned char> ops;
	std::vector<signed char> rows;
	while (p.next_fill()) {
		const int nrows = p.nrows(), ncols = p.ncols();
		int nops = 0, remj = nrows, remk = 0, score = -p.border_i() * nrows;
		if (ncols > 0) {
			ops.assign((size_t)nrows + ncols + 64, 0);
			rows.resize((size_t)nrows);
			p.debug_rowcodes(rows.data());
			rc = fill(user, nrows, ncols, p.nprev(), p.debug_sv(), rows.data(), p.debug_border_top(), p.border_i(),
			          ops.data(), &nops, &remj, &remk, &score);
			if (rc != CSADP_OK) { result->status = rc; return rc; }
		}
		rc = p.apply_trace(ops.data(), nops, remj, remk, &score);
		if (rc != CSADP_OK) { result->status = rc; return rc; }
	}
	rc = p.finish(result);
	result->status = rc;
	return rc;
}

/* The same for a BATCH of tasks, through the product's own round driver (align_batch_rounds: lock-step rounds, tasks dealt over
 * round groups, one host thread per group, per-task host work on the pool) with the caller's filler in place of the device step:
 * what the thread sanitizer needs to see of csadp_align_batch.  The filler is called from several threads at once. */
int csadp_debug_align_batch_with_filler(const csadp_task *tasks, int ntasks, csadp_debug_fill_fn fill, void *user, csadp_result *results)
{
	if (!tasks || !fill || !results || ntasks < 0) return CSADP_ERR_ARG;
	struct Scratch {
		std::vector<std::vector<unsigned char>> ops;
		std::vector<int32_t> score;
	};
	auto fills_of = [&](int) -> RoundFills {
		std::shared_ptr<Scratch> S = std::make_shared<Scratch>();
		return [S, fill, user](std::vector<Progressive> &prog, const std::vector<int> &active, std::vector<RoundTrace> &out, double *) -> int {
			if (S->ops.size() < active.size()) S->ops.resize(active.size());
			S->score.assign(active.size(), 0);
			std::vector<int> rcs(active.size(), CSADP_OK);
			parallel_for((int)active.size(), [&](int j) {
				Progressive &p = prog[(size_t)active[(size_t)j]];
				const int nrows = p.nrows(), ncols = p.ncols();
				std::vector<unsigned char> &ops = S->ops[(size_t)j];
				ops.assign((size_t)nrows + ncols + 64, 0);
				std::vector<signed char> rows((size_t)nrows);
				p.debug_rowcodes(rows.data());
				int nops = 0, remj = nrows, remk = 0, score = -p.border_i() * nrows;
				rcs[(size_t)j] = fill(user, nrows, ncols, p.nprev(), p.debug_sv(), rows.data(), p.debug_border_top(), p.border_i(), ops.data(), &nops, &remj,
				                      &remk, &score);
				S->score[(size_t)j] = score;
				out[(size_t)j].ops = ops.data();
				out[(size_t)j].nops = nops;
				out[(size_t)j].remj = remj;
				out[(size_t)j].remk = remk;
				out[(size_t)j].score = &S->score[(size_t)j];
			});
			for (int rc : rcs)
				if (rc != CSADP_OK) return rc;
			return CSADP_OK;
		};
	};
	return align_batch_rounds(tasks, ntasks, results, 8, [](int) { return CSADP_OK; }, fills_of, []() { return CSADP_OK; });
}

}  // extern "C"
