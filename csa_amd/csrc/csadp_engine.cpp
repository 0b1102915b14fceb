/*
 * csadp_engine.cpp -- device runtime of libcsadp (see csadp_engine.h).
 */
#include "csadp_config.h"
#include "csadp_engine.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "csadp_kernels.h"

namespace csadp {

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define HIP_TRY(expr)                                                                         \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess) {                                                               \
			fprintf(stderr, "csadp: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_),   \
			        __FILE__, __LINE__);                                                      \
			return CSADP_ERR_HIP;                                                             \
		}                                                                                     \
	} while (0)

}  // namespace

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_registry_mutex;
Engine *g_engines[kMaxDevices] = {};
Engine *g_primary = nullptr;
}  // namespace

Engine *Engine::open(int device, const csadp_config *cfg, int *rc)
{
	std::lock_guard<std::mutex> lock(g_registry_mutex);
	/* An engine runs two fill streams, their side streams and a copy stream; the HIP runtime multiplexes all streams of a
	 * process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one queue run in order -- a fill
	 * behind somebody else's traceback (rocprofv3 trace of tools/history/stream_probe2.py).  Ask for 8 unless the caller has
	 * chosen: ONCE per process, under the registry lock, in front of the library's first HIP call -- later calls would
	 * change nothing (the runtime reads it when it initialises) and would write the environment next to other threads'
	 * getenv.  Without effect when the process has initialised HIP before (then the caller sets it, as bench.py does;
	 * INTEGRATION.md). */
	static bool queues_asked = false;
	if (!queues_asked) {
		setenv("GPU_MAX_HW_QUEUES", "8", 0);
		queues_asked = true;
	}
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		fprintf(stderr, "csadp: no HIP device available (this library has no CPU fallback)\n");
		*rc = CSADP_ERR_NO_DEVICE;
		return nullptr;
	}
	if (device < 0 || device >= kMaxDevices) { *rc = CSADP_ERR_ARG; return nullptr; }
	if (device >= count) {
		/* several ranks on one GPU give per-GPU numbers that are not per-GPU: only on request (rehearsals) */
		if (!config().share_device) {
			fprintf(stderr, "csadp: HIP device %d requested, %d visible (CSADP_SHARE_DEVICE=1 maps ranks onto the visible ones)\n", device, count);
			*rc = CSADP_ERR_NO_DEVICE;
			return nullptr;
		}
		device %= count;
	}
	if (!g_engines[device]) g_engines[device] = new Engine;       /* never deleted: no HIP calls at process exit */
	Engine *e = g_engines[device];
	*rc = e->ready_ ? e->bind() : e->init(device, cfg);
	if (*rc != CSADP_OK) return nullptr;
	if (!g_primary) g_primary = e;
	return e;
}

Engine *Engine::primary(const csadp_config *cfg, int *rc)
{
	{
		std::lock_guard<std::mutex> lock(g_registry_mutex);
		if (g_primary && g_primary->ready_) {
			*rc = g_primary->bind();
			return *rc == CSADP_OK ? g_primary : nullptr;
		}
	}
	int dev = cfg ? cfg->device : -1;
	if (dev < 0) dev = config().local_rank;
	return open(dev, cfg, rc);
}

Engine *Engine::primary_if_ready()
{
	std::lock_guard<std::mutex> lock(g_registry_mutex);
	return (g_primary && g_primary->ready_) ? g_primary : nullptr;
}

void Engine::shutdown_all()
{
	std::lock_guard<std::mutex> lock(g_registry_mutex);
	for (Engine *e : g_engines)
		if (e) e->shutdown();
	g_primary = nullptr;
}

namespace {
std::atomic<uint32_t> g_epoch_counter{0};
}

uint32_t Engine::next_epoch()
{
	uint32_t e;
	do e = (g_epoch_counter.fetch_add(1) + 1) & 0xffffffu; while (e == 0);
	return e;
}

uint32_t Engine::set_epoch_counter(uint32_t v) { return g_epoch_counter.exchange(v); }

int Engine::bind() const
{
	HIP_TRY(hipSetDevice(device_));
	return CSADP_OK;
}

int Engine::init(int dev, const csadp_config *cfg)
{
	HIP_TRY(hipSetDevice(dev));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, dev));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		fprintf(stderr, "csadp: device %d is %s; the kernels are built for gfx950 only\n", dev, prop.gcnArchName);
		return CSADP_ERR_NO_DEVICE;
	}
	device_ = dev;
	snprintf(name_, sizeof(name_), "%s (%s)", prop.name, prop.gcnArchName);
	cus_ = prop.multiProcessorCount;
	slots_ = config().slots;
	if (slots_ < 1 || slots_ > kMaxSlots) return CSADP_ERR_ARG;
	nstreams_ = 2 * main_streams();
	for (int i = 0; i < nstreams_; ++i) HIP_TRY(hipStreamCreateWithFlags(&streams_[i], hipStreamNonBlocking));
	for (int i = 0; i < nstreams_; ++i) HIP_TRY(hipEventCreateWithFlags(&busy_ev_[i], hipEventDisableTiming));
	HIP_TRY(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&upload_stream_, hipStreamNonBlocking));
	HIP_TRY(configure_kernels());                     /* per-device function attributes (csadp_bits.hip) */
	HIP_TRY(configure_traceback_cells());
	verbose_ = cfg && cfg->verbose;
	ready_ = true;
	return CSADP_OK;
}

long primary_engine_recoveries()
{
	Engine *E = Engine::primary_if_ready();
	return E ? E->recoveries.load() : 0;
}

int Engine::mark_busy(int stream_index, hipStream_t st)
{
	if (stream_index < 0 || stream_index >= nstreams_) return CSADP_ERR_ARG;
	HIP_TRY(hipEventRecord(busy_ev_[stream_index], st));
	busy_mask_.fetch_or(1ull << stream_index);
	return CSADP_OK;
}

bool Engine::device_idle()
{
	const unsigned long long m = busy_mask_.load();
	for (int i = 0; i < nstreams_; ++i)
		if (((m >> i) & 1ull) && hipEventQuery(busy_ev_[i]) != hipSuccess) return false;
	return true;
}

int Engine::warm_copy_paths()
{
	{ const int brc = bind(); if (brc != CSADP_OK) return brc; }
	constexpr size_t kBytes = (size_t)8 << 20;
	uint8_t *dev = nullptr, *pin = nullptr;
	HIP_TRY(hipMalloc((void **)&dev, kBytes));
	if (hipHostMalloc((void **)&pin, kBytes, hipHostMallocDefault) != hipSuccess) { (void)hipFree(dev); return CSADP_ERR_HIP; }
	memset(pin, 0, kBytes);
	std::vector<hipStream_t> all(streams_, streams_ + nstreams_);
	all.push_back(copy_stream_);
	all.push_back(upload_stream_);
	int rc = CSADP_OK;
	for (hipStream_t st : all)
		for (size_t bytes : {(size_t)4096, (size_t)256 << 10, kBytes}) {
			if (hipMemsetAsync(dev, 0, bytes, st) != hipSuccess || hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
			    hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
				rc = CSADP_ERR_HIP;
		}
	for (hipStream_t st : all)
		if (hipStreamSynchronize(st) != hipSuccess) rc = CSADP_ERR_HIP;
	(void)hipHostFree(pin);
	(void)hipFree(dev);
	return rc;
}

/* Pools of released HBM arenas and pinned staging buffers: consecutive batches of similar size
 * (bench steps, the drop-in's ~50 calls, a streaming caller with several pair batches in flight) skip
 * hipMalloc / hipHostMalloc, which cost 0.1 .. 1 ms each.  Best fit; a full pool drops its smallest entry. */
namespace {
constexpr size_t kPoolEntries = 6;

uint8_t *pool_take(std::vector<std::pair<uint8_t *, size_t>> &pool, size_t need, size_t *got)
{
	/* best fit, but never a buffer far larger than asked for: the grow-only arenas of csadp_align_batch once took the 41 GB arenas a
	 * streaming caller's pair batches had just released for rounds that need a few MB, and kept them for the life of the process
	 * (round 5: bench.py's config 5 leg then found 143 of 288 GB free) */
	const size_t most = 4 * need + ((size_t)256 << 20);
	int best = -1;
	for (size_t i = 0; i < pool.size(); ++i)
		if (pool[i].second >= need && pool[i].second <= most && (best < 0 || pool[i].second < pool[(size_t)best].second)) best = (int)i;
	if (best < 0) return nullptr;
	uint8_t *p = pool[(size_t)best].first;
	*got = pool[(size_t)best].second;
	pool.erase(pool.begin() + best);
	return p;
}

/* returns the buffer the pool lets go of (to be freed by the caller), or nullptr */
uint8_t *pool_give(std::vector<std::pair<uint8_t *, size_t>> &pool, uint8_t *ptr, size_t bytes)
{
	pool.emplace_back(ptr, bytes);
	if (pool.size() <= kPoolEntries) return nullptr;
	size_t smallest = 0;
	for (size_t i = 1; i < pool.size(); ++i)
		if (pool[i].second < pool[smallest].second) smallest = i;
	uint8_t *drop = pool[smallest].first;
	pool.erase(pool.begin() + (long)smallest);
	return drop;
}
}  // namespace

void Engine::give_arena(uint8_t *ptr, size_t bytes)
{
	if (!ptr) return;
	std::lock_guard<std::mutex> lock(pool_mutex_);
	uint8_t *drop = ready_ ? pool_give(arena_pool_, ptr, bytes) : ptr;
	if (drop) (void)hipFree(drop);
}

uint8_t *Engine::take_arena(size_t need, size_t *got)
{
	std::lock_guard<std::mutex> lock(pool_mutex_);
	return pool_take(arena_pool_, need, got);
}

void Engine::give_pinned(uint8_t *ptr, size_t bytes)
{
	if (!ptr) return;
	std::lock_guard<std::mutex> lock(pool_mutex_);
	uint8_t *drop = ready_ ? pool_give(pinned_pool_, ptr, bytes) : ptr;
	if (drop) (void)hipHostFree(drop);
}

uint8_t *Engine::take_pinned(size_t need, size_t *got)
{
	std::lock_guard<std::mutex> lock(pool_mutex_);
	return pool_take(pinned_pool_, need, got);
}

void Engine::drop_arena_cache()
{
	std::lock_guard<std::mutex> lock(pool_mutex_);
	for (auto &e : arena_pool_) (void)hipFree(e.first);
	arena_pool_.clear();
}

void Engine::shutdown()
{
	if (!ready_) return;
	(void)hipSetDevice(device_);
	{
		std::lock_guard<std::mutex> lock(batch_mutex);
		delete cached_batch;
		cached_batch = nullptr;
		for (FillBatch *b : extra_batches) delete b;
		extra_batches.clear();
	}
	drop_arena_cache();
	{
		std::lock_guard<std::mutex> lock(pool_mutex_);
		for (auto &e : pinned_pool_) (void)hipHostFree(e.first);
		pinned_pool_.clear();
	}
	for (int i = 0; i < nstreams_; ++i) {
		(void)hipStreamSynchronize(streams_[i]);
		(void)hipStreamDestroy(streams_[i]);
		streams_[i] = nullptr;
		if (busy_ev_[i]) (void)hipEventDestroy(busy_ev_[i]);
		busy_ev_[i] = nullptr;
	}
	busy_mask_.store(0);
	if (copy_stream_) {
		(void)hipStreamSynchronize(copy_stream_);
		(void)hipStreamDestroy(copy_stream_);
		copy_stream_ = nullptr;
	}
	if (upload_stream_) {
		(void)hipStreamSynchronize(upload_stream_);
		(void)hipStreamDestroy(upload_stream_);
		upload_stream_ = nullptr;
	}
	ready_ = false;
}

/* ---- FillBatch ----------------------------------------------------------------------- */

FillBatch::~FillBatch()
{
	(void)E_->bind();
	settle_pull();
	/* the arena and the pinned mirrors go back to the engine's pools: nothing may still be using them */
	if (laid_out_ && bits_) {
		(void)wait_batch();
	} else if (laid_out_) {
		for (int sl = 0; sl < E_->nstreams(); ++sl) (void)hipStreamSynchronize(E_->stream(sl));
	}
	if (arena_) E_->give_arena(arena_, arena_cap_);
	if (h_in_) E_->give_pinned(h_in_, h_in_cap_);
	if (h_res_) E_->give_pinned(h_res_, h_res_cap_);
	if (ev_up_) (void)hipEventDestroy(ev_up_);
	for (auto &slot : ev_)
		for (auto &e : slot)
			if (e) (void)hipEventDestroy(e);
}

/* A pull upload (upload(): a kernel reads the pinned staging on the batch's stream) is not waited for where it is launched;
 * download() waits for that stream on the way.  Every other road to the staging -- clear() and a new layout after an error
 * between upload() and download(), the destructor -- settles it first. */
void FillBatch::settle_pull()
{
	if (!pull_pending_) return;
	if (E_->bind() == CSADP_OK) (void)hipStreamSynchronize(home_stream(0));
	pull_pending_ = false;
}

void FillBatch::clear()
{
	settle_pull();
	jobs_.clear();
	extra_.clear();
	tiles_.clear();
	diag_off_.clear();
	bits_ = false;
	bjobs_.clear();
	bextra_.clear();
	texts_.clear();
	pairio_.clear();
	io_ = false;
	cells_mode_ = false;
	cjobs_.clear();
	laid_out_ = false;
	ran_ = false;
}

int FillBatch::add_text(const char *text, int size)
{
	for (size_t i = 0; i < texts_.size(); ++i)
		if (texts_[i].ptr == text && texts_[i].size == size) return (int)i;
	texts_.push_back(TextRef{text, size, 0});
	return (int)texts_.size() - 1;
}

void FillBatch::set_pair_io(int j, int text_col, int first_col, int text_row, int first_row)
{
	if ((int)pairio_.size() <= j) pairio_.resize((size_t)j + 1, PairIo{{-1, -1}, {0, 0}});
	pairio_[(size_t)j] = PairIo{{text_col, text_row}, {first_col, first_row}};
}

uint8_t *FillBatch::text_staging(int id) { return h_in_ + texts_[(size_t)id].off; }

const uint8_t *FillBatch::out_row(int j, int which) const { return h_res_ + bextra_[(size_t)j].res_out[which]; }

int FillBatch::add(int nrows, int ncols, int nprev, int left_i)
{
	FillJob j;
	memset(&j, 0, sizeof(j));
	j.nrows = nrows;
	j.ncols = ncols;
	j.nprev = nprev;
	j.leftmul = 4 * (nprev - left_i);
	jobs_.push_back(j);
	laid_out_ = false;
	return (int)jobs_.size() - 1;
}

/* A few large pairs with the device to themselves: every chain of nw_fill_cells then has its own compute units (the fetcher layout,
 * csadp_cells.hip) and a matrix takes (rows + 0.66 cols) x 41 ns there, its band-parallel walk a fifth of the bit-parallel path's serial
 * one -- against 86 ns per row plus 17 for the walk (tools/single_probe.py: 16 384^2 1.23 ms on the device against 1.75, 100 000^2
 * 7.5 against 10.2).  The bit-parallel kernels are built for batches; these are not one.  (A pair's columns are its shorter sequence.) */
/* The helper-wave layout of nw_fill_cells wants one workgroup per compute unit (every chain alone on its units): the limit follows the
 * device's compute units -- a partitioned or smaller part has fewer than the 256 of a whole MI355X (round-4 ADVICE) -- unless
 * CSADP_CELLS_FETCH was set explicitly. */
/* A workgroup of that layout has a compute unit to itself (six waves, the LDS rings), and its consumer waves spin on their producers: when
 * several launches share the device (`sharers`: round groups, passes of one batch in flight) and together hold more workgroups than there
 * are compute units, a producer can wait for a unit behind consumers that wait for it, until their bounded waits run out (seen with the
 * layout forced onto four round groups of a thousand workgroups each: seconds per launch).  The limit is therefore per sharer. */
int cells_fetch_limit(const Engine &E, int sharers)
{
	const Config &cfg = config();
	if (cfg.cells_fetch_forced) return cfg.cells_fetch_wgs;
	return std::min(cfg.cells_fetch_wgs, E.compute_units()) / std::max(1, sharers);
}

bool FillBatch::lone_pairs_take_cells() const
{
	const Config &cfg = config();
	if (!cfg.lone_cells || jobs_.empty() || jobs_.size() > 8) return false;
	long chunks = 0;
	for (const FillJob &J : jobs_) {
		if (J.nprev != 1 || J.leftmul != 0 || J.nrows < 4096 || J.ncols > 2L * J.nrows || J.ncols <= 0) return false;
		chunks += (J.ncols + kCellStripCols * kCellWaves - 1) / (kCellStripCols * kCellWaves);
	}
	return chunks <= cells_fetch_limit(*E_, 1);   /* (run_slot_cells: when a launch takes the fetcher layout) */
}

int FillBatch::layout()
{
	Engine &E = *E_;
	if (!E.ready()) return CSADP_ERR_NO_DEVICE;
	{ const int brc = E.bind(); if (brc != CSADP_OK) return brc; }
	const int nj = (int)jobs_.size();
	extra_.assign(nj, Extra());
	cells_ = dir_bytes_ = border_bytes_ = 0;
	wide_ = false;
	bits_ = false;
	cells_mode_ = false;
	if (bits_allowed_ && nj >= 1 && config().bits) {
		bits_ = true;
		for (const FillJob &J : jobs_)
			if (J.nprev != 1 || J.leftmul != 0 || J.nrows <= 0 || J.ncols <= 0) bits_ = false;
	}
	if (bits_ && lone_pairs_take_cells()) bits_ = false;
	if (bits_) return layout_bits();
	/* every other fill (profile steps, stale borders, pairs with the bit-parallel path switched off by CSADP_BITS=0 --
	 * the 32-bit cross-check of the bit-parallel kernels): the persistent cell-per-lane wavefront */
	return layout_cells();
}

uint32_t *FillBatch::coltab(int j) { return reinterpret_cast<uint32_t *>(h_in_ + extra_[j].in_coltab); }
int32_t *FillBatch::leftc(int j) { return reinterpret_cast<int32_t *>(h_in_ + extra_[j].in_leftc); }
uint8_t *FillBatch::rowshift(int j) { return h_in_ + extra_[j].in_rowshift; }
int32_t *FillBatch::top(int j) { return reinterpret_cast<int32_t *>(h_in_ + extra_[j].in_top); }
int FillBatch::ncols_pad(int j) const { return extra_[j].ncols_pad; }

/* Cell-per-lane mode (csadp_cells.hip): one CellJob per fill, a work list of (job, chunk of kCellWaves
 * strips) -- the longest jobs first, a job's chunks in ascending order -- one launch per pass. */
int FillBatch::layout_cells()
{
	Engine &E = *E_;
	const int nj = (int)jobs_.size();
	cells_mode_ = true;
	const Config &cfg = config();
	const int band_min = cfg.tb_band_min;                         /* rows; 0x7fffffff: never */
	pull_uploads_ = cfg.pull_uploads;
	const bool band_forced = cfg.tb_band_forced;
	const int tb_corridor = cfg.tb_corridor;                      /* groups of 1024 start columns scouted per band */
	tb_max_bands_ = tb_max_groups_ = 0;
	long banded_bands = 0;                                        /* bands of all banded jobs: the scouts' workgroups per corridor group */
	cjobs_.assign((size_t)nj, CellJob());
	tiles_.clear();
	diag_off_.assign(2, 0);                       /* "one launch" for timing() */
	for (int j = 0; j < nj; ++j) {
		FillJob &J = jobs_[(size_t)j];
		if (J.nrows <= 0 || J.ncols <= 0) return CSADP_ERR_ARG;
		const long long nprev = J.nprev;
		if (nprev < 1 || nprev > 63) return CSADP_ERR_ARG;
		if (nprev * (2LL * J.nrows + J.ncols) * 4 + 64 >= (1LL << config().test_range_log2)) return CSADP_ERR_RANGE;
		if (nprev > 21) wide_ = true;                  /* nw_fill_cells folds leftc into the gain bytes: 12 * nprev + 1 <= 255 */
		CellJob &C = cjobs_[(size_t)j];
		memset(&C, 0, sizeof(C));
		C.nrows = J.nrows;
		C.ncols = J.ncols;
		C.nprev = J.nprev;
		C.leftmul = J.leftmul;
		C.nstrips = (J.ncols + kCellStripCols - 1) / kCellStripCols;
		C.nchunks = (C.nstrips + kCellWaves - 1) / kCellWaves;
		C.steps_pad = (int)align_up((size_t)J.nrows + 64, kCellBlock);
		J.nstrips = C.nstrips;
		J.steps_pad = C.steps_pad;
		/* the walk band-parallel from band_min rows on (csadp_cells_tb.hip); smaller matrices by one serial walk */
		C.nbands = (J.nrows + kBandRows - 1) / kBandRows;
		/* (a profile more than twice as long as the row sequence: every band is crossed by more L moves than a scout takes, all of
		 * them would be walked twice -- one serial walk then, unless the tests ask for bands) */
		C.banded = (J.nrows >= band_min && (band_forced || J.ncols <= 2 * J.nrows)) ? 1 : 0;
		if (C.banded) {
			tb_max_bands_ = std::max(tb_max_bands_, C.nbands);
			banded_bands += C.nbands;
		}
		extra_[(size_t)j].ncols_pad = C.nstrips * kCellStripCols;
		cells_ += (long long)J.nrows * J.ncols;
		dir_bytes_ += (long long)C.nstrips * kCellCols * (C.steps_pad / 16) * kLanes * 4;
		border_bytes_ += (long long)std::max(C.nchunks - 1, 0) * C.steps_pad * 8 * 2;      /* written once, read once */
	}
	/* The corridor of start columns the scouts cover per band (groups of 1024 columns around the straight line between the matrix' corners):
	 * `tb_corridor` groups, and more while the scouts of the whole batch still fit ONE workgroup per compute unit -- a wider corridor then
	 * costs no time, and a path that leaves the corridor costs a serial walk of every band from there on (one round of Mammals: 0.40 ms in
	 * nw_tb_resolve with three groups of seven scouted).  CSADP_TB_CORRIDOR set explicitly is taken as it is. */
	/* The band-parallel walk buys latency with work: a scout workgroup per band and corridor group, most of whose walks are never used.  A
	 * batch of MANY banded matrices has its parallelism in the jobs: past ~5 000 scout workgroups one serial walk per job is shorter
	 * (tools/r04/fetch_threshold_probe.py, matrices of 5 000 x 6 187: the walk of 48 of them 0.50 ms either way, of 96 0.77 banded and
	 * 0.53 serial; of 8 0.25 against 0.49).  Not when the tests ask for bands (CSADP_TB_BAND_MIN set). */
	const long cus = std::max(1, E.compute_units());
	if (!band_forced && banded_bands * std::max(1, tb_corridor) > 5000 * cus / 256) {
		for (CellJob &C : cjobs_) C.banded = 0;
		tb_max_bands_ = 0;
		banded_bands = 0;
	}
	for (int j = 0; j < nj; ++j) {
		CellJob &C = cjobs_[(size_t)j];
		if (!C.banded) continue;
		const int ngroups = (C.ncols / kBandStride + 1 + kScoutStarts - 1) / kScoutStarts;
		int want = std::max(1, tb_corridor);
		if (!cfg.tb_corridor_forced) want = std::max(want, (int)(cus / std::max(1L, banded_bands)));
		/* at most as many groups as let one band's table row fit the resolve kernel's LDS table (64 KiB of u16) */
		C.tb_groups = std::min(std::min(ngroups, want), 64 * 1024 / 2 / kScoutStarts);
		C.tb_pitch = C.tb_groups * kScoutStarts;
		tb_max_groups_ = std::max(tb_max_groups_, C.tb_groups);
	}
	{
		std::vector<int> order((size_t)nj);
		for (int j = 0; j < nj; ++j) order[(size_t)j] = j;
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
			const CellJob &A = cjobs_[(size_t)a], &B = cjobs_[(size_t)b];
			return (long long)A.steps_pad + 96LL * A.nstrips > (long long)B.steps_pad + 96LL * B.nstrips;   /* wavefront length */
		});
		for (int j : order)
			for (int c = 0; c < cjobs_[(size_t)j].nchunks; ++c) {
				TileRef t;
				t.job = j;
				t.a = c;
				t.s = 0;
				t.first = 0;
				tiles_.push_back(t);
			}
		serial_tiles_ = tiles_;
		std::stable_sort(serial_tiles_.begin(), serial_tiles_.end(), [](const TileRef &a, const TileRef &b) { return a.a < b.a; });
		chunk_first_.clear();
		for (size_t i = 0; i < serial_tiles_.size(); ++i)
			if (i == 0 || serial_tiles_[i].a != serial_tiles_[i - 1].a) chunk_first_.push_back(i);
		chunk_first_.push_back(serial_tiles_.size());
		/* Job by job, a launch that does not fit the device keeps the later chunks of every resident job spinning until their wavefront
		 * arrives -- chunk c of a job starts 3.6 us x 4 strips x c behind its first: 0.66 ncols of nrows + 0.66 ncols steps on average, a
		 * quarter to 40 % of the slots a batch of families holds -- while the jobs behind them wait for those slots.  Level by level
		 * (every job's first chunk, then every second one, ...; a job's chunks still in ascending order, so a producer is still dispatched
		 * before its consumer) a chunk gets its slot when its producer is nearly or wholly through: resident workgroups work.  The price is
		 * a job's own latency (its levels queue behind everyone's), so only launches beyond what the device holds at two workgroups per
		 * compute unit -- shared with the other round groups -- take it (tools/r05/profile_batch_probe.py, profiles/r05_cells_order.txt). */
		const int sharers = std::max(1, E.cells_sharers.load(std::memory_order_relaxed));
		const bool by_level = cfg.cells_order >= 0 ? cfg.cells_order == 1 : (long)tiles_.size() * sharers > 2 * cus;
		if (by_level) tiles_ = serial_tiles_;
	}

	nslots_ = pipelined_ ? E.slots() : 1;
	next_slot_ = 0;
	size_t off = 0;
	for (int sl = 0; sl < nslots_; ++sl) {
		jobs_off_[sl] = off;
		off += (size_t)nj * sizeof(CellJob);
	}
	off = align_up(off, 256);
	abort_off_ = off;
	off += 256;
	tiles_off_ = off;
	off = align_up(off + tiles_.size() * sizeof(TileRef), 256);
	serial_off_ = off;
	off = align_up(off + serial_tiles_.size() * sizeof(TileRef), 256);
	for (int j = 0; j < nj; ++j) {
		CellJob &C = cjobs_[(size_t)j];
		Extra &X = extra_[(size_t)j];
		X.in_coltab = C.coltab = off;
		off = align_up(off + (size_t)X.ncols_pad * 4, 256);
		X.in_leftc = C.leftc = off;
		off = align_up(off + (size_t)X.ncols_pad * 4, 256);
		X.in_rowshift = C.rowshift = off;
		off = align_up(off + (size_t)C.steps_pad + 64 + 512, 256);   /* the first strip reads 256-byte groups, one group ahead */
		X.in_top = C.top = off;
		off = align_up(off + ((size_t)X.ncols_pad + 1) * 4, 256);
	}
	in_bytes_ = off;
	std::vector<std::vector<CellJob>> slot_jobs((size_t)nslots_, cjobs_);
	for (int sl = 0; sl < nslots_; ++sl) {
		res_off_[sl] = off;
		for (int j = 0; j < nj; ++j) {
			CellJob &C = slot_jobs[(size_t)sl][(size_t)j];
			Extra &X = extra_[(size_t)j];
			C.summary = off;
			X.res_summary = off - res_off_[sl];
			off += 64;
			C.ops = off;
			X.res_ops = off - res_off_[sl];
			off = align_up(off + (size_t)C.nrows + C.ncols + 64, 256);
		}
		res_bytes_ = off - res_off_[sl];
		sum_bytes_ = res_bytes_;
		for (int j = 0; j < nj; ++j) {
			CellJob &C = slot_jobs[(size_t)sl][(size_t)j];
			C.dirs = off;
			off = align_up(off + (size_t)C.nstrips * kCellCols * (C.steps_pad / 16) * kLanes * 4, 256);
		}
		for (int j = 0; j < nj; ++j) {
			CellJob &C = slot_jobs[(size_t)sl][(size_t)j];
			if (!C.banded) continue;
			C.tb_tab = off;
			off = align_up(off + (size_t)C.nbands * C.tb_pitch * 2, 256);
			C.tb_ent = off;
			off = align_up(off + ((size_t)C.nbands + 2) * 4, 256);
			C.tb_cnt = off;
			off = align_up(off + (size_t)C.nbands * 4, 256);
			C.tb_scratch = off;
			off = align_up(off + (size_t)C.nrows + C.ncols + 64, 256);
		}
		hand_off_[sl] = off;                          /* the hand-off granules of all jobs, contiguous: zeroed by upload() */
		for (int j = 0; j < nj; ++j) {
			CellJob &C = slot_jobs[(size_t)sl][(size_t)j];
			C.hand = off;
			off = align_up(off + (size_t)std::max(C.nchunks - 1, 0) * C.steps_pad * 8, 256);
		}
		hand_bytes_ = off - hand_off_[sl];
	}
	total_bytes_ = off;
	const int rc = finish_layout();
	if (rc != CSADP_OK) return rc;
	for (int sl = 0; sl < nslots_; ++sl)
		memcpy(h_in_ + jobs_off_[sl], slot_jobs[(size_t)sl].data(), (size_t)nj * sizeof(CellJob));
	memcpy(h_in_ + tiles_off_, tiles_.data(), tiles_.size() * sizeof(TileRef));
	memcpy(h_in_ + serial_off_, serial_tiles_.data(), serial_tiles_.size() * sizeof(TileRef));
	cjobs_ = slot_jobs[0];
	return CSADP_OK;
}

/* Bit-parallel mode: one BitJob per fill, the whole matrix in one launch (csadp_bits.hip). */
int FillBatch::layout_bits()
{
	Engine &E = *E_;
	const int nj = (int)jobs_.size();
	bjobs_.assign((size_t)nj, BitJob());
	bextra_.assign((size_t)nj, BitExtra());
	tiles_.clear();
	diag_off_.assign(2, 0);                       /* "one launch" for timing() */
	bits_maxstrips_ = 1;
	bits_wide_ = false;
	const Config &cfg = config();
	test_abort_ = cfg.test_force_abort;
	bits_lds_pad_ = -1;                                             /* chosen below, once the launch shape is known */
	/* Words of 32 columns per lane (1, 2 or 3 by default; 4 on request: CSADP_BITS_WORDS).  More words per lane amortise what a step spends on its neighbours (the three
	 * borrow instructions, the letter chain, the accumulators: 11 of W = 1's 31 instructions, 11 of W = 2's 53) and halve the
	 * strips -- and with them the waves -- of a matrix.  DESIGN.md section 3 has the measurements behind the default. */
	{
		/* two words per lane when that still puts two waves on every SIMD: strips of 64 lanes x 2 words, times the passes a
		 * pipelined batch keeps in flight (2 launches of up to 4 passes, chosen below by the same rule).  Batches smaller than
		 * that -- down to one matrix -- are bound by the latency of a step: one word per lane, twice the strips. */
		long long strips[4] = {0, 0, 0, 0}, cost[4] = {0, 0, 0, 0}, widest[4] = {0, 0, 0, 0};
		static const int kStepValu[4] = {0, 31, 52, 73};              /* VALU instructions per step (tools/count_valu.py) */
		for (const FillJob &J : jobs_) {
			const long long words = (J.ncols + 31) / 32;
			for (int w = 1; w <= 3; ++w) {
				const long long sw = (words + 64 * w - 1) / (64 * w);
				strips[w] += sw;
				widest[w] = std::max(widest[w], sw);
				/* strips times the step's instructions.  (Until round 5 a job of up to 16 strips was ONE workgroup: a fifth strip
				 * doubled up on the first strip's SIMD, and strips beyond four counted in fours.  Jobs wider than four strips are
				 * chains of four-strip workgroups now -- below -- and a fifth strip is a workgroup of its own.) */
				cost[w] += sw * kStepValu[w];
			}
		}
		const long long simds = 4LL * std::max(E.compute_units(), 1);
		const int group2 = pipelined_ ? std::max(1, std::min(std::max(E.compute_units(), 1) / std::max(nj, 1), 4)) : 1;
		const int passes = pipelined_ ? 2 * group2 : 1;
		/* two or three words per lane when that still keeps a wave per SIMD in flight (three: real mitochondrial genomes: 16.4-17.0 k
		 * letters are three strips of 6144 columns instead of five of 4096); one word per lane only when it is clearly cheaper (at
		 * equal cost two words measure 14 % faster: 64 jobs of 33 000 columns) or the batch is a handful of matrices whose chains of
		 * strips are all there is to wait for.  tools/r05/words_probe.py, profiles/r05_words_probe.txt: 15 shapes from 2 pairs of
		 * 200 kbp to 1000 of 5 kbp at 1 / 2 / 3 words; round 4's rule (1.5 waves per SIMD for two words, strips in fours) chose one
		 * word for 16 pairs of 33 kbp and 32 of 17 kbp (18.7 against 22.7 TCUPS) and two for 40 of 50 kbp and 64 of 100 kbp
		 * (33.3 / 41.0 against 36.0 / 44.0 at three). */
		/* ... and where every job is at most two strips the jobs share four-wave workgroups (bits_pack_ below) and a launch holds up to eight
		 * passes of them: sixteen passes in flight (tools/r05/pack_words.py, profiles/r05_pack_words.txt: 200 pairs of 7 kbp 31.6 at one word,
		 * 35.6 at two; 128 of 4 kbp 25.0 / 30.8; 96 of 8 kbp 33.2 / 36.9) */
		auto in_flight = [&](int w) { return (pipelined_ && cfg.bits_pack != 0 && widest[w] <= 2) ? 16 : passes; };
		const bool ok2 = strips[2] * in_flight(2) >= simds, ok3 = strips[3] * in_flight(3) >= simds;
		int w = 1;
		long long best = cost[1] * 100 / 93;
		if (ok2 && cost[2] <= best) { w = 2; best = cost[2]; }
		if (ok3 && cost[3] < best) { w = 3; best = cost[3]; }
		if (cfg.bits_words >= 0) w = cfg.bits_words;
		bits_words_ = (w >= 2 && w <= 4) ? w : 1;
	}
	const int wpl = bits_words_;
	for (int j = 0; j < nj; ++j) {
		const FillJob &J = jobs_[(size_t)j];
		BitJob &B = bjobs_[(size_t)j];
		memset(&B, 0, sizeof(B));
		B.nrows = J.nrows;
		B.ncols = J.ncols;
		B.wpl = wpl;
		const int words = (J.ncols + 31) / 32;
		B.nstrips = (words + wpl * kLanes - 1) / (wpl * kLanes);
		B.nwords_pad = B.nstrips * kLanes * wpl;
		B.steps_pad = (int)align_up((size_t)J.nrows + 64, kBitBlock);
		B.rowwords = B.steps_pad / 32;
		bits_maxstrips_ = std::max(bits_maxstrips_, B.nstrips);
		extra_[(size_t)j].ncols_pad = B.nwords_pad * 32;
		cells_ += (long long)J.nrows * J.ncols;
		/* no direction planes: lane state + carry history per block of 32 steps */
		border_bytes_ += (long long)B.nstrips * (B.steps_pad / kBitBlock) * kLanes * (16 * wpl + 8);
	}
	dir_bytes_ = 0;
	/* device-side I/O: every job has its two texts registered */
	io_ = (int)pairio_.size() == nj && nj > 0;
	for (const PairIo &P : pairio_)
		if (P.text[0] < 0 || P.text[1] < 0) io_ = false;
	/* A job is one workgroup of up to 16 waves; consecutive passes are MERGED: `group` passes (slots) form one launch of
	 * group * nj workgroups, and `streams` such fill launches are kept in flight on as many streams, each with its traceback
	 * on a side stream (launch_bits_pass), so that fills never pause for a traceback.  Round 2 (one word per lane, 8 waves per
	 * 16 kbp job): group x streams = 2 x 2.  The target is the same number of WAVES in flight whatever the words per lane:
	 * four per SIMD. */
	bits_group_ = 1;
	nslots_ = 1;
	/* Jobs narrower than four strips share workgroups of four waves (nw_fill_bits<.., PACK>): first fit, widest and longest first, a job's
	 * strips on consecutive waves; `shared_table_` is the launch's table, one {job, strip} per wave.  The launch then has the shape of
	 * four-strip jobs.  Not when nothing is gained (every job takes a workgroup anyway: three- and four-strip jobs only). */
	bits_pack_ = 1;
	shared_table_.clear();
	if (cfg.bits_pack != 0 && bits_maxstrips_ <= 4) {
		std::vector<int> order((size_t)nj);
		for (int j = 0; j < nj; ++j) order[(size_t)j] = j;
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
			const BitJob &A = bjobs_[(size_t)a], &B = bjobs_[(size_t)b];
			return A.nstrips != B.nstrips ? A.nstrips > B.nstrips : A.steps_pad > B.steps_pad;
		});
		std::vector<int> fill;                                 /* waves taken per workgroup */
		std::vector<int> open_with[4];                         /* workgroups with 1, 2, 3 free waves (index = free waves) */
		std::vector<TileRef> table;
		for (int j : order) {
			const int need = bjobs_[(size_t)j].nstrips;
			int wg = -1;
			for (int room = need; room <= 3 && wg < 0; ++room)
				if (!open_with[room].empty()) { wg = open_with[room].back(); open_with[room].pop_back(); }
			if (wg < 0) {
				wg = (int)fill.size();
				fill.push_back(0);
				TileRef none;
				none.job = -1; none.a = 0; none.s = 0; none.first = 0;
				table.insert(table.end(), 4, none);
			}
			for (int c = 0; c < need; ++c) {
				TileRef &t = table[(size_t)wg * 4 + fill[(size_t)wg] + c];
				t.job = j;
				t.a = c;
			}
			fill[(size_t)wg] += need;
			if (fill[(size_t)wg] < 4) open_with[4 - fill[(size_t)wg]].push_back(wg);
		}
		if ((int)fill.size() < nj) {
			shared_table_.swap(table);
			bits_pack_ = 2;                                    /* (any value > 1: workgroups are shared) */
		}
	}
	const int nwgs = bits_pack_ > 1 ? (int)shared_table_.size() / 4 : nj;       /* workgroups of a pass */
	if (pipelined_) {
		/* workgroups of at most three strips (real mitochondrial genomes at three words per lane: 16.3-17.7 k columns) leave a
		 * SIMD of their compute unit idle and a pass is few waves (66 pairs: 198): four launches in flight, and no LDS
		 * reservation below, so that a compute unit takes three or four of them (profiles/r04_sweep_real.txt, 48 steps, with
		 * round 3's traceback: 66 Mammals pairs 24.6 -> 29.6 TCUPS, 120 Primates pairs 31.2 -> 33.7; with the windowed traceback
		 * 32-33 and 36-38 in every shape of four launches; synthetic pairs of the same shape: profiles/r04_shape_probe.txt) */
		const int wg_strips = bits_pack_ > 1 ? 4 : bits_maxstrips_;      /* waves of a workgroup */
		const bool small_wgs = wg_strips <= 3;
		const int dflt_streams = small_wgs ? 4 : 2;
		/* passes per launch: one workgroup per compute unit (tools/history/sweep_words.sh, profiles/r03_sweep_words.txt: with two words
		 * per lane 2 streams x 2 passes = two waves per SIMD runs 44 TCUPS, 4 x 2 39-43, 2 x 4 37-40: more workgroups than compute
		 * units per launch are not spread evenly over them) */
		/* ... so a launch of workgroups of four strips holds as many passes as give every compute unit at most ONE of them
		 * (120 jobs: 2 passes = 240 workgroups run 39 TCUPS, 3 passes = 360 run 29: the compute units that get two take twice
		 * as long); workgroups of more strips than SIMDs are uneven anyway and do better queued deep (120 jobs of 9 strips: 4
		 * passes 27.5 TCUPS, 2 passes 22.6) */
		const int want = std::max(E.compute_units(), 1);
		/* ... and workgroups of one or two strips fill a compute unit a quarter or half as much: their launches hold as many passes as put two
		 * waves on every SIMD, up to seven (tools/r05/group_probe.py, profiles/r05_group_probe.txt: 128 pairs of 12 kbp 31.0 -> 38.4 TCUPS,
		 * 256 of them 32.0 -> 37.6, 512 of 5 kbp 23.1 -> 26.3, 128 of 6 kbp at one word per lane 27.0 -> 30.1; three-strip workgroups -- the
		 * real mitochondrial sets -- keep the shape round 4 measured for them) */
		if (wg_strips <= 2) {
			const int two_per_simd = 8 * want / std::max(nj * bits_maxstrips_, 1);        /* passes that put two waves on every SIMD */
			/* fewer jobs than even eight passes fill: four launches of four passes keep more in flight than two of seven (64 pairs of 12 kbp:
			 * 32.1 against 28.4 TCUPS) */
			bits_group_ = two_per_simd > 8 ? 4 : std::max(1, std::min(two_per_simd, 7));
		}
		else if (wg_strips == 3) bits_group_ = std::max(1, std::min(want / std::max(nj, 1), 4));
		else if (wg_strips == 4) {
			/* Four-strip workgroups (16 kbp pairs at two words per lane): a launch should be whole "waves" of workgroups -- a multiple of the
			 * compute units.  Round 3 found one workgroup per compute unit per launch (128 jobs x 2 passes, 64 x 4: 360 workgroups ran 29 TCUPS
			 * where 240 ran 39) and took floor(CUs / jobs) passes; job counts that do not divide the compute units were left with launches of
			 * 192 or 128 or 64 workgroups: 96 jobs ran 35 TCUPS, 192 jobs 36.5, 32 jobs 24 (tools/r05/streams_probe.py).  Now: the FEWEST
			 * passes (up to eight) whose workgroups are a multiple of the compute units, else the count with the smallest idle share of its
			 * last wave: 96 jobs x 8 = 768 = 3 x 256: 47.0 TCUPS; 192 x 4: 48; 32 x 8: 45.6; 128 x 2 and 64 x 4 as before.  Not simply "eight":
			 * gap-rich pairs (the unrelated 16 kbp variant) lose a quarter in launches of eight passes -- their tracebacks are as long as
			 * their fills and a launch's traceback holds its slot range (profiles/r05_streams_probe.txt: 128 unrelated pairs 39.9 at two passes,
			 * 28.9 at eight).  Held back by memory: a slot is a pass' checkpoints (5.3 MB per 16 kbp pair). */
			int best_rel = 1 << 30;
			bits_group_ = 8;
			for (int g = 1; g <= 8; ++g) {
				const long total = (long)nwgs * g;                /* workgroups of a launch of g passes */
				if (total < want && g < 8) continue;                      /* not even one workgroup per compute unit */
				const long waves = (total + want - 1) / want;
				const int rel = (int)((waves * want - total) * 1000 / (waves * want));
				if (rel < best_rel) { best_rel = rel; bits_group_ = g; }
			}
			while (bits_group_ > 1 && 4.0 * bits_group_ * (double)border_bytes_ > 48e9) --bits_group_;
		}
		else {
			/* jobs wider than four strips: two workgroups per compute unit per launch, as before -- up to eight passes, not four, for the
			 * same reason as above (16 pairs of 16 kbp at one word per lane: 16.7 -> 30.6 TCUPS), within the same memory */
			bits_group_ = std::max(1, std::min((2 * want + nj - 1) / nj, 8));
			const int floor_g = std::min(bits_group_, 4);                  /* what round 4 took whatever the memory (config 5: two passes, 168 GB) */
			while (bits_group_ > floor_g && 4.0 * bits_group_ * (double)border_bytes_ > 48e9) --bits_group_;
		}
		bits_group_ = std::max(1, std::min(cfg.bits_group >= 0 ? cfg.bits_group : bits_group_, 8));
		bits_streams_ = std::max(1, std::min(cfg.bits_streams >= 0 ? cfg.bits_streams : dflt_streams, E.main_streams()));
		/* every stream alternates between TWO slot ranges: the traceback of a launch runs on the stream's side
		 * stream while the next fill of the stream already works on the other range */
		bits_streams_ = std::max(1, std::min(bits_streams_, (Engine::kMaxSlots - 1) / (2 * bits_group_)));
		nslots_ = bits_streams_ * 2 * bits_group_;
	}
	/* How many strips share a workgroup.  A job of at most FOUR strips is one workgroup, a wave per SIMD.  A wider one is a chain of
	 * four-strip workgroups (`bits_chunk_`) handing over through granules in HBM, so that its strips spread over compute units: one
	 * workgroup of up to 16 strips keeps a job's whole wavefront on the four SIMDs of ONE unit, and the longest jobs of a batch -- the
	 * chain every pass waits for -- then run four waves to a SIMD while other units idle.  Until round 5 only launches of few strips
	 * were spread this way (4 per workgroup up to one wave per SIMD in all, 8 up to two, else 16); measured over batch shapes
	 * (tools/r05/chunk_probe.py, profiles/r05_chunk_probe.txt, TCUPS at 16 / 8 / 4 strips per workgroup): config 5's 256 pairs of
	 * 1-200 kbp 30.7 / 33.3 / 41.8-44.6, 64 pairs of 33 kbp 27.4 / 36.7 / 36.5, 40 of 50 kbp 20.3 / 26.4 / 28.8, 8 of 200 kbp
	 * 12.7 / 12.7 / 16.4; batches of a hundred and more equal jobs the same within the run-to-run spread (+-5 %). */
	bits_chunk_ = 4;
	{
		const int forced = cfg.bits_chunk;
		if (forced == 4 || forced == 8 || forced == 16) bits_chunk_ = forced;
	}
	bits_wide_ = false;
	for (const BitJob &B : bjobs_)
		if (B.nstrips > bits_chunk_) bits_wide_ = true;
	if (!bits_wide_) bits_chunk_ = kBitMaxStrips;
	/* Pipelined launches of one workgroup per job: the dispatcher hands a compute unit as many workgroups as fit, not one of each
	 * launch in flight -- three fills on one unit and one on the next run at the pace of the fuller one.  Every fill workgroup
	 * reserves dynamic LDS so that exactly two of them fit next to one traceback workgroup (tools/history/sweep_pad.sh,
	 * profiles/r03_sweep_pad.txt: +2-4 %, and a collapse of 25 % as soon as two fills and a traceback no longer fit). */
	bits_lds_pad_ = 0;
	if (pipelined_ && !bits_wide_ && (bits_maxstrips_ > 3 || bits_pack_ > 1)) {
		const int waves = bits_maxstrips_ <= 4 ? 4 : bits_maxstrips_ <= 8 ? 8 : 16;
		const int room = (160 * 1024 - traceback_bits_lds_bytes(bits_words_)) / 2 - fill_bits_lds_bytes(waves) - 2048;
		bits_lds_pad_ = std::max(0, std::min(room, 60 * 1024)) & ~255;
	}
	{
		const int forced = cfg.bits_lds_pad;
		if (forced >= 0) bits_lds_pad_ = std::min(forced, 60) * 1024;
	}
	next_slot_ = 0;
	/* the lone shape (csadp_engine.h): device-I/O batches whose planes are per-slot scratch, more than one word per lane, and
	 * no more strips at one word per lane than the chip has SIMDs */
	lone_ = LoneShape();
	last_lone_ = false;
	std::vector<BitJob> lone_jobs;
	{
		const bool io_batch = (int)pairio_.size() == nj && nj > 0;
		long long strips1 = 0;
		for (const FillJob &J : jobs_) strips1 += ((J.ncols + 31) / 32 + kLanes - 1) / kLanes;
		const long long simds = 4LL * std::max(E.compute_units(), 1);
		lone_.on = pipelined_ && io_batch && bits_words_ > 1 && !bits_wide_ && strips1 <= simds && nslots_ < Engine::kMaxSlots &&
		           cfg.lone_shape;
	}
	if (lone_.on) {
		lone_.slot = nslots_;
		lone_.chunk = 4;
		lone_jobs = bjobs_;
		for (BitJob &B : lone_jobs) {
			const int words = (B.ncols + 31) / 32;
			B.wpl = 1;
			B.nstrips = (words + kLanes - 1) / kLanes;
			B.nwords_pad = B.nstrips * kLanes;
			if (B.nstrips > lone_.chunk) lone_.wide = true;
		}
	}
	size_t off = 0;
	for (int sl = 0; sl < nslots_ + (lone_.on ? 1 : 0); ++sl) {
		jobs_off_[sl] = off;
		off += (size_t)nj * sizeof(BitJob);
	}
	off = align_up(off, 256);
	abort_off_ = off;                             /* one abort word for every launch of the batch, zeroed by upload() */
	off += 256;
	if (lone_.on && lone_.wide) {
		/* its work list: (job, chunk of four strips), the longest jobs first, a job's chunks ascending; and once more grouped by
		 * chunk index for the wait-free repeat (as for the batch's own chunked shape below) */
		std::vector<int> order((size_t)nj);
		for (int j = 0; j < nj; ++j) order[(size_t)j] = j;
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
			return (long long)lone_jobs[(size_t)a].steps_pad * lone_jobs[(size_t)a].nstrips > (long long)lone_jobs[(size_t)b].steps_pad * lone_jobs[(size_t)b].nstrips;
		});
		for (int j : order)
			for (int c = 0; c * lone_.chunk < lone_jobs[(size_t)j].nstrips; ++c) {
				TileRef t;
				t.job = j;
				t.a = c;
				t.s = 0;
				t.first = 0;
				lone_.tiles.push_back(t);
			}
		lone_.tiles_off = off;
		off = align_up(off + lone_.tiles.size() * sizeof(TileRef), 256);
		lone_.serial_off = off;
		lone_.serial_tiles = lone_.tiles;
		std::stable_sort(lone_.serial_tiles.begin(), lone_.serial_tiles.end(), [](const TileRef &a, const TileRef &b) { return a.a < b.a; });
		for (size_t i = 0; i < lone_.serial_tiles.size(); ++i)
			if (i == 0 || lone_.serial_tiles[i].a != lone_.serial_tiles[i - 1].a) lone_.chunk_first.push_back(i);
		lone_.chunk_first.push_back(lone_.serial_tiles.size());
		off = align_up(off + lone_.serial_tiles.size() * sizeof(TileRef), 256);
	}
	tiles_off_ = off;
	chunk_first_.clear();
	wide_shared_ = false;
	if (!bits_wide_ && bits_pack_ > 1) {                 /* the table of the shared workgroups travels where a chunked launch has its work list */
		tiles_ = shared_table_;
		serial_tiles_.clear();
		off = align_up(off + tiles_.size() * sizeof(TileRef), 256);
		serial_off_ = off;
	}
	if (bits_wide_) {
		/* work list of the chunked kernel: (job, chunk of bits_chunk_ strips), the longest jobs first -- they
		 * are the critical path of a mixed batch -- and a job's chunks in ascending order, so that the
		 * workgroup a chunk waits for is always dispatched before it */
		std::vector<int> order((size_t)nj);
		for (int j = 0; j < nj; ++j) order[(size_t)j] = j;
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
			return (long long)bjobs_[(size_t)a].steps_pad * bjobs_[(size_t)a].nstrips > (long long)bjobs_[(size_t)b].steps_pad * bjobs_[(size_t)b].nstrips;
		});
		/* ... in batches whose widest job is at most three chunks: with longer chains in the batch -- config 5: up to 33 strips -- the shared
		 * form measures 3 % LOWER over eight alternating runs (41.4 against 42.6 TCUPS, profiles/r05_config5_pack_ab.txt) where batches of
		 * jobs of up to ten strips gain 10-24 % (profiles/r05_pack_wide.txt); CSADP_BITS_PACK=2 shares whatever the widths */
		wide_shared_ = cfg.bits_pack != 0 && bits_chunk_ == 4 && (bits_maxstrips_ <= 12 || cfg.bits_pack >= 2);
		if (wide_shared_) {
			/* Shared workgroups in a chunked launch (round 5): whole chunks as before; the LAST, partial chunk of a job and the jobs narrower
			 * than a chunk go side by side into workgroups of their own, first fit in the order of the list, and such a workgroup follows the
			 * chunks of the job that completed it (its members' producers are then all in front of it).  `tiles_` = four {job, strip}
			 * entries per workgroup (job < 0: empty); `level` = the highest chunk index in a workgroup, for the repeat path. */
			/* The tails of LONG chains stay alone: config 5 (jobs of up to 33 strips; its step is its longest chains) measured 43.3 TCUPS with
			 * one workgroup per chunk and 37-42 with every tail shared (tools/r05/config5_pack_ab.py); batches of jobs of up to ten strips gain
			 * 10-24 % (tools/r05/pack_wide.py). */
			constexpr int kSharedTailChunks = 2;
			std::vector<int> level;
			TileRef none;
			none.job = -1; none.a = 0; none.s = 0; none.first = 0;
			struct Open { int at; int used; };
			std::vector<Open> open;                               /* shared workgroups with room, by position in pending */
			std::vector<std::vector<TileRef>> pending;            /* shared workgroups not yet emitted */
			std::vector<int> pending_level;
			auto emit = [&](const TileRef *four, int lv) {
				tiles_.insert(tiles_.end(), four, four + 4);
				level.push_back(lv);
			};
			for (int j : order) {
				const int ns = bjobs_[(size_t)j].nstrips, full = ns / 4, rest = ns % 4;
				for (int c = 0; c < full; ++c) {
					TileRef four[4];
					for (int w = 0; w < 4; ++w) { four[w] = none; four[w].job = j; four[w].a = 4 * c + w; }
					emit(four, c);
				}
				if (rest == 0) continue;
				if (full > kSharedTailChunks) {                   /* a long chain keeps its tail to itself (below) */
					TileRef four[4] = {none, none, none, none};
					for (int w = 0; w < rest; ++w) { four[w].job = j; four[w].a = 4 * full + w; }
					emit(four, full);
					continue;
				}
				int slot = -1;
				for (size_t o = 0; o < open.size(); ++o)
					if (4 - open[o].used >= rest) { slot = (int)o; break; }
				if (slot < 0) {
					pending.push_back(std::vector<TileRef>(4, none));
					pending_level.push_back(0);
					open.push_back(Open{(int)pending.size() - 1, 0});
					slot = (int)open.size() - 1;
				}
				Open &O = open[(size_t)slot];
				for (int w = 0; w < rest; ++w) {
					TileRef &t = pending[(size_t)O.at][(size_t)(O.used + w)];
					t.job = j;
					t.a = 4 * full + w;
				}
				O.used += rest;
				pending_level[(size_t)O.at] = std::max(pending_level[(size_t)O.at], full);
				if (O.used == 4) {                                /* complete: it goes out behind this job's chunks */
					emit(pending[(size_t)O.at].data(), pending_level[(size_t)O.at]);
					pending[(size_t)O.at].clear();
					open.erase(open.begin() + slot);
				}
			}
			for (const Open &O : open) emit(pending[(size_t)O.at].data(), pending_level[(size_t)O.at]);
			off = align_up(off + tiles_.size() * sizeof(TileRef), 256);
			serial_off_ = off;
			std::vector<int> by_level(level.size());
			for (size_t i = 0; i < by_level.size(); ++i) by_level[i] = (int)i;
			std::stable_sort(by_level.begin(), by_level.end(), [&](int a, int b) { return level[(size_t)a] < level[(size_t)b]; });
			std::vector<TileRef> serial;
			for (size_t i = 0; i < by_level.size(); ++i) {
				if (i == 0 || level[(size_t)by_level[i]] != level[(size_t)by_level[i - 1]]) chunk_first_.push_back(4 * i);
				serial.insert(serial.end(), tiles_.begin() + 4 * by_level[i], tiles_.begin() + 4 * by_level[i] + 4);
			}
			chunk_first_.push_back(serial.size());
			serial_tiles_.swap(serial);
			off = align_up(off + serial_tiles_.size() * sizeof(TileRef), 256);
		} else {
		for (int j : order)
			for (int c = 0; c * bits_chunk_ < bjobs_[(size_t)j].nstrips; ++c) {
				TileRef t;
				t.job = j;
				t.a = c;
				t.s = 0;
				t.first = 0;
				tiles_.push_back(t);
			}
		off = align_up(off + tiles_.size() * sizeof(TileRef), 256);
		/* the same items once more, grouped by chunk index: the serial fallback launches one group at a
		 * time, so no workgroup ever waits for another (recover_bits) */
		serial_off_ = off;
		std::vector<TileRef> serial(tiles_);
		std::stable_sort(serial.begin(), serial.end(), [](const TileRef &a, const TileRef &b) { return a.a < b.a; });
		for (size_t i = 0; i < serial.size(); ++i)
			if (i == 0 || serial[i].a != serial[i - 1].a) chunk_first_.push_back(i);
		chunk_first_.push_back(serial.size());
		serial_tiles_.swap(serial);
		off = align_up(off + serial_tiles_.size() * sizeof(TileRef), 256);
		}
	}
	if (io_) {
		/* inputs = the raw texts (each once) + one status word per job; the planes are per-slot scratch
		 * written by nw_pack_planes */
		for (TextRef &T : texts_) {
			T.off = off;
			off = align_up(off + (size_t)T.size + 16, 256);
		}
		for (int j = 0; j < nj; ++j) {
			BitJob &B = bjobs_[(size_t)j];
			const PairIo &P = pairio_[(size_t)j];
			for (int w = 0; w < 2; ++w) {
				B.text[w] = texts_[(size_t)P.text[w]].off;
				B.size[w] = texts_[(size_t)P.text[w]].size;
				B.first[w] = P.first[w];
			}
			B.istatus = off;
			off += 4;
		}
		off = align_up(off, 256);
	} else {
		for (int j = 0; j < nj; ++j) {
			BitJob &B = bjobs_[(size_t)j];
			bextra_[(size_t)j].in_cols = B.colplanes = off;
			off = align_up(off + (size_t)2 * B.nwords_pad * 4, 256);
			bextra_[(size_t)j].in_rows = B.rowplanes = off;
			off = align_up(off + (size_t)2 * B.rowwords * 4, 256);
		}
	}
	in_bytes_ = off;
	std::vector<std::vector<BitJob>> slot_jobs((size_t)nslots_, bjobs_);
	if (lone_.on) {
		for (int j = 0; j < nj; ++j) {                     /* the texts' places and status words: assigned above, after lone_jobs was copied */
			BitJob &L = lone_jobs[(size_t)j];
			const BitJob &B = bjobs_[(size_t)j];
			for (int w = 0; w < 2; ++w) {
				L.text[w] = B.text[w];
				L.size[w] = B.size[w];
				L.first[w] = B.first[w];
			}
			L.istatus = B.istatus;
		}
		slot_jobs.push_back(lone_jobs);
	}
	const int nslots_all = (int)slot_jobs.size();
	for (int sl = 0; sl < nslots_all; ++sl) {
		const bool lone_slot = lone_.on && sl == lone_.slot;
		const int chunk_of_slot = lone_slot ? lone_.chunk : bits_chunk_;
		const bool wide_of_slot = lone_slot ? lone_.wide : bits_wide_;
		res_off_[sl] = off;
		if (io_) {
			/* results = [summaries][aligned rows]: the host never sees the op lists */
			for (int j = 0; j < nj; ++j) {
				slot_jobs[(size_t)sl][(size_t)j].summary = off;
				extra_[(size_t)j].res_summary = off - res_off_[sl];
				off += 64;
			}
			off = align_up(off, 256);
			sum_bytes_ = off - res_off_[sl];
			for (int j = 0; j < nj; ++j) {
				BitJob &B = slot_jobs[(size_t)sl][(size_t)j];
				for (int w = 0; w < 2; ++w) {
					B.out[w] = off;
					bextra_[(size_t)j].res_out[w] = off - res_off_[sl];
					off = align_up(off + (size_t)B.nrows + B.ncols + 1, 16);
				}
			}
			off = align_up(off, 256);
		} else {
			for (int j = 0; j < nj; ++j) {
				BitJob &B = slot_jobs[(size_t)sl][(size_t)j];
				Extra &X = extra_[(size_t)j];
				B.summary = off;
				X.res_summary = off - res_off_[sl];
				off += 64;
				B.ops = off;
				X.res_ops = off - res_off_[sl];
				off = align_up(off + (size_t)B.nrows + B.ncols + 64, 256);
			}
			sum_bytes_ = off - res_off_[sl];
		}
		res_bytes_ = off - res_off_[sl];
		if (io_) {
			for (int j = 0; j < nj; ++j) {               /* scratch of this slot: op list and bit planes */
				BitJob &B = slot_jobs[(size_t)sl][(size_t)j];
				B.ops = off;
				off = align_up(off + (size_t)B.nrows + B.ncols + 64, 256);
				B.colplanes = off;
				off = align_up(off + (size_t)2 * B.nwords_pad * 4, 256);
				B.rowplanes = off;
				off = align_up(off + (size_t)2 * B.rowwords * 4, 256);
			}
		}
		for (int j = 0; j < nj; ++j) {
			BitJob &B = slot_jobs[(size_t)sl][(size_t)j];
			const size_t blocks = (size_t)B.nstrips * (B.steps_pad / kBitBlock);
			B.ckpt = off;
			off = align_up(off + blocks * B.wpl * kLanes * 16, 256);
			B.hand = off;
			off = align_up(off + blocks * kLanes * 8, 256);
		}
		const size_t hand0 = off;                 /* the granules between the chunks of all jobs, contiguous: zeroed by upload() */
		for (int j = 0; j < nj; ++j) {
			BitJob &B = slot_jobs[(size_t)sl][(size_t)j];
			B.xhand = off;
			const int nchunks = (B.nstrips + chunk_of_slot - 1) / chunk_of_slot;
			if (wide_of_slot && nchunks > 1) off = align_up(off + (size_t)(nchunks - 1) * (B.steps_pad / kBitBlock) * 24, 256);
		}
		if (lone_slot) {
			lone_.hand_off = hand0;
			lone_.hand_bytes = off - hand0;
		} else {
			hand_off_[sl] = hand0;
			hand_bytes_ = off - hand0;
		}
	}
	total_bytes_ = off;
	const int rc = finish_layout();
	if (rc != CSADP_OK) return rc;
	for (int sl = 0; sl < nslots_all; ++sl)
		memcpy(h_in_ + jobs_off_[sl], slot_jobs[(size_t)sl].data(), (size_t)nj * sizeof(BitJob));
	if (lone_.on && !lone_.tiles.empty()) {
		memcpy(h_in_ + lone_.tiles_off, lone_.tiles.data(), lone_.tiles.size() * sizeof(TileRef));
		memcpy(h_in_ + lone_.serial_off, lone_.serial_tiles.data(), lone_.serial_tiles.size() * sizeof(TileRef));
	}
	if (!tiles_.empty()) {
		memcpy(h_in_ + tiles_off_, tiles_.data(), tiles_.size() * sizeof(TileRef));
		memcpy(h_in_ + serial_off_, serial_tiles_.data(), serial_tiles_.size() * sizeof(TileRef));
	}
	next_stream_ = 0;
	launch_no_ = 0;
	/* batches of one engine start on different streams, so that a streaming caller's batches (each one
	 * pass, several in flight) overlap instead of queueing behind each other */
	/* ... unless ONE pass of the batch already covers the chip: then batches take turns on the same stream (first in,
	 * first out, each traceback under the next batch's fill).  Side by side, three such batches progress at equal rates,
	 * all complete late and together, and the caller's pipeline runs dry in between: 4.9 vs 4.25 ms per 512-pair batch. */
	const bool rotate = cfg.stream_rotate >= 0 ? cfg.stream_rotate != 0 : nj < E.compute_units();
	base_stream_ = rotate ? E.rotate_stream() % E.main_streams() : 0;
	issued_ = 0;
	bjobs_ = slot_jobs[0];
	return CSADP_OK;
}

/* HBM arena and pinned staging mirrors, grow-only, taken from / returned to the engine's pools */
int FillBatch::alloc_buffers()
{
	Engine &E = *E_;
	settle_pull();
	if (total_bytes_ > arena_cap_) {
		if (arena_) { E.give_arena(arena_, arena_cap_); arena_ = nullptr; arena_cap_ = 0; }
		arena_ = E.take_arena(total_bytes_, &arena_cap_);
	}
	if (total_bytes_ > arena_cap_) {
		size_t free_b = 0, total_b = 0;
		const size_t pretend = (size_t)config().test_hbm_limit_mb << 20;       /* test seam: a device with little memory left */
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		if (pretend) free_b = std::min(free_b, pretend);
		if (total_bytes_ + (256u << 20) > free_b) {
			E.drop_arena_cache();
			HIP_TRY(hipMemGetInfo(&free_b, &total_b));
			if (pretend) free_b = std::min(free_b, pretend);
		}
		if (total_bytes_ + (256u << 20) > free_b) {
			fprintf(stderr, "csadp: batch needs %.1f GiB of HBM, %.1f GiB free\n",
			        total_bytes_ / 1073741824.0, free_b / 1073741824.0);
			return CSADP_ERR_RANGE;
		}
		HIP_TRY(hipMalloc((void **)&arena_, total_bytes_));
		arena_cap_ = total_bytes_;
	}
	/* the last 64 bytes of the upload staging receive the abort word (a pinned allocation of its own per batch cost a
	 * hipHostMalloc / hipHostFree pair, and hipHostFree waits for the whole device: 1.6 ms per destroyed batch while the
	 * next ones were running) */
	const size_t need_in = align_up(in_bytes_, 64) + 64;
	if (need_in > h_in_cap_) {
		if (h_in_) E.give_pinned(h_in_, h_in_cap_);
		h_in_ = E.take_pinned(need_in, &h_in_cap_);
		if (!h_in_) {
			HIP_TRY(hipHostMalloc((void **)&h_in_, need_in, hipHostMallocDefault));
			h_in_cap_ = need_in;
		}
	}
	h_abort_ = reinterpret_cast<int *>(h_in_ + h_in_cap_ - 64);
	if (res_bytes_ > h_res_cap_) {
		if (h_res_) E.give_pinned(h_res_, h_res_cap_);
		h_res_ = E.take_pinned(res_bytes_, &h_res_cap_);
		if (!h_res_) {
			HIP_TRY(hipHostMalloc((void **)&h_res_, res_bytes_, hipHostMallocDefault));
			h_res_cap_ = res_bytes_;
		}
	}
	return CSADP_OK;
}

int FillBatch::reserve(size_t arena_bytes, size_t in_bytes, size_t res_bytes)
{
	/* best effort: on a device that is short of memory (another process holds most of it) the reservation is skipped -- a real batch
	 * then allocates what it needs, or reports what it lacks */
	{
		size_t free_b = 0, total_b = 0;
		if (E_->bind() != CSADP_OK || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return CSADP_OK;
		if (arena_bytes > arena_cap_ && free_b < 4 * arena_bytes) return CSADP_OK;
	}
	const size_t t = total_bytes_, i = in_bytes_, r = res_bytes_;
	total_bytes_ = std::max(t, arena_bytes);
	in_bytes_ = std::max(i, in_bytes);
	res_bytes_ = std::max(r, res_bytes);
	const int rc = alloc_buffers();
	total_bytes_ = t;
	in_bytes_ = i;
	res_bytes_ = r;
	return rc;
}

/* buffers, zeroed inputs, events */
int FillBatch::finish_layout()
{
	settle_pull();
	{ const int arc = alloc_buffers(); if (arc != CSADP_OK) return arc; }
	/* events: the bit-parallel path keeps a launch's three events at the first slot of its range only */
	for (int sl = 0; sl < nslots_; ++sl) {
		if (bits_ && nslots_ > 1 && sl % bits_group_ != 0) continue;
		for (auto &e : ev_[sl])
			if (!e) HIP_TRY(hipEventCreate(&e));
	}
	if (bits_ && lone_.on)
		for (auto &e : ev_[lone_.slot])
			if (!e) HIP_TRY(hipEventCreate(&e));
	/* zeroed inputs (padding the kernels rely on, the abort word, the status words) -- except the raw texts of
	 * a device-I/O batch, which the caller overwrites letter for letter: a streaming caller's create() would
	 * otherwise clear megabytes of pinned memory twice */
	if (io_ && !texts_.empty()) {
		const size_t tbeg = texts_.front().off, tend = texts_.back().off + (size_t)texts_.back().size;
		memset(h_in_, 0, tbeg);
		memset(h_in_ + tend, 0, in_bytes_ - tend);
	} else {
		memset(h_in_, 0, in_bytes_);
	}
	laid_out_ = true;
	ran_ = false;
	pending_ = 0;
	memset(slot_used_, 0, sizeof(slot_used_));
	return CSADP_OK;
}

uint32_t *FillBatch::bit_cols(int j) { return reinterpret_cast<uint32_t *>(h_in_ + bextra_[(size_t)j].in_cols); }
int FillBatch::bit_nwords(int j) const { return bjobs_[(size_t)j].nwords_pad; }
uint32_t *FillBatch::bit_rows(int j) { return reinterpret_cast<uint32_t *>(h_in_ + bextra_[(size_t)j].in_rows); }
int FillBatch::bit_rowwords(int j) const { return bjobs_[(size_t)j].rowwords; }

int FillBatch::upload()
{
	{ const int brc = E_->bind(); if (brc != CSADP_OK) return brc; }
	if (!laid_out_) return CSADP_ERR_STATE;
	if ((cells_mode_ || (bits_ && bits_wide_)) && hand_bytes_ > 0)
		for (int sl = 0; sl < nslots_; ++sl) HIP_TRY(hipMemsetAsync(arena_ + hand_off_[sl], 0, hand_bytes_, home_stream(0)));
	if (bits_ && lone_.on && lone_.hand_bytes > 0) HIP_TRY(hipMemsetAsync(arena_ + lone_.hand_off, 0, lone_.hand_bytes, home_stream(0)));
	/* every slot's stream must see the inputs: copy on the batch's first stream and wait (upload is not on the
	 * timed path; run() calls may follow on any stream) */
	if (cells_mode_ && nslots_ == 1 && pull_uploads_ && in_bytes_ <= (size_t)4 << 20) {
		/* a lock-step round's tables: pulled by a kernel, and NOT waited for here -- the fill follows on the same stream */
		HIP_TRY(launch_pull_pinned(arena_, h_in_, in_bytes_, home_stream(0)));
		pull_pending_ = true;                          /* the kernel reads h_in_: settle_pull() before the staging is touched again */
		return CSADP_OK;
	}
	HIP_TRY(hipMemcpyAsync(arena_, h_in_, in_bytes_, hipMemcpyHostToDevice, home_stream(0)));
	HIP_TRY(hipStreamSynchronize(home_stream(0)));
	return CSADP_OK;
}

/* inputs -> HBM without waiting.  The copy runs on the engine's upload stream (profile batches: stream 0) and an
 * event makes every compute stream of the engine wait for it, so passes may be enqueued right away (device-I/O pair batches: the
 * host never blocks between create and fetch). */
int FillBatch::upload_async()
{
	if (!laid_out_) return CSADP_ERR_STATE;
	{ const int brc = E_->bind(); if (brc != CSADP_OK) return brc; }
	const int nst = E_->nstreams();
	/* pair batches upload on a stream of their own: on the fill stream the copy of batch n+2 queued behind the fill
	 * of batch n+1 (streams are FIFO) and the device idled through it; a batch that is uploaded AGAIN first lets
	 * its own earlier passes drain */
	hipStream_t s0 = bits_ ? E_->upload_stream() : E_->stream(0);
	if (bits_)
		for (int first = 0; first < Engine::kMaxSlots && first < 64; ++first)
			if ((issued_ >> first) & 1ull) HIP_TRY(hipStreamWaitEvent(s0, ev_[first][2], 0));
	if (bits_ && bits_wide_ && hand_bytes_ > 0)
		for (int sl = 0; sl < nslots_; ++sl) HIP_TRY(hipMemsetAsync(arena_ + hand_off_[sl], 0, hand_bytes_, s0));
	if (bits_ && lone_.on && lone_.hand_bytes > 0) HIP_TRY(hipMemsetAsync(arena_ + lone_.hand_off, 0, lone_.hand_bytes, s0));
	HIP_TRY(hipMemcpyAsync(arena_, h_in_, in_bytes_, hipMemcpyHostToDevice, s0));
	if (!ev_up_) HIP_TRY(hipEventCreateWithFlags(&ev_up_, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(ev_up_, s0));
	for (int sl = 0; sl < nst; ++sl)
		if (E_->stream(sl) != s0) HIP_TRY(hipStreamWaitEvent(E_->stream(sl), ev_up_, 0));
	return CSADP_OK;
}

int FillBatch::run()
{
	if (!laid_out_) return CSADP_ERR_STATE;
	++pending_;
	ran_ = true;
	return CSADP_OK;
}

int FillBatch::flush()
{
	{ const int brc = E_->bind(); if (brc != CSADP_OK) return brc; }
	if (pending_ == 0) return CSADP_OK;
	const int k = pending_;
	pending_ = 0;
	if (bits_) return flush_bits(k);
	/* cell-per-lane batches: pass i goes to slot i % slots on that slot's own stream; kernels of different streams overlap */
	for (int i = 0; i < k; ++i) {
		const int sl = next_slot_;
		next_slot_ = (next_slot_ + 1) % nslots_;
		last_slot_ = sl;
		const int rc = run_slot_cells(sl, false);
		if (rc != CSADP_OK) return rc;
	}
	return CSADP_OK;
}

/* Bit-parallel mode: enqueue k passes as merged launches of up to bits_group_ passes.  Launches
 * rotate over bits_streams_ streams; stream q owns slots [q * group, (q + 1) * group) and every launch
 * starts at its stream's first slot, so the passes per launch never depend on how many passes earlier
 * flushes carried (a slot is reused in stream order). */
int FillBatch::flush_bits(int k)
{
	Engine &E = *E_;
	const int mainN = E.main_streams();
	if (k == 1 && lone_.on && idle_now() && E.device_idle()) {
		const int q = base_stream_ % mainN;
		int rc = launch_bits_pass(lone_.slot, 1, E.stream(q), E.stream(mainN + q), false, true);
		if (rc == CSADP_OK) rc = E.mark_busy(mainN + q, E.stream(mainN + q));
		if (rc != CSADP_OK) return rc;
		last_stream_ = mainN + q;
		last_first_ = last_slot_ = lone_.slot;
		last_group_ = 1;
		return CSADP_OK;
	}
	while (k > 0) {
		const bool piped = nslots_ > 1;
		const int qi = piped ? next_stream_ : 0;                        /* which of the batch's streams */
		const int parity = piped ? (launch_no_ / bits_streams_) & 1 : 0;  /* which of that stream's two slot ranges */
		const int q = (base_stream_ + qi) % mainN;
		const int qs = piped ? mainN + q : q;                           /* its side stream */
		const int first = (qi * 2 + parity) * bits_group_;
		const int g = std::min(k, bits_group_);
		int rc = launch_bits_pass(first, g, E.stream(q), E.stream(qs), false);
		if (rc == CSADP_OK) rc = E.mark_busy(qs, E.stream(qs));
		last_stream_ = qs;
		if (rc != CSADP_OK) return rc;
		last_first_ = first;
		last_slot_ = first + g - 1;
		last_group_ = g;
		next_stream_ = (next_stream_ + 1) % bits_streams_;
		++launch_no_;
		k -= g;
	}
	return CSADP_OK;
}

/* g merged passes (slots first .. first+g-1): [pack planes] and fill on stream st, then traceback and [expand
 * rows] on `side` behind an event, so st is free for the fill of its other slot range at once; a range is
 * re-entered only after its own previous traceback has finished.  The three events of a launch live at
 * ev_[first].  serial = the wait-free form of the chunked fill: one launch per chunk index (check_abort). */
bool FillBatch::idle_now()
{
	for (int first = 0; first < Engine::kMaxSlots && first < 64; ++first)
		if (((issued_ >> first) & 1ull) && hipEventQuery(ev_[first][2]) != hipSuccess) return false;
	return true;
}

int FillBatch::launch_bits_pass(int first, int g, hipStream_t st, hipStream_t side, bool serial, bool lone)
{
	const int nj = (int)bjobs_.size();
	/* the shape of this launch: the batch's own, or the lone pass's (csadp_engine.h) */
	const int words = lone ? 1 : bits_words_, chunk = lone ? lone_.chunk : bits_chunk_;
	const bool wide = lone ? lone_.wide : bits_wide_;
	const size_t tiles_off = lone ? lone_.tiles_off : tiles_off_, serial_off = lone ? lone_.serial_off : serial_off_;
	const size_t ntiles = lone ? lone_.tiles.size() : tiles_.size();
	const std::vector<size_t> &chunk_first = lone ? lone_.chunk_first : chunk_first_;
	last_lone_ = lone;
	hipEvent_t *ev = ev_[first];
	const BitJob *bj = reinterpret_cast<const BitJob *>(arena_ + jobs_off_[first]);
	int *abort_word = reinterpret_cast<int *>(arena_ + abort_off_);
	if (slot_used_[first] && side != st) HIP_TRY(hipStreamWaitEvent(st, ev[2], 0));
	HIP_TRY(hipEventRecord(ev[0], st));
	if (io_) HIP_TRY(launch_pack_planes(arena_, bj, g * nj, st));
	if (wide) {
		/* hand-off words between the chunks of a job carry this pass' epoch (see nw_fill_bits_wide) */
		const uint32_t epoch = Engine::next_epoch();
		/* (shared workgroups: four table entries per workgroup, the lists count entries) */
		const bool shared = wide_shared_ && !lone;
		const int per = shared ? 4 : 1;
		if (!serial) {
			HIP_TRY(launch_fill_bits_wide(words, chunk, arena_, bj, nj, g, reinterpret_cast<const TileRef *>(arena_ + tiles_off), (int)ntiles / per, epoch,
			                              abort_word, st, shared));
		} else {
			for (size_t c = 0; c + 1 < chunk_first.size(); ++c)
				HIP_TRY(launch_fill_bits_wide(words, chunk, arena_, bj, nj, g, reinterpret_cast<const TileRef *>(arena_ + serial_off) + chunk_first[c],
				                              (int)(chunk_first[c + 1] - chunk_first[c]) / per, epoch, abort_word, st, shared));
		}
	} else {
		if (!lone && bits_pack_ > 1)
			HIP_TRY(launch_fill_bits_shared(words, arena_, bj, nj, g, reinterpret_cast<const TileRef *>(arena_ + tiles_off_), (int)tiles_.size() / 4, bits_lds_pad_,
			                                abort_word, st));
		else HIP_TRY(launch_fill_bits(words, arena_, bj, g * nj, lone ? 4 : bits_maxstrips_, lone ? 0 : bits_lds_pad_, abort_word, st));
	}
	if (!serial && wide && test_abort_)                   /* testing: see run_slot_cells */
		HIP_TRY(hipMemsetAsync(arena_ + abort_off_, 1, 4, st));
	HIP_TRY(hipEventRecord(ev[1], st));
	if (side != st) HIP_TRY(hipStreamWaitEvent(side, ev[1], 0));
	HIP_TRY(launch_traceback_bits(words, arena_, bj, g * nj, side));
	if (io_) HIP_TRY(launch_expand_rows(arena_, bj, g * nj, side));
	HIP_TRY(hipEventRecord(ev[2], side));
	slot_used_[first] = true;
	issued_ |= 1ull << first;
	return CSADP_OK;
}

/* Did a bounded wait inside a bit-parallel fill run out (the word is shared by every launch since
 * upload())?  Then no pass since then can be trusted: repeat the LAST pass -- the one whose results
 * fetch() hands out -- on the wait-free path.  Chunked fills (nw_fill_bits_wide) wait across
 * workgroups and rely on dispatch order for speed, never for results: launched chunk by chunk the
 * producer of every hand-off has finished before its consumer starts. */
int FillBatch::check_abort()
{
	if ((!bits_ && !cells_mode_) || !h_abort_) return CSADP_OK;
	if (cells_mode_) {
		hipStream_t st = home_stream(last_slot_);
		HIP_TRY(hipMemcpyAsync(h_abort_, arena_ + abort_off_, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (*h_abort_ == 0) return CSADP_OK;
		fprintf(stderr, "csadp: a wait of the cell-per-lane fill timed out; repeating the pass chunk by chunk\n");
		++recoveries_;
	++E_->recoveries;
		HIP_TRY(hipMemsetAsync(arena_ + abort_off_, 0, 4, st));
		const int rc = run_slot_cells(last_slot_, true);
		if (rc != CSADP_OK) return rc;
		HIP_TRY(hipMemcpyAsync(h_abort_, arena_ + abort_off_, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (*h_abort_ != 0) {
			fprintf(stderr, "csadp: the chunk-by-chunk repeat timed out as well\n");
			return CSADP_ERR_HIP;
		}
		return CSADP_OK;
	}
	/* the callers have waited for the batch's launches: read the word on the copy stream, not behind later batches */
	HIP_TRY(hipMemcpyAsync(h_abort_, arena_ + abort_off_, 4, hipMemcpyDeviceToHost, E_->copy_stream()));
	HIP_TRY(hipStreamSynchronize(E_->copy_stream()));
	if (*h_abort_ == 0) return CSADP_OK;
	hipStream_t st = E_->stream(last_stream_);
	if (!(last_lone_ ? lone_.wide : bits_wide_)) {
		fprintf(stderr, "csadp: a wait inside the bit-parallel fill kernel timed out\n");
		return CSADP_ERR_HIP;
	}
	fprintf(stderr, "csadp: a cross-workgroup wait of the chunked fill timed out; repeating the pass chunk by chunk\n");
	++recoveries_;
	++E_->recoveries;
	HIP_TRY(hipMemsetAsync(arena_ + abort_off_, 0, 4, st));
	const int rc = launch_bits_pass(last_first_, last_slot_ - last_first_ + 1, st, st, true, last_lone_);
	if (rc != CSADP_OK) return rc;
	HIP_TRY(hipMemcpyAsync(h_abort_, arena_ + abort_off_, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (*h_abort_ != 0) {
		fprintf(stderr, "csadp: the chunk-by-chunk repeat timed out as well\n");
		return CSADP_ERR_HIP;
	}
	return CSADP_OK;
}

/* Enqueue ONE pass (fill + traceback) of slot sl on stream sl. */
/* one pass of the cell-per-lane kernels on slot sl: fill (one launch, or one per chunk index on the
 * wait-free path), traceback */
int FillBatch::run_slot_cells(int sl, bool serial)
{
	hipStream_t st = home_stream(sl);
	hipEvent_t *ev = ev_[sl];
	const CellJob *cj = reinterpret_cast<const CellJob *>(arena_ + jobs_off_[sl]);
	int *abort_word = reinterpret_cast<int *>(arena_ + abort_off_);
	if (slot_used_[sl]) HIP_TRY(hipStreamWaitEvent(st, ev[2], 0));
	HIP_TRY(hipEventRecord(ev[0], st));
	/* the hand-off granules between chunks are valid when they carry this pass' epoch: unique per process,
	 * and the hand regions are zeroed when a batch is laid out (epoch 0 is never used), so whatever an
	 * earlier pass or an earlier owner of the arena left there is never mistaken for this pass' data */
	const uint32_t epoch = E_->next_epoch();
	/* Few workgroups (one per compute unit at most): every chain is alone on its units and the fill takes as long as its hand-offs do --
	 * the layout with a fetcher and a publisher wave (csadp_cells.hip, fetch_granules, publish_halves).  More: a compute unit holds two
	 * workgroups of four waves, but only one of six.  (ONE matrix of 391 chunks in that layout, its later chunks starting as the first ones end: a 200 kbp pair fills in
	 * 19.15 ms, as in the plain layout, and the bit-parallel path stays ahead host to host: 20.5 against 22.3 ms.) */
	/* (sharers: the other round groups of the batch in flight, and the fills of this batch's other slots that have not finished) */
	int own_in_flight = 1;
	for (int o = 0; o < nslots_; ++o)
		if (o != sl && slot_used_[o] && hipEventQuery(ev_[o][1]) != hipSuccess) ++own_in_flight;
	const bool fetch = (int)tiles_.size() <= cells_fetch_limit(*E_, std::max(own_in_flight, E_->cells_sharers.load(std::memory_order_relaxed)));
	if (!serial) {
		HIP_TRY(launch_fill_cells(wide_, fetch, arena_, cj, reinterpret_cast<const TileRef *>(arena_ + tiles_off_), (int)tiles_.size(), epoch, abort_word, st,
		                          config().test_slow_publisher));
	} else {
		for (size_t c = 0; c + 1 < chunk_first_.size(); ++c)
			HIP_TRY(launch_fill_cells(wide_, fetch, arena_, cj, reinterpret_cast<const TileRef *>(arena_ + serial_off_) + chunk_first_[c],
			                          (int)(chunk_first_[c + 1] - chunk_first_[c]), epoch, abort_word, st));
	}
	if (!serial && test_abort_)                           /* testing: pretend a bounded wait ran out, so that the repeat path runs */
		HIP_TRY(hipMemsetAsync(arena_ + abort_off_, 1, 4, st));
	HIP_TRY(hipEventRecord(ev[1], st));
	HIP_TRY(launch_traceback_cells(arena_, cj, (int)cjobs_.size(), tb_max_bands_, tb_max_groups_, st));
	HIP_TRY(hipEventRecord(ev[2], st));
	slot_used_[sl] = true;
	return CSADP_OK;
}

int FillBatch::sync()
{
	{ const int brc = E_->bind(); if (brc != CSADP_OK) return brc; }
	const int rc = flush();
	if (rc != CSADP_OK) return rc;
	/* bit-parallel batches wait for THEIR launches (the event behind each one's traceback), not for the streams
	 * they ran on: batches of one engine share a few streams, and a streaming caller's later batches are queued
	 * on them long before an earlier one is fetched -- waiting for the streams made every fetch wait for all of
	 * them, the device then idled through the host's part of the fetch (trace: one batch on the device at a time) */
	if (bits_) {
		const int wrc = wait_batch();
		if (wrc != CSADP_OK) return wrc;
	} else {
		for (int sl = 0; sl < E_->main_streams(); ++sl) HIP_TRY(hipStreamSynchronize(E_->stream(sl)));
	}
	return ran_ ? check_abort() : CSADP_OK;
}

int FillBatch::wait_batch()
{
	if (ev_up_) HIP_TRY(hipEventSynchronize(ev_up_));             /* the upload (a batch may never have run) */
	for (int first = 0; first < Engine::kMaxSlots && first < 64; ++first)
		if ((issued_ >> first) & 1ull) HIP_TRY(hipEventSynchronize(ev_[first][2]));
	return CSADP_OK;
}

int FillBatch::download()
{
	{ const int brc = E_->bind(); if (brc != CSADP_OK) return brc; }
	if (!ran_) return CSADP_ERR_STATE;
	if (cells_mode_ && nslots_ == 1) {
		/* the lock-step rounds of N-sequence tasks: ONE wait per round.  Abort word and results are queued behind
		 * the traceback and waited for together (sync() + check_abort() + the copy were three round trips to the
		 * device, ~20 us each, 15 times per alignment) */
		const int frc = flush();
		if (frc != CSADP_OK) return frc;
		hipStream_t st = home_stream(last_slot_);
		HIP_TRY(hipMemcpyAsync(h_abort_, arena_ + abort_off_, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipMemcpyAsync(h_res_, arena_ + res_off_[last_slot_], res_bytes_, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		pull_pending_ = false;                         /* one slot: st is the stream the pull ran on */
		if (*h_abort_ == 0) return CSADP_OK;
		const int arc = check_abort();                 /* repeats the pass chunk by chunk */
		if (arc != CSADP_OK) return arc;
		HIP_TRY(hipMemcpyAsync(h_res_, arena_ + res_off_[last_slot_], res_bytes_, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		return CSADP_OK;
	}
	{
		const int rc = sync();            /* flush pending passes; results of the LAST pass are wanted */
		if (rc != CSADP_OK) return rc;
	}
	hipStream_t st = bits_ ? E_->stream(last_stream_) : home_stream(last_slot_);
	const size_t want = (io_ && !want_strings_) ? sum_bytes_ : res_bytes_;
	if (bits_) st = E_->copy_stream();            /* sync() has waited for the batch: nothing to order the copy behind */
	HIP_TRY(hipMemcpyAsync(h_res_, arena_ + res_off_[last_slot_], want, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	return CSADP_OK;
}

const uint8_t *FillBatch::ops(int j) const { return h_res_ + extra_[j].res_ops; }
const int32_t *FillBatch::summary(int j) const { return reinterpret_cast<const int32_t *>(h_res_ + extra_[j].res_summary); }

int FillBatch::timing(csadp_timing *t)
{
	{ const int brc = E_->bind(); if (brc != CSADP_OK) return brc; }
	if (!ran_) return CSADP_ERR_STATE;
	{
		const int rc = flush();
		if (rc != CSADP_OK) return rc;
	}
	memset(t, 0, sizeof(*t));
	hipEvent_t *ev = ev_[bits_ ? last_first_ : last_slot_];
	HIP_TRY(hipEventSynchronize(ev[2]));
	{
		const int before = recoveries_;
		const int rc = check_abort();                 /* a launch that gave up early must not be timed as if it had run */
		if (rc != CSADP_OK) return rc;
		if (recoveries_ != before) HIP_TRY(hipEventSynchronize(ev[2]));
	}
	HIP_TRY(hipEventElapsedTime(&t->fill_ms, ev[0], ev[1]));
	HIP_TRY(hipEventElapsedTime(&t->traceback_ms, ev[1], ev[2]));
	HIP_TRY(hipEventElapsedTime(&t->total_ms, ev[0], ev[2]));
	t->launch_passes = bits_ ? last_group_ : 1;
	t->merge_group = bits_ ? bits_group_ : 1;
	t->recoveries = recoveries_;
	t->device_io = io_ ? 1 : 0;
	t->bit_parallel = bits_ ? 2 : 0;
	t->words_per_lane = bits_ ? (last_lone_ ? 1 : bits_words_) : 0;
	t->streams = bits_ ? bits_streams_ : 1;
	t->cells = cells_;
	t->fill_launches = (int)diag_off_.size() - 1;
	t->fill_tiles = (long long)tiles_.size();
	t->dir_bytes = dir_bytes_;
	t->border_bytes = border_bytes_;
	return CSADP_OK;
}

}  // namespace csadp
