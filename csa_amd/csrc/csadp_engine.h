/*
 * csadp_engine.h -- device runtime of libcsadp: device selection, the batch arena in HBM,
 * tile scheduling (one launch per tile anti-diagonal) and HIP-event timing.
 */
#ifndef CSADP_ENGINE_H
#define CSADP_ENGINE_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "csadp.h"
#include "csadp_device.h"

namespace csadp {

class FillBatch;

/*
 * One Engine per HIP device of the process (streams, arena cache, the cached csadp_align_batch
 * arena).  The process-wide entry points (csadp_init, csadp_align_batch, ...) use the PRIMARY
 * engine -- the device csadp_init selected -- and the *_on(device) entry points address any other
 * one, so a host program can drive every GPU of a node from one thread per GPU.  HIP's current
 * device is a per-thread setting: every entry point that touches HIP calls bind() first.
 */
class Engine {
public:
	/* the primary engine, created by the first csadp_init / first use; nullptr + rc on failure */
	static Engine *primary(const csadp_config *cfg, int *rc);
	/* get-or-create the engine of HIP device `device` (thread-safe) */
	static Engine *open(int device, const csadp_config *cfg, int *rc);
	static Engine *primary_if_ready();
	static void shutdown_all();
	int bind() const;                /* hipSetDevice(device_) on the calling thread */
	int rotate_stream() { return stream_rr_.fetch_add(1); }
	/* 1, 2, ... (never 0 modulo 2^24 within 16 M passes): tags of the hand-off granules of nw_fill_cells */
	static uint32_t next_epoch();
	static uint32_t set_epoch_counter(uint32_t v);   /* test seam (csadp_debug_set_epoch) */
	bool ready() const { return ready_; }
	static constexpr int kMaxSlots = 33;             /* 32 pipelined passes (streams x two slot ranges x passes per launch) + the slot of a lone pass's shape */
	hipStream_t stream(int slot = 0) const { return streams_[slot]; }
	int slots() const { return slots_; }
	/* streams [0, main_streams()) carry fills; stream main_streams() + q is the side stream of main stream q:
	 * the traceback of a bit-parallel launch runs there, so the next fill of stream q need not wait for it */
	int main_streams() const { return slots_ < 2 ? 2 : slots_; }
	int nstreams() const { return nstreams_; }
	/* result downloads and abort-word reads of batches that have finished: a stream of its own, so that they never
	 * queue behind what LATER batches have put on the compute streams */
	hipStream_t copy_stream() const { return copy_stream_; }
	hipStream_t upload_stream() const { return upload_stream_; }
	int device() const { return device_; }
	const char *name() const { return name_; }
	int compute_units() const { return cus_; }
	bool verbose() const { return verbose_; }
	/* pools of released HBM arenas / pinned staging buffers (csadp_engine.cpp) */
	void give_arena(uint8_t *ptr, size_t bytes);
	uint8_t *take_arena(size_t need, size_t *got);
	void give_pinned(uint8_t *ptr, size_t bytes);
	uint8_t *take_pinned(size_t need, size_t *got);
	void drop_arena_cache();
	/* csadp_warmup: the first memset / copy of either direction and of either size class (blit kernel, copy engine) on every
	 * stream of the engine, paid now (profiles/r05_dropin_hiptrace: 3-9 ms each inside a process' first batch) */
	int warm_copy_paths();
	/* Engine-wide idleness for the launch shaping of pair batches: every pass of EVERY batch records the end of its work here (one event
	 * per stream, re-recorded by each launch on it); device_idle() is true when all of them have been reached.  A batch's own events say
	 * nothing about the device: a streaming caller's batches are each "idle" by themselves (round-4 ADVICE). */
	int mark_busy(int stream_index, hipStream_t st);
	bool device_idle();

	/* csadp_align_batch keeps one FillBatch (HBM arena + pinned staging, grow-only) per device alive
	 * between calls: the drop-in adapter calls it once per un-anchored gap (~50 times per input set) */
	/* launches of nw_fill_cells that may share the device: the round groups of the batch in flight (csadp_align_batch) -- the helper-wave
	 * layout is taken only while ALL of them together fit one workgroup per compute unit (cells_fetch_limit) */
	std::atomic<int> cells_sharers{1};
	std::atomic<long> recoveries{0};              /* passes of any batch of this engine repeated chunk by chunk (FillBatch::check_abort) */
	std::mutex batch_mutex;
	FillBatch *cached_batch = nullptr;
	std::vector<FillBatch *> extra_batches;       /* ... and the arenas of the further round groups (guarded by batch_mutex too) */

private:
	int init(int device, const csadp_config *cfg);
	void shutdown();
	bool ready_ = false;
	bool verbose_ = false;
	int device_ = 0;
	int cus_ = 0;
	char name_[256] = {0};
	int slots_ = 2, nstreams_ = 0;
	hipStream_t streams_[2 * kMaxSlots] = {};
	hipStream_t copy_stream_ = nullptr;
	hipStream_t upload_stream_ = nullptr;
	hipEvent_t busy_ev_[2 * kMaxSlots] = {};
	std::atomic<unsigned long long> busy_mask_{0};       /* streams whose busy event has ever been recorded */
	std::atomic<int> stream_rr_{0};
	std::mutex pool_mutex_;
	std::vector<std::pair<uint8_t *, size_t>> arena_pool_, pinned_pool_;
};

/*
 * A batch of independent matrix fills executed in lock-step: all tiles on the same tile
 * anti-diagonal (a + s) of every job go into one kernel launch.
 *
 * Arena layout in HBM (one allocation, offsets in FillJob):
 *   [ jobs | tiles | per-job inputs: coltab, rowshift, top ]   <- one H2D copy
 *   [ per-job results: summary, ops ]                          <- one D2H copy      } x slots
 *   [ per-job scratch: state, handoff, dirs ]                  <- never leaves HBM  } x slots
 *
 * Slots: consecutive run() calls rotate over `slots` independent result+scratch sets, each
 * on its own HIP stream, so the tail of one pass (few tiles per launch, then the latency-
 * bound traceback) overlaps the head of the next.  Inputs are read-only and shared.
 */
class FillBatch {
public:
	explicit FillBatch(Engine *engine) : E_(engine) {}
	~FillBatch();
	Engine *engine() const { return E_; }
	FillBatch(const FillBatch &) = delete;
	FillBatch &operator=(const FillBatch &) = delete;

	/* pipelined = rotate run() calls over Engine::slots() result/scratch sets; otherwise one */
	void set_pipelined(bool on) { pipelined_ = on; }
	/* Batches that run side by side from different host threads (the round groups of csadp_align_batch) take different
	 * streams of the engine: slot s of this batch uses stream (base + s) modulo the engine's main streams. */
	void set_stream_base(int base) { stream_base_ = base; }
	void clear();
	/* register a fill; returns its job index */
	int add(int nrows, int ncols, int nprev, int left_i);
	int njobs() const { return (int)jobs_.size(); }
	/* compute the arena layout and the tile schedule, (re)allocate HBM and pinned staging */
	int layout();
	/* grow the (grow-only) arena and the two pinned staging buffers to at least these sizes now, so that the layouts of a
	 * first real batch find them (csadp_warmup) */
	int reserve(size_t arena_bytes, size_t in_bytes, size_t res_bytes);
	/* host staging pointers for the inputs of job j (valid after layout) */
	uint32_t *coltab(int j);
	int32_t *leftc(int j);
	bool wide() const { return wide_; }   /* table format of this batch (csadp_device.h) */
	/* bit-parallel mode (BitJob): every job is a first fill with unit borders (the caller vouches
	 * for the borders with allow_bits) and at most kBitMaxStrips*2048 columns wide.  Host tables:
	 * two bit planes of the column letters and two of the row letters. */
	void allow_bits(bool on) { bits_allowed_ = on; }
	bool bits() const { return bits_; }
	bool lone_pairs_take_cells() const;                            /* before layout(): these jobs would leave the bit-parallel path */
	uint32_t *bit_cols(int j);         /* [2][bit_nwords(j)] */
	int bit_nwords(int j) const;
	uint32_t *bit_rows(int j);         /* [2][bit_rowwords(j)] */
	int bit_rowwords(int j) const;
	/* Device-side I/O of 2-sequence tasks (bit-parallel mode only; csadp_pairio.hip): the batch holds the
	 * raw texts, the planes are packed and the aligned rows written on the device.  add_text() registers a
	 * text once per (pointer, size); set_pair_io() names the column / row sequence of job j and the text
	 * index of each region's first letter.  After layout(): copy the letters to text_staging(id). */
	int add_text(const char *text, int size);
	void set_pair_io(int j, int text_col, int first_col, int text_row, int first_row);
	bool device_io() const { return io_; }
	int ntexts() const { return (int)texts_.size(); }
	uint8_t *text_staging(int id);
	const char *text_source(int id) const { return texts_[(size_t)id].ptr; }
	int text_size(int id) const { return texts_[(size_t)id].size; }
	/* results of the last pass in device-I/O mode: the aligned row of the column (0) / row (1) sequence */
	const uint8_t *out_row(int j, int which) const;
	void want_strings(bool on) { want_strings_ = on; }   /* false: download() fetches the summaries only */
	uint8_t *rowshift(int j);        /* points at row 1 (index padl) */
	int32_t *top(int j);
	int ncols_pad(int j) const;
	int upload();                    /* inputs -> HBM, waits for the copy */
	int upload_async();              /* the same without waiting: later work of every engine stream is ordered behind it */
	/* Request one more pass.  Passes are enqueued at the next sync()/timing()/download(): tiled
	 * kernels: pass i on slot i % slots and that slot's stream; bit-parallel kernels: consecutive
	 * passes merged into launches that rotate over 2-3 streams (flush_bits). */
	int run();
	int flush();                     /* enqueue what run() requested */
	int sync();                      /* flush, then wait for all streams */
	int download();                  /* results of the LAST run() -> host (blocking) */
	const uint8_t *ops(int j) const;
	const int32_t *summary(int j) const;   /* nops, remj, remk, 0 */
	/* the traceback kernel summed the move scores of its path (checkpoint mode of the bit-parallel path) */
	int timing(csadp_timing *t);

private:
	Engine *E_;
	int stream_base_ = 0;
	hipStream_t home_stream(int slot) const { return E_->stream((stream_base_ + slot) % E_->main_streams()); }
	struct Extra { int ncols_pad; size_t in_coltab, in_leftc, in_rowshift, in_top, res_summary, res_ops; };
	struct BitExtra { size_t in_cols, in_rows; size_t res_out[2]; };
	struct TextRef { const char *ptr; int size; size_t off; };
	struct PairIo { int text[2], first[2]; };
	std::vector<TextRef> texts_;
	std::vector<PairIo> pairio_;
	bool io_ = false, want_strings_ = true;
	size_t sum_bytes_ = 0;             /* leading part of a slot's result region that holds the summaries */
	int layout_bits();
	int layout_cells();
	int run_slot_cells(int sl, bool serial);
	std::vector<CellJob> cjobs_;
	bool cells_mode_ = false;
	size_t hand_off_[Engine::kMaxSlots] = {}, hand_bytes_ = 0;
	int flush_bits(int k);
	int launch_bits_pass(int first, int g, hipStream_t st, hipStream_t side, bool serial, bool lone = false);
	int check_abort();
	int bits_group_ = 1, last_group_ = 1, bits_streams_ = 2, next_stream_ = 0, recoveries_ = 0, bits_pack_ = 1;
	bool wide_shared_ = false;                    /* chunked launch: tiles_ holds four {job, strip} entries per workgroup */
	std::vector<TileRef> shared_table_;           /* bits_pack_ > 1: one {job, strip} per wave of the shared workgroups of a pass */
	int bits_words_ = 1;                          /* words of 32 columns per lane of this batch's bit-parallel kernels */
	int bits_lds_pad_ = 0;                        /* dynamic LDS a one-workgroup-per-job fill launch reserves on top (bounds the workgroups per compute unit) */
	bool test_abort_ = false;                     /* CSADP_TEST_FORCE_ABORT, read when the batch is laid out */
	bool pull_uploads_ = true;                    /* CSADP_PULL_UPLOADS: profile steps' tables are read from pinned memory by a kernel */
	int base_stream_ = 0, last_stream_ = 0, last_first_ = 0, launch_no_ = 0;
	bool pull_pending_ = false;                   /* a pull upload of the staging may still be running on home_stream(0) */
	void settle_pull();
	int tb_max_bands_ = 0, tb_max_groups_ = 0;     /* band-parallel traceback: most bands / scout groups of a banded job */
	unsigned long long issued_ = 0;             /* bit-parallel path: slot ranges (by first slot) with a launch on record */
	int wait_batch();                           /* ... and the wait for exactly those launches (events, not streams) */
	size_t abort_off_ = 0, serial_off_ = 0;
	std::vector<TileRef> serial_tiles_;
	std::vector<size_t> chunk_first_;
	int alloc_buffers();             /* arena / staging allocation shared by the layouts */
	int finish_layout();
	std::vector<BitJob> bjobs_;
	std::vector<BitExtra> bextra_;
	bool bits_ = false, bits_allowed_ = false, bits_wide_ = false;
	int bits_maxstrips_ = 1, bits_chunk_ = kBitMaxStrips;
	/* A pipelined batch whose single pass does not fill the chip keeps a SECOND shape for a pass that is flushed alone onto an
	 * idle device: one word per lane, its strips spread four to a workgroup over every compute unit (one wave per SIMD), on
	 * one extra slot with its own job table, work list and scratch; texts and results are the batch's own.  The batch's
	 * regular shape (two or three words per lane, several passes in flight) is the fast one in a steady state, a third of
	 * it for one pass of 128 pairs (BENCH_r03 one_shot: 0.315 of value). */
	struct LoneShape {
		bool on = false;
		int slot = -1, chunk = 4;
		bool wide = false;
		size_t tiles_off = 0, serial_off = 0, hand_off = 0, hand_bytes = 0;
		std::vector<TileRef> tiles, serial_tiles;
		std::vector<size_t> chunk_first;
	} lone_;
	bool last_lone_ = false;                      /* the last launch took the lone shape */
	bool idle_now();                              /* every launch of the batch on record has finished */
	int *h_abort_ = nullptr;
	std::vector<FillJob> jobs_;
	std::vector<Extra> extra_;
	std::vector<TileRef> tiles_;
	std::vector<size_t> diag_off_;   /* tiles_ index of the first tile of each diagonal, +1 sentinel */
	size_t in_bytes_ = 0, res_bytes_ = 0, total_bytes_ = 0;
	size_t jobs_off_[Engine::kMaxSlots] = {}, res_off_[Engine::kMaxSlots] = {};
	size_t tiles_off_ = 0;
	int nslots_ = 1, next_slot_ = 0, last_slot_ = 0, pending_ = 0;
	bool slot_used_[Engine::kMaxSlots] = {};
	uint8_t *arena_ = nullptr;
	size_t arena_cap_ = 0;
	uint8_t *h_in_ = nullptr;        /* pinned mirror of the input region */
	size_t h_in_cap_ = 0;
	uint8_t *h_res_ = nullptr;       /* pinned mirror of the result region */
	size_t h_res_cap_ = 0;
	hipEvent_t ev_[Engine::kMaxSlots][3] = {};
	hipEvent_t ev_up_ = nullptr;
	bool laid_out_ = false, ran_ = false, pipelined_ = false, wide_ = false;
	long long cells_ = 0, dir_bytes_ = 0, border_bytes_ = 0;
};

}  // namespace csadp

#endif
