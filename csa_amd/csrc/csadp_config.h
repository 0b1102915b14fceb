/*
 * csadp_config.h -- every environment switch of libcsadp.so, read ONCE per process into one struct (INTEGRATION.md lists them).
 * config() returns the cached values; csadp_debug_reload_config() (include/csadp_debug.h) reads the environment again -- the seam
 * through which the test-suite flips switches between calls inside one process (tests/conftest.py calls it before every test and
 * after every monkeypatched variable).  -1 / INT_MIN in an int field = "not set: the library chooses".
 */
#ifndef CSADP_CONFIG_H
#define CSADP_CONFIG_H

namespace csadp {

struct Config {
	/* which kernels */
	bool bits = true;               /* CSADP_BITS: first fills on the bit-parallel path (0: every fill on the cell-per-lane path) */
	bool device_io = true;          /* CSADP_DEVICE_IO: 2-sequence batches packed / expanded on the device */
	int bits_words = -1;            /* CSADP_BITS_WORDS: words of 32 columns per lane (1-4) instead of the batch's own choice */
	int bits_chunk = 0;             /* CSADP_BITS_CHUNK: strips per workgroup of a chunked launch (4, 8, 16) */
	int bits_group = -1;            /* CSADP_BITS_GROUP: passes merged into one launch */
	int bits_streams = -1;          /* CSADP_BITS_STREAMS: launches in flight */
	int bits_lds_pad = -1;          /* CSADP_BITS_LDS_PAD: KiB of dynamic LDS a fill workgroup reserves */
	bool lone_shape = true;         /* CSADP_LONE_SHAPE: a pass flushed alone takes the spread shape (csadp_engine.h) */
	int stream_rotate = -1;         /* CSADP_STREAM_ROTATE: batches of one engine start on different streams (default: by batch size) */
	bool cells_fetch_forced = false; /* CSADP_CELLS_FETCH was set: taken as it is; else min(256, the device's compute units) */
	int bits_pack = 1;              /* CSADP_BITS_PACK: jobs of one or two strips share four-wave workgroups of nw_fill_bits (1, the default; 0: a workgroup per job) */
	int cells_order = -1;           /* CSADP_CELLS_ORDER: work list of a cell-per-lane launch: 0 job by job, 1 chunk level by chunk level, -1 by size (layout_cells) */
	int cells_fetch_wgs = 256;      /* CSADP_CELLS_FETCH: cell-per-lane launches of at most this many workgroups carry a fetcher wave (0: none) */
	bool lone_cells = true;         /* CSADP_LONE_CELLS: at most 8 large square-ish pairs alone take the cell-per-lane path (FillBatch::layout) */
	int slots = 4;                  /* CSADP_SLOTS: result / scratch sets of a pipelined cell-per-lane batch */
	/* band-parallel traceback of the profile steps */
	int tb_band_min = 512;          /* CSADP_TB_BAND_MIN: rows from which a matrix' walk is cut into bands */
	bool tb_band_forced = false;    /* ... was set explicitly (then also for matrices more than twice as wide as high) */
	int tb_corridor = 3;            /* CSADP_TB_CORRIDOR: groups of 1024 start columns scouted per band (at least; more while the scouts fit the chip) */
	bool tb_corridor_forced = false; /* ... was set explicitly: exactly that many */
	bool pull_uploads = true;       /* CSADP_PULL_UPLOADS: a round's tables are pulled from pinned memory by a kernel */
	/* host */
	int round_groups = 2;           /* CSADP_ROUND_GROUPS: task groups whose lock-step rounds run side by side */
	bool round_groups_forced = false; /* ... was set explicitly; else batches of 128 tasks and more take 4 (csadp_api.cpp) */
	int refine_speculate = 0;       /* CSADP_REFINE_SPECULATE: one-task entry points speculate DeleteGappedColumns too (1: same thread, 2: threads) */
	int host_threads = 0;           /* CSADP_HOST_THREADS: size of the host pool (0: by the machine); read when the pool starts */
	bool trace_host = false;        /* CSADP_TRACE_HOST: per-stage host timings on stderr */
	/* devices */
	bool share_device = false;      /* CSADP_SHARE_DEVICE: several ordinals may name one device (rehearsals) */
	int local_rank = 0;             /* LOCAL_RANK: default device ordinal */
	/* testing */
	int test_slow_publisher = 0;    /* CSADP_TEST_SLOW_PUBLISHER: the publisher wave of nw_fill_cells sleeps this many x 127 x 64 cycles per half block (tests) */
	int test_range_log2 = 31;       /* CSADP_TEST_RANGE_LOG2: the gain form's score range as a power of two (31; tests lower it to reach CSADP_ERR_RANGE) */
	int test_hbm_limit_mb = 0;      /* CSADP_TEST_HBM_LIMIT_MB: pretend the device has only this much free memory (0: ask the device) */
	bool test_force_abort = false;  /* CSADP_TEST_FORCE_ABORT: pretend a bounded wait of a chunked fill ran out */
};

const Config &config();
void reload_config();
/* passes the primary engine's batches have repeated chunk by chunk so far (csadp_engine.cpp); 0 before the library is initialised */
long primary_engine_recoveries();

}  // namespace csadp

#endif
