/*
 * csadp_device.h -- device-side data layout shared by the HIP kernels and the host engine.
 *
 * Score representation used on the device (all int32), "gain form":
 *     X[r][k] = 4 * H[r][k] + 4 * i * r                (low 2 bits: direction tag)
 * where H is the reference's dpmatrix (dynamicprogramming.c:996-1025) and i the number of
 * already aligned sequences.  Adding 4*i per row cancels the row-gap score, so the three
 * candidate moves become
 *     up   : + 0                              tag 0   ('U')   -- no instruction at all
 *     left : + 4 * (sv[k][4] - i)  + 1        tag 1   ('L')
 *     diag : + 8 * sv[k][c]        + 2        tag 2   ('D')
 * and max3 over (X | tag) picks the reference's H and the reference's direction with the
 * reference's tie-break (D >= L >= U, :1014-1025) in one instruction.
 */
#ifndef CSADP_DEVICE_H
#define CSADP_DEVICE_H

#include <stdint.h>

namespace csadp {

constexpr int kLanes = 64;           /* wave64 */

/* direction codes stored 2 bit per cell */
enum : int { DIR_U = 0, DIR_L = 1, DIR_D = 2 };

/* One matrix fill as the host registers it (one progressive step of one task); the engine turns it into a BitJob or a
 * CellJob when the batch is laid out. */
struct FillJob {
	int32_t nrows, ncols;
	int32_t nstrips;
	int32_t steps_pad;
	int32_t nprev;            /* i                                                                             */
	int32_t leftmul;          /* 4*(i - left_i): X of border column 0 is leftmul * r (0 on fresh borders)      */
};

/* work item of a chunked launch: chunk `a` of job `job` (csadp_bits.hip, csadp_cells.hip) */
struct TileRef {
	int32_t job;
	int32_t a;
	int32_t s;                /* unused */
	int32_t first;            /* unused */
};

/*
 * Bit-parallel job: ONE first fill (i = 1, fresh borders H[0][k] = -k, H[r][0] = -r, scores
 * +1 / -1 / -1).  Adjacent cells of such a matrix differ by -1..2, so a row is held as three
 * thermometer bit planes of its horizontal differences and a lane advances `wpl` words of 32 columns per
 * step with word-wide logic and carry-propagating additions (csadp_bits.hip: nw_fill_bits).
 * A wave owns a strip of 64 * wpl words; lane L of a strip computes row (l - L) at its local step l.
 * The fill stores no directions: after every block of 32 steps a lane saves its state and the 3 x 32
 * carries it has put out, and the traceback replays the 16-lane x 32-step pieces its path crosses (csadp_bits.hip, K2c).
 */
constexpr int kBitMaxStrips = 16;    /* waves per workgroup */
constexpr int kBitBlock = 32;        /* steps per hand-off block between strips */
constexpr int kBitMaxWords = 4;      /* words per lane: 1, 2 or 3 chosen per batch, 4 on request (csadp_engine.cpp: layout_bits) */
struct BitJob {
	uint64_t colplanes;       /* u32 [2][nwords_pad] bit b of word w of plane p = bit p of the letter code of column 32w+b */
	uint64_t rowplanes;       /* u32 [2][rowwords]   same for the rows (0-based), zero padded                             */
	uint64_t ckpt;            /* u32 [nstrips][steps_pad/32][wpl][64][4] lane state after every block of 32 steps, per word */
	                          /*     of a lane: nH0, H1, H2; the fourth word of a lane's word 0: the carries its ">= 2"     */
	                          /*     plane put out during the block, first step in bit 31                                   */
	uint64_t hand;            /* u32 [nstrips][steps_pad/32][64][2] the same for the planes ">= 1" and ">= 0": with them a   */
	                          /*     replay can start at any lane of a strip                                                */
	uint64_t xhand;           /* u64 [nchunks-1][steps_pad/32][3] chunked launches: {carries, epoch} of lane 63 of a chunk's   */
	                          /*     last strip per block and plane, valid when they carry the launch's epoch; zeroed at upload */
	uint64_t ops;             /* u8 traceback ops, walk order                                                              */
	uint64_t summary;         /* i32 [4] nops, remaining rows, remaining cols, 0                                           */
	int32_t nrows, ncols;
	int32_t nstrips;
	int32_t steps_pad;        /* local steps per strip, multiple of kBitBlock, >= nrows + 64                               */
	int32_t nwords_pad;       /* 64 * wpl * nstrips                                                                        */
	int32_t rowwords;         /* steps_pad / 32                                                                            */
	int32_t wpl;              /* words of 32 columns per lane                                                              */
	int32_t pad_;
	/* device-side input packing / output expansion of 2-sequence tasks (nw_pack_planes, nw_expand_rows): the batch
	 * holds the raw circular texts; [0] = the column sequence (the shorter region, dynamicprogramming.c:290-307),
	 * [1] = the row sequence.  text = 0 (no arena offset is 0: the job table sits there) marks a job whose planes
	 * the host wrote.                                                                                            */
	uint64_t text[2];         /* u8 [size] the sequence as loaded (un-rotated)                                             */
	uint64_t out[2];          /* u8 [nrows + ncols + 1] the aligned row of that sequence, NUL-terminated (per slot)        */
	uint64_t istatus;         /* i32 input status: bit 0 = a letter other than A,C,G,T inside a region (survey Q4)         */
	int32_t size[2];          /* textsizes                                                                                 */
	int32_t first[2];         /* rotation + start, wrapped once: text index of the region's first letter (CharAt)          */
};

/*
 * Cell-per-lane job (csadp_cells.hip): ONE matrix fill of any progressive step (any i, stale borders
 * included) as a persistent wavefront.  A lane owns kCellCols adjacent columns, a wave a strip of 128, a
 * workgroup a chunk of kCellWaves strips; lane L of a strip computes row (l - L + 1) of its columns at its
 * local step l, so the whole matrix takes nrows + ncols / 2 steps of 13 VALU instructions instead of the
 * tiled kernel's 16 columns x 2 rows per lane-step and a launch per tile anti-diagonal: the latency of a
 * single large matrix -- what the reference's own use (mode N: a few wide gaps, up to 63 sequential profile
 * steps each) is bound by.  Waves of a workgroup hand the right edge over through an LDS ring, workgroups of
 * a job through epoch-tagged granules in `hand` in HBM.  Directions: the columns A (even) of strip s are
 * "virtual strip" 2s, the columns B 2s + 1; a virtual strip is [steps_pad / 16][64 lanes] words of 16 tags.
 */
constexpr int kCellWaves = 4;        /* strips per workgroup: one wave per SIMD of a compute unit      */
constexpr int kCellBlock = 32;       /* steps per hand-off block                                       */
constexpr int kCellCols = 2;         /* adjacent columns per lane                                      */
constexpr int kCellStripCols = kLanes * kCellCols;   /* columns per strip (wave)                     */

struct CellJob {
	uint64_t coltab;          /* u32 [ncols_pad] diag gains per row letter (narrow bytes / wide 6-bit counts, as FillJob) */
	uint64_t leftc;           /* i32 [ncols_pad] left gain 4*(sv[4]-i) + 1                                                */
	uint64_t rowshift;        /* u8  [steps_pad + 576] bfe offset of the letter of row j at index j-1, zero padded         */
	uint64_t top;             /* i32 [ncols_pad + 1] X of border row 0 (possibly stale, survey Q1)                        */
	uint64_t dirs;            /* u32 [nstrips][2][steps_pad/16][64]: the 2-bit tags of 16 consecutive local steps of one   */
	                          /*     column (half 0: the lane's column A, half 1: B); local step l of lane L is row l - L + 1; */
	                          /*     first step in bits 1:0                                                                 */
	uint64_t hand;            /* u64 [nchunks-1][steps_pad] granules {X, epoch << 8} leaving the last column of each chunk   */
	                          /*     but the last: valid when they carry the launch's epoch                                  */
	uint64_t ops;             /* u8 traceback ops, walk order                                                              */
	uint64_t summary;         /* i32 [4] nops, remaining rows, remaining cols, 0                                           */
	int32_t nrows, ncols;
	int32_t nstrips;          /* ceil(ncols / 128)                                                                         */
	int32_t nchunks;          /* ceil(nstrips / kCellWaves)                                                                */
	int32_t steps_pad;        /* local steps per strip, multiple of kCellBlock, >= nrows + 64                              */
	int32_t nprev;            /* i                                                                                          */
	int32_t leftmul;          /* X of border column 0 is leftmul * r                                                       */
	int32_t banded;           /* 1: the band-parallel traceback (csadp_cells_tb.hip); 0: one serial walk                   */
	/* band-parallel traceback: bands of kBandRows rows, band b = rows (b*R, min((b+1)*R, nrows)], entered in its bottom row */
	uint64_t tb_tab;          /* u16 [nbands][tb_pitch]: the walk from the x-th scouted start column of band b leaves the   */
	                          /*     band `value` columns further left; kBandUnknown: not finished inside the band        */
	uint64_t tb_ent;          /* i32 [nbands + 2]: column in which the path enters band b (-1: never); then end row, end column */
	uint64_t tb_cnt;          /* i32 [nbands]: ops of band b                                                               */
	uint64_t tb_scratch;      /* u8 [nrows + ncols + 64]: the ops of band b from (nrows - entry row) + (ncols - entry column) on */
	int32_t nbands;           /* ceil(nrows / kBandRows)                                                                   */
	int32_t tb_pitch;         /* entries per band in tb_tab: the scouted starts, tb_groups * kScoutStarts                  */
	int32_t tb_groups;        /* groups of kScoutStarts start columns scouted per band, around the corner-to-corner line    */
	int32_t pad_;
};

/* band-parallel traceback of the profile steps */
constexpr int kBandRows = 128;       /* rows per band (multiple of 16: a band starts on a direction-word boundary)          */
constexpr int kBandStride = 16;      /* columns between the scouts' start columns                                           */
constexpr int kBandWords = kBandRows / 16 + 4;   /* direction words of one column that hold a band's rows, lane skew included */
constexpr int kScoutStarts = 64;     /* start columns per scout workgroup (a lane each): 1024 columns = 8 strips            */
constexpr int kScoutStrips = kScoutStarts * kBandStride / 128 + 3;     /* strips staged in LDS by a scout workgroup: 72 KiB */
constexpr int kScoutPitch = kBandWords | 1;   /* words per column of the scouts' window: odd, so that starts 16 columns apart read different banks */
constexpr int kScoutGuard = 16;      /* words in front of the window's guard column (a parked lane reads up to one word below its column) */
constexpr int kScoutMaxLeft = 128;   /* L moves after which a scout gives up (a band the path crosses so is walked exactly) */
constexpr int kScoutCap = kBandRows + kScoutMaxLeft;                   /* steps a scout can take (a multiple of 16)         */
constexpr int kEmitStrips = 4;       /* strips staged by an emitting workgroup                                              */
constexpr unsigned kBandUnknown = 0xffffu;
constexpr unsigned kBandMoved = 0x8000u;      /* a scout's `moved` is below this */

}  // namespace csadp

#endif
