/*
 * csadp_progressive.h -- host side of one ProgressiveDP task: everything of
 * /root/reference/source/dynamicprogramming.c:906-1171 that is NOT the matrix fill or the
 * direction walk (those run on the GPU): sequence ordering (:276-308), profile seeding
 * (:926-944), the border-refresh rule (:957, survey quirk Q1), applying a traceback to the
 * strings and the profile (:1050-1155) and the gapped-column refinement (:643-899).
 */
#ifndef CSADP_PROGRESSIVE_H
#define CSADP_PROGRESSIVE_H

#include <stdint.h>

#include <string>
#include <vector>

#include "csadp.h"

namespace csadp {

class Progressive {
public:
	/* Validate the task, order the sequences, seed the profile.  CSADP_OK or an error. */
	int init(const csadp_task &task);

	/* Move to the next progressive step that needs a matrix fill; steps whose row sequence
	 * is empty are completed here (:950-956).  False when the task is finished. */
	bool next_fill();

	/* geometry of the pending fill */
	int nrows() const { return nrows_; }
	int ncols() const { return consensus_; }
	int nprev() const { return step_; }            /* the reference's loop variable i */
	int border_i() const { return border_i_; }     /* i in force when the borders were last refreshed */
	bool stale_borders() const { return stale_; }

	/* Device-format inputs of the pending fill (gain form, csadp_device.h).
	 *   coltab[0..ncols_pad)      diag gains: bytes 8*sv[c]+2 (narrow) or 6-bit sv[c] (wide); 0 beyond ncols
	 *   leftc[0..ncols_pad)       left gain 4*(sv[4]-i)+1; 0 beyond ncols
	 *   rowshift[0..nrows)        bfe offset of each row letter: 8*code (narrow) / 6*code (wide)
	 *   top[0..ncols_pad]         X of border row 0 = 4*H[0][k]; beyond ncols: last value */
	void write_tables(uint32_t *coltab, int32_t *leftc, int ncols_pad, uint8_t *rowshift, int32_t *top, bool wide) const;

	/* first fill with fresh borders: H[0][k] = -k, H[r][0] = -r, every column one letter (the
	 * conditions of the bit-parallel kernel, csadp_bits.hip) */
	bool unit_borders() const { return step_ == 1 && border_i_ == 1 && !stale_; }
	/* bit planes of the column letters (profile of one sequence) and of the row letters */
	void write_tables_bits(uint32_t *cols, int nwords, uint32_t *rows, int rowwords) const;

	/* Apply the GPU traceback of the pending fill: ops in walk order (DIR_* codes, from cell
	 * (nrows,ncols) backwards), remj/remk = rows/columns left when the walk hit a border.
	 * The DP score H[nrows][ncols] is re-derived on the way as border value + sum of the move
	 * scores along the path; if expect_score is given it must agree (test seam). */
	int apply_trace(const uint8_t *ops, int nops, int remj, int remk, const int *expect_score = nullptr, bool defer_refinement = false);

	/* DeleteGappedColumns (:643-899) of a step applied with defer_refinement, in three parts so that a caller
	 * with many tasks can spread the second one over all of its threads:
	 *   refine_prepare()      number of independent chunks of candidate columns (0: nothing to refine)
	 *   refine_speculate(c)   scores the candidates of chunk c on the alignment as it stands -- read-only, any
	 *                         number of threads at once
	 *   refine_commit()       the reference's left-to-right pass; a candidate whose neighbourhood no slide has
	 *                         touched takes its score from the speculation, every other one is scored afresh */
	int refine_prepare();
	void refine_speculate(int chunk);
	void refine_commit();

	/* DP score H[nrows][ncols] of the pending fill from its traced path (border value + sum of
	 * move scores, as apply_trace derives it) without touching the strings or the profile. */
	int score_from_trace(const uint8_t *ops, int nops, int remj, int remk, int *score) const;

	/* Publish the result (malloc'd strings, original index order). */
	int finish(csadp_result *res);

	/* H-domain views of the pending fill for the test seam (include/csadp_debug.h) */
	const int *debug_sv() const { return sv_.data(); }
	const int *debug_border_top() const { return border_top_.data(); }
	void debug_rowcodes(signed char *out) const;

	long long cells() const { return cells_; }
	int fills() const { return fills_; }
	int nseq() const { return nseq_; }

private:
	char char_at(int pos, int seq) const;
	void delete_gapped_columns(int numseqs, int maxnongaps);

	/* scratch of one candidate evaluation (one per thread) */
	struct RefineScratch {
		struct Run { bool valid; int a, b; bool at, bt; int gl, gr; };
		std::vector<int> movers, block, nextgaps, affected, statv, movv, vacp;
		std::vector<signed char> codev;
		std::vector<Run> runs;                   /* per row: what the last walk found around the last candidate */
		std::vector<int> keep_affected, keep_best;   /* the winning slide: bestnposaffected, bestworkingsv */
		int keep_maxaffected = 0;
		void size_for(int consensus, int nseq);
		void reserve(int columns);
	};
	struct RefineEval { int nmov = 0, bestshift = 0, lo = 0; };
	/* scores candidate column `col` (split of the gap buffer at col - 1, columns >= col live `gap` higher) */
	void refine_evaluate(int col, int gap, int numseqs, RefineScratch &S, RefineEval &R) const;

	int nseq_ = 0;
	/* private copy of the task's small arrays; the sequence texts themselves are borrowed
	 * and must outlive this object */
	std::vector<const char *> texts_;
	std::vector<int> textsizes_, rotations_, starts_, ends_;
	std::vector<int> order_, len_;
	std::vector<int> sv_;                        /* (consensus+1) x 5, column 0 unused (:931) */
	std::vector<std::string> str_;               /* aligned strings, ORIGINAL index order       */
	std::vector<char> have_;                     /* str_[s] assigned                            */
	int consensus_ = 0;
	int step_ = 0;                               /* i                                           */
	int nrows_ = 0;
	bool pending_ = false;
	bool empty_task_ = false;
	/* survey Q1: borders are refreshed only if consensus != prevconsensus || nrows > prevnrows */
	int prevconsensus_ = 0, prevnrows_ = 0;
	std::vector<int> border_top_;                /* H[0][k] as of the last refresh              */
	int border_i_ = 0;
	bool stale_ = false;
	int last_score_ = 0;
	long long cells_ = 0;
	int fills_ = 0;
	double dgc_ms_ = 0;                          /* CSADP_TRACE_HOST: time spent in delete_gapped_columns */
	std::string tokens_;                         /* '.' per fill (:1156), '!' per all-gap column met (:689) */
	/* a refinement that was deferred / speculated on */
	int refine_numseqs_ = 0;                     /* 0: none pending */
	std::vector<unsigned char> spec_state_;      /* per original column: 0 unknown, 1 no movers, 2 no slide, 3 slides */
	std::vector<int> spec_lo_;                   /* leftmost column the evaluation read */
	static constexpr int kRefineChunk = 256;     /* columns per speculation chunk */
};

}  // namespace csadp

#endif
