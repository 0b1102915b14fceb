/*
 * csadp_progressive.cpp -- host side of one ProgressiveDP task (see csadp_progressive.h).
 * Line numbers cite /root/reference/source/dynamicprogramming.c unless stated otherwise.
 */
#include "csadp_progressive.h"
#include "csadp_hostpar.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <stdio.h>

#include "csadp_device.h"
#include "csadp_config.h"

namespace csadp {

namespace {

constexpr int kSym = 5;   /* A C G T - (:9-12) */
constexpr int kGap = 4;

/* compiled-in scores of the reference (:16-19) */
constexpr int kMatch = +1, kDoubleGap = 0, kMismatch = -1, kIndel = -1;

inline int code_of(char ch)   /* :74-85 */
{
	switch (ch) {
	case 'A': return 0;
	case 'C': return 1;
	case 'G': return 2;
	case 'T': return 3;
	case '-': return 4;
	default: return -1;
	}
}

}  // namespace

/* CharAt, alignment.c:16-20 */
char Progressive::char_at(int pos, int seq) const
{
	int i = rotations_[seq] + pos;
	if (i >= textsizes_[seq]) i -= textsizes_[seq];
	return texts_[seq][i];
}

int Progressive::init(const csadp_task &task)
{
	nseq_ = task.nseq;
	if (nseq_ < 2 || nseq_ > CSADP_MAX_SEQS) return CSADP_ERR_ARG;
	if (!task.texts || !task.textsizes || !task.rotations || !task.starts || !task.ends) return CSADP_ERR_ARG;
	texts_.assign(task.texts, task.texts + nseq_);
	textsizes_.assign(task.textsizes, task.textsizes + nseq_);
	rotations_.assign(task.rotations, task.rotations + nseq_);
	starts_.assign(task.starts, task.starts + nseq_);
	ends_.assign(task.ends, task.ends + nseq_);
	int maxgap = 0;
	for (int s = 0; s < nseq_; ++s) {
		const int size = task.textsizes[s];
		if (!task.texts[s] || size < 0) return CSADP_ERR_ARG;
		if (task.starts[s] < 0 || task.ends[s] < task.starts[s] || task.ends[s] > size) return CSADP_ERR_ARG;
		if (task.ends[s] > task.starts[s] && (task.rotations[s] < 0 || task.rotations[s] >= size)) return CSADP_ERR_ARG;
		maxgap = std::max(maxgap, task.ends[s] - task.starts[s]);
		for (int p = task.starts[s]; p < task.ends[s]; ++p) {
			const int c = code_of(char_at(p, s));
			if (c < 0 || c > 3) return CSADP_ERR_ALPHABET;   /* reference: scorevector[][-1], UB */
		}
	}
	order_.resize(nseq_);
	len_.resize(nseq_);
	str_.assign(nseq_, std::string());
	have_.assign(nseq_, 0);
	if (maxgap == 0) {           /* :916 */
		empty_task_ = true;
		step_ = nseq_;
		return CSADP_OK;
	}
	/* SortSequencesForDP :286-307 -- selection sort with swap (not stable) */
	for (int i = 0; i < nseq_; ++i) {
		order_[i] = i;
		len_[i] = task.ends[i] - task.starts[i];
	}
	for (int i = 0; i < nseq_ - 1; ++i) {
		int minpos = i;
		for (int j = i + 1; j < nseq_; ++j)
			if (len_[j] < len_[minpos]) minpos = j;
		if (minpos != i) {
			std::swap(order_[i], order_[minpos]);
			std::swap(len_[i], len_[minpos]);
		}
	}
	/* seed the profile with the shortest sequence, :926-944 */
	consensus_ = len_[0];
	prevconsensus_ = 0;
	prevnrows_ = 0;
	sv_.assign((size_t)(consensus_ + 1) * kSym, 0);
	const int n = order_[0];
	str_[n].resize(consensus_);
	have_[n] = 1;
	for (int m = 1; m <= consensus_; ++m) {
		const char ch = char_at(task.starts[n] + m - 1, n);
		str_[n][m - 1] = ch;
		sv_[(size_t)m * kSym + code_of(ch)]++;
	}
	step_ = 1;
	pending_ = false;
	return CSADP_OK;
}

bool Progressive::next_fill()
{
	if (pending_) return true;
	while (step_ < nseq_) {
		const int n = order_[step_];
		nrows_ = len_[step_];
		if (nrows_ == 0) {                                 /* :950-956: all gaps, profile untouched */
			str_[n].assign((size_t)consensus_, '-');
			have_[n] = 1;
			++step_;
			continue;
		}
		if (consensus_ != prevconsensus_ || nrows_ > prevnrows_) {   /* :957 */
			border_top_.resize((size_t)consensus_ + 1);
			int colgap = 0;
			border_top_[0] = 0;
			for (int k = 1; k <= consensus_; ++k) {               /* :969-973 */
				const int g = sv_[(size_t)k * kSym + kGap];
				colgap += kDoubleGap * g + kIndel * (step_ - g);
				border_top_[k] = colgap;
			}
			border_i_ = step_;                                    /* :963,:967 */
			prevnrows_ = nrows_;                                  /* :986 */
			stale_ = false;
		} else {
			stale_ = true;
		}
		pending_ = true;
		return true;
	}
	return false;
}

void Progressive::write_tables(uint32_t *coltab, int32_t *leftc, int ncols_pad, uint8_t *rowshift, int32_t *top,
                               bool wide) const
{
	const int i = step_;
	const int ncols = consensus_;
	for (int k = 1; k <= ncols; ++k) {
		const int *col = &sv_[(size_t)k * kSym];
		uint32_t w = 0;
		for (int c = 0; c < 4; ++c)
			w |= wide ? (uint32_t)(col[c] & 63) << (6 * c) : (uint32_t)((8 * col[c] + 2) & 255) << (8 * c);
		coltab[k - 1] = w;
		leftc[k - 1] = 4 * (col[kGap] - i) + 1;
	}
	for (int k = ncols; k < ncols_pad; ++k) { coltab[k] = 0; leftc[k] = 0; }
	const int n = order_[step_];
	const int start = starts_[n];
	const int width = wide ? 6 : 8;
	for (int j = 0; j < nrows_; ++j) rowshift[j] = (uint8_t)(width * code_of(char_at(start + j, n)));
	for (int k = 0; k <= ncols; ++k) top[k] = 4 * border_top_[k];
	for (int k = ncols + 1; k <= ncols_pad; ++k) top[k] = top[ncols];
}

void Progressive::write_tables_bits(uint32_t *cols, int nwords, uint32_t *rows, int rowwords) const
{
	/* the destination is pinned staging memory: one store per finished word, no read-modify-write */
	const int ncols = consensus_;
	for (int w = 0; w * 32 < ncols; ++w) {
		uint32_t p0 = 0, p1 = 0;
		const int kend = std::min(ncols, 32 * w + 32);
		for (int k = 32 * w + 1; k <= kend; ++k) {
			const int *col = &sv_[(size_t)k * kSym];
			int c = 0;
			while (c < 3 && col[c] == 0) ++c;             /* the one letter of this column */
			p0 |= (uint32_t)(c & 1) << ((k - 1) & 31);
			p1 |= (uint32_t)(c >> 1) << ((k - 1) & 31);
		}
		cols[w] = p0;
		cols[nwords + w] = p1;
	}
	const int n = order_[step_];
	const int start = starts_[n];
	for (int w = 0; w * 32 < nrows_; ++w) {
		uint32_t p0 = 0, p1 = 0;
		const int jend = std::min(nrows_, 32 * w + 32);
		for (int j = 32 * w; j < jend; ++j) {
			const int c = code_of(char_at(start + j, n));
			p0 |= (uint32_t)(c & 1) << (j & 31);
			p1 |= (uint32_t)(c >> 1) << (j & 31);
		}
		rows[w] = p0;
		rows[rowwords + w] = p1;
	}
}

void Progressive::debug_rowcodes(signed char *out) const
{
	const int n = order_[step_];
	for (int j = 0; j < nrows_; ++j) out[j] = (signed char)code_of(char_at(starts_[n] + j, n));
}

int Progressive::score_from_trace(const uint8_t *ops, int nops, int remj, int remk, int *score) const
{
	if (!pending_ || !score) return CSADP_ERR_STATE;
	const int i = step_;
	const int n = order_[i];
	int j = nrows_, k = consensus_;
	int pos = ends_[n] - 1;
	long long sc = 0;
	for (int t = 0; t < nops; ++t) {
		const int op = ops[t];
		if (j <= 0 || k <= 0) return CSADP_ERR_HIP;
		if (op == DIR_D) {
			const int c = code_of(char_at(pos, n));
			const int *col = &sv_[(size_t)k * kSym];
			sc += kMatch * col[c] + kIndel * col[kGap] + kMismatch * (i - (col[c] + col[kGap]));
			--pos; --j; --k;
		} else if (op == DIR_L) {
			const int g = sv_[(size_t)k * kSym + kGap];
			sc += kDoubleGap * g + kIndel * (i - g);
			--k;
		} else if (op == DIR_U) {
			sc += kIndel * i;
			--pos; --j;
		} else {
			return CSADP_ERR_HIP;
		}
	}
	if (j != remj || k != remk) return CSADP_ERR_HIP;
	sc += (j > 0) ? -(long long)border_i_ * j : (long long)border_top_[(size_t)k];
	*score = (int)sc;
	return CSADP_OK;
}

/* Traceback application, :1033-1155, driven by the op list instead of dpdirs. */
int Progressive::apply_trace(const uint8_t *ops, int nops, int remj, int remk, const int *expect_score, bool defer_refinement)
{
	if (!pending_) return CSADP_ERR_STATE;
	const int i = step_;
	const int n = order_[i];
	const int nrows = nrows_;
	const int ncols = consensus_;
	const int newcons = nops + remj + remk;             /* :1034-1049 */
	const bool inplace = (newcons == ncols);            /* :1050 / :1059 */

	/* consistency of the walk with the matrix geometry */
	{
		int dj = remj, dk = remk;
		for (int t = 0; t < nops; ++t) {
			if (ops[t] == DIR_D) { ++dj; ++dk; }
			else if (ops[t] == DIR_L) ++dk;
			else if (ops[t] == DIR_U) ++dj;
			else return CSADP_ERR_HIP;
		}
		if (dj != nrows || dk != ncols) return CSADP_ERR_HIP;
	}

	/* The profile is only read again by a later step of the same task (and by
	 * DeleteGappedColumns, i > 1): the last step of a 2-sequence task need not maintain it. */
	const bool profile = !(nseq_ == 2 && i == 1);
	/* The reference writes, for every cell of the path, one letter into each of the i old strings and five
	 * counts (:1075-1105): i scattered writes per op.  Here the walk only records where every new column
	 * comes from (`src`: the old column, 0 = a column the row sequence opens) and the new row; profile and
	 * strings are then rebuilt string by string -- the same bytes, sequential access. */
	std::string cur((size_t)newcons, '\0');
	std::vector<int> src;
	if (!inplace) src.assign((size_t)newcons, 0);
	int j = nrows, k = ncols, m = newcons - 1;
	int pos = ends_[n] - 1;
	long long score = 0;                                    /* sum of move scores, :993-998 */
	for (int t = 0; t < nops; ++t, --m) {                   /* :1072-1114 */
		const int op = ops[t];
		if (op == DIR_D) {
			const char ch = char_at(pos, n);
			const int c = code_of(ch);
			const int *col = &sv_[(size_t)k * kSym];
			score += kMatch * col[c] + kIndel * col[kGap] + kMismatch * (i - (col[c] + col[kGap]));
			if (!inplace) src[(size_t)m] = k;               /* :1075-1079 */
			cur[(size_t)m] = ch;
			--pos; --j; --k;
		} else if (op == DIR_L) {
			const int g = sv_[(size_t)k * kSym + kGap];
			score += kDoubleGap * g + kIndel * (i - g);
			if (!inplace) src[(size_t)m] = k;
			cur[(size_t)m] = '-';
			--k;
		} else {                                            /* :1100-1105: a new column, old sequences get '-' */
			score += kIndel * i;
			cur[(size_t)m] = char_at(pos, n);
			--pos; --j;
		}
	}
	/* the walk stopped on a border cell: H[j][0] = -border_i*j (:967) or H[0][k] (:972) */
	score += (j > 0) ? -(long long)border_i_ * j : (long long)border_top_[(size_t)k];
	if (expect_score && *expect_score != (int)score) return CSADP_ERR_HIP;
	for (; j > 0; --j, --m) {                               /* :1115-1127 */
		cur[(size_t)m] = char_at(pos, n);
		--pos;
	}
	for (; k > 0; --k, --m) {                               /* :1128-1138 */
		if (!inplace) src[(size_t)m] = k;
		cur[(size_t)m] = '-';
	}
	if (inplace) {                                          /* no new column: the row's symbols join the old columns */
		if (profile)
			for (int c = 0; c < newcons; ++c) sv_[(size_t)(c + 1) * kSym + code_of(cur[(size_t)c])]++;
	} else {                                                /* :1139-1153 */
		/* old columns keep their order, so src[] is a few long runs of consecutive columns: block copies */
		struct Span { int dst, src, len; };
		std::vector<Span> spans;
		for (int c = 0; c < newcons;) {
			if (!src[(size_t)c]) { ++c; continue; }
			int e = c + 1;
			while (e < newcons && src[(size_t)e] == src[(size_t)e - 1] + 1) ++e;
			spans.push_back(Span{c, src[(size_t)c], e - c});
			c = e;
		}
		if (profile) {
			std::vector<int> newsv((size_t)(newcons + 1) * kSym, 0);
			for (const Span &sp : spans)
				memcpy(&newsv[(size_t)(sp.dst + 1) * kSym], &sv_[(size_t)sp.src * kSym], (size_t)sp.len * kSym * sizeof(int));
			for (int c = 0; c < newcons; ++c) {
				int *dst = &newsv[(size_t)(c + 1) * kSym];
				if (!src[(size_t)c]) dst[kGap] = i;
				dst[code_of(cur[(size_t)c])]++;
			}
			sv_.swap(newsv);
		}
		std::string nw;
		for (int l = 0; l < i; ++l) {
			std::string &old = str_[order_[l]];
			nw.assign((size_t)newcons, '-');
			for (const Span &sp : spans) memcpy(&nw[(size_t)sp.dst], &old[(size_t)sp.src - 1], (size_t)sp.len);
			old.swap(nw);
		}
	}
	str_[n].swap(cur);
	have_[n] = 1;
	prevconsensus_ = ncols;                                 /* :1033 */
	consensus_ = newcons;
	last_score_ = (int)score;
	cells_ += (long long)nrows * (long long)ncols;
	++fills_;
	tokens_.push_back('.');                                 /* :1156 */
	if (i > 1) {
		if (defer_refinement) refine_numseqs_ = i + 1;                /* :1157, run by refine_prepare / _speculate / _commit */
		else delete_gapped_columns(i + 1, (i + 1) / 2);
	}
	++step_;
	pending_ = false;
	return CSADP_OK;
}

/*
 * DeleteGappedColumns, :643-899.  For every column with at least numseqs-maxnongaps gaps:
 * try to slide the residue blocks of the non-gap sequences into the gaps that follow them
 * (right first, then left), score each slide with the sum-of-pairs style column formula,
 * take the LAST slide whose gain is >= the best so far (:791), apply it, drop the all-gap
 * columns it leaves behind and re-examine from the left-most dropped column.
 *
 * Three parts.  refine_evaluate scores ONE candidate column and reads only; refine_commit is the
 * reference's left-to-right pass with its side effects; refine_speculate runs refine_evaluate for a chunk
 * of candidates on the alignment as it stands before the pass -- any number of threads at once.  Slides
 * are rare (one candidate in hundreds), and a candidate's score depends only on the columns it reads, so the
 * pass takes the speculated outcome of every candidate whose neighbourhood lies right of everything a slide
 * has touched so far and scores the others afresh: same decisions, same order, but the scoring -- all of the
 * time of this function -- spread over the caller's threads instead of one per task.
 */
void Progressive::RefineScratch::size_for(int consensus, int nseq)
{
	(void)consensus;                                                  /* the column arrays grow with the widest search met (reserve) */
	runs.assign((size_t)nseq, Run{false, 0, 0, false, false, -1, -1});
}

void Progressive::RefineScratch::reserve(int columns)
{
	if ((int)codev.size() >= columns + 2) return;
	const size_t n = (size_t)std::max(columns + 2, 2 * (int)codev.size());
	codev.resize(n);
	vacp.resize(n + 1);
	statv.resize(n * kSym);
	movv.reserve(n * kSym);
}

void Progressive::refine_evaluate(int col, int gap, int numseqs, RefineScratch &S, RefineEval &R) const
{
	std::vector<int> &movers = S.movers, &block = S.block, &nextgaps = S.nextgaps, &affected = S.affected;
	std::vector<int> &statv = S.statv, &movv = S.movv, &vacp = S.vacp;
	std::vector<signed char> &codev = S.codev;
	typedef RefineScratch::Run Run;
	movers.clear();
	for (int t = 0; t < numseqs; ++t) {
		const int s = order_[t];
		if (str_[s][(size_t)(col + gap) - 1] != '-') movers.push_back(s);
	}
	const int nmov = (int)movers.size();
	R.nmov = nmov;
	R.bestshift = 0;
	R.lo = col;
	if (nmov == 0) return;
	int bestscore = 0, bestshift = 0;
	block.assign(nmov, 0);
	nextgaps.assign(nmov, 0);
	affected.assign(nmov, 0);
	for (int dir = +1; dir >= -1; dir -= 2) {
		const int limit = (dir > 0) ? consensus_ + 1 : 0;
		int farthest = 0, minnext = consensus_;
		bool blocked = false;
		/* column `col` itself is the head of the right segment (offset gap); the columns the leftward
		 * search reads beyond it are in the left segment (offset 0) */
		const int off = dir > 0 ? gap : 0;
		for (int t = 0; t < nmov; ++t) {                          /* :699-715 */
			/* The reference walks the residue run of the mover from `col` to its end for every candidate
			 * column (:700-712); consecutive candidates inside one long run -- e.g. the newest, longest
			 * sequence over a short gap: 6944 candidates x 6993 letters on the third example set -- re-walk
			 * the same letters.  What the walk found is remembered per row: the residues known around the
			 * last candidate, [a, b), whether a / b are the run's true ends, and the lengths of the gap
			 * runs beyond them; a row's memory is dropped when the row changes, everybody's when columns
			 * are deleted.  Same block / nextgaps / blocked values, one walk per run instead of one per
			 * column. */
			const char *row = str_[movers[t]].data() + off - 1;       /* row[j] = logical column j on this side */
			Run &W = S.runs[(size_t)movers[t]];
			if (!(W.valid && W.a <= col && col < W.b)) W = Run{true, col, col + 1, false, false, -1, -1};
			if (dir > 0) {
				if (!W.bt) {
					int j = W.b;
					while (j != limit && row[j] != '-') ++j;
					W.b = j;
					W.bt = true;
					W.gr = -1;
				}
				block[t] = W.b - col;
				if (W.b == limit) { blocked = true; break; }
				if (W.gr < 0) {
					int j = W.b, g = 0;
					while (j != limit && row[j] == '-') { ++g; ++j; }
					W.gr = g;
				}
				nextgaps[t] = W.gr;
			} else {
				if (!W.at) {
					int j = W.a - 1;
					while (j != limit && row[j] != '-') --j;
					W.a = j + 1;
					W.at = true;
					W.gl = -1;
				}
				block[t] = col - W.a + 1;
				if (W.a - 1 == limit) { blocked = true; break; }
				if (W.gl < 0) {
					int j = W.a - 1, g = 0;
					while (j != limit && row[j] == '-') { ++g; --j; }
					W.gl = g;
				}
				nextgaps[t] = W.gl;
				R.lo = std::min(R.lo, W.a - 1 - W.gl);               /* the column that ended the gap run was read too */
			}
			farthest = std::max(farthest, block[t]);
			minnext = std::min(minnext, nextgaps[t]);
		}
		if (blocked) {                                            /* :716-721 */
			if (dir < 0) R.lo = 0;                                /* the walk went to the left end */
			continue;
		}
		for (int t = 0; t < nmov; ++t) affected[t] = block[t] + minnext;
		const int maxaff = farthest + minnext;
		S.reserve(maxaff);
		/* statv = the columns without the movers, movv = the movers' symbols, current = what the movers
		 * score where they stand (:739-761) */
		auto build_arrays = [&]() {
			movv.assign((size_t)maxaff * kSym, 0);
			for (int j = 0; j < maxaff; ++j) memcpy(&statv[(size_t)j * kSym], &sv_[(size_t)(col + dir * j + (j ? off : gap)) * kSym], kSym * sizeof(int));
			int cur = 0;
			for (int t = 0; t < nmov; ++t) {
				const char *row = str_[movers[t]].data() + off - 1;
				for (int j = 0; j < affected[t]; ++j) {
					const int jj = col + dir * j, o = j ? off : gap;
					const int c = code_of(row[jj + o - off]);
					const int *svjj = &sv_[(size_t)(jj + o) * kSym];
					movv[(size_t)j * kSym + c]++;
					statv[(size_t)j * kSym + c]--;
					cur += (c != kGap) ? kMatch * (svjj[c] - 1) + kMismatch * (numseqs - (svjj[c] + svjj[kGap])) + kIndel * svjj[kGap]
					                   : kDoubleGap * (svjj[kGap] - 1) + kIndel * (numseqs - svjj[kGap]);
				}
			}
			return cur;
		};
		if (nmov == 1) {
			/*
			 * One mover (most candidates: a long insertion of one sequence): the same sums without the
			 * per-column count vectors.  Relative column q holds the mover's letter for q < B and one of its
			 * gaps for B <= q < B + G.  After a slide by sh the vacated columns [0, sh) and the columns
			 * [B + sh, B + G) that keep one of the mover's gaps score vac(q) (a prefix sum), and letter q
			 * scores against column q + sh without the mover's own symbol there.
			 */
			const int B = block[0], G = minnext, n = B + G;
			const char *row = str_[movers[0]].data() + off - 1;
			vacp[0] = 0;
			int current = 0;
			for (int q = 0; q < n; ++q) {
				const int o = q ? off : gap;
				const int *sq = &sv_[(size_t)(col + dir * q + o) * kSym];
				int wg;                                               /* gaps of the column once the mover has a gap there */
				if (q < B) {
					const int c = code_of(row[col + dir * q + o - off]);
					codev[(size_t)q] = (signed char)c;
					current += kMatch * (sq[c] - 1) + kMismatch * (numseqs - (sq[c] + sq[kGap])) + kIndel * sq[kGap];
					wg = sq[kGap] + 1;
				} else {
					codev[(size_t)q] = kGap;
					current += kDoubleGap * (sq[kGap] - 1) + kIndel * (numseqs - sq[kGap]);
					wg = sq[kGap];
				}
				vacp[(size_t)q + 1] = vacp[(size_t)q] + (wg == numseqs ? 0 : kDoubleGap * (wg - 1) + kIndel * (numseqs - wg));
			}
			for (int sh = 1; sh <= G; ++sh) {
				int shifted = vacp[(size_t)sh] + vacp[(size_t)n] - vacp[(size_t)B + sh];
				for (int q = 0; q < B; ++q) {
					const int j = q + sh, c = codev[(size_t)q], cj = codev[(size_t)j];
					const int *sj = &sv_[(size_t)(col + dir * j + off) * kSym];
					const int wc = sj[c] + 1 - (cj == c), wg = sj[kGap] - (cj == kGap);
					if (wg == numseqs) continue;
					shifted += kMatch * (wc - 1) + kMismatch * (numseqs - (wc + wg)) + kIndel * wg;
				}
				shifted -= current;
				if (shifted >= bestscore) {                           /* :791 */
					bestshift = dir * sh;
					bestscore = shifted;
				}
			}
			if (bestshift != 0 && bestshift * dir > 0) {              /* the arrays the winner is applied from */
				build_arrays();
				for (int y = 0; y < G; ++y) movv[(size_t)(B + y) * kSym + kGap]--;
			}
		} else {
			const int current = build_arrays();
			/* the columns a slide vacates hold the movers' gaps only: a prefix sum over the slide length */
			vacp[0] = 0;
			for (int j = 0; j < minnext; ++j) {
				const int wg = statv[(size_t)j * kSym + kGap] + nmov;
				vacp[(size_t)j + 1] = vacp[(size_t)j] + (wg == numseqs ? 0 : nmov * (kDoubleGap * (wg - 1) + kIndel * (numseqs - wg)));
			}
			for (int sh = 1; sh <= minnext; ++sh) {               /* :762-795 (workingsv is not kept: only its score is used) */
				for (int t = 0; t < nmov; ++t) {
					movv[(size_t)(affected[t] - 1) * kSym + kGap]--;
					affected[t]--;
				}
				int shifted = vacp[(size_t)sh];
				for (int j = sh; j < maxaff; ++j) {
					const int *mvp = &movv[(size_t)(j - sh) * kSym];
					const int *st = &statv[(size_t)j * kSym];
					const int wg = st[kGap] + mvp[kGap];
					if (wg == numseqs) continue;
					int colscore = 0;
					for (int y = 0; y < kGap; ++y)
						if (mvp[y] != 0) {
							const int wy = st[y] + mvp[y];
							colscore += mvp[y] * (kMatch * (wy - 1) + kMismatch * (numseqs - (wy + wg)) + kIndel * wg);
						}
					if (mvp[kGap] != 0) colscore += mvp[kGap] * (kDoubleGap * (wg - 1) + kIndel * (numseqs - wg));
					shifted += colscore;
				}
				shifted -= current;
				if (shifted >= bestscore) {                       /* :791 */
					bestshift = dir * sh;
					bestscore = shifted;
				}
			}
		}
		if (bestshift != 0 && bestshift * dir > 0) {              /* :796-818 */
			const int sh = bestshift * dir;
			const int back = minnext - sh;
			S.keep_maxaffected = maxaff;
			S.keep_affected.assign(nmov, 0);
			for (int t = 0; t < nmov; ++t) {
				for (int y = 0; y < back; ++y) movv[(size_t)(block[t] + y) * kSym + kGap]++;
				S.keep_affected[t] = block[t] + sh;
			}
			S.keep_best.assign((size_t)maxaff * kSym, 0);
			for (int j = 0; j < maxaff; ++j) {
				for (int y = 0; y < kSym; ++y) {
					int v = statv[(size_t)j * kSym + y];
					if (j >= sh) v += movv[(size_t)(j - sh) * kSym + y];
					S.keep_best[(size_t)j * kSym + y] = v;
				}
				if (j < sh) S.keep_best[(size_t)j * kSym + kGap] += nmov;
			}
		}
	}
	R.bestshift = bestshift;
}

int Progressive::refine_prepare()
{
	if (refine_numseqs_ == 0) return 0;
	spec_state_.assign((size_t)consensus_ + 2, 0);
	spec_lo_.assign((size_t)consensus_ + 2, 0);
	return (consensus_ + kRefineChunk - 1) / kRefineChunk;
}

void Progressive::refine_speculate(int chunk)
{
	const int numseqs = refine_numseqs_, mingaps = numseqs - numseqs / 2;
	const int first = chunk * kRefineChunk + 1, last = std::min(consensus_, first + kRefineChunk - 1);
	static thread_local RefineScratch S;                             /* its column arrays are kept from chunk to chunk */
	bool sized = false;
	RefineEval R;
	for (int col = first; col <= last; ++col) {
		if (sv_[(size_t)col * kSym + kGap] < mingaps) continue;      /* :678 */
		if (!sized) { S.size_for(consensus_, nseq_); sized = true; }
		refine_evaluate(col, 0, numseqs, S, R);
		spec_state_[(size_t)col] = R.nmov == 0 ? 1 : R.bestshift == 0 ? 2 : 3;
		spec_lo_[(size_t)col] = R.lo;
	}
}

void Progressive::refine_commit()
{
	const int numseqs = refine_numseqs_, maxnongaps = numseqs / 2;
	if (numseqs == 0) return;
	refine_numseqs_ = 0;
	const bool trace = config().trace_host;
	const auto t0 = std::chrono::steady_clock::now();
	/*
	 * Storage during the pass: a gap buffer.  The reference deletes a run of all-gap columns by moving
	 * every later column of the profile and of every string down (:865-887), O(consensus x numseqs) per
	 * deletion -- the 19-sequence example set spends more time there than in its 20 000 x 17 000 fills
	 * on the GPU.  Here logical column j lives at physical index j for j <= split and at j + gap behind
	 * it; a deletion moves the split to the run (the scan only moves forward, so those moves add up to
	 * one pass over the columns) and widens the gap.  The logical content after every operation is the
	 * reference's; the arrays are compacted once at the end.
	 */
	int split = consensus_, gap = 0;
	auto phys = [&](int j) { return j <= split ? j : j + gap; };
	auto SV = [&](int col, int sym) -> int & { return sv_[(size_t)phys(col) * kSym + sym]; };
	/* the rows' storage does not move during the pass (they are resized once, at the end) */
	std::vector<char *> rowp((size_t)numseqs), seqp((size_t)nseq_, nullptr);
	for (int t = 0; t < numseqs; ++t) rowp[(size_t)t] = seqp[(size_t)order_[t]] = &str_[order_[t]][0];
	auto CH = [&](int seq, int col) -> char & { return seqp[(size_t)seq][(size_t)phys(col) - 1]; };
	auto move_split = [&](int to) {                 /* make logical columns 1..to the left segment: one block move per array */
		if (gap != 0 && to < split) {               /* columns to+1..split join the right segment */
			const size_t n = (size_t)(split - to);
			memmove(&sv_[(size_t)(to + 1 + gap) * kSym], &sv_[(size_t)(to + 1) * kSym], n * kSym * sizeof(int));
			for (int t = 0; t < numseqs; ++t) memmove(rowp[(size_t)t] + to + gap, rowp[(size_t)t] + to, n);
		} else if (gap != 0 && to > split) {        /* columns split+1..to join the left segment */
			const size_t n = (size_t)(to - split);
			memmove(&sv_[(size_t)(split + 1) * kSym], &sv_[(size_t)(split + 1 + gap) * kSym], n * kSym * sizeof(int));
			for (int t = 0; t < numseqs; ++t) memmove(rowp[(size_t)t] + split, rowp[(size_t)t] + split + gap, n);
		}
		split = to;
	};
	const int mingaps = numseqs - maxnongaps;
	RefineScratch S;
	S.size_for(consensus_, nseq_);
	RefineEval R;
	/* speculation bookkeeping, in the column numbers of the alignment before the pass: `deleted` columns are gone so
	 * far, and nothing right of `dirty` has been read or written by a slide.  A logical column c was column
	 * <= c + deleted then, exactly so when it lies right of `dirty`. */
	const bool speculated = !spec_state_.empty();
	const int spec_n = speculated ? (int)spec_state_.size() - 2 : 0;
	int deleted = 0, dirty = 0;

	for (int col = 1; col <= consensus_; ++col) {
		if (SV(col, kGap) < mingaps) continue;                       /* :678 */
		const int then = col + deleted;
		if (speculated && then <= spec_n && spec_state_[(size_t)then] != 0 && spec_lo_[(size_t)then] > dirty) {
			const int st = spec_state_[(size_t)then];
			if (st == 1) { tokens_.push_back('!'); continue; }        /* :688-690 */
			if (st == 2) continue;                                    /* :823 */
		}
		/* the split follows the columns that are scored here: everything at or right of `col` is in the right
		 * segment (physical index + gap), everything left of it in the left one, so each direction of the
		 * search reads its side with ONE fixed offset instead of a comparison per access */
		move_split(col - 1);
		refine_evaluate(col, gap, numseqs, S, R);
		if (R.nmov == 0) { tokens_.push_back('!'); continue; }        /* :688-690 */
		const int bestshift = R.bestshift;
		if (bestshift == 0) continue;                                 /* :823 */
		const int nmov = R.nmov;
		const int dir = (bestshift < 0) ? -1 : +1;
		const int sh = (bestshift < 0) ? -bestshift : bestshift;
		for (int j = 0; j < S.keep_maxaffected; ++j)                  /* :837-840 */
			for (int y = 0; y < kSym; ++y) SV(col + dir * j, y) = S.keep_best[(size_t)j * kSym + y];
		for (int t = 0; t < nmov; ++t) {                              /* :841-852 */
			const int sq = S.movers[t];
			S.runs[(size_t)sq].valid = false;
			for (int j = S.keep_affected[t] - 1; j >= 0; --j) {
				const int c = col + dir * j;
				CH(sq, c) = (j < sh) ? '-' : CH(sq, c - dir * sh);
			}
		}
		int right = 0, left = 0;                                      /* :853-864 */
		for (int j = col; j <= consensus_; ++j) { if (SV(j, kGap) != numseqs) break; ++right; }
		for (int j = col - 1; j >= 1; --j) { if (SV(j, kGap) != numseqs) break; ++left; }
		const int drop = right + left;
		if (drop > 0) {                                               /* :865-887 */
			for (auto &W : S.runs) W.valid = false;
			move_split(col - left - 1);                               /* the run becomes the head of the right segment */
			gap += drop;
			consensus_ -= drop;
			deleted += drop;
		}
		/* columns touched: col .. col + maxaffected - 1 (a slide to the right) or col - maxaffected + 1 .. col (to the
		 * left), and the deleted run from col on; with `deleted` after the drop this bounds their old numbers */
		dirty = std::max(dirty, col + deleted + std::max(dir > 0 ? S.keep_maxaffected : 1, right));
		col -= left + 1;                                              /* :888 */
	}
	if (gap > 0) {                                                    /* compact: the right segment moves down by `gap` */
		const size_t from = (size_t)split + 1 + gap, count = (size_t)consensus_ - split;
		memmove(&sv_[((size_t)split + 1) * kSym], &sv_[from * kSym], count * kSym * sizeof(int));
		sv_.resize((size_t)(consensus_ + 1) * kSym);
		for (int t = 0; t < numseqs; ++t) {
			std::string &r = str_[order_[t]];
			memmove(&r[(size_t)split], &r[from - 1], count);
			r.resize((size_t)consensus_);
		}
	}
	spec_state_.clear();
	spec_lo_.clear();
	if (trace) {
		dgc_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		if (step_ + (pending_ ? 1 : 0) == nseq_ && cells_ > 100000000)
			fprintf(stderr, "csadp task: %d sequences, %lld cells: DeleteGappedColumns (commit pass) %.2f ms in total\n", nseq_, cells_, dgc_ms_);
	}
}

/* the whole refinement in one call (callers with one task at a time).  CSADP_REFINE_SPECULATE=1 runs the speculation
 * too, on this thread, =2 on several: the tests use it to hold the speculated pass against the plain one (and run it
 * under the thread sanitizer). */
void Progressive::delete_gapped_columns(int numseqs, int maxnongaps)
{
	(void)maxnongaps;                                                 /* = numseqs / 2 (:1157) */
	refine_numseqs_ = numseqs;
	const int speculate = config().refine_speculate;              /* 1: on this thread, 2: chunks over the host pool */
	if (speculate) {
		const int chunks = refine_prepare();
		if (speculate >= 2) host_parallel_for(chunks, [this](int c) { refine_speculate(c); });
		else for (int c = 0; c < chunks; ++c) refine_speculate(c);
	}
	refine_commit();
}

int Progressive::finish(csadp_result *res)
{
	res->status = CSADP_OK;
	res->score = last_score_;
	res->consensus = empty_task_ ? 0 : consensus_;
	res->cells = cells_;
	res->fills = fills_;
	res->aligned = nullptr;
	res->progress = nullptr;
	if (empty_task_) return CSADP_OK;
	if (step_ < nseq_ || pending_) return CSADP_ERR_STATE;
	char **out = (char **)calloc((size_t)nseq_, sizeof(char *));
	if (!out) return CSADP_ERR_NOMEM;
	res->progress = strdup(tokens_.c_str());
	if (!res->progress) { free(out); return CSADP_ERR_NOMEM; }
	for (int s = 0; s < nseq_; ++s) {
		out[s] = (char *)malloc(str_[s].size() + 1);
		if (!out[s]) {
			for (int t = 0; t < s; ++t) free(out[t]);
			free(out);
			return CSADP_ERR_NOMEM;
		}
		memcpy(out[s], str_[s].data(), str_[s].size());
		out[s][str_[s].size()] = '\0';
	}
	res->aligned = out;
	return CSADP_OK;
}

}  // namespace csadp
