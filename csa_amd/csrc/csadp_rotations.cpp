/*
 * csadp_rotations.cpp -- the rotation finder: which rotation of every
 * circular sequence the DP should start from.
 *
 * Reference behaviour (csamsa.c:69-308 over the generalized cyclic suffix tree of
 * gencycsuffixtrees.c): collect the deepest tree nodes present in all sequences, drop those
 * that are suffixes of others and those occurring more than once in some sequence ("blocks"),
 * link blocks that follow each other in every sequence into chains, take the chain with the
 * largest summed block length; rotations[s] = position of its first block in sequence s.
 *
 * This file reaches the same blocks without a suffix tree.  With sequence 0 as the anchor,
 * M[p] = length of the longest string starting at cyclic position p of sequence 0 that occurs
 * in every sequence (minimum over sequences of the matching statistics, computed with one
 * suffix automaton per doubled sequence).  A block is unique in sequence 0, so it equals
 * seq0[p, p+M[p]) for its position p (right-maximal), it is not a suffix of a longer common
 * string ending at the same place iff M[p-1] <= M[p] (left-maximal), and it must occur exactly
 * once (cyclically) in every sequence (end-position counts of the automata).  Block order,
 * chain linking, chain weights and the final selection restate csamsa.c:132-226,:260-267 and
 * nodeslinkedlists.c:34-79 literally, including the order in which equal-depth blocks enter
 * the list (the suffix tree's child order = order of first insertion, i.e. of the first
 * occurrence in sequence 0).  oracle/rot_oracle.py states the same semantics by brute force.
 */
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include <algorithm>
#include <string>
#include <vector>

#include "csadp.h"
#include "csadp_hostpar.h"
#include "csadp_config.h"

using csadp::host_parallel_for;

namespace {

/* letters as the reference's followChar sees them (gencycsuffixtrees.c:321): ACGT or "other" */
inline int code_of(char c)
{
	switch (c) {
	case 'A': return 0;
	case 'C': return 1;
	case 'G': return 2;
	case 'T': return 3;
	default: return 4;
	}
}

/* suffix automaton of a doubled circular sequence */
struct Sam {
	struct State {
		int next[5];
		int link, len;
		int cnt;      /* end positions in [n, 2n): cyclic occurrences of the state's strings */
		int maxend;   /* largest end position */
	};
	std::vector<State> st;
	int last = 0;
	int n = 0;        /* length of the (single) sequence */

	int add_state(int len)
	{
		State s;
		for (int &x : s.next) x = -1;
		s.link = -1;
		s.len = len;
		s.cnt = 0;
		s.maxend = -1;
		st.push_back(s);
		return (int)st.size() - 1;
	}

	void build(const std::vector<unsigned char> &seq)
	{
		n = (int)seq.size();
		st.clear();
		st.reserve((size_t)4 * n + 4);
		last = add_state(0);
		for (int i = 0; i < 2 * n; ++i) {
			const int c = seq[(size_t)(i % n)];
			const int cur = add_state(st[(size_t)last].len + 1);
			st[(size_t)cur].cnt = (i >= n) ? 1 : 0;
			st[(size_t)cur].maxend = i;
			int p = last;
			while (p != -1 && st[(size_t)p].next[c] == -1) {
				st[(size_t)p].next[c] = cur;
				p = st[(size_t)p].link;
			}
			if (p == -1) {
				st[(size_t)cur].link = 0;
			} else {
				const int q = st[(size_t)p].next[c];
				if (st[(size_t)p].len + 1 == st[(size_t)q].len) {
					st[(size_t)cur].link = q;
				} else {
					const int clone = add_state(st[(size_t)p].len + 1);
					for (int x = 0; x < 5; ++x) st[(size_t)clone].next[x] = st[(size_t)q].next[x];
					st[(size_t)clone].link = st[(size_t)q].link;
					while (p != -1 && st[(size_t)p].next[c] == q) {
						st[(size_t)p].next[c] = clone;
						p = st[(size_t)p].link;
					}
					st[(size_t)q].link = clone;
					st[(size_t)cur].link = clone;
				}
			}
			last = cur;
		}
		/* propagate counts and largest end positions up the suffix links, longest first: a counting sort by length
		 * (lengths are at most 2n; a comparison sort of the 4n states took as long as building them) */
		std::vector<int> first((size_t)2 * n + 2, 0), order(st.size());
		for (const State &x : st) ++first[(size_t)x.len + 1];
		for (int l = 0; l <= 2 * n; ++l) first[(size_t)l + 1] += first[(size_t)l];
		for (size_t i = 0; i < st.size(); ++i) order[(size_t)first[(size_t)st[i].len]++] = (int)i;
		std::reverse(order.begin(), order.end());
		for (int v : order) {
			const int l = st[(size_t)v].link;
			if (l >= 0) {
				st[(size_t)l].cnt += st[(size_t)v].cnt;
				st[(size_t)l].maxend = std::max(st[(size_t)l].maxend, st[(size_t)v].maxend);
			}
		}
	}

	/* state of the suffix of length `len` of the string whose longest match sits in state v */
	int shrink(int v, int len) const
	{
		while (st[(size_t)v].link >= 0 && st[(size_t)st[(size_t)v].link].len >= len) v = st[(size_t)v].link;
		return v;
	}
};

struct Block {
	int depth = 0;
	int p0 = 0;                    /* position in sequence 0 */
	std::vector<int> pos;          /* position in every sequence */
	int size = 0, total = 0;
	int next = -1;                 /* nextblock (index into the list) */
	long long first_end = 0;       /* scratch for the child-order comparison */
};

}  // namespace

extern "C" {

int csadp_find_rotations(int nseq, const char *const *texts, const int *sizes, int *rotations, csadp_rotation_info *info)
{
	if (nseq < 2 || nseq > CSADP_MAX_SEQS || !texts || !sizes || !rotations) return CSADP_ERR_ARG;
	std::vector<std::vector<unsigned char>> seq((size_t)nseq);
	int minlen = 0x7fffffff;
	for (int s = 0; s < nseq; ++s) {
		if (!texts[s] || sizes[s] < 1) return CSADP_ERR_ARG;
		seq[(size_t)s].resize((size_t)sizes[s]);
		for (int i = 0; i < sizes[s]; ++i) seq[(size_t)s][(size_t)i] = (unsigned char)code_of(texts[s][i]);
		minlen = std::min(minlen, sizes[s]);
	}
	if (info) memset(info, 0, sizeof(*info));
	const int n0 = sizes[0];
	/* blocks as long as the shortest sequence are leaves of the reference's tree and follow other
	 * rules there; they do not occur on real data and are not considered */
	const int cap = minlen - 1;
	if (cap < 1) return CSADP_ERR_ARG;

	const bool trace = csadp::config().trace_host;
	auto tp = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!trace) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "  rotations: %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
		tp = now;
	};
	/* ---- matching statistics of sequence 0 against every sequence ------------------------- */
	std::vector<Sam> sam((size_t)nseq);
	std::vector<std::vector<int>> state_at((size_t)nseq);     /* automaton state after end position e */
	std::vector<int> M((size_t)n0, cap);
	std::vector<std::vector<int>> Ms((size_t)nseq);           /* per sequence, reduced below */
	const int qlen = 2 * n0;                                  /* doubled query */
	host_parallel_for(nseq, [&](int s) {
		sam[(size_t)s].build(seq[(size_t)s]);
		Ms[(size_t)s].assign((size_t)n0, 0);
		const Sam &A = sam[(size_t)s];
		std::vector<int> ms_end((size_t)qlen);
		state_at[(size_t)s].assign((size_t)qlen, 0);
		int v = 0, l = 0;
		for (int e = 0; e < qlen; ++e) {
			const int c = seq[0][(size_t)(e % n0)];
			while (v != 0 && A.st[(size_t)v].next[c] == -1) {
				v = A.st[(size_t)v].link;
				l = A.st[(size_t)v].len;
			}
			if (A.st[(size_t)v].next[c] != -1) {
				v = A.st[(size_t)v].next[c];
				++l;
			}
			ms_end[(size_t)e] = l;
			state_at[(size_t)s][(size_t)e] = v;
		}
		/* longest match STARTING at p: the matches [e - ms_end[e] + 1, e] have non-decreasing starts */
		int e = 0;
		for (int p = 0; p < n0; ++p) {
			if (e < p) e = p;
			while (e + 1 < qlen && (e + 1) - ms_end[(size_t)(e + 1)] + 1 <= p) ++e;
			int len = 0;
			if (e - ms_end[(size_t)e] + 1 <= p) len = e - p + 1;
			len = std::min(len, std::min(cap, sizes[s] - 1));
			Ms[(size_t)s][(size_t)p] = len;
		}
	});
	for (int s = 0; s < nseq; ++s)
		for (int p = 0; p < n0; ++p) M[(size_t)p] = std::min(M[(size_t)p], Ms[(size_t)s][(size_t)p]);

	lap("automata + matching statistics");
	/* ---- blocks ---------------------------------------------------------------------------- */
	/* every position is looked at on its own (suffix-link walks in nseq automata): slices of positions over the pool, joined in order */
	std::vector<Block> blocks;
	{
		const int T = std::max(1, std::min(32, n0 / 256));
		std::vector<std::vector<Block>> part((size_t)T);
		host_parallel_for(T, [&](int t) {
			std::vector<Block> &mine = part[(size_t)t];
			for (int p = (int)((long long)n0 * t / T), pe = (int)((long long)n0 * (t + 1) / T); p < pe; ++p) {
				const int d = M[(size_t)p];
				if (d < 1) continue;
				if (M[(size_t)((p + n0 - 1) % n0)] > d) continue;     /* suffix of the longer common string one to the left */
				Block b;
				b.depth = d;
				b.p0 = p;
				b.pos.assign((size_t)nseq, 0);
				bool unique = true;
				const int e = p + d - 1;                              /* end in the doubled query, < 2*n0 */
				for (int s = 0; s < nseq && unique; ++s) {
					const Sam &A = sam[(size_t)s];
					const int v = A.shrink(state_at[(size_t)s][(size_t)e], d);
					if (A.st[(size_t)v].cnt != 1) { unique = false; break; }
					b.pos[(size_t)s] = ((A.st[(size_t)v].maxend - d + 1) % sizes[s] + sizes[s]) % sizes[s];
				}
				if (unique) mine.push_back(std::move(b));
			}
		});
		for (std::vector<Block> &v : part)
			for (Block &b : v) blocks.push_back(std::move(b));
	}
	if (blocks.empty()) return CSADP_ERR_RANGE;               /* reference: "No unique subsequences found" */

	lap("blocks");
	/* ---- list order: decreasing depth; equal depths in reverse order of the tree's DFS -------- */
	const std::vector<unsigned char> &s0 = seq[0];
	auto letter = [&](const Block &b, int i) { return s0[(size_t)((b.p0 + i) % n0)]; };
	auto first_end_of_prefix = [&](const Block &b, int len) -> long long {
		/* end position of the first occurrence of b[0, len) in the doubled sequence 0 */
		for (int i = 0; i + len <= 2 * n0; ++i) {
			int k = 0;
			while (k < len && s0[(size_t)((i + k) % n0)] == letter(b, k)) ++k;
			if (k == len) return i + len;
		}
		return 2LL * n0 + 1;
	};
	auto dfs_before = [&](const Block &a, const Block &b) {
		int l = 0;
		const int lim = std::min(a.depth, b.depth);
		while (l < lim && letter(a, l) == letter(b, l)) ++l;
		if (l == lim) return a.depth < b.depth;
		return first_end_of_prefix(a, l + 1) < first_end_of_prefix(b, l + 1);
	};
	std::stable_sort(blocks.begin(), blocks.end(), [&](const Block &a, const Block &b) {
		if (a.depth != b.depth) return a.depth > b.depth;
		return dfs_before(b, a);                              /* later inserted first (insertSortedItem) */
	});
	const int nb = (int)blocks.size();

	lap("list order");
	/* ---- chain links, csamsa.c:143-178 --------------------------------------------------------- */
	std::vector<int> by_pos((size_t)nb);
	for (int k = 0; k < nseq; ++k) {
		for (int i = 0; i < nb; ++i) by_pos[(size_t)i] = i;
		std::sort(by_pos.begin(), by_pos.end(), [&](int a, int b) { return blocks[(size_t)a].pos[(size_t)k] < blocks[(size_t)b].pos[(size_t)k]; });
		long long limit = sizes[k];
		int prev = -1;
		for (int idx : by_pos) {
			Block &b = blocks[(size_t)idx];
			const int p = b.pos[(size_t)k];
			if ((long long)p + b.depth >= limit) continue;        /* the scan stops before this block ends */
			if (prev >= 0) {
				Block &pb = blocks[(size_t)prev];
				if (pb.size == 0) {
					if (pb.next < 0) pb.next = idx;
					else if (pb.next != idx) { pb.next = -1; pb.size = -1; }
				}
			} else {
				limit += p;                                       /* wrap-around part up to the first block */
			}
			prev = idx;
		}
	}
	/* ---- chain weights, csamsa.c:180-224 -------------------------------------------------------- */
	for (int i = 0; i < nb; ++i) {
		Block &b = blocks[(size_t)i];
		if (b.total == -1) continue;
		b.size = b.depth;
		int prev = i, cur = b.next, guard = 0;
		while (cur >= 0) {
			if (++guard > 4 * nb + 8) return CSADP_ERR_RANGE;     /* the reference does not terminate here */
			Block &c = blocks[(size_t)cur];
			const Block &pb = blocks[(size_t)prev];
			long long interval = 0x7fffffff;
			for (int k = 0; k < nseq; ++k) {
				long long cnt = 0;
				if (c.pos[(size_t)k] < pb.pos[(size_t)k]) cnt += sizes[k];
				cnt += c.pos[(size_t)k] - (pb.pos[(size_t)k] + pb.depth);
				interval = std::min(interval, cnt);
			}
			if (c.total > 0) {
				b.size += c.size;
				b.total += c.total;
				b.total += (int)interval;
				c.size = c.depth;
				c.total = -1;
				break;
			}
			c.size = c.depth;
			b.size += c.size;
			b.total += (int)interval;
			c.total = -1;
			prev = cur;
			cur = c.next;
		}
		b.total += b.size;
	}
	lap("chains");
	/* ---- the first strictly largest chain wins (sortList, nodeslinkedlists.c:55-79) ------------- */
	int best = 0;
	for (int i = 1; i < nb; ++i)
		if (blocks[(size_t)i].size > blocks[(size_t)best].size) best = i;
	for (int s = 0; s < nseq; ++s) rotations[s] = blocks[(size_t)best].pos[(size_t)s];
	if (info) {
		info->blocks = nb;
		info->chain_size = blocks[(size_t)best].size;
		info->chain_span = blocks[(size_t)best].total;
		info->first_block_depth = blocks[(size_t)best].depth;
	}
	return CSADP_OK;
}

}  // extern "C"
