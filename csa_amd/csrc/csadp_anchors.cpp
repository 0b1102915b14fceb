/*
 * csadp_anchors.cpp -- the anchor stage that feeds the DP: which stretches of the rotated
 * sequences are fixed as common anchors and which gaps between them go to ProgressiveDP.
 * Host code (the reference spends ~1 s of a 5 s run here; the gaps it emits are what the
 * device batches).
 *
 * Replaces, result for result, alignment.c:69-86 (PrepareTreeForAlignment) and :163-214
 * (RunAlignment) with their helpers in morenodeslinkedlists.c and alignmentmap.c.  No suffix
 * tree is built.  After MarkUsedNodes/DeleteUnusedNodes (morenodeslinkedlists.c:561-641) the
 * reference's tree holds exactly the suffixes of the rotated LINEAR sequences, and
 * CollectBorderNodes (:302-330) credits every suffix start p of sequence s to the deepest node
 * on its path that belongs to all sequences.  In string terms: to w = the longest prefix of
 * T_s[p..] that occurs in every sequence.  Those lengths are matching statistics, computed
 * here with one suffix automaton per (reversed) sequence; equal strings are recognised by
 * their state in sequence 0's automaton.
 *
 * The anchor loop itself (UpdateActiveBorderNodes, SortBorderNodes, the heaviest increasing
 * chain, SetAlignmentMapSegments) is order dependent down to which list neighbour a node is
 * parked in while hidden, so it is restated operation by operation on index-linked arrays.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "csadp.h"
#include "csadp_hostpar.h"
#include "csadp_config.h"

using csadp::host_parallel_for;

namespace {

inline int code_of(char ch)
{
	switch (ch) {
	case 'A': return 0;
	case 'C': return 1;
	case 'G': return 2;
	case 'T': return 3;
	default: return 4;      /* every other letter is one symbol for the tree (gencycsuffixtrees.c:320) */
	}
}

/* suffix automaton of one linear string */
struct Automaton {
	struct State {
		int next[5];
		int link, len;
	};
	std::vector<State> st;

	int add(int len)
	{
		State s;
		for (int &x : s.next) x = -1;
		s.link = -1;
		s.len = len;
		st.push_back(s);
		return (int)st.size() - 1;
	}

	void build(const unsigned char *seq, int n)
	{
		st.clear();
		st.reserve((size_t)2 * n + 2);
		int last = add(0);
		for (int i = 0; i < n; ++i) {
			const int c = seq[i];
			const int cur = add(st[(size_t)last].len + 1);
			int p = last;
			for (; p != -1 && st[(size_t)p].next[c] == -1; p = st[(size_t)p].link) st[(size_t)p].next[c] = cur;
			if (p == -1) {
				st[(size_t)cur].link = 0;
			} else {
				const int q = st[(size_t)p].next[c];
				if (st[(size_t)p].len + 1 == st[(size_t)q].len) {
					st[(size_t)cur].link = q;
				} else {
					const int clone = add(st[(size_t)p].len + 1);
					memcpy(st[(size_t)clone].next, st[(size_t)q].next, sizeof(st[0].next));
					st[(size_t)clone].link = st[(size_t)q].link;
					for (; p != -1 && st[(size_t)p].next[c] == q; p = st[(size_t)p].link) st[(size_t)p].next[c] = clone;
					st[(size_t)q].link = clone;
					st[(size_t)cur].link = clone;
				}
			}
			last = cur;
		}
	}
};

/* ---- border nodes ------------------------------------------------------------------------ */

struct Border {
	int nseq = 0;
	std::vector<int> size;        /* per node */
	std::vector<int> head, tail;  /* per node*nseq: live positions are pool[head..tail) */
	std::vector<int> pool;
};

/* rev[s] = reversed rotated sequence s in symbol codes */
int collect_border_nodes(const std::vector<std::vector<unsigned char>> &rev, Border *out)
{
	const int N = (int)rev.size();
	const bool trace = csadp::config().trace_host;
	auto tp = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (!trace) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "  border nodes: %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
		tp = now;
	};
	std::vector<Automaton> sam((size_t)N);
	host_parallel_for(N, [&](int t) { sam[(size_t)t].build(rev[(size_t)t].data(), (int)rev[(size_t)t].size()); });
	lap("automata");

	/* common[s][p] = length of the longest prefix of T_s[p..] found in every sequence;
	 * ident[s][p]  = state of that prefix (reversed) in sequence 0's automaton.
	 * One item per (text s, automaton t): N x N independent walks (round 2 ran one item per text: 12-19 items on a host
	 * with 256 hardware threads), then one item per text takes the minimum over the automata. */
	std::vector<std::vector<int>> common((size_t)N), ident((size_t)N);
	std::vector<std::vector<int>> st0((size_t)N);
	/* Every item walks its text through its automaton and min-reduces what it finds into common[s], a chunk of positions at a
	 * time under the text's lock: the transient memory stays O(N n) (round 3 kept every walk whole until a second pass reduced
	 * them: N x N vectors of text length, 280 MB for 64 genomes of 17 kbp, over 3 GB for 200 kbp inputs). */
	std::vector<std::mutex> guard((size_t)N);
	host_parallel_for(N, [&](int s) {
		const int n = (int)rev[(size_t)s].size();
		std::vector<int> &L = common[(size_t)s];
		L.resize((size_t)n);
		for (int p = 0; p < n; ++p) L[(size_t)p] = n - p;         /* own bound: the suffix itself */
		st0[(size_t)s].assign((size_t)n, 0);
	});
	host_parallel_for(N * N, [&](int item) {
		const int s = item / N, t = item % N;
		if (t == s && t != 0) return;                              /* the text's own bound is the suffix itself */
		const std::vector<unsigned char> &R = rev[(size_t)s];
		const int n = (int)R.size();
		constexpr int kChunk = 4096;
		int part[kChunk];                                          /* matching statistics of positions p0 .. p0 + len - 1, descending q */
		std::vector<int> &L = common[(size_t)s];
		int *S0 = t == 0 ? st0[(size_t)s].data() : nullptr;        /* the states in automaton 0: one writer per text */
		const Automaton &A = sam[(size_t)t];
		int v = 0, l = 0;
		for (int q0 = 0; q0 < n; q0 += kChunk) {
			const int len = std::min(kChunk, n - q0);
			for (int i = 0; i < len; ++i) {
				const int q = q0 + i;
				const int c = R[(size_t)q];
				while (v != 0 && A.st[(size_t)v].next[c] == -1) {
					v = A.st[(size_t)v].link;
					l = A.st[(size_t)v].len;
				}
				if (A.st[(size_t)v].next[c] != -1) {
					v = A.st[(size_t)v].next[c];
					++l;
				}
				part[i] = l;
				if (S0) S0[(size_t)(n - 1 - q)] = v;
			}
			std::lock_guard<std::mutex> hold(guard[(size_t)s]);
			for (int i = 0; i < len; ++i) {
				int &dst = L[(size_t)(n - 1 - (q0 + i))];
				if (part[i] < dst) dst = part[i];
			}
		}
	});
	host_parallel_for(N, [&](int s) {
		const int n = (int)rev[(size_t)s].size();
		const std::vector<int> &L = common[(size_t)s];
		std::vector<int> &id = ident[(size_t)s];
		id.assign((size_t)n, 0);
		const Automaton &A0 = sam[0];
		for (int p = 0; p < n; ++p) {
			int v = st0[(size_t)s][(size_t)p];
			const int len = L[(size_t)p];
			if (len == 0) continue;
			while (A0.st[(size_t)A0.st[(size_t)v].link].len >= len) v = A0.st[(size_t)v].link;
			id[(size_t)p] = v;
		}
	});
	lap("matching statistics");
	struct Credit {
		long long key;
		int seq, pos;
	};
	std::vector<Credit> credits;
	{
		/* credits in (seq, pos) order: every text counts its own, then writes them at its offset */
		std::vector<size_t> first((size_t)N + 1, 0);
		host_parallel_for(N, [&](int s) {
			size_t c = 0;
			for (int len : common[(size_t)s]) c += len != 0;      /* len 0: credited to the root, the list's sentinel (alignment.c:47) */
			first[(size_t)s + 1] = c;
		});
		for (int s = 0; s < N; ++s) first[(size_t)s + 1] += first[(size_t)s];
		credits.resize(first[(size_t)N]);
		const long long span0 = (long long)rev[0].size() + 2;
		host_parallel_for(N, [&](int s) {
			size_t at = first[(size_t)s];
			const int n = (int)rev[(size_t)s].size();
			for (int p = 0; p < n; ++p) {
				const int len = common[(size_t)s][(size_t)p];
				if (len == 0) continue;
				credits[at++] = {(long long)ident[(size_t)s][(size_t)p] * span0 + len, s, p};
			}
		});
	}
	const long long span = (long long)rev[0].size() + 2;
	/* order by (key, seq, pos): the credits were generated in (seq, pos) order, so a STABLE sort on
	 * the key alone does it -- LSD radix, 11 bits a pass, every pass split over the pool: each thread counts its
	 * slice, the slices' counts are turned into write offsets digit by digit (slice order inside a digit keeps the
	 * sort stable), each thread scatters its slice */
	{
		long long maxkey = 0;
		for (const Credit &c : credits) maxkey = std::max(maxkey, c.key);
		std::vector<Credit> tmp(credits.size());
		const int T = (int)std::max<size_t>(1, std::min<size_t>(16, credits.size() / 8192));
		std::vector<size_t> count((size_t)T * 2048);
		const size_t n = credits.size();
		for (int shift = 0; (maxkey >> shift) != 0; shift += 11) {
			std::fill(count.begin(), count.end(), 0);
			host_parallel_for(T, [&](int t) {
				size_t *cnt = &count[(size_t)t * 2048];
				for (size_t i = n * t / T, e = n * (t + 1) / T; i < e; ++i) ++cnt[(credits[i].key >> shift) & 2047];
			}, T);
			size_t run = 0;
			for (int d = 0; d < 2048; ++d)
				for (int t = 0; t < T; ++t) {
					const size_t c = count[(size_t)t * 2048 + d];
					count[(size_t)t * 2048 + d] = run;
					run += c;
				}
			host_parallel_for(T, [&](int t) {
				size_t *at = &count[(size_t)t * 2048];
				for (size_t i = n * t / T, e = n * (t + 1) / T; i < e; ++i) tmp[at[(credits[i].key >> shift) & 2047]++] = credits[i];
			}, T);
			credits.swap(tmp);
		}
	}

	lap("credits + radix sort");
	struct Group {
		size_t from, to;
		int first0;
	};
	std::vector<Group> groups;
	{
		/* runs of equal keys, slices of the sorted credits over the pool: a slice starts at the first run that begins inside it */
		const size_t nc = credits.size();
		const int T = (int)std::max<size_t>(1, std::min<size_t>(32, nc / 4096));
		std::vector<std::vector<Group>> part((size_t)T);
		host_parallel_for(T, [&](int t) {
			size_t i = nc * (size_t)t / (size_t)T;
			const size_t stop = nc * (size_t)(t + 1) / (size_t)T;
			while (i > 0 && i < nc && credits[i].key == credits[i - 1].key) ++i;      /* the run that straddles the slice's start is the left slice's */
			while (i < stop) {
				size_t j = i;
				int seen = 0, prev = -1;
				while (j < nc && credits[j].key == credits[i].key) {
					if (credits[j].seq != prev) {
						++seen;
						prev = credits[j].seq;
					}
					++j;
				}
				if (seen == N) part[(size_t)t].push_back({i, j, credits[i].pos});     /* morenodeslinkedlists.c:325-328 */
				i = j;
			}
		}, T);
		for (const std::vector<Group> &v : part) groups.insert(groups.end(), v.begin(), v.end());
	}
	std::sort(groups.begin(), groups.end(), [](const Group &a, const Group &b) { return a.first0 < b.first0; });

	Border &B = *out;
	B.nseq = N;
	const size_t nodes = groups.size() + 1;
	B.size.assign(nodes, 0);
	B.head.assign(nodes * N, 0);
	B.tail.assign(nodes * N, 0);
	B.pool.clear();
	{
		size_t total = (size_t)N;
		for (const Group &g : groups) total += g.to - g.from;
		B.pool.reserve(total);
	}
	for (int s = 0; s < N; ++s) {                   /* node 0: the sentinel, position -1 everywhere */
		B.head[(size_t)s] = (int)B.pool.size();
		B.pool.push_back(-1);
		B.tail[(size_t)s] = (int)B.pool.size();
	}
	for (size_t g = 0; g < groups.size(); ++g) {
		const size_t node = g + 1;
		B.size[node] = (int)(credits[groups[g].from].key % span);
		size_t i = groups[g].from;
		for (int s = 0; s < N; ++s) {
			B.head[node * N + s] = (int)B.pool.size();
			for (; i < groups[g].to && credits[i].seq == s; ++i) B.pool.push_back(credits[i].pos);
			B.tail[node * N + s] = (int)B.pool.size();
		}
	}
	lap("groups + nodes");
	return CSADP_OK;
}

/* ---- the anchor loop ---------------------------------------------------------------------- */

const int NIL = -1;

struct Loop {
	int N;
	const int *sizes;
	Border B;
	std::vector<int> nxt, prv, parked, act;
	std::vector<char> hidden;
	std::vector<int> start, end;

	/* alignment map */
	struct Seg {
		int size, mingap, maxgap, dp, next;
	};
	std::vector<Seg> seg;
	std::vector<int> segpos;            /* seg index * N + s */

	/* chain of the current gap */
	struct Item {
		int size, weight, back, next, prev;
	};
	std::vector<Item> item;
	std::vector<int> itempos;
	int chain = NIL;

	int first_pos(int node, int s) const { return B.pool[(size_t)B.head[(size_t)node * N + s]]; }
	int k0(int node) const { return first_pos(node, 0); }
	bool empty(int node, int s) const { return B.head[(size_t)node * N + s] == B.tail[(size_t)node * N + s]; }

	void init()
	{
		const size_t nodes = B.size.size();
		nxt.assign(nodes, NIL);
		prv.assign(nodes, NIL);
		parked.assign(nodes, NIL);
		hidden.assign(nodes, 0);
		act.assign(nodes * N, 0);
		for (size_t i = 0; i < nodes; ++i) {
			nxt[i] = (i + 1 < nodes) ? (int)(i + 1) : NIL;
			prv[i] = (i > 0) ? (int)(i - 1) : NIL;
		}
		start.assign((size_t)N, 0);
		end.assign((size_t)N, 0);
		/* alignment.c:57-65: a fake first segment of size 1 at -1 and a fake last one at the ends */
		seg.push_back({1, 0, 0, 0, 1});
		seg.push_back({0, std::numeric_limits<int>::max(), std::numeric_limits<int>::max(), 0, NIL});
		segpos.assign((size_t)2 * N, -1);
		for (int s = 0; s < N; ++s) segpos[(size_t)N + s] = sizes[s];
		gap_sizes(0);
	}

	/* alignmentmap.c:239-256 */
	void gap_sizes(int g)
	{
		int lo = std::numeric_limits<int>::max(), hi = std::numeric_limits<int>::min();
		const int r = seg[(size_t)g].next;
		for (int s = 0; s < N; ++s) {
			int d = segpos[(size_t)r * N + s] - (segpos[(size_t)g * N + s] + seg[(size_t)g].size);
			if (d < 0) d += sizes[s];
			lo = std::min(lo, d);
			hi = std::max(hi, d);
		}
		seg[(size_t)g].mingap = lo;
		seg[(size_t)g].maxgap = hi;
	}

	/* morenodeslinkedlists.c:62-68 */
	void unlink(int node)
	{
		if (prv[(size_t)node] != NIL) nxt[(size_t)prv[(size_t)node]] = nxt[(size_t)node];
		if (nxt[(size_t)node] != NIL) prv[(size_t)nxt[(size_t)node]] = prv[(size_t)node];
	}

	/* morenodeslinkedlists.c:106-128: the node leaves the list and is stacked inside its left
	 * neighbour */
	void park(int node)
	{
		if (hidden[(size_t)node]) return;
		const int keeper = prv[(size_t)node];
		nxt[(size_t)keeper] = nxt[(size_t)node];
		if (nxt[(size_t)node] != NIL) prv[(size_t)nxt[(size_t)node]] = keeper;
		nxt[(size_t)node] = NIL;
		prv[(size_t)node] = parked[(size_t)keeper];
		if (parked[(size_t)keeper] != NIL) nxt[(size_t)parked[(size_t)keeper]] = node;
		parked[(size_t)keeper] = node;
		hidden[(size_t)node] = 1;
	}

	/* morenodeslinkedlists.c:131-146: everything stacked inside the node returns right after it */
	void unpark(int keeper)
	{
		int h = parked[(size_t)keeper];
		if (h == NIL) return;
		hidden[(size_t)h] = 0;
		nxt[(size_t)h] = nxt[(size_t)keeper];
		if (nxt[(size_t)keeper] != NIL) prv[(size_t)nxt[(size_t)keeper]] = h;
		while (prv[(size_t)h] != NIL) {
			h = prv[(size_t)h];
			hidden[(size_t)h] = 0;
		}
		prv[(size_t)h] = keeper;
		nxt[(size_t)keeper] = h;
		parked[(size_t)keeper] = NIL;
	}

	/* morenodeslinkedlists.c:398-443 */
	void sort_front()
	{
		const int e0 = end[0];
		int c = nxt[0];
		while (c != NIL && k0(c) < e0) {
			const int left = prv[(size_t)c];
			int after;
			if (k0(c) < k0(left)) {
				int back = left;
				while (back != NIL && k0(back) > k0(c)) back = prv[(size_t)back];
				const int following = nxt[(size_t)back];
				nxt[(size_t)back] = c;
				prv[(size_t)c] = back;
				int run = c;                                   /* an ascending run moves as a whole */
				while (nxt[(size_t)run] != NIL && k0(nxt[(size_t)run]) > k0(run) && k0(nxt[(size_t)run]) < k0(following))
					run = nxt[(size_t)run];
				after = nxt[(size_t)run];
				nxt[(size_t)run] = following;
				prv[(size_t)following] = run;
				nxt[(size_t)left] = after;
				if (after != NIL) prv[(size_t)after] = left;
			} else {
				after = nxt[(size_t)c];
			}
			c = after;
		}
	}

	/* morenodeslinkedlists.c:446-462 */
	void move_right(int node)
	{
		if (nxt[(size_t)node] == NIL || k0(nxt[(size_t)node]) > k0(node)) return;
		int cur = nxt[(size_t)node];
		while (nxt[(size_t)cur] != NIL && k0(nxt[(size_t)cur]) < k0(node)) cur = nxt[(size_t)cur];
		const int l = prv[(size_t)node], r = nxt[(size_t)node];
		if (l != NIL) nxt[(size_t)l] = r;
		if (r != NIL) prv[(size_t)r] = l;
		const int after = nxt[(size_t)cur];
		nxt[(size_t)cur] = node;
		prv[(size_t)node] = cur;
		if (after != NIL) prv[(size_t)after] = node;
		nxt[(size_t)node] = after;
	}

	/* morenodeslinkedlists.c:465-534.  Positions dropped by the chaining step stay dropped: the
	 * reference's call that should bring them back (:481) returns at once because its guard
	 * (:175) tests the parked-node pointer that :480 has just cleared. */
	int refresh_active()
	{
		const int e0 = end[0];
		int b = nxt[0];
		while (b != NIL && k0(b) < e0) {
			if (parked[(size_t)b] != NIL) unpark(b);
			const int after = nxt[(size_t)b];
			for (int s = 0; s < N; ++s) {
				int &h = B.head[(size_t)b * N + s];
				const int t = B.tail[(size_t)b * N + s];
				while (h < t && B.pool[(size_t)h] < start[(size_t)s]) {
					++h;
					--act[(size_t)b * N + s];
				}
				if (h == t) {
					unlink(b);
					break;
				}
			}
			b = after;
		}
		sort_front();
		int active = 0;
		b = nxt[0];
		while (b != NIL && k0(b) < e0) {
			++active;
			int s = 0;
			for (; s < N; ++s) {
				int count = 0;
				for (int i = B.head[(size_t)b * N + s]; i < B.tail[(size_t)b * N + s] && B.pool[(size_t)i] < end[(size_t)s]; ++i) ++count;
				if (count == 0) break;
				act[(size_t)b * N + s] = count;
			}
			const int after = nxt[(size_t)b];
			if (s != N) {                              /* no position of some sequence in this gap (yet) */
				park(b);
				--active;
				b = after;
				continue;
			}
			for (s = 1; s < N; ++s)
				if (act[(size_t)b * N + s] != act[(size_t)b * N]) {  /* occurrence counts differ */
					park(b);
					--active;
					break;
				}
			b = after;
		}
		return active;
	}

	/* alignmentmap.c:9-31 */
	int new_item(int node)
	{
		const int id = (int)item.size();
		int size = B.size[(size_t)node];
		for (int s = 0; s < N; ++s) {
			const int pos = first_pos(node, s);
			itempos.push_back(pos);
			if (pos + B.size[(size_t)node] >= end[(size_t)s]) size = std::min(size, end[(size_t)s] - pos);
		}
		item.push_back({size, size, NIL, NIL, NIL});
		return id;
	}

	/* alignmentmap.c:59-67 */
	bool to_the_right(int a, int b) const
	{
		for (int s = 0; s < N; ++s)
			if (itempos[(size_t)a * N + s] < itempos[(size_t)b * N + s] + item[(size_t)b].size) return false;
		return true;
	}

	/* alignmentmap.c:70-105: the chain list is kept in decreasing weight */
	void heaviest_chain()
	{
		item.clear();
		itempos.clear();
		chain = NIL;
		int b = nxt[0];
		while (b != NIL && k0(b) < end[0]) {
			const int fresh = new_item(b);
			int cur = NIL, probe = chain;
			while (probe != NIL && !to_the_right(fresh, probe)) {
				cur = probe;
				probe = item[(size_t)cur].next;
			}
			if (probe != NIL) {
				item[(size_t)fresh].weight += item[(size_t)probe].weight;
				item[(size_t)fresh].back = probe;
			}
			int before = cur;
			cur = probe;
			while (before != NIL && item[(size_t)fresh].weight >= item[(size_t)before].weight) {
				cur = before;
				before = item[(size_t)cur].prev;
			}
			if (before == NIL) chain = fresh;
			else item[(size_t)before].next = fresh;
			item[(size_t)fresh].prev = before;
			if (cur != NIL) item[(size_t)cur].prev = fresh;
			item[(size_t)fresh].next = cur;

			int after = nxt[(size_t)b];
			if (act[(size_t)b * N] > 1) {             /* another occurrence in this gap: drop the used one, re-place */
				for (int s = 0; s < N; ++s) {
					++B.head[(size_t)b * N + s];
					--act[(size_t)b * N + s];
				}
				move_right(b);
				if (nxt[(size_t)b] == after) after = b;
			}
			b = after;
		}
	}

	static int half_open_div(long long a, long long b) { return (int)(a / b); }   /* C division, truncating */

	/* alignmentmap.c:259-316 */
	int fix_segments(int startseg, int endseg)
	{
		int right = endseg, count = 0;
		for (int it = chain; it != NIL; it = item[(size_t)it].back) {
			const int g = (int)seg.size();
			seg.push_back({item[(size_t)it].size, 0, 0, 0, right});
			for (int s = 0; s < N; ++s) segpos.push_back(itempos[(size_t)it * N + s]);
			gap_sizes(g);
			long long sum = 0;
			for (int s = 0; s < N; ++s) {
				int d = segpos[(size_t)right * N + s] - (segpos[(size_t)g * N + s] + seg[(size_t)g].size);
				if (d < 0) d += sizes[s];
				sum += d;
			}
			const int lo = seg[(size_t)g].mingap, hi = seg[(size_t)g].maxgap;
			const int avg_wo_min = half_open_div((int)(sum - lo), N - 1);
			const int avg_wo_max = half_open_div((int)(sum - hi), N - 1);
			if (lo < avg_wo_min / 2 || hi > (avg_wo_max * 3) / 2) {
				seg.pop_back();                    /* unbalanced gap to the right: not an anchor */
				segpos.resize(segpos.size() - (size_t)N);
			} else {
				right = g;
				++count;
			}
		}
		seg[(size_t)startseg].next = right;
		gap_sizes(startseg);
		chain = NIL;
		return count;
	}

	/* alignment.c:163-214 */
	void run()
	{
		int startseg = 0;
		const int lastseg = 1;
		while (startseg != lastseg) {
			const int endseg = seg[(size_t)startseg].next;
			if (seg[(size_t)startseg].mingap == 0) {
				startseg = endseg;
				continue;
			}
			for (int s = 0; s < N; ++s) {
				start[(size_t)s] = segpos[(size_t)startseg * N + s] + seg[(size_t)startseg].size;
				end[(size_t)s] = segpos[(size_t)endseg * N + s];
			}
			int count = refresh_active();
			if (count > 0) {
				heaviest_chain();
				count = fix_segments(startseg, endseg);
			}
			if (count == 0) {
				seg[(size_t)startseg].dp = 1;
				startseg = seg[(size_t)startseg].next;
			}
		}
	}
};

}  // namespace

extern "C" {

int csadp_build_anchor_map(int nseq, const char *const *texts, const int *sizes, const int *rotations, csadp_anchor_map *out)
{
	if (nseq < 2 || nseq > CSADP_MAX_SEQS || !texts || !sizes || !rotations || !out) return CSADP_ERR_ARG;
	memset(out, 0, sizeof(*out));
	std::vector<std::vector<unsigned char>> fwd((size_t)nseq), rev((size_t)nseq);
	for (int s = 0; s < nseq; ++s) {
		const int n = sizes[s];
		if (!texts[s] || n < 1 || rotations[s] < 0 || rotations[s] >= n) return CSADP_ERR_ARG;
		fwd[(size_t)s].resize((size_t)n);
		rev[(size_t)s].resize((size_t)n);
		for (int p = 0; p < n; ++p) {
			int i = rotations[s] + p;
			if (i >= n) i -= n;
			const unsigned char c = (unsigned char)code_of(texts[s][i]);
			fwd[(size_t)s][(size_t)p] = c;
			rev[(size_t)s][(size_t)(n - 1 - p)] = c;
		}
	}
	/* A proper suffix of one rotated sequence that is a whole rotation of another one sends the
	 * reference's suffix walk (morenodeslinkedlists.c:590-617) into the other sequence's rotation
	 * leaves: it records positions past the end of the text, depends on the processing order and
	 * need not end.  No defined result to reproduce. */
	{
		std::vector<char> collides((size_t)nseq, 0);
		/* a rotation of sequence i has i's letter counts: the suffix of j is only searched for when its counts are the same (prefix counts
		 * of every sequence, one pass each) -- on real data never, and the search itself (a 16 k needle in a 33 k haystack per pair of
		 * sequences) was 1-2 ms of the stage */
		std::vector<std::vector<int>> upto((size_t)nseq);            /* upto[s][5 p + c] = letters c among the first p of sequence s */
		host_parallel_for(nseq, [&](int s) {
			const std::vector<unsigned char> &F = fwd[(size_t)s];
			std::vector<int> &U = upto[(size_t)s];
			U.assign(5 * (F.size() + 1), 0);
			for (size_t p = 0; p < F.size(); ++p) {
				for (int c = 0; c < 5; ++c) U[5 * (p + 1) + c] = U[5 * p + c];
				++U[5 * (p + 1) + F[p]];
			}
		});
		host_parallel_for(nseq, [&](int i) {
			const size_t ni = fwd[(size_t)i].size();
			std::vector<unsigned char> twice;
			for (int j = 0; j < nseq; ++j) {
				const size_t nj = fwd[(size_t)j].size();
				if (i == j || ni >= nj) continue;
				bool same = true;
				for (int c = 0; c < 5; ++c)
					same = same && upto[(size_t)j][5 * nj + c] - upto[(size_t)j][5 * (nj - ni) + c] == upto[(size_t)i][5 * ni + c];
				if (!same) continue;
				if (twice.empty()) {
					twice = fwd[(size_t)i];
					twice.insert(twice.end(), fwd[(size_t)i].begin(), fwd[(size_t)i].end());
				}
				if (memmem(twice.data(), twice.size(), fwd[(size_t)j].data() + (nj - ni), ni) != NULL) collides[(size_t)i] = 1;
			}
		});
		for (int i = 0; i < nseq; ++i)
			if (collides[(size_t)i]) return CSADP_ERR_RANGE;
	}

	const bool trace = csadp::config().trace_host;
	auto t0 = std::chrono::steady_clock::now();
	Loop loop;
	loop.N = nseq;
	loop.sizes = sizes;
	const int rc = collect_border_nodes(rev, &loop.B);
	if (rc != CSADP_OK) return rc;
	out->border_nodes = (int)loop.B.size.size() - 1;
	auto t1 = std::chrono::steady_clock::now();
	loop.init();
	loop.run();
	if (trace)
		fprintf(stderr, "csadp_build_anchor_map: border nodes %.1f ms, anchor loop %.1f ms\n",
		        std::chrono::duration<double, std::milli>(t1 - t0).count(),
		        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());

	int count = 0;
	for (int g = 0; g != NIL; g = loop.seg[(size_t)g].next) ++count;
	out->nseq = nseq;
	out->nsegs = count;
	out->size = (int *)malloc(sizeof(int) * (size_t)count);
	out->dp = (int *)malloc(sizeof(int) * (size_t)count);
	out->positions = (int *)malloc(sizeof(int) * (size_t)count * (size_t)nseq);
	if (!out->size || !out->dp || !out->positions) {
		csadp_free_anchor_map(out);
		return CSADP_ERR_NOMEM;
	}
	int k = 0;
	for (int g = 0; g != NIL; g = loop.seg[(size_t)g].next, ++k) {
		out->size[k] = loop.seg[(size_t)g].size;
		out->dp[k] = loop.seg[(size_t)g].dp;
		for (int s = 0; s < nseq; ++s) out->positions[(size_t)k * nseq + s] = loop.segpos[(size_t)g * nseq + s];
	}
	return CSADP_OK;
}

void csadp_free_anchor_map(csadp_anchor_map *m)
{
	if (!m) return;
	free(m->size);
	free(m->dp);
	free(m->positions);
	memset(m, 0, sizeof(*m));
}

}  // extern "C"
