/*
 * host_sanitize.cpp -- TEST INFRASTRUCTURE: drives the product's HOST code (no GPU) under
 * -fsanitize=address,undefined and, built a second time, under -fsanitize=thread (SURVEY.md 5:
 * "host code under sanitizers in CPU tests").  Covered: the FASTA loader and the -Rotated.fasta
 * writer/reader, the rotation finder (suffix automata, threaded), the anchor map (index-linked list
 * surgery, threaded), the whole progressive host logic of ProgressiveDP through the test seam
 * (csadp_debug_align_with_filler: ordering, stale-border rule, traceback application, the gap-buffer
 * DeleteGappedColumns) with fills supplied by the oracle, the round driver of csadp_align_batch (round groups on their own host
 * threads: csadp_debug_align_batch_with_filler), the LPT partitioner, and the persistent host thread pool hammered from several
 * caller threads.  Every result is also checked against the oracle.
 *
 *   host_sanitize <tests/golden/data/Primates.txt> <tmpdir>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "csadp.h"
#include "csadp_debug.h"
#include "../../oracle/csa_dp_oracle.h"

static int failures = 0;
#define CHECK(cond)                                                                  \
	do {                                                                             \
		if (!(cond)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
	} while (0)

/* one fill + walk in the reference's terms, from the oracle (csadp_debug.h) */
static int oracle_fill(void *, int nrows, int ncols, int nprev, const int *sv, const signed char *rowcodes, const int *top, int left_i,
                       unsigned char *ops, int *nops, int *remj, int *remk, int *score)
{
	std::vector<int> H((size_t)(nrows + 1) * (ncols + 1));
	std::vector<char> D((size_t)(nrows + 1) * (ncols + 1));
	if (odp_fill(nrows, ncols, rowcodes, sv, nprev, top, left_i, H.data(), D.data()) != ODP_OK) return CSADP_ERR_HIP;
	const size_t pitch = (size_t)ncols + 1;
	int j = nrows, k = ncols, n = 0;
	while (j > 0 && k > 0) {
		const char d = D[(size_t)j * pitch + k];
		if (d == 'D') { ops[n] = 2; --j; --k; }
		else if (d == 'L') { ops[n] = 1; --k; }
		else { ops[n] = 0; --j; }
		++n;
	}
	*nops = n; *remj = j; *remk = k;
	*score = H[(size_t)nrows * pitch + ncols];
	return CSADP_OK;
}

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd()
{
	rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
	return (unsigned)(rng_state >> 32);
}

static std::vector<std::string> family(int n, int len, double mut, double indel)
{
	std::string root;
	for (int i = 0; i < len; ++i) root.push_back("ACGT"[rnd() & 3]);
	std::vector<std::string> out;
	for (int s = 0; s < n; ++s) {
		std::string t;
		for (char c : root) {
			const double u = (rnd() & 0xffff) / 65536.0;
			if (u < indel) continue;
			if (u < 2 * indel) t.push_back("ACGT"[rnd() & 3]);
			t.push_back(((rnd() & 0xffff) / 65536.0 < mut) ? "ACGT"[rnd() & 3] : c);
		}
		if (t.empty()) t = "A";
		out.push_back(t);
	}
	return out;
}

static void check_task(const std::vector<std::string> &fam, const std::vector<int> &rot, const std::vector<int> &st, const std::vector<int> &en)
{
	const int n = (int)fam.size();
	std::vector<const char *> txt(n);
	std::vector<int> sz(n);
	for (int s = 0; s < n; ++s) { txt[s] = fam[s].c_str(); sz[s] = (int)fam[s].size(); }
	csadp_task task = {n, txt.data(), sz.data(), rot.data(), st.data(), en.data()};
	csadp_result res;
	const int rc = csadp_debug_align_with_filler(&task, oracle_fill, NULL, &res);
	std::vector<char *> want(n, nullptr);
	odp_stats os;
	const int cons = odp_progressive_dp(n, txt.data(), sz.data(), rot.data(), st.data(), en.data(), want.data(), &os);
	CHECK(rc == CSADP_OK && cons >= 0);
	if (rc == CSADP_OK && cons >= 0) {
		CHECK(res.consensus == cons);
		for (int s = 0; s < n; ++s) {
			if (want[s] == nullptr) { CHECK(res.aligned == nullptr); continue; }
			CHECK(res.aligned != nullptr && strcmp(res.aligned[s], want[s]) == 0);
		}
		if (want[0]) CHECK(res.score == os.last_score && res.fills == os.fills);
	}
	for (int s = 0; s < n; ++s) odp_free(want[s]);
	csadp_free_result(&res, n);
}

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: host_sanitize <Primates.txt> <tmpdir>\n"); return 2; }

	/* wire formats */
	char **texts = NULL, **descs = NULL;
	int *sizes = NULL, nseq = 0;
	CHECK(csadp_load_fasta(argv[1], &texts, &descs, &sizes, &nseq) == CSADP_OK && nseq == 16);

	/* rotation finder + anchor map on the real set (threads inside) */
	std::vector<int> rot(nseq);
	csadp_rotation_info ri;
	CHECK(csadp_find_rotations(nseq, (const char *const *)texts, sizes, rot.data(), &ri) == CSADP_OK);
	CHECK(rot[0] == 1947 && rot[3] == 2530 && ri.blocks == 58);
	const std::string path = std::string(argv[2]) + "/x-Rotated.fasta";
	CHECK(csadp_write_rotated_fasta(path.c_str(), (const char *const *)descs, (const char *const *)texts, sizes, rot.data(), nseq) == CSADP_OK);
	std::vector<int> back(nseq);
	int nread = 0;
	CHECK(csadp_read_rotations(path.c_str(), back.data(), nseq, &nread) == CSADP_OK && nread == nseq && back == rot);
	csadp_anchor_map map;
	CHECK(csadp_build_anchor_map(nseq, (const char *const *)texts, sizes, rot.data(), &map) == CSADP_OK);
	CHECK(map.nsegs == 52);

	/* the progressive host logic on real gaps of that map (the small ones: the oracle fills on the CPU) */
	int gaps = 0;
	for (int k = 0; k + 1 < map.nsegs && gaps < 12; ++k) {
		if (!map.dp[k]) continue;
		std::vector<int> st(nseq), en(nseq);
		int widest = 0;
		for (int s = 0; s < nseq; ++s) {
			st[s] = map.positions[(size_t)k * nseq + s] + map.size[k];
			en[s] = map.positions[(size_t)(k + 1) * nseq + s];
			widest = en[s] - st[s] > widest ? en[s] - st[s] : widest;
		}
		if (widest > 400) continue;
		std::vector<std::string> fam(nseq);
		for (int s = 0; s < nseq; ++s) fam[s].assign(texts[s], (size_t)sizes[s]);
		check_task(fam, rot, st, en);
		++gaps;
	}
	CHECK(gaps >= 8);
	csadp_free_anchor_map(&map);

	/* fuzzed families: equal lengths (stale borders), empty regions, many gaps (DeleteGappedColumns) */
	for (int it = 0; it < 60; ++it) {
		const int n = 2 + (int)(rnd() % 9);
		std::vector<std::string> fam = family(n, 20 + (int)(rnd() % 160), 0.15, it % 3 == 0 ? 0.25 : 0.06);
		if (it % 5 == 0) {
			size_t m = fam[0].size();
			for (auto &f : fam) m = f.size() < m ? f.size() : m;
			for (auto &f : fam) f.resize(m);
		}
		std::vector<int> r(n), st(n, 0), en(n);
		for (int s = 0; s < n; ++s) { r[s] = (int)(rnd() % fam[s].size()); en[s] = (int)fam[s].size(); }
		if (it % 7 == 0) st[0] = en[0] = (int)(rnd() % (fam[0].size() + 1));
		check_task(fam, r, st, en);
	}

	/* the round driver of csadp_align_batch itself (lock-step rounds, round groups on their own host threads, per-task host work
	 * on the pool) with the oracle's fills in place of the device step: a batch of families, large ones included so that the
	 * speculated refinement runs, through 1, 2 and 3 groups -- equal results, equal to the oracle's */
	{
		const int ntasks = 9;
		std::vector<std::vector<std::string>> fams;
		std::vector<std::vector<int>> rots, sts, ens;
		std::vector<std::vector<const char *>> txts;
		std::vector<std::vector<int>> szs;
		std::vector<csadp_task> tasks;
		for (int t = 0; t < ntasks; ++t) {
			const int n = 3 + (int)(rnd() % 6);
			fams.push_back(family(n, t < 2 ? 1100 + (int)(rnd() % 200) : 30 + (int)(rnd() % 200), 0.12, t % 3 == 0 ? 0.2 : 0.05));
			std::vector<int> r(n), st(n, 0), en(n);
			for (int q = 0; q < n; ++q) { r[q] = (int)(rnd() % fams.back()[q].size()); en[q] = (int)fams.back()[q].size(); }
			rots.push_back(r); sts.push_back(st); ens.push_back(en);
		}
		for (int t = 0; t < ntasks; ++t) {
			const int n = (int)fams[t].size();
			txts.emplace_back(n); szs.emplace_back(n);
			for (int q = 0; q < n; ++q) { txts[t][q] = fams[t][q].c_str(); szs[t][q] = (int)fams[t][q].size(); }
		}
		for (int t = 0; t < ntasks; ++t) tasks.push_back({(int)fams[t].size(), txts[t].data(), szs[t].data(), rots[t].data(), sts[t].data(), ens[t].data()});
		std::vector<std::vector<std::string>> first;
		for (int groups = 1; groups <= 3; ++groups) {
			char g[8];
			snprintf(g, sizeof g, "%d", groups);
			setenv("CSADP_ROUND_GROUPS", g, 1);
			std::vector<csadp_result> res(ntasks);
			CHECK(csadp_debug_align_batch_with_filler(tasks.data(), ntasks, oracle_fill, NULL, res.data()) == CSADP_OK);
			for (int t = 0; t < ntasks; ++t) {
				const int n = tasks[t].nseq;
				CHECK(res[t].status == CSADP_OK && res[t].aligned != nullptr);
				if (res[t].status != CSADP_OK || !res[t].aligned) continue;
				if (groups == 1) {
					std::vector<char *> want(n, nullptr);
					odp_stats os;
					const int cons = odp_progressive_dp(n, txts[t].data(), szs[t].data(), rots[t].data(), sts[t].data(), ens[t].data(), want.data(), &os);
					CHECK(cons == res[t].consensus);
					first.emplace_back();
					for (int q = 0; q < n; ++q) {
						CHECK(want[q] && strcmp(want[q], res[t].aligned[q]) == 0);
						first.back().push_back(res[t].aligned[q]);
						odp_free(want[q]);
					}
				} else {
					for (int q = 0; q < n; ++q) CHECK(first[t][q] == res[t].aligned[q]);
				}
				csadp_free_result(&res[t], n);
			}
		}
		unsetenv("CSADP_ROUND_GROUPS");
	}

	/* partitioner and digests */
	{
		std::vector<long long> cost(200);
		for (auto &c : cost) c = 1 + rnd() % 100000;
		std::vector<int> part(200);
		long long maxload = 0, total = 0;
		CHECK(csadp_partition_lpt(cost.data(), 200, 8, part.data(), &maxload) == CSADP_OK);
		for (long long c : cost) total += c;
		CHECK(maxload * 8 <= total * 105 / 100 + 100000);
		const char *two[2] = {"ACGT-", "AC-TT"};
		CHECK(csadp_fnv1a(two, 2) != 0);
	}

	/* the host thread pool from several caller threads at once */
	{
		std::vector<std::thread> th;
		std::vector<long long> sums(4, 0);
		for (int t = 0; t < 4; ++t)
			th.emplace_back([&sums, t] {
				for (int rep = 0; rep < 20; ++rep) {
					long long s = 0;
					if (csadp_debug_pool_selftest(1000 + t, &s) != CSADP_OK) s = -1;
					sums[(size_t)t] += s;
				}
			});
		for (auto &x : th) x.join();
		for (int t = 0; t < 4; ++t) CHECK(sums[(size_t)t] == 20LL * (1000 + t) * (999 + t) / 2);
	}

	csadp_free_fasta(texts, descs, sizes, nseq);
	if (failures) { fprintf(stderr, "host_sanitize: %d check(s) failed\n", failures); return 1; }
	printf("host_sanitize ok\n");
	return 0;
}
