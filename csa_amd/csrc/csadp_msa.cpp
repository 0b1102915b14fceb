/*
 * csadp_msa.cpp -- the reference's mode-N alignment stage end to end behind the C-ABI:
 * rotations (csamsa.c:610) -> anchor map (alignment.c:69-86, :163-214) -> ONE device batch
 * holding every gap the anchor loop hands to ProgressiveDP (alignment.c:201) -> rows as
 * SaveAlignment prints them (alignment.c:88-160).
 *
 * The reference runs its gaps one after the other inside the anchor loop.  The loop never looks
 * at a DP result, so the map is built first and the gaps then advance together in lock step on
 * the device (batch across gaps).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "csadp.h"
#include "csadp_config.h"

namespace {

double ms_since(const std::chrono::steady_clock::time_point &t0)
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void append_rotated(std::string *row, const char *text, int n, int rot, int from, int count)
{
	int i = rot + from;
	if (i >= n) i -= n;
	for (int k = 0; k < count; ++k) {           /* CharAt, alignment.c:16-20 */
		row->push_back(text[i]);
		if (++i == n) i = 0;
	}
}

}  // namespace

extern "C" {

int csadp_msa(int nseq, const char *const *texts, const int *sizes, const int *rotations_in, int *rotations_out,
              char ***rows_out, csadp_msa_stats *stats)
{
	if (nseq < 2 || nseq > CSADP_MAX_SEQS || !texts || !sizes || !rows_out) return CSADP_ERR_ARG;
	*rows_out = NULL;
	csadp_msa_stats st;
	memset(&st, 0, sizeof(st));
	st.nseq = nseq;

	/* the HIP runtime takes ~0.15 s to come up in a fresh process: start it now, beside the host
	 * stages (a no-op when the library is already initialised) */
	int init_rc = CSADP_OK;
	std::thread warm([&init_rc] { init_rc = csadp_init(NULL); });
	struct Joiner {
		std::thread &t;
		~Joiner() { if (t.joinable()) t.join(); }
	} joiner{warm};

	std::vector<int> rot((size_t)nseq);
	auto t0 = std::chrono::steady_clock::now();
	if (rotations_in) {
		for (int s = 0; s < nseq; ++s) rot[(size_t)s] = rotations_in[s];
	} else {
		const int rc = csadp_find_rotations(nseq, texts, sizes, rot.data(), NULL);
		if (rc != CSADP_OK) return rc;
	}
	st.rotations_ms = ms_since(t0);
	if (rotations_out)
		for (int s = 0; s < nseq; ++s) rotations_out[s] = rot[(size_t)s];

	t0 = std::chrono::steady_clock::now();
	csadp_anchor_map map;
	int rc = csadp_build_anchor_map(nseq, texts, sizes, rot.data(), &map);
	if (rc != CSADP_OK) return rc;
	st.anchors_ms = ms_since(t0);
	st.segments = map.nsegs;
	st.border_nodes = map.border_nodes;

	/* one task per gap flagged for DP */
	t0 = std::chrono::steady_clock::now();
	std::vector<int> gap_of;                              /* task -> segment index */
	for (int k = 0; k + 1 < map.nsegs; ++k)
		if (map.dp[k]) gap_of.push_back(k);
	const int ntasks = (int)gap_of.size();
	std::vector<int> starts((size_t)ntasks * nseq), ends((size_t)ntasks * nseq);
	std::vector<csadp_task> tasks((size_t)ntasks);
	std::vector<csadp_result> results((size_t)ntasks);
	for (int t = 0; t < ntasks; ++t) {
		const int k = gap_of[(size_t)t];
		for (int s = 0; s < nseq; ++s) {
			starts[(size_t)t * nseq + s] = map.positions[(size_t)k * nseq + s] + map.size[k];
			ends[(size_t)t * nseq + s] = map.positions[(size_t)(k + 1) * nseq + s];
		}
		tasks[(size_t)t].nseq = nseq;
		tasks[(size_t)t].texts = texts;
		tasks[(size_t)t].textsizes = sizes;
		tasks[(size_t)t].rotations = rot.data();
		tasks[(size_t)t].starts = &starts[(size_t)t * nseq];
		tasks[(size_t)t].ends = &ends[(size_t)t * nseq];
	}
	warm.join();
	if (init_rc != CSADP_OK) {
		csadp_free_anchor_map(&map);
		return init_rc;
	}
	if (ntasks > 0) {
		const long rec0 = csadp::primary_engine_recoveries();
		rc = csadp_align_batch(tasks.data(), ntasks, results.data());
		st.recoveries = (int)(csadp::primary_engine_recoveries() - rec0);
		if (rc == CSADP_OK)
			for (int t = 0; t < ntasks && rc == CSADP_OK; ++t) rc = results[(size_t)t].status;
		if (rc != CSADP_OK) {
			for (int t = 0; t < ntasks; ++t) csadp_free_result(&results[(size_t)t], nseq);
			csadp_free_anchor_map(&map);
			return rc;
		}
	}
	st.dp_gaps = ntasks;
	for (int t = 0; t < ntasks; ++t) {
		st.cells += results[(size_t)t].cells;
		st.fills += results[(size_t)t].fills;
	}
	st.dp_ms = ms_since(t0);

	/* rows, alignment.c:104-154 */
	t0 = std::chrono::steady_clock::now();
	char **rows = (char **)calloc((size_t)nseq, sizeof(char *));
	if (!rows) rc = CSADP_ERR_NOMEM;
	for (int s = 0; s < nseq && rc == CSADP_OK; ++s) {
		std::string row;
		int task = 0;
		for (int k = 0; k + 1 < map.nsegs; ++k) {
			if (k > 0) append_rotated(&row, texts[s], sizes[s], rot[(size_t)s], map.positions[(size_t)k * nseq + s], map.size[k]);
			if (!map.dp[k]) continue;                  /* skipped gap: nothing is printed (alignment.c:139) */
			const csadp_result &r = results[(size_t)task++];
			if (r.aligned && r.aligned[s]) {
				row.append(r.aligned[s]);
			} else if (r.aligned) {                    /* alignment.c:145-158 */
				const int from = map.positions[(size_t)k * nseq + s] + map.size[k];
				append_rotated(&row, texts[s], sizes[s], rot[(size_t)s], from, map.positions[(size_t)(k + 1) * nseq + s] - from);
			}
		}
		rows[s] = strdup(row.c_str());
		if (!rows[s]) rc = CSADP_ERR_NOMEM;
		if (s == 0) st.alignment_length = (int)row.size();
	}
	for (int t = 0; t < ntasks; ++t) csadp_free_result(&results[(size_t)t], nseq);
	csadp_free_anchor_map(&map);
	if (rc != CSADP_OK) {
		csadp_free_rows(rows, nseq);
		return rc;
	}
	st.rows_ms = ms_since(t0);
	*rows_out = rows;
	if (stats) *stats = st;
	return CSADP_OK;
}

void csadp_free_rows(char **rows, int nseq)
{
	if (!rows) return;
	for (int s = 0; s < nseq; ++s) free(rows[s]);
	free(rows);
}

/* "<base>-Aligned.fasta" as SaveAlignment writes it (alignment.c:97-105, :160) */
int csadp_write_aligned_fasta(const char *path, const char *const *descs, const int *rotations, const char *const *rows, int nseq)
{
	if (!path || !descs || !rows || nseq < 1) return CSADP_ERR_ARG;
	FILE *f = fopen(path, "wb");
	if (!f) return CSADP_ERR_ARG;
	for (int s = 0; s < nseq; ++s) {
		if (rotations) fprintf(f, ">%s @ %d\n", descs[s], rotations[s]);
		else fprintf(f, ">%s\n", descs[s]);
		fputs(rows[s], f);
		fputc('\n', f);
	}
	return fclose(f) == 0 ? CSADP_OK : CSADP_ERR_ARG;
}

}  // extern "C"
