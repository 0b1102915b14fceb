/*
 * refine_bench.cpp -- host-side profile harness (CPU only): the product's host logic of ProgressiveDP (tables, trace application,
 * DeleteGappedColumns) on families of N sequences, with the matrix fills REPLAYED from op lists the oracle produced once.
 *   g++ -O2 -pg ... (tools/r05/refine_bench.sh) ; refine_bench <families> <nseq> <length> <repeats>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "csadp.h"
#include "csadp_debug.h"
#include "../../oracle/csa_dp_oracle.h"

struct Recorded { std::vector<unsigned char> ops; int nops, remj, remk, score; };
struct Tape { std::vector<Recorded> fills; size_t next = 0; bool recording = true; };

static int fill(void *user, int nrows, int ncols, int nprev, const int *sv, const signed char *rowcodes, const int *top, int left_i,
                unsigned char *ops, int *nops, int *remj, int *remk, int *score)
{
	Tape &T = *(Tape *)user;
	if (!T.recording) {
		const Recorded &R = T.fills[T.next++];
		memcpy(ops, R.ops.data(), (size_t)R.nops);
		*nops = R.nops; *remj = R.remj; *remk = R.remk; *score = R.score;
		return CSADP_OK;
	}
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
	Recorded R;
	R.ops.assign(ops, ops + n);
	R.nops = n; R.remj = j; R.remk = k; R.score = H[(size_t)nrows * pitch + ncols];
	T.fills.push_back(R);
	*nops = n; *remj = j; *remk = k; *score = R.score;
	return CSADP_OK;
}

static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 32); }

int main(int argc, char **argv)
{
	const int nfam = argc > 1 ? atoi(argv[1]) : 4, nseq = argc > 2 ? atoi(argv[2]) : 8, len = argc > 3 ? atoi(argv[3]) : 4000, reps = argc > 4 ? atoi(argv[4]) : 20;
	double total = 0;
	for (int f = 0; f < nfam; ++f) {
		std::string base((size_t)len, 'A');
		for (char &c : base) c = "ACGT"[rnd() & 3];
		std::vector<std::string> seqs;
		for (int s = 0; s < nseq; ++s) {
			std::string t;
			for (char c : base) {
				const unsigned u = rnd() % 1000;
				if (u < 20) continue;
				if (u < 40) t.push_back("ACGT"[rnd() & 3]);
				t.push_back(u < 120 ? "ACGT"[rnd() & 3] : c);
			}
			seqs.push_back(t);
		}
		std::vector<const char *> ptr;
		std::vector<int> size, zero, end;
		for (auto &t : seqs) { ptr.push_back(t.c_str()); size.push_back((int)t.size()); zero.push_back(0); }
		end = size;
		csadp_task task{nseq, ptr.data(), size.data(), zero.data(), zero.data(), end.data()};
		Tape T;
		csadp_result res;
		if (csadp_debug_align_with_filler(&task, fill, &T, &res) != CSADP_OK || res.status != CSADP_OK) { fprintf(stderr, "record failed\n"); return 1; }
		const int cons = res.consensus;
		std::string first = res.aligned[0];
		csadp_free_result(&res, nseq);
		T.recording = false;
		const auto t0 = std::chrono::steady_clock::now();
		for (int r = 0; r < reps; ++r) {
			T.next = 0;
			if (csadp_debug_align_with_filler(&task, fill, &T, &res) != CSADP_OK || res.status != CSADP_OK || first != res.aligned[0]) { fprintf(stderr, "replay failed\n"); return 1; }
			csadp_free_result(&res, nseq);
		}
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
		total += ms;
		printf("family %d: %d x %d, consensus %d: host logic %.2f ms per task (%d steps)\n", f, nseq, len, cons, ms, nseq - 1);
	}
	printf("mean %.2f ms per task\n", total / nfam);
	return 0;
}
