/*
 * oracle/csa_dp_oracle.c -- TEST INFRASTRUCTURE ONLY (see csa_dp_oracle.h).
 *
 * Plain-C restatement of the reference's progressive sequence-vs-profile DP.
 * Every function cites the lines of /root/reference/source it follows.  The
 * data structures are flat arrays and an explicit context instead of the
 * reference's header-defined globals; the arithmetic, the tie-breaks, the
 * border-refresh rule (survey quirk Q1) and the column-shifting heuristic are
 * the reference's.
 *
 * Parity status: PINNED (see header).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "csa_dp_oracle.h"

/* compiled-in scores, dynamicprogramming.c:16-19 */
#define S_MATCH     (+1)
#define S_DOUBLEGAP (0)
#define S_MISMATCH  (-1)
#define S_INDEL     (-1)
#define NSYM 5        /* A C G T -  (ALPHABETSIZE+1, :9-12) */
#define GAP  4

typedef struct {
	int nseq;
	const char *const *texts;
	const int *textsizes, *rotations, *starts, *ends;
	int *order, *len;        /* orderedseqs / seqlengths, :30-31          */
	int *sv;                 /* scorevector, (consensus+1) x 5, col 0 unused */
	char **strings;          /* per ORIGINAL index, capacity >= consensus+1 */
	int consensus;
	int **H;                 /* dpmatrix (:28): one malloc per row, as :964-966 */
	char **D;                /* dpdirs   (:914,:979-981)                        */
	int matrows;             /* rows currently allocated in H/D                 */
	int prevconsensus, prevnrows;
	odp_stats st;
} ctx_t;

static double now_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* alignment.c:16-20 */
static char char_at(const ctx_t *c, int pos, int seq)
{
	int i = c->rotations[seq] + pos;
	if (i >= c->textsizes[seq]) i -= c->textsizes[seq];
	return c->texts[seq][i];
}

/* dynamicprogramming.c:74-85 (and :63-69 without the gap) */
static int char_code(char ch)
{
	switch (ch) {
	case 'A': return 0;
	case 'C': return 1;
	case 'G': return 2;
	case 'T': return 3;
	case '-': return 4;
	default:  return -1;
	}
}

/* dynamicprogramming.c:286-307: selection sort by region length, ascending, with swap */
static void sort_sequences(ctx_t *c)
{
	int i, j, min, minpos, aux;
	for (i = 0; i < c->nseq; i++) {
		c->order[i] = i;
		c->len[i] = c->ends[i] - c->starts[i];
	}
	for (i = 0; i < c->nseq - 1; i++) {
		min = c->len[i];
		minpos = i;
		for (j = i + 1; j < c->nseq; j++) {
			if (c->len[j] < min) { min = c->len[j]; minpos = j; }
		}
		if (minpos != i) {
			aux = c->order[i]; c->order[i] = c->order[minpos]; c->order[minpos] = aux;
			aux = c->len[i];   c->len[i] = c->len[minpos];     c->len[minpos] = aux;
		}
	}
}

/* ---- DeleteGappedColumns, dynamicprogramming.c:643-899 ------------------- */

#define SV(col, n) sv[(size_t)(col) * NSYM + (n)]

typedef struct {
	int cap;                /* tempsvsize (:650) */
	int *stat, *mov, *work, *best, *shifted;
} dgc_scratch;

static int dgc_reserve(dgc_scratch *s, int need)
{
	if (need <= s->cap) return 0;
	s->stat = (int *)realloc(s->stat, (size_t)need * NSYM * sizeof(int));
	s->mov = (int *)realloc(s->mov, (size_t)need * NSYM * sizeof(int));
	s->work = (int *)realloc(s->work, (size_t)need * NSYM * sizeof(int));
	s->best = (int *)realloc(s->best, (size_t)need * NSYM * sizeof(int));
	s->shifted = (int *)realloc(s->shifted, (size_t)need * sizeof(int));
	if (!s->stat || !s->mov || !s->work || !s->best || !s->shifted) return -1;
	s->cap = need;
	return 0;
}

static int delete_gapped_columns(ctx_t *c, int numseqs, int maxnongaps)
{
	int *sv = c->sv;
	char **str = c->strings;
	const int *usable = c->order;
	int i, j, k, n, m, col, mingaps, ii, jj;
	int ntoshift, postofarthestgap, minnextgaps, maxposaffected;
	int charcode, currentscore, bestscore, bestshift, colscore;
	int looplimit, dirsignal, bestmaxposaffected = 0;
	int consize = c->consensus;
	dgc_scratch s = {0, NULL, NULL, NULL, NULL, NULL};
	int *nposaffected = (int *)malloc((size_t)(maxnongaps + 1) * sizeof(int));
	int *seqstoshift = (int *)malloc((size_t)(numseqs + 1) * sizeof(int));
	int *postonextgap = (int *)malloc((size_t)(maxnongaps + 1) * sizeof(int));
	int *nnextgaps = (int *)malloc((size_t)(maxnongaps + 1) * sizeof(int));
	int *bestnposaffected = (int *)malloc((size_t)(maxnongaps + 1) * sizeof(int));
	int *tmpcol = NULL;
	int rc = 0;

	mingaps = numseqs - maxnongaps;                                   /* :668 */
	for (col = 1; col <= consize; col++) {                            /* :677 */
		if (SV(col, GAP) < mingaps) continue;                         /* :678 */
		ntoshift = 0;
		for (i = 0; i < numseqs; i++) {                               /* :681-687 */
			ii = usable[i];
			if (str[ii][col - 1] != '-') seqstoshift[ntoshift++] = ii;
		}
		if (ntoshift == 0) continue;                                  /* :688-691 ("!" token) */
		bestscore = 0;
		bestshift = 0;
		looplimit = consize + 1;
		dirsignal = +1;
		for (;;) {                                                    /* :696 */
			postofarthestgap = 0;
			minnextgaps = consize;
			for (k = 0; k < ntoshift; k++) {                          /* :699-715 */
				i = seqstoshift[k];
				j = col;
				postonextgap[k] = 0;
				while (j != looplimit && str[i][j - 1] != '-') { postonextgap[k]++; j += dirsignal; }
				if (j == looplimit) break;
				if (postonextgap[k] > postofarthestgap) postofarthestgap = postonextgap[k];
				nnextgaps[k] = 0;
				while (j != looplimit && str[i][j - 1] == '-') { nnextgaps[k]++; j += dirsignal; }
				if (nnextgaps[k] < minnextgaps) minnextgaps = nnextgaps[k];
			}
			if (k != ntoshift) {                                      /* :716-721 */
				if (dirsignal == -1) break;
				looplimit = 0;
				dirsignal = -1;
				continue;
			}
			for (k = 0; k < ntoshift; k++) nposaffected[k] = postonextgap[k] + minnextgaps;  /* :722 */
			maxposaffected = postofarthestgap + minnextgaps;
			if (dgc_reserve(&s, maxposaffected) != 0) { rc = ODP_ERR_NOMEM; goto done; }
			currentscore = 0;
			for (j = 0; j < maxposaffected; j++) {                    /* :739-761 */
				jj = col + dirsignal * j;
				for (n = 0; n < NSYM; n++) {
					s.stat[j * NSYM + n] = SV(jj, n);
					s.mov[j * NSYM + n] = 0;
				}
				for (k = 0; k < ntoshift; k++) {
					if (j < nposaffected[k]) {
						i = seqstoshift[k];
						charcode = char_code(str[i][jj - 1]);
						s.mov[j * NSYM + charcode]++;
						s.stat[j * NSYM + charcode]--;
					}
				}
				colscore = 0;
				for (n = 0; n < GAP; n++) {
					if (s.mov[j * NSYM + n] != 0)
						colscore += s.mov[j * NSYM + n] * (S_MATCH * (SV(jj, n) - 1)
						          + S_MISMATCH * (numseqs - (SV(jj, n) + SV(jj, GAP)))
						          + S_INDEL * SV(jj, GAP));
				}
				if (s.mov[j * NSYM + GAP] != 0)
					colscore += s.mov[j * NSYM + GAP] * (S_DOUBLEGAP * (SV(jj, GAP) - 1)
					          + S_INDEL * (numseqs - SV(jj, GAP)));
				currentscore += colscore;
			}
			for (i = 1; i <= minnextgaps; i++) {                      /* :762-795 */
				s.shifted[i - 1] = 0;
				for (k = 0; k < ntoshift; k++) {
					j = nposaffected[k] - 1;
					s.mov[j * NSYM + GAP]--;
					nposaffected[k]--;
				}
				for (j = 0; j < maxposaffected; j++) {
					int *w = &s.work[j * NSYM];
					colscore = 0;
					if (j < i) {
						for (n = 0; n < GAP; n++) w[n] = 0;
						w[GAP] = s.stat[j * NSYM + GAP] + ntoshift;
						if (w[GAP] == numseqs) continue;
						colscore += ntoshift * (S_DOUBLEGAP * (w[GAP] - 1) + S_INDEL * (numseqs - w[GAP]));
						s.shifted[i - 1] += colscore;
						continue;
					}
					for (n = 0; n < GAP; n++) w[n] = s.stat[j * NSYM + n] + s.mov[(j - i) * NSYM + n];
					w[GAP] = s.stat[j * NSYM + GAP] + s.mov[(j - i) * NSYM + GAP];
					if (w[GAP] == numseqs) continue;
					for (n = 0; n < GAP; n++) {
						if (s.mov[(j - i) * NSYM + n] != 0)
							colscore += s.mov[(j - i) * NSYM + n] * (S_MATCH * (w[n] - 1)
							          + S_MISMATCH * (numseqs - (w[n] + w[GAP])) + S_INDEL * w[GAP]);
					}
					if (s.mov[(j - i) * NSYM + GAP] != 0)
						colscore += s.mov[(j - i) * NSYM + GAP] * (S_DOUBLEGAP * (w[GAP] - 1)
						          + S_INDEL * (numseqs - w[GAP]));
					s.shifted[i - 1] += colscore;
				}
				s.shifted[i - 1] -= currentscore;
				if (s.shifted[i - 1] >= bestscore) {                  /* :791: ">=" keeps the LAST best */
					bestshift = dirsignal * i;
					bestscore = s.shifted[i - 1];
				}
			}
			if (bestshift != 0 && (bestshift * dirsignal) > 0) {      /* :796-818 */
				bestmaxposaffected = maxposaffected;
				i = bestshift * dirsignal;
				n = minnextgaps - i;
				for (k = 0; k < ntoshift; k++) {
					m = postonextgap[k];
					for (j = 0; j < n; j++) { s.mov[m * NSYM + GAP]++; m++; }
					bestnposaffected[k] = postonextgap[k] + i;
				}
				for (j = 0; j < maxposaffected; j++) {
					if (j < i) {
						for (n = 0; n < NSYM; n++) s.best[j * NSYM + n] = s.stat[j * NSYM + n];
						s.best[j * NSYM + GAP] += ntoshift;
						continue;
					}
					for (n = 0; n < NSYM; n++) s.best[j * NSYM + n] = s.stat[j * NSYM + n] + s.mov[(j - i) * NSYM + n];
				}
			}
			if (dirsignal == -1) break;                               /* :819-821 */
			looplimit = 0;
			dirsignal = -1;
		}
		if (bestshift == 0) continue;                                 /* :823 */
		dirsignal = +1;
		if (bestshift < 0) { dirsignal = -1; bestshift = -bestshift; }
		for (j = 0; j < bestmaxposaffected; j++)                      /* :837-840 */
			for (n = 0; n < NSYM; n++) SV(col + dirsignal * j, n) = s.best[j * NSYM + n];
		for (k = 0; k < ntoshift; k++) {                              /* :841-852 */
			i = seqstoshift[k];
			m = dirsignal * bestshift;
			for (j = bestnposaffected[k] - 1; j >= 0; j--) {
				n = col + dirsignal * j;
				if (j < bestshift) { str[i][n - 1] = '-'; continue; }
				str[i][n - 1] = str[i][n - m - 1];
			}
		}
		n = consize;                                                  /* :853-864 */
		m = 0;
		for (j = col; j <= n; j++) { if (SV(j, GAP) != numseqs) break; m++; }
		k = 0;
		for (j = col - 1; j >= 1; j--) { if (SV(j, GAP) != numseqs) break; k++; }
		m = m + k;
		/* :865-886: the m all-gap columns starting at (col-k) are rotated to the
		 * end of the vector (with their gap count reset to 0) and the strings
		 * are closed up and NUL-filled at the tail. */
		if (m > 0) {
			tmpcol = (int *)realloc(tmpcol, (size_t)m * NSYM * sizeof(int));
			if (!tmpcol) { rc = ODP_ERR_NOMEM; goto done; }
			memcpy(tmpcol, &SV(col - k, 0), (size_t)m * NSYM * sizeof(int));
		}
		for (j = col - k; j <= n - m; j++) {
			memmove(&SV(j, 0), &SV(j + m, 0), NSYM * sizeof(int));
			for (i = 0; i < numseqs; i++) {
				ii = usable[i];
				str[ii][j - 1] = str[ii][j + m - 1];
			}
		}
		for (j = 0; j < m; j++) {
			tmpcol[j * NSYM + GAP] = 0;
			memcpy(&SV(n - j, 0), &tmpcol[j * NSYM], NSYM * sizeof(int));
			for (i = 0; i < numseqs; i++) {
				ii = usable[i];
				str[ii][n - j - 1] = '\0';
			}
		}
		consize = consize - m;                                        /* :887 */
		col = col - (k + 1);                                          /* :888 */
	}
done:
	c->consensus = consize;
	free(nposaffected); free(seqstoshift); free(postonextgap); free(nnextgaps); free(bestnposaffected);
	free(s.stat); free(s.mov); free(s.work); free(s.best); free(s.shifted); free(tmpcol);
	return rc;
}

/* ---- one fill, dynamicprogramming.c:990-1029 ---------------------------- */

static void fill_matrix(int nrows, int ncols, const signed char *rowcodes, const int *sv, int nprev,
                        int **H, char **D)
{
	int j, k;
	for (j = 1; j <= nrows; j++) {
		const int charcode = rowcodes[j - 1];
		const int *hp = H[j - 1];
		int *hc = H[j];
		char *dc = D[j];
		for (k = 1; k <= ncols; k++) {
			const int *col = &sv[(size_t)k * NSYM];
			int score = S_MATCH * col[charcode] + S_INDEL * col[GAP]
			          + S_MISMATCH * (nprev - (col[charcode] + col[GAP]));          /* :993 */
			int rowgap = S_INDEL * nprev;                                           /* :994 */
			int colgap = S_DOUBLEGAP * col[GAP] + S_INDEL * (nprev - col[GAP]);     /* :995 */
			int diag = hp[k - 1] + score;
			int up = hp[k] + rowgap;
			int left = hc[k - 1] + colgap;
			if (diag >= up && diag >= left) { hc[k] = diag; dc[k] = 'D'; continue; } /* :1014 */
			if (left >= up) { hc[k] = left; dc[k] = 'L'; continue; }                 /* :1019 */
			hc[k] = up; dc[k] = 'U';
		}
	}
}

/* fresh borders, dynamicprogramming.c:963-985 */
static void init_borders(int nrows, int ncols, const int *sv, int nprev, int **H, char **D)
{
	int j, colgap = 0;
	int rowgap = S_INDEL * nprev;
	for (j = 0; j <= nrows; j++) {
		H[j][0] = j * rowgap;
		D[j][0] = 'U';
	}
	for (j = 1; j <= ncols; j++) {
		colgap += S_DOUBLEGAP * sv[(size_t)j * NSYM + GAP] + S_INDEL * (nprev - sv[(size_t)j * NSYM + GAP]);
		H[0][j] = colgap;
		D[0][j] = 'L';
	}
	D[0][0] = 'D';
}

static void free_matrix(ctx_t *c)
{
	int j;
	if (c->H) { for (j = 0; j < c->matrows; j++) free(c->H[j]); free(c->H); c->H = NULL; }
	if (c->D) { for (j = 0; j < c->matrows; j++) free(c->D[j]); free(c->D); c->D = NULL; }
	c->matrows = 0;
}

/* :958-966,:974-981: the matrices are released and re-created row by row */
static int alloc_matrix(ctx_t *c, int nrows, int ncols)
{
	int j;
	free_matrix(c);
	c->H = (int **)calloc((size_t)nrows + 1, sizeof(int *));
	c->D = (char **)calloc((size_t)nrows + 1, sizeof(char *));
	if (!c->H || !c->D) return -1;
	c->matrows = nrows + 1;
	for (j = 0; j <= nrows; j++) {
		c->H[j] = (int *)malloc(((size_t)ncols + 1) * sizeof(int));
		c->D[j] = (char *)malloc((size_t)ncols + 1);
		if (!c->H[j] || !c->D[j]) return -1;
	}
	return 0;
}

int odp_fill(int nrows, int ncols, const signed char *rowcodes, const int *sv, int nprev,
             const int *top, int left_i, int *H, char *dirs)
{
	size_t pitch = (size_t)ncols + 1;
	int j;
	int **hr;
	char **dr;
	if (nrows < 0 || ncols < 0 || !H || !dirs) return ODP_ERR_ARG;
	hr = (int **)malloc(((size_t)nrows + 1) * sizeof(int *));
	dr = (char **)malloc(((size_t)nrows + 1) * sizeof(char *));
	if (!hr || !dr) { free(hr); free(dr); return ODP_ERR_NOMEM; }
	for (j = 0; j <= nrows; j++) { hr[j] = H + (size_t)j * pitch; dr[j] = dirs + (size_t)j * pitch; }
	init_borders(nrows, ncols, sv, nprev, hr, dr);
	if (top != NULL) {
		for (j = 0; j <= ncols; j++) hr[0][j] = top[j];
		for (j = 0; j <= nrows; j++) hr[j][0] = -left_i * j;
	}
	fill_matrix(nrows, ncols, rowcodes, sv, nprev, hr, dr);
	free(hr); free(dr);
	return ODP_OK;
}

/* ---- ProgressiveDP, dynamicprogramming.c:906-1171 ------------------------ */

static void free_strings(char **strings, int nseq)
{
	int s;
	if (!strings) return;
	for (s = 0; s < nseq; s++) free(strings[s]);
	free(strings);
}

int odp_progressive_dp(int nseq, const char *const *texts, const int *textsizes,
                       const int *rotations, const int *starts, const int *ends,
                       char **out, odp_stats *stats)
{
	ctx_t c;
	int i, j, k, l, m, n, p, s, nrows, ncols, pos, maxgap = 0;
	int rc = ODP_OK;
	signed char *rowcodes = NULL;

	if (nseq < 2 || !texts || !textsizes || !rotations || !starts || !ends || !out) return ODP_ERR_ARG;
	memset(&c, 0, sizeof(c));
	c.nseq = nseq; c.texts = texts; c.textsizes = textsizes; c.rotations = rotations;
	c.starts = starts; c.ends = ends;
	for (s = 0; s < nseq; s++) {
		out[s] = NULL;
		if (ends[s] < starts[s] || starts[s] < 0 || ends[s] > textsizes[s]) return ODP_ERR_ARG;
		if (ends[s] - starts[s] > maxgap) maxgap = ends[s] - starts[s];
		/* survey Q4: the reference indexes scorevector[][-1] for non-ACGT letters */
		for (p = starts[s]; p < ends[s]; p++) {
			int code = char_code(char_at(&c, p, s));
			if (code < 0 || code > 3) return ODP_ERR_ALPHABET;
		}
	}
	if (stats) memset(stats, 0, sizeof(*stats));
	if (maxgap == 0) return 0;                                        /* :916 */

	c.order = (int *)calloc((size_t)nseq, sizeof(int));
	c.len = (int *)calloc((size_t)nseq, sizeof(int));
	c.strings = (char **)calloc((size_t)nseq, sizeof(char *));
	if (!c.order || !c.len || !c.strings) { rc = ODP_ERR_NOMEM; goto fail; }
	sort_sequences(&c);                                               /* :918 */

	c.prevconsensus = 0;                                              /* :924-928 */
	c.prevnrows = 0;
	c.consensus = c.len[0];
	ncols = c.consensus;
	c.sv = (int *)calloc((size_t)(ncols + 1) * NSYM, sizeof(int));    /* :929-932 */
	n = c.order[0];
	c.strings[n] = (char *)malloc((size_t)ncols + 1);                 /* :933-944 */
	if (!c.sv || !c.strings[n]) { rc = ODP_ERR_NOMEM; goto fail; }
	c.strings[n][ncols] = '\0';
	pos = starts[n];
	for (m = 1; m <= ncols; m++) {
		char ch = char_at(&c, pos, n);
		c.strings[n][m - 1] = ch;
		c.sv[(size_t)m * NSYM + char_code(ch)]++;
		pos++;
	}

	for (i = 1; i < nseq; i++) {                                      /* :945 */
		int newcons, inplace;
		int *newsv;
		char **newstr;
		char *string;
		ncols = c.consensus;
		nrows = c.len[i];
		n = c.order[i];
		if (nrows == 0) {                                             /* :950-956 */
			c.strings[n] = (char *)malloc((size_t)ncols + 1);
			if (!c.strings[n]) { rc = ODP_ERR_NOMEM; goto fail; }
			memset(c.strings[n], '-', (size_t)ncols);
			c.strings[n][ncols] = '\0';
			continue;
		}
		if (c.consensus != c.prevconsensus || nrows > c.prevnrows) {  /* :957-987 */
			if (alloc_matrix(&c, nrows, ncols) != 0) { rc = ODP_ERR_NOMEM; goto fail; }
			init_borders(nrows, ncols, c.sv, i, c.H, c.D);
			c.prevnrows = nrows;
		} else {
			c.st.stale_border_fills++;                                /* survey Q1 */
		}
		rowcodes = (signed char *)realloc(rowcodes, (size_t)nrows);
		if (!rowcodes) { rc = ODP_ERR_NOMEM; goto fail; }
		pos = starts[n];                                              /* :988-991 */
		for (j = 0; j < nrows; j++) rowcodes[j] = (signed char)char_code(char_at(&c, pos + j, n));
		{
			double t0 = now_s();
			fill_matrix(nrows, ncols, rowcodes, c.sv, i, c.H, c.D);
			c.st.fill_seconds += now_s() - t0;
		}
		c.st.fills++;
		c.st.cells += (long long)nrows * (long long)ncols;
		c.st.last_score = c.H[nrows][ncols];

		c.prevconsensus = c.consensus;                                /* :1033-1049 */
		newcons = 0;
		j = nrows; k = ncols;
		while (j > 0 && k > 0) {
			char d = c.D[j][k];
			if (d == 'D') { j--; k--; }
			else if (d == 'L') { k--; }
			else if (d == 'U') { j--; }
			newcons++;
		}
		if (j > 0) newcons += j;
		if (k > 0) newcons += k;
		inplace = (newcons == c.prevconsensus);
		if (!inplace) {                                               /* :1050-1062 */
			newsv = (int *)calloc((size_t)(newcons + 1) * NSYM, sizeof(int));
			newstr = (char **)calloc((size_t)nseq, sizeof(char *));
			if (!newsv || !newstr) { free(newsv); free(newstr); rc = ODP_ERR_NOMEM; goto fail; }
			for (j = 0; j < i; j++) {
				p = c.order[j];
				newstr[p] = (char *)malloc((size_t)newcons + 1);
				if (!newstr[p]) { free(newsv); free_strings(newstr, nseq); rc = ODP_ERR_NOMEM; goto fail; }
				newstr[p][newcons] = '\0';
			}
		} else {
			newsv = c.sv;
			newstr = c.strings;
		}
		newstr[n] = (char *)malloc((size_t)newcons + 1);              /* :1063-1071 */
		if (!newstr[n]) { rc = ODP_ERR_NOMEM; goto fail; }
		newstr[n][newcons] = '\0';
		string = newstr[n];
		j = nrows; k = ncols;
		m = newcons - 1;
		pos = ends[n] - 1;   /* the reference keeps a look-ahead charcode (:1071,:1084); the code of the
		                      * char just emitted is the same value without reading position start-1 */
		while (j > 0 && k > 0) {                                      /* :1072-1114 */
			char d = c.D[j][k];
			if (d == 'D') {
				if (!inplace) {
					for (l = 0; l < NSYM; l++) newsv[(size_t)(m + 1) * NSYM + l] = c.sv[(size_t)k * NSYM + l];
					for (l = 0; l < i; l++) { p = c.order[l]; newstr[p][m] = c.strings[p][k - 1]; }
				}
				string[m] = char_at(&c, pos, n);
				newsv[(size_t)(m + 1) * NSYM + char_code(string[m])]++;
				pos--;
				j--; k--;
			} else if (d == 'L') {
				if (!inplace) {
					for (l = 0; l < NSYM; l++) newsv[(size_t)(m + 1) * NSYM + l] = c.sv[(size_t)k * NSYM + l];
					for (l = 0; l < i; l++) { p = c.order[l]; newstr[p][m] = c.strings[p][k - 1]; }
				}
				string[m] = '-';
				newsv[(size_t)(m + 1) * NSYM + GAP]++;
				k--;
			} else { /* 'U' */
				if (!inplace) {
					newsv[(size_t)(m + 1) * NSYM + GAP] = 0;
					for (l = 0; l < i; l++) {
						p = c.order[l];
						newstr[p][m] = '-';
						newsv[(size_t)(m + 1) * NSYM + GAP]++;
					}
				}
				string[m] = char_at(&c, pos, n);
				newsv[(size_t)(m + 1) * NSYM + char_code(string[m])]++;
				pos--;
				j--;
			}
			m--;
		}
		while (j > 0) {                                               /* :1115-1127 */
			for (l = 0; l < i; l++) {
				p = c.order[l];
				newstr[p][m] = '-';
				newsv[(size_t)(m + 1) * NSYM + GAP]++;
			}
			string[m] = char_at(&c, pos, n);
			newsv[(size_t)(m + 1) * NSYM + char_code(string[m])]++;
			pos--;
			j--; m--;
		}
		while (k > 0) {                                               /* :1128-1138 */
			for (l = 0; l < NSYM; l++) newsv[(size_t)(m + 1) * NSYM + l] = c.sv[(size_t)k * NSYM + l];
			for (l = 0; l < i; l++) { p = c.order[l]; newstr[p][m] = c.strings[p][k - 1]; }
			string[m] = '-';
			newsv[(size_t)(m + 1) * NSYM + GAP]++;
			k--; m--;
		}
		if (!inplace) {                                               /* :1139-1153 */
			free(c.sv);
			for (j = 0; j < i; j++) { p = c.order[j]; free(c.strings[p]); }
			free(c.strings);
		}
		c.sv = newsv;
		c.strings = newstr;
		c.consensus = newcons;
		if (i > 1) {                                                  /* :1157 */
			rc = delete_gapped_columns(&c, i + 1, (i + 1) / 2);
			if (rc != ODP_OK) goto fail;
		}
	}

	for (s = 0; s < nseq; s++) { out[s] = c.strings[s]; c.strings[s] = NULL; }   /* :1160 */
	c.st.consensus = c.consensus;
	if (stats) *stats = c.st;
	rc = c.consensus;
fail:
	free(rowcodes);
	free(c.order); free(c.len); free(c.sv);
	free_matrix(&c);
	free_strings(c.strings, nseq);
	return rc;
}

/* ---- linear-space score of the pairwise case ------------------------------------------------ */

/*
 * One row of :990-1029 for i = 1 without the direction matrix.  With a = max(diag, up) the
 * left move only ever subtracts 1 per column, so new[k] = max over t <= k of (a[t] - (k - t)):
 * a running maximum of a[t] + t.  The first loop has no loop-carried dependency (the compiler
 * vectorises it), the second is one max per cell.
 */
#define PAIR_ROW_BODY                                                                       \
	int k, run;                                                                             \
	for (k = 1; k <= ncols; k++) {                                                          \
		const int sc = (colcodes[k - 1] == rc) ? S_MATCH : S_MISMATCH;                      \
		const int diag = prev[k - 1] + sc, up = prev[k] + S_INDEL;                          \
		tmp[k] = (diag > up ? diag : up) + k;                                               \
	}                                                                                       \
	run = cur0;                         /* a[0] + 0 = H[j][0] */                            \
	cur[0] = cur0;                                                                          \
	for (k = 1; k <= ncols; k++) {                                                          \
		run = tmp[k] > run ? tmp[k] : run;                                                  \
		cur[k] = run - k;                                                                   \
	}

static void pair_row_plain(int ncols, const signed char *colcodes, int rc, const int *prev, int *cur, int *tmp, int cur0)
{
	PAIR_ROW_BODY
}

__attribute__((target("avx2"), optimize("O3"))) static void pair_row_avx2(int ncols, const signed char *colcodes, int rc,
                                                                           const int *prev, int *cur, int *tmp, int cur0)
{
	PAIR_ROW_BODY
}

int odp_pair_score_linear(const char *const *texts, const int *textsizes, const int *rotations,
                          const int *starts, const int *ends, long long *score)
{
	ctx_t c;
	int s, j, k, col, row, nrows, ncols, rc = ODP_OK;
	int *a = NULL, *b = NULL, *tmp = NULL;
	signed char *colcodes = NULL;
	const int fast = __builtin_cpu_supports("avx2");
	if (!texts || !textsizes || !rotations || !starts || !ends || !score) return ODP_ERR_ARG;
	memset(&c, 0, sizeof(c));
	c.nseq = 2;
	c.texts = texts; c.textsizes = textsizes; c.rotations = rotations; c.starts = starts; c.ends = ends;
	for (s = 0; s < 2; s++)
		if (!texts[s] || starts[s] < 0 || ends[s] < starts[s] || ends[s] > textsizes[s]) return ODP_ERR_ARG;
	/* the shorter region seeds the profile = the columns (:290-307, first strict minimum) */
	col = (ends[1] - starts[1] < ends[0] - starts[0]) ? 1 : 0;
	row = 1 - col;
	ncols = ends[col] - starts[col];
	nrows = ends[row] - starts[row];
	a = (int *)malloc(((size_t)ncols + 1) * sizeof(int));
	b = (int *)malloc(((size_t)ncols + 1) * sizeof(int));
	tmp = (int *)malloc(((size_t)ncols + 1) * sizeof(int));
	colcodes = (signed char *)malloc((size_t)ncols + 1);
	if (!a || !b || !tmp || !colcodes) { rc = ODP_ERR_NOMEM; goto done; }
	for (k = 0; k < ncols; k++) {
		colcodes[k] = (signed char)char_code(char_at(&c, starts[col] + k, col));
		if (colcodes[k] < 0 || colcodes[k] > 3) { rc = ODP_ERR_ALPHABET; goto done; }
	}
	for (k = 0; k <= ncols; k++) a[k] = k * S_INDEL;                        /* :969-973, i = 1, no gaps in the seed */
	for (j = 1; j <= nrows; j++) {
		const int code = char_code(char_at(&c, starts[row] + j - 1, row));
		int *t;
		if (code < 0 || code > 3) { rc = ODP_ERR_ALPHABET; goto done; }
		if (fast) pair_row_avx2(ncols, colcodes, code, a, b, tmp, j * S_INDEL);   /* H[j][0], :967 */
		else pair_row_plain(ncols, colcodes, code, a, b, tmp, j * S_INDEL);
		t = a; a = b; b = t;
	}
	*score = a[ncols];
done:
	free(a); free(b); free(tmp); free(colcodes);
	return rc;
}

/* tools.c:259-281: per column gaps, conservation (all characters equal, gaps included) and the pair loop */
int odp_sp_stats(int nseq, const char *const *aligned, int *consensus, long long *gaps, int *conserved, long long *sp)
{
	size_t n, len;
	int i, j;
	if (nseq < 2 || !aligned || !aligned[0]) return ODP_ERR_ARG;
	len = strlen(aligned[0]);
	for (i = 1; i < nseq; i++)
		if (!aligned[i] || strlen(aligned[i]) != len) return ODP_ERR_ARG;             /* :248-254 */
	*consensus = (int)len; *gaps = 0; *conserved = 0; *sp = 0;
	for (n = 0; n < len; n++) {
		for (i = 0; i < nseq; i++)
			if (aligned[i][n] == '-') (*gaps)++;                                       /* :266 */
		for (i = 1; i < nseq; i++)
			if (aligned[i][n] != aligned[0][n]) break;
		if (i == nseq) (*conserved)++;                                                 /* :268-272 */
		for (i = 0; i <= nseq - 2; i++)
			for (j = i + 1; j <= nseq - 1; j++) {
				char x = aligned[i][n], y = aligned[j][n];
				if (x == '-' && y == '-') continue;
				if (x == y) (*sp)++; else (*sp)--;
			}
	}
	return ODP_OK;
}

/* tools.c:274-280: gap/gap scores nothing, equal +1, anything else -1 */
long long odp_sp_score(int nseq, const char *const *aligned)
{
	long long score = 0;
	size_t n, len;
	int i, j;
	if (nseq < 2 || !aligned || !aligned[0]) return 0;
	len = strlen(aligned[0]);
	for (n = 0; n < len; n++) {
		for (i = 0; i <= nseq - 2; i++) {
			for (j = i + 1; j <= nseq - 1; j++) {
				char a = aligned[i][n], b = aligned[j][n];
				if (a == '-' && b == '-') continue;
				if (a == b) score++; else score--;
			}
		}
	}
	return score;
}

unsigned odp_fnv1a(int nseq, const char *const *aligned)
{
	unsigned h = 0x811c9dc5u;
	int s;
	for (s = 0; s < nseq; s++) {
		const unsigned char *p = (const unsigned char *)aligned[s];
		if (!p) continue;
		for (; *p; p++) { h ^= *p; h *= 0x01000193u; }
	}
	return h;
}

void odp_free(void *p) { free(p); }
