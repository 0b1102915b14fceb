/*
 * csadp_hostutil.cpp -- host helpers of the C-ABI: work partitioning over GPUs and the
 * FASTA reader (wire format of the reference's loader, csamsa.c:433-519).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <string>
#include <vector>

#include <mutex>
#include "csadp.h"
#include "csadp_config.h"
#include "csadp_debug.h"

extern "C" {

/* Longest-processing-time-first: sort by cost descending, give each task to the least
 * loaded part (ties: lowest part index; equal costs keep task order) -- SURVEY.md 8(e). */
int csadp_partition_lpt(const long long *cost, int n, int nparts, int *assign, long long *maxload)
{
	if (n < 0 || nparts <= 0 || (n > 0 && (!cost || !assign))) return CSADP_ERR_ARG;
	std::vector<int> idx((size_t)n);
	std::iota(idx.begin(), idx.end(), 0);
	std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cost[a] > cost[b]; });
	std::vector<long long> load((size_t)nparts, 0);
	for (int t : idx) {
		if (cost[t] < 0) return CSADP_ERR_ARG;
		const int p = (int)(std::min_element(load.begin(), load.end()) - load.begin());
		assign[t] = p;
		load[(size_t)p] += cost[t];
	}
	if (maxload) *maxload = *std::max_element(load.begin(), load.end());
	return CSADP_OK;
}

/* FNV-1a-32 over strings[0], strings[1], ...: the digest of SURVEY.md 8(c)'s golden values and of the
 * fixed-size result records ranks exchange (bench.py, csa_amd/dist.py) */
unsigned csadp_fnv1a(const char *const *strings, int n)
{
	unsigned h = 0x811c9dc5u;
	if (!strings) return h;
	for (int s = 0; s < n; ++s) {
		if (!strings[s]) continue;
		for (const unsigned char *p = (const unsigned char *)strings[s]; *p; ++p) {
			h ^= *p;
			h *= 0x01000193u;
		}
	}
	return h;
}

/*
 * csamsa.c:433-519.  Records start at '>', the description runs to end of line; sequence
 * bytes: \n \r NUL '-' ' ' skipped; ACGT and the IUPAC codes RYSWKMDHBVN kept (lower case
 * upper-cased); any other byte drops the whole record ("INVALID_CHARS"); empty records are
 * dropped; at most 64 records are kept; fewer than 2 valid records is an error.
 */
int csadp_load_fasta(const char *path, char ***texts, char ***descs, int **sizes, int *nseq)
{
	if (!path || !texts || !descs || !sizes || !nseq) return CSADP_ERR_ARG;
	FILE *f = fopen(path, "rb");
	if (!f) return CSADP_ERR_ARG;
	std::string data;
	char buf[1 << 16];
	size_t got;
	while ((got = fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, got);
	fclose(f);

	std::vector<std::string> seqs, names;
	size_t p = data.find('>');
	while (p != std::string::npos && (int)seqs.size() < CSADP_MAX_SEQS) {
		size_t eol = p + 1;
		while (eol < data.size() && data[eol] != '\n' && data[eol] != '\r') ++eol;
		std::string name = data.substr(p + 1, eol - (p + 1));
		size_t next = data.find('>', eol);
		const size_t end = (next == std::string::npos) ? data.size() : next;
		std::string seq;
		bool valid = true;
		for (size_t q = eol; q < end; ++q) {
			unsigned char c = (unsigned char)data[q];
			if (c == '\n' || c == '\r' || c == '\0' || c == '-' || c == ' ') continue;
			if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
			if (strchr("ACGTRYSWKMDHBVN", c) != NULL) seq.push_back((char)c);
			else { valid = false; break; }
		}
		if (valid && !seq.empty()) {
			seqs.push_back(seq);
			names.push_back(name);
		}
		p = next;
	}
	if (seqs.size() < 2) return CSADP_ERR_ARG;
	const int n = (int)seqs.size();
	char **t = (char **)calloc((size_t)n, sizeof(char *));
	char **d = (char **)calloc((size_t)n, sizeof(char *));
	int *z = (int *)calloc((size_t)n, sizeof(int));
	if (!t || !d || !z) { free(t); free(d); free(z); return CSADP_ERR_NOMEM; }
	for (int i = 0; i < n; ++i) {
		t[i] = strdup(seqs[(size_t)i].c_str());
		d[i] = strdup(names[(size_t)i].c_str());
		z[i] = (int)seqs[(size_t)i].size();
		if (!t[i] || !d[i]) { csadp_free_fasta(t, d, z, n); return CSADP_ERR_NOMEM; }
	}
	*texts = t;
	*descs = d;
	*sizes = z;
	*nseq = n;
	return CSADP_OK;
}

/* saveRotatedSequences, csamsa.c:416-431 */
int csadp_write_rotated_fasta(const char *path, const char *const *descs, const char *const *texts,
                              const int *sizes, const int *rotations, int nseq)
{
	if (!path || !descs || !texts || !sizes || !rotations || nseq < 1) return CSADP_ERR_ARG;
	for (int i = 0; i < nseq; ++i)
		if (rotations[i] < 0 || rotations[i] > sizes[i]) return CSADP_ERR_ARG;
	FILE *f = fopen(path, "wb");
	if (!f) return CSADP_ERR_ARG;
	for (int i = 0; i < nseq; ++i) {
		fprintf(f, ">%s @ %d\n", descs[i], rotations[i]);
		fwrite(texts[i] + rotations[i], 1, (size_t)(sizes[i] - rotations[i]), f);
		fwrite(texts[i], 1, (size_t)rotations[i], f);
		fputc('\n', f);
	}
	return fclose(f) == 0 ? CSADP_OK : CSADP_ERR_ARG;
}

int csadp_read_rotations(const char *path, int *rotations, int nmax, int *nread)
{
	if (!path || !rotations || !nread || nmax < 0) return CSADP_ERR_ARG;
	FILE *f = fopen(path, "rb");
	if (!f) return CSADP_ERR_ARG;
	std::string line;
	int n = 0, c;
	bool header = false, any = false;
	while ((c = fgetc(f)) != EOF) {
		if (!any && c != '>') continue;              /* only a '>' opens a header (first byte of a line) */
		if (c == '>' && !header) { header = true; any = true; line.clear(); continue; }
		if (header) {
			if (c == '\n' || c == '\r') {
				header = false;
				const size_t at = line.rfind(" @ ");
				if (at == std::string::npos) { fclose(f); return CSADP_ERR_ARG; }
				if (n < nmax) rotations[n] = atoi(line.c_str() + at + 3);
				++n;
			} else {
				line.push_back((char)c);
			}
		}
	}
	fclose(f);
	*nread = n < nmax ? n : nmax;
	return n > 0 ? CSADP_OK : CSADP_ERR_ARG;
}

void csadp_free_fasta(char **texts, char **descs, int *sizes, int nseq)
{
	for (int i = 0; i < nseq; ++i) {
		if (texts) free(texts[i]);
		if (descs) free(descs[i]);
	}
	free(texts);
	free(descs);
	free(sizes);
}

}  // extern "C"

/* ---- environment switches: read once (csadp_config.h) ------------------------------------------------------------ */
namespace csadp {
namespace {
int env_int(const char *name, int dflt)
{
	const char *v = getenv(name);
	return v && *v ? atoi(v) : dflt;
}
Config read_config()
{
	Config c;
	c.bits = env_int("CSADP_BITS", 1) != 0;
	c.device_io = env_int("CSADP_DEVICE_IO", 1) != 0;
	c.bits_words = env_int("CSADP_BITS_WORDS", -1);
	c.bits_chunk = env_int("CSADP_BITS_CHUNK", 0);
	c.bits_group = env_int("CSADP_BITS_GROUP", -1);
	c.bits_streams = env_int("CSADP_BITS_STREAMS", -1);
	c.bits_lds_pad = env_int("CSADP_BITS_LDS_PAD", -1);
	c.lone_shape = env_int("CSADP_LONE_SHAPE", 1) != 0;
	c.stream_rotate = env_int("CSADP_STREAM_ROTATE", -1);
	c.bits_pack = env_int("CSADP_BITS_PACK", 1);
	c.cells_order = env_int("CSADP_CELLS_ORDER", -1);
	c.cells_fetch_wgs = env_int("CSADP_CELLS_FETCH", 256);
	c.cells_fetch_forced = getenv("CSADP_CELLS_FETCH") != nullptr;
	c.lone_cells = env_int("CSADP_LONE_CELLS", 1) != 0;
	c.slots = env_int("CSADP_SLOTS", 4);
	c.tb_band_min = env_int("CSADP_TB_BAND_MIN", 512);
	c.tb_band_forced = getenv("CSADP_TB_BAND_MIN") != nullptr;
	c.tb_corridor = env_int("CSADP_TB_CORRIDOR", 3);
	c.tb_corridor_forced = getenv("CSADP_TB_CORRIDOR") != nullptr;
	c.pull_uploads = env_int("CSADP_PULL_UPLOADS", 1) != 0;
	c.round_groups = env_int("CSADP_ROUND_GROUPS", 2);
	c.round_groups_forced = getenv("CSADP_ROUND_GROUPS") != nullptr;
	c.refine_speculate = env_int("CSADP_REFINE_SPECULATE", 0);
	c.host_threads = env_int("CSADP_HOST_THREADS", 0);
	c.trace_host = getenv("CSADP_TRACE_HOST") != nullptr;
	c.share_device = env_int("CSADP_SHARE_DEVICE", 0) != 0;
	c.local_rank = env_int("LOCAL_RANK", 0);
	c.test_force_abort = env_int("CSADP_TEST_FORCE_ABORT", 0) != 0;
	c.test_range_log2 = std::max(8, std::min(31, env_int("CSADP_TEST_RANGE_LOG2", 31)));
	c.test_hbm_limit_mb = std::max(0, env_int("CSADP_TEST_HBM_LIMIT_MB", 0));
	c.test_slow_publisher = std::max(0, std::min(255, env_int("CSADP_TEST_SLOW_PUBLISHER", 0)));
	return c;
}
std::mutex g_config_mutex;
Config *g_config = nullptr;        /* published once; a reload publishes a new one and leaks the old (a test seam: a handful per process) */
}  // namespace

const Config &config()
{
	Config *c = __atomic_load_n(&g_config, __ATOMIC_ACQUIRE);
	if (c) return *c;
	std::lock_guard<std::mutex> lock(g_config_mutex);
	if (!g_config) __atomic_store_n(&g_config, new Config(read_config()), __ATOMIC_RELEASE);
	return *g_config;
}

void reload_config()
{
	std::lock_guard<std::mutex> lock(g_config_mutex);
	__atomic_store_n(&g_config, new Config(read_config()), __ATOMIC_RELEASE);
}
}  // namespace csadp

extern "C" void csadp_debug_reload_config(void) { csadp::reload_config(); }
