/*
 * csadp_engine.cpp -- device runtime of libcsadp (see csadp_engine.h).
 */
#include "csadp_engine.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "csadp_kernels.h"

namespace csadp {

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int env_int(const char *name, int dflt)
{
	const char *v = getenv(name);
	if (!v || !*v) return dflt;
	return atoi(v);
}

#define HIP_TRY(expr)                                                                         \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess) {                                                               \
			fprintf(stderr, "csadp: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_),   \
			        __FILE__, __LINE__);                                                      \
			return CSADP_ERR_HIP;                                                             \
		}                                                                                     \
	} while (0)

}  // namespace

Engine &Engine::get()
{
	static Engine e;
	return e;
}

int Engine::init(const csadp_config *cfg)
{
	if (ready_) return CSADP_OK;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		fprintf(stderr, "csadp: no HIP device available (this library has no CPU fallback)\n");
		return CSADP_ERR_NO_DEVICE;
	}
	int dev = cfg ? cfg->device : -1;
	if (dev < 0) dev = env_int("LOCAL_RANK", 0);
	if (dev >= count) dev %= count;
	HIP_TRY(hipSetDevice(dev));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, dev));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		fprintf(stderr, "csadp: device %d is %s; the kernels are built for gfx950 only\n", dev, prop.gcnArchName);
		return CSADP_ERR_NO_DEVICE;
	}
	device_ = dev;
	snprintf(name_, sizeof(name_), "%s (%s)", prop.name, prop.gcnArchName);
	cus_ = prop.multiProcessorCount;
	HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
	C_ = env_int("CSADP_COLS_PER_LANE", 16);
	TR_ = (cfg && cfg->tile_rows > 0) ? cfg->tile_rows : env_int("CSADP_TILE_ROWS", 128);
	if (C_ != 16 && C_ != 32) return CSADP_ERR_ARG;
	if (TR_ != 64 && TR_ != 128 && TR_ != 256) return CSADP_ERR_ARG;
	verbose_ = cfg && cfg->verbose;
	ready_ = true;
	return CSADP_OK;
}

void Engine::shutdown()
{
	if (!ready_) return;
	(void)hipStreamSynchronize(stream_);
	(void)hipStreamDestroy(stream_);
	stream_ = nullptr;
	ready_ = false;
}

/* ---- FillBatch ----------------------------------------------------------------------- */

FillBatch::~FillBatch()
{
	if (arena_) (void)hipFree(arena_);
	if (h_in_) (void)hipHostFree(h_in_);
	if (h_res_) (void)hipHostFree(h_res_);
	for (auto &e : ev_)
		if (e) (void)hipEventDestroy(e);
}

void FillBatch::clear()
{
	jobs_.clear();
	extra_.clear();
	tiles_.clear();
	diag_off_.clear();
	laid_out_ = false;
	ran_ = false;
}

int FillBatch::add(int nrows, int ncols, int nprev, int left_i)
{
	FillJob j;
	memset(&j, 0, sizeof(j));
	j.nrows = nrows;
	j.ncols = ncols;
	j.upc = 8 * nprev + 2;
	j.leftmul = 4 * (left_i + nprev);
	jobs_.push_back(j);
	laid_out_ = false;
	return (int)jobs_.size() - 1;
}

int FillBatch::layout()
{
	Engine &E = Engine::get();
	if (!E.ready()) return CSADP_ERR_NO_DEVICE;
	const int C = E.C(), TR = E.TR(), W = C / 16;
	const int nj = (int)jobs_.size();
	extra_.assign(nj, Extra());
	cells_ = dir_bytes_ = border_bytes_ = 0;

	/* geometry + tile schedule */
	int ndiag = 0;
	std::vector<int> diag_count;
	for (int j = 0; j < nj; ++j) {
		FillJob &J = jobs_[j];
		if (J.nrows <= 0 || J.ncols <= 0) return CSADP_ERR_ARG;
		const long long nprev = (J.upc - 2) / 8;
		if (nprev * (2LL * J.nrows + J.ncols) * 4 + 64 >= (1LL << 31)) return CSADP_ERR_RANGE;
		const int lanes = (J.ncols + C - 1) / C;
		J.nstrips = (lanes + kLanes - 1) / kLanes;
		extra_[j].ncols_pad = J.nstrips * kLanes * C;
		J.steps_pad = (int)align_up((size_t)J.nrows + (size_t)kLanes * J.nstrips, TR);
		J.hpitch = J.steps_pad + 64;
		J.padl = kLanes * J.nstrips + 64;
		J.lf = (J.ncols - 1) / C;
		J.tf = J.nrows - 1 + J.lf;
		for (int s = 0; s < J.nstrips; ++s) {
			const int a0 = (kLanes * s) / TR;
			const int a1 = (J.nrows - 1 + kLanes * s + 63) / TR;
			ndiag = std::max(ndiag, a1 + s + 1);
			if ((int)diag_count.size() < a1 + s + 1) diag_count.resize(a1 + s + 1, 0);
			for (int a = a0; a <= a1; ++a) diag_count[a + s]++;
		}
		cells_ += (long long)J.nrows * J.ncols;
		dir_bytes_ += (long long)J.nrows * (long long)lanes * 4 * W;
	}
	diag_off_.assign((size_t)ndiag + 1, 0);
	for (int d = 0; d < ndiag; ++d) diag_off_[d + 1] = diag_off_[d] + (size_t)diag_count[d];
	tiles_.assign(diag_off_[ndiag], TileRef());
	{
		std::vector<size_t> cur(diag_off_.begin(), diag_off_.end() - 1);
		for (int j = 0; j < nj; ++j) {
			const FillJob &J = jobs_[j];
			for (int s = 0; s < J.nstrips; ++s) {
				const int a0 = (kLanes * s) / TR;
				const int a1 = (J.nrows - 1 + kLanes * s + 63) / TR;
				for (int a = a0; a <= a1; ++a) {
					TileRef t;
					t.job = j;
					t.a = a;
					t.s = s;
					t.first = (a == a0);
					tiles_[cur[a + s]++] = t;
				}
				/* hand-off ints written once and read once, lane state saved + restored per tile */
				border_bytes_ += 2LL * 4 * (a1 - a0 + 1) * TR + 2LL * 4 * (a1 - a0 + 1) * (C + 2) * kLanes;
			}
		}
	}

	/* arena offsets */
	size_t off = 0;
	jobs_off_ = off;
	off = align_up(off + (size_t)nj * sizeof(FillJob), 256);
	tiles_off_ = off;
	off = align_up(off + tiles_.size() * sizeof(TileRef), 256);
	for (int j = 0; j < nj; ++j) {
		FillJob &J = jobs_[j];
		Extra &X = extra_[j];
		X.in_coltab = J.coltab = off;
		off = align_up(off + (size_t)X.ncols_pad * 4, 256);
		X.in_rowshift = J.rowshift = off;
		off = align_up(off + (size_t)J.padl + J.steps_pad + 64, 256);
		X.in_top = J.top = off;
		off = align_up(off + ((size_t)X.ncols_pad + 1) * 4, 256);
	}
	in_bytes_ = off;
	res_off_ = off;
	for (int j = 0; j < nj; ++j) {
		FillJob &J = jobs_[j];
		Extra &X = extra_[j];
		J.summary = off;
		X.res_summary = off - res_off_;
		off += 64;
		J.ops = off;
		X.res_ops = off - res_off_;
		off = align_up(off + (size_t)J.nrows + J.ncols + 64, 256);
	}
	res_bytes_ = off - res_off_;
	for (int j = 0; j < nj; ++j) {
		FillJob &J = jobs_[j];
		J.final_row = off;
		off = align_up(off + (size_t)C * 4, 256);
		J.state = off;
		off = align_up(off + (size_t)J.nstrips * (C + 2) * kLanes * 4, 256);
		J.handoff = off;
		off = align_up(off + (size_t)J.nstrips * J.hpitch * 4, 256);
		J.dirs = off;
		off = align_up(off + (size_t)J.nstrips * J.steps_pad * W * kLanes * 4, 256);
	}
	total_bytes_ = off;

	if (total_bytes_ > arena_cap_) {
		if (arena_) { (void)hipFree(arena_); arena_ = nullptr; arena_cap_ = 0; }
		size_t free_b = 0, total_b = 0;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		if (total_bytes_ + (256u << 20) > free_b) {
			fprintf(stderr, "csadp: batch needs %.1f GiB of HBM, %.1f GiB free\n",
			        total_bytes_ / 1073741824.0, free_b / 1073741824.0);
			return CSADP_ERR_RANGE;
		}
		HIP_TRY(hipMalloc((void **)&arena_, total_bytes_));
		arena_cap_ = total_bytes_;
	}
	if (in_bytes_ > h_in_cap_) {
		if (h_in_) (void)hipHostFree(h_in_);
		h_in_ = nullptr;
		HIP_TRY(hipHostMalloc((void **)&h_in_, in_bytes_, hipHostMallocDefault));
		h_in_cap_ = in_bytes_;
	}
	if (res_bytes_ > h_res_cap_) {
		if (h_res_) (void)hipHostFree(h_res_);
		h_res_ = nullptr;
		HIP_TRY(hipHostMalloc((void **)&h_res_, res_bytes_, hipHostMallocDefault));
		h_res_cap_ = res_bytes_;
	}
	for (auto &e : ev_)
		if (!e) HIP_TRY(hipEventCreate(&e));
	memset(h_in_, 0, in_bytes_);
	memcpy(h_in_ + jobs_off_, jobs_.data(), (size_t)nj * sizeof(FillJob));
	memcpy(h_in_ + tiles_off_, tiles_.data(), tiles_.size() * sizeof(TileRef));
	laid_out_ = true;
	ran_ = false;
	return CSADP_OK;
}

uint32_t *FillBatch::coltab(int j) { return reinterpret_cast<uint32_t *>(h_in_ + extra_[j].in_coltab); }
uint8_t *FillBatch::rowshift(int j) { return h_in_ + extra_[j].in_rowshift + jobs_[j].padl; }
int32_t *FillBatch::top(int j) { return reinterpret_cast<int32_t *>(h_in_ + extra_[j].in_top); }
int FillBatch::ncols_pad(int j) const { return extra_[j].ncols_pad; }

int FillBatch::upload()
{
	if (!laid_out_) return CSADP_ERR_STATE;
	HIP_TRY(hipMemcpyAsync(arena_, h_in_, in_bytes_, hipMemcpyHostToDevice, Engine::get().stream()));
	return CSADP_OK;
}

int FillBatch::run()
{
	if (!laid_out_) return CSADP_ERR_STATE;
	Engine &E = Engine::get();
	hipStream_t st = E.stream();
	const FillJob *djobs = reinterpret_cast<const FillJob *>(arena_ + jobs_off_);
	const TileRef *dtiles = reinterpret_cast<const TileRef *>(arena_ + tiles_off_);
	HIP_TRY(hipEventRecord(ev_[0], st));
	const int ndiag = (int)diag_off_.size() - 1;
	for (int d = 0; d < ndiag; ++d) {
		const int cnt = (int)(diag_off_[d + 1] - diag_off_[d]);
		HIP_TRY(launch_fill(E.C(), E.TR(), arena_, djobs, dtiles + diag_off_[d], cnt, st));
	}
	HIP_TRY(hipEventRecord(ev_[1], st));
	HIP_TRY(launch_traceback(E.C(), arena_, djobs, (int)jobs_.size(), st));
	HIP_TRY(hipEventRecord(ev_[2], st));
	ran_ = true;
	return CSADP_OK;
}

int FillBatch::sync()
{
	HIP_TRY(hipStreamSynchronize(Engine::get().stream()));
	return CSADP_OK;
}

int FillBatch::download()
{
	if (!ran_) return CSADP_ERR_STATE;
	hipStream_t st = Engine::get().stream();
	HIP_TRY(hipMemcpyAsync(h_res_, arena_ + res_off_, res_bytes_, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	return CSADP_OK;
}

const uint8_t *FillBatch::ops(int j) const { return h_res_ + extra_[j].res_ops; }
const int32_t *FillBatch::summary(int j) const { return reinterpret_cast<const int32_t *>(h_res_ + extra_[j].res_summary); }

int FillBatch::timing(csadp_timing *t)
{
	if (!ran_) return CSADP_ERR_STATE;
	memset(t, 0, sizeof(*t));
	HIP_TRY(hipEventSynchronize(ev_[2]));
	HIP_TRY(hipEventElapsedTime(&t->fill_ms, ev_[0], ev_[1]));
	HIP_TRY(hipEventElapsedTime(&t->traceback_ms, ev_[1], ev_[2]));
	HIP_TRY(hipEventElapsedTime(&t->total_ms, ev_[0], ev_[2]));
	t->cells = cells_;
	t->fill_launches = (int)diag_off_.size() - 1;
	t->fill_tiles = (long long)tiles_.size();
	t->dir_bytes = dir_bytes_;
	t->border_bytes = border_bytes_;
	return CSADP_OK;
}

}  // namespace csadp
