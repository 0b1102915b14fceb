/*
 * csadp_hostpar.h -- the host side's fork-join helper: ONE persistent pool of threads per process
 * (CSADP_HOST_THREADS, default min(64, hardware threads)) shared by the per-task work around the device
 * passes (csadp_api.cpp) and the host stages around the DP (rotation finder, anchor stage).  Creating
 * threads per call cost ~2 ms per parallel region on the 256-thread hosts of the GPU boxes (measured in
 * round 2 for the per-task work; the host stages had kept their own short-lived threads until round 3).
 * Items are handed out one at a time through an atomic counter, so uneven items balance.  Regions started by different host
 * threads run side by side on the workers that are idle when each starts.
 */
#ifndef CSADP_HOSTPAR_H
#define CSADP_HOSTPAR_H

#include "csadp_config.h"
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace csadp {

class HostPool {
public:
	static HostPool &get()
	{
		static HostPool *pool = new HostPool;        /* never destroyed: workers may outlive static destructors */
		return *pool;
	}
	int size() const { return nthreads_; }
	/* run body() on up to `workers` - 1 pool threads and on the caller, return when every one has finished.  Only the threads
	 * that take part are woken (each waits on its own condition variable): a region of 16 on a pool of 64 does not pay
	 * for 48 wake-ups that go straight back to sleep.
	 * Regions of different callers run SIDE BY SIDE (round 5): a region takes the workers that are idle when it starts -- possibly
	 * none: then its items run on the caller alone -- and never waits for another region.  Until round 4 the pool served one region at
	 * a time, and the round groups of csadp_align_batch (a host thread each) spent 5-9 ms of a round of 64 tasks waiting for each
	 * other's regions (profiles/r05_profile_batch_probe.txt). */
	void run(const std::function<void()> &body, int workers)
	{
		Region region;
		region.body = &body;
		std::vector<int> mine;
		{
			std::lock_guard<std::mutex> lock(m_);
			const int want = std::min(workers, nthreads_) - 1;
			for (int w = 1; w < nthreads_ && (int)mine.size() < want; ++w)
				if (!job_[(size_t)w]) {
					job_[(size_t)w] = &region;
					mine.push_back(w);
				}
			region.pending = (int)mine.size();
		}
		for (int w : mine) cv_[(size_t)w].notify_one();
		body();
		std::unique_lock<std::mutex> lock(m_);
		cv_done_.wait(lock, [&] { return region.pending == 0; });
	}

private:
	struct Region {
		const std::function<void()> *body = nullptr;
		int pending = 0;
	};
	HostPool()
	{
		int t = config().host_threads > 0 ? config().host_threads : (int)std::thread::hardware_concurrency();
		nthreads_ = t < 1 ? 1 : (t > 64 ? 64 : t);
		job_.assign((size_t)nthreads_, nullptr);
		cv_ = std::vector<std::condition_variable>((size_t)nthreads_);
		for (int w = 1; w < nthreads_; ++w) std::thread([this, w] { loop(w); }).detach();
	}
	void loop(int w)
	{
		for (;;) {
			Region *region = nullptr;
			{
				std::unique_lock<std::mutex> lock(m_);
				cv_[(size_t)w].wait(lock, [&] { return job_[(size_t)w] != nullptr; });
				region = job_[(size_t)w];
			}
			(*region->body)();
			{
				std::lock_guard<std::mutex> lock(m_);
				job_[(size_t)w] = nullptr;                  /* idle again, before the region's owner may return */
				--region->pending;
			}
			cv_done_.notify_all();                          /* several owners may wait: each looks at its own region */
		}
	}
	int nthreads_ = 1;
	std::mutex m_;
	std::vector<std::condition_variable> cv_;
	std::condition_variable cv_done_;
	std::vector<Region *> job_;                             /* per worker: the region it serves, nullptr = idle */
};

inline bool &host_in_region()
{
	static thread_local bool inside = false;
	return inside;
}

/* fn(i) for i in [0, n) on at most `cap` threads of the pool (the caller included); n small -> inline.  A call
 * from inside a parallel region (an item that is itself parallel) runs its items on the calling thread: the workers
 * are for the outermost regions. */
template <class F>
void host_parallel_for(int n, F &&fn, int cap = 64)
{
	const int nthreads = std::min(HostPool::get().size(), cap);
	const int workers = (n < 2 || host_in_region()) ? 1 : std::min(nthreads, n);
	if (workers <= 1) {
		for (int i = 0; i < n; ++i) fn(i);
		return;
	}
	std::atomic<int> next(0);
	const std::function<void()> body = [&]() {
		host_in_region() = true;
		for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
		host_in_region() = false;
	};
	HostPool::get().run(body, workers);
}

}  // namespace csadp

#endif
