/*
 * csadp_hostpar.h -- the host side's fork-join helper: ONE persistent pool of threads per process
 * (CSADP_HOST_THREADS, default min(64, hardware threads)) shared by the per-task work around the device
 * passes (csadp_api.cpp) and the host stages around the DP (rotation finder, anchor stage).  Creating
 * threads per call cost ~2 ms per parallel region on the 256-thread hosts of the GPU boxes (measured in
 * round 2 for the per-task work; the host stages had kept their own short-lived threads until round 3).
 * Items are handed out one at a time through an atomic counter, so uneven items balance.
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
	/* run body() on `workers` - 1 pool threads and on the caller, return when every one has finished.  Only the threads
	 * that take part are woken (each waits on its own condition variable): a region of 16 on a pool of 64 does not pay
	 * for 48 wake-ups that go straight back to sleep. */
	void run(const std::function<void()> &body, int workers)
	{
		std::unique_lock<std::mutex> busy(run_mutex_);       /* one parallel region at a time */
		const int want = std::min(workers, nthreads_) - 1;
		{
			std::lock_guard<std::mutex> lock(m_);
			body_ = &body;
			done_ = 0;
			for (int w = 1; w <= want; ++w) ++go_[(size_t)w];
		}
		for (int w = 1; w <= want; ++w) cv_[(size_t)w].notify_one();
		body();
		std::unique_lock<std::mutex> lock(m_);
		cv_done_.wait(lock, [&] { return done_ == want; });
		body_ = nullptr;
	}

private:
	HostPool()
	{
		int t = config().host_threads > 0 ? config().host_threads : (int)std::thread::hardware_concurrency();
		nthreads_ = t < 1 ? 1 : (t > 64 ? 64 : t);
		go_.assign((size_t)nthreads_, 0);
		cv_ = std::vector<std::condition_variable>((size_t)nthreads_);
		for (int w = 1; w < nthreads_; ++w) std::thread([this, w] { loop(w); }).detach();
	}
	void loop(int w)
	{
		unsigned long long seen = 0;
		for (;;) {
			const std::function<void()> *body = nullptr;
			{
				std::unique_lock<std::mutex> lock(m_);
				cv_[(size_t)w].wait(lock, [&] { return go_[(size_t)w] != seen; });
				seen = go_[(size_t)w];
				body = body_;
			}
			(*body)();
			{
				std::lock_guard<std::mutex> lock(m_);
				++done_;
			}
			cv_done_.notify_one();
		}
	}
	int nthreads_ = 1;
	std::mutex run_mutex_, m_;
	std::vector<std::condition_variable> cv_;
	std::condition_variable cv_done_;
	std::vector<unsigned long long> go_;
	const std::function<void()> *body_ = nullptr;
	int done_ = 0;
};

inline bool &host_in_region()
{
	static thread_local bool inside = false;
	return inside;
}

/* fn(i) for i in [0, n) on at most `cap` threads of the pool (the caller included); n small -> inline.  A call
 * from inside a parallel region (an item that is itself parallel) runs its items on the calling thread: the pool
 * serves one region at a time. */
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
