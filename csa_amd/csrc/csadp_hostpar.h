/*
 * csadp_hostpar.h -- tiny fork-join helper for the host stages around the DP (rotation finder,
 * anchor stage): independent per-sequence work spread over short-lived threads
 * (CSADP_HOST_THREADS, default min(64, hardware threads)).
 */
#ifndef CSADP_HOSTPAR_H
#define CSADP_HOSTPAR_H

#include <stdlib.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace csadp {

inline int host_stage_threads()
{
	const char *e = getenv("CSADP_HOST_THREADS");
	const int n = e && *e ? atoi(e) : (int)std::thread::hardware_concurrency();
	return std::max(1, std::min(n, 64));
}

template <class F>
void host_parallel_for(int n, F &&fn)
{
	const int nt = std::min(host_stage_threads(), n);
	if (nt <= 1) {
		for (int i = 0; i < n; ++i) fn(i);
		return;
	}
	std::vector<std::thread> pool;
	pool.reserve((size_t)nt);
	for (int t = 0; t < nt; ++t)
		pool.emplace_back([&fn, t, n, nt]() {
			for (int i = t; i < n; i += nt) fn(i);
		});
	for (auto &th : pool) th.join();
}

}  // namespace csadp

#endif
