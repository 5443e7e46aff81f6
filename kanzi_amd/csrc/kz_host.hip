// kz_host.hip -- host-side helpers shared by the batched calls and the stream code: a persistent pool of host threads for the
// per-block CPU work that surrounds the GPU path (the TEXT / UTF stages, bit-granular payload copies, staging copies).  The
// reference runs that work on its task pool (K/io/CompressedOutputStream.java:541-566, jobs <= 64); here the threads live for
// the life of the process, so their thread_local scratch (dictionaries, alias maps, stage buffers) is reused from call to call
// and several callers (a batch's host stages, the stream writer's assembly of the previous batch) share one set of threads.
#include "kz_internal.h"
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {
struct PfJob {
  int n = 0, maxWorkers = 0;
  void (*fn)(int, void*) = nullptr;
  void* arg = nullptr;
  std::atomic<int> next{0};
  int workers = 0;                 // pool threads inside the job (under the pool mutex)
  int done = 0;                    // items finished (under the pool mutex)
};
struct Pool {
  std::mutex mu;
  std::condition_variable wake, finished;
  std::vector<PfJob*> active;
  int threads = 0, cap = 1;
};
int usable_cpus() {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) return c; }
  const int hw = (int)std::thread::hardware_concurrency();
  return hw > 0 ? hw : 1;
}
Pool& pool() {
  static Pool* p = new Pool();     // never destroyed: its detached threads may outlive static destructors
  return *p;
}
void worker(Pool* P) {
  std::unique_lock<std::mutex> lk(P->mu);
  for (;;) {
    PfJob* j = nullptr;
    for (PfJob* c : P->active)
      if (c->workers < c->maxWorkers && c->next.load(std::memory_order_relaxed) < c->n) { j = c; break; }
    if (!j) { P->wake.wait(lk); continue; }
    j->workers++;
    lk.unlock();
    int did = 0;
    for (;;) { const int i = j->next.fetch_add(1); if (i >= j->n) break; j->fn(i, j->arg); did++; }
    lk.lock();
    j->done += did;
    j->workers--;
    if (j->workers == 0 && j->done >= j->n) P->finished.notify_all();
  }
}
}  // namespace

// the CPUs this process may run on (N ranks on one host are pinned to their GPU's NUMA node: kz_pin_to_device_numa)
int kz_usable_cpus() { return usable_cpus(); }

// fn(i, arg) for i in [0, n) on up to maxThreads threads (the caller is one of them); returns when all are done.  May be called
// from several host threads at once.
void kz_parallel_for(int n, int maxThreads, void (*fn)(int, void*), void* arg) {
  if (n <= 0) return;
  const int T = std::max(1, std::min(std::min(n, maxThreads), usable_cpus()));
  if (T == 1) { for (int i = 0; i < n; i++) fn(i, arg); return; }
  Pool& P = pool();
  PfJob job;
  job.n = n; job.maxWorkers = T - 1; job.fn = fn; job.arg = arg;
  {
    std::lock_guard<std::mutex> g(P.mu);
    // grow the pool up to the widest request seen (bounded by the CPUs this process may use)
    const int want = std::min(std::max(P.threads, T - 1), std::max(1, usable_cpus() - 1));
    while (P.threads < want) { std::thread(worker, &P).detach(); P.threads++; }
    P.active.push_back(&job);
  }
  P.wake.notify_all();
  int did = 0;
  for (;;) { const int i = job.next.fetch_add(1); if (i >= n) break; fn(i, arg); did++; }
  std::unique_lock<std::mutex> lk(P.mu);
  job.done += did;
  P.finished.wait(lk, [&] { return job.workers == 0 && job.done >= n; });
  P.active.erase(std::find(P.active.begin(), P.active.end(), &job));
}
