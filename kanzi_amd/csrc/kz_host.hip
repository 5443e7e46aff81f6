// kz_host.hip -- host-side helpers shared by the batched calls and the stream code: a persistent pool of host threads for the
// per-block CPU work that surrounds the GPU path (the TEXT / UTF stages, bit-granular payload copies, staging copies).  The
// reference runs that work on its task pool (K/io/CompressedOutputStream.java:541-566, jobs <= 64); here the threads live for
// the life of the process, so their thread_local scratch (dictionaries, alias maps, stage buffers) is reused from call to call
// and several callers (a batch's host stages, the stream writer's assembly of the previous batch) share one set of threads.
#include "kz_internal.h"
#include <sched.h>
#include <stdlib.h>
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
// CPUs this process can actually use: its affinity mask, cut down to the cgroup's CPU bandwidth quota when there is one.  A
// container may see every CPU of the host and still be allowed only N CPU-seconds per second (cgroup v2 cpu.max "quota period",
// v1 cpu.cfs_quota_us / cpu.cfs_period_us); running more busy threads than that makes the kernel freeze ALL threads of the group
// for the rest of each period -- including the one that drives the GPU (measured on the MI355X box: 256 CPUs visible, quota 16:
// 128 host-stage threads stretched every GPU call of the pipeline from 0.21 s to 0.55 s).
int quota_cpus() {
  long long quota = -1, period = 100000;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0};
    if (fscanf(f, "%63s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    if (fscanf(g, "%lld", &quota) != 1) quota = -1;
    fclose(g);
    if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
  }
  if (quota <= 0 || period <= 0) return 1 << 30;
  return (int)std::max<long long>(1, (quota + period - 1) / period);
}
std::atomic<int> g_share{1};      // processes on this host that share the quota (kz_host_share)
int usable_cpus() {
  static const int quota0 = quota_cpus();
  const int sh = std::max(1, g_share.load(std::memory_order_relaxed));
  const int quota = quota0 >= (1 << 30) ? quota0 : std::max(1, (quota0 + sh - 1) / sh);
  cpu_set_t set;
  CPU_ZERO(&set);
  int c = 0;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) c = CPU_COUNT(&set);
  if (c <= 0) { const int hw = (int)std::thread::hardware_concurrency(); c = hw > 0 ? hw : 1; }
  return std::min(c, quota);
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
extern "C" int32_t kz_host_cpus(void) { return usable_cpus(); }
extern "C" int32_t kz_host_share(int32_t ranksOnHost) { g_share.store(ranksOnHost > 1 ? ranksOnHost : 1); return usable_cpus(); }

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

// grow-only staging buffer kept by the context: pinned host memory or HBM outside the arena
int kz_stage_reserve(kz_ctx* ctx, kz_ctx::Stage& s, size_t need, bool pinned) {
  if (need <= s.cap) return 0;
  if (s.p) { if (pinned) hipHostFree(s.p); else hipFree(s.p); s.p = nullptr; s.cap = 0; }
  need = kz_align(need + (need >> 4), 1 << 20);
  if (pinned) KZ_HIP(hipHostMalloc((void**)&s.p, need, hipHostMallocDefault)); else KZ_HIP(hipMalloc((void**)&s.p, need));
  s.cap = need;
  return 0;
}
