// Micro-benchmark behind tools/pmc_traffic.py's per-access-class FETCH_SIZE / WRITE_SIZE factors and behind the inverse BWT's
// ceiling: known byte counts in the access patterns the pipeline really has.
//   stream16  : coalesced 16 B / lane streaming read of the whole array          (the guide's calibrated case: FETCH_SIZE x 2)
//   gather4   : independent random 4-byte loads, one line each                    (k_lz_fwd's hash probes, rank stores' read side)
//   chase4    : dependent random 4-byte loads, every lane the same chain length   (k_bwti_walk1 with perfect lane occupancy)
//   chase4geo : the same with geometric chain lengths per lane (mean = steps)      (k_bwti_walk1 as it is: lanes idle behind the longest)
//   scatter4  : random 4-byte stores                                              (rank / suffix-array stores of the suffix sort)
//   stream16w : coalesced 16 B / lane streaming write
//   chaseblk  : ONE lane per wave follows a dependent chain inside its own 2 MiB region (waves = blocks): the latency of a
//               serial per-block coder that probes a per-block hash map in HBM (what a TEXT inverse on the GPU would do per word)
// Usage: ubench_gather <kernel> [array MiB = 4096] [steps per lane = 256] [waves = 65536]
// Prints one line: kernel, elapsed ms, accesses, G accesses / s, useful GB/s.  Run it under
//   rocprofv3 --pmc FETCH_SIZE -- ...   and   rocprofv3 --pmc WRITE_SIZE -- ...
// and divide the counters by the access count printed here (tools/pmc_calibrate.sh does).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;
__device__ __forceinline__ u64 mix(u64 z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
__global__ void k_init(u32* a, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) a[i] = (u32)(mix(i) % n);
}
__global__ void k_stream16(const uint4* __restrict__ a, u64 n16, u32* sink) {
  u32 acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_stream16w(uint4* __restrict__ a, u64 n16) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) a[i] = make_uint4((u32)i, 1, 2, 3);
}
__global__ void k_gather4(const u32* __restrict__ a, u64 n, int steps, u32* sink) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 acc = 0;
  for (int j = 0; j < steps; j += 4) {
    u32 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(&a[mix(t * 1000003ULL + (u64)(j + q)) % n]);
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_chase4(const u32* __restrict__ a, u64 n, int steps, int geo, u32* sink, unsigned long long* total) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 p = (u32)(mix(t) % n);
  int len = steps;
  if (geo) {                                  // geometric with mean `steps`: -steps * ln(u)
    const double u = ((double)(mix(t ^ 0xABCDEFULL) >> 11) + 1.0) / 9007199254740993.0;
    len = (int)(-(double)steps * log(u)) + 1;
  }
  for (int j = 0; j < len; j++) p = __builtin_nontemporal_load(&a[p]);
  if (p == 0xFFFFFFFFu) sink[0] = p;
  if (geo) {
    unsigned long long s = (unsigned long long)len;
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(total, s);
  }
}
__global__ void k_chaseblk(const u32* __restrict__ a, u64 n, int steps, u32* sink) {
  const u64 region = 512 * 1024;                                // 2 MiB of u32 per wave
  const u64 base = ((u64)blockIdx.x * region) % (n - region);
  u32 p = (u32)(mix(blockIdx.x) % region);
  if (threadIdx.x == 0) {
    for (int j = 0; j < steps; j++) p = (__builtin_nontemporal_load(&a[base + p]) + (u32)j * 2654435761u) % (u32)region;   // (+ j: a plain x -> f(x) walk falls into a cycle of a few hundred lines)
    sink[blockIdx.x & 15] = p;                                  // (keeps the chain alive)
  }
}
__global__ void k_scatter4(u32* __restrict__ a, u64 n, int steps) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < steps; j++) a[mix(t * 1000003ULL + (u64)j) % n] = (u32)j;
}
int main(int argc, char** argv) {
  const char* k = argc > 1 ? argv[1] : "gather4";
  const u64 mib = argc > 2 ? strtoull(argv[2], 0, 10) : 4096;
  const int steps = argc > 3 ? atoi(argv[3]) : 256;
  const int waves = argc > 4 ? atoi(argv[4]) : 65536;
  const u64 n = mib * (1ULL << 20) / 4;
  u32 *a, *sink; unsigned long long* total;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&total, 8)); CK(hipMemset(total, 0, 8));
  hipLaunchKernelGGL(k_init, dim3(65536), dim3(256), 0, 0, a, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double acc = 0, useful = 0;
  for (int rep = 0; rep < 2; rep++) {        // rep 0 = warm-up (code load), rep 1 is the one reported; both show up in a profile
    CK(hipMemset(total, 0, 8));
    CK(hipEventRecord(e0, 0));
    if (!strcmp(k, "stream16")) { hipLaunchKernelGGL(k_stream16, dim3(16384), dim3(256), 0, 0, (const uint4*)a, n / 4, sink); acc = (double)n / 4; useful = (double)n * 4; }
    else if (!strcmp(k, "stream16w")) { hipLaunchKernelGGL(k_stream16w, dim3(16384), dim3(256), 0, 0, (uint4*)a, n / 4); acc = (double)n / 4; useful = (double)n * 4; }
    else if (!strcmp(k, "gather4")) { hipLaunchKernelGGL(k_gather4, dim3(waves / 4), dim3(256), 0, 0, a, n, steps, sink); acc = (double)waves * 64 * steps; useful = acc * 4; }
    else if (!strcmp(k, "chase4")) { hipLaunchKernelGGL(k_chase4, dim3(waves), dim3(64), 0, 0, a, n, steps, 0, sink, total); acc = (double)waves * 64 * steps; useful = acc * 4; }
    else if (!strcmp(k, "chase4geo")) { hipLaunchKernelGGL(k_chase4, dim3(waves), dim3(64), 0, 0, a, n, steps, 1, sink, total); }
    else if (!strcmp(k, "chaseblk")) { hipLaunchKernelGGL(k_chaseblk, dim3(waves), dim3(64), 0, 0, a, n, steps, sink); acc = (double)waves * steps; useful = acc * 4; }
    else if (!strcmp(k, "scatter4")) { hipLaunchKernelGGL(k_scatter4, dim3(waves / 4), dim3(256), 0, 0, a, n, steps); acc = (double)waves * 64 * steps; useful = acc * 4; }
    else { fprintf(stderr, "unknown kernel %s\n", k); return 2; }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    if (!strcmp(k, "chase4geo")) { unsigned long long t; CK(hipMemcpy(&t, total, 8, hipMemcpyDeviceToHost)); acc = (double)t; useful = acc * 4; }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1 && !strcmp(k, "chaseblk")) printf("chaseblk: %.1f ns per dependent load of one lane, %d waves side by side\n", ms * 1e6 / steps, waves);
    if (rep == 1) printf("%s array %llu MiB steps %d waves %d: %.2f ms, %.0f accesses, %.2f G accesses/s, %.1f useful GB/s\n", k, (unsigned long long)mib, steps, waves, ms, acc, acc / ms / 1e6, useful / ms / 1e6);
  }
  return 0;
}
