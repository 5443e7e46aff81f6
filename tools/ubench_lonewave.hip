// Micro-benchmark: cost of dependent instruction patterns for ONE wave per CU (the regime of the serial
// per-block kernels). Prints shader cycles (s_memtime) per iteration and the effective clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 200000
__global__ void k_valu(long long* out, int* sink) {
  int x = threadIdx.x;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) { x = x * 3 + 1; x = x ^ (x >> 3); x = x + i; x = x * 5 + 7; }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = x;
}
__global__ void k_pingpong(long long* out, int* sink) {   // VALU -> SGPR -> VALU chain via readlane
  int x = threadIdx.x;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    int s = __builtin_amdgcn_readlane(x, (i & 63));
    s = (s >> 3) + i;
    x = x + s;
    int s2 = __builtin_amdgcn_readlane(x, (s & 63));
    x = x ^ s2;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = x;
}
__global__ void k_ballot(long long* out, int* sink) {     // v_cmp -> s_bcnt -> v_cmp chain
  unsigned x = threadIdx.x * 2654435761u;
  int acc = 0;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    unsigned long long b = __ballot(x > (unsigned)(acc * 977 + i));
    acc += __popcll(b);
    unsigned long long b2 = __ballot((x ^ 0x5555) > (unsigned)(acc * 31));
    acc += __popcll(b2);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = acc + x;
}
__global__ void k_branch(long long* out, int* sink) {     // uniform data-dependent branches
  int x = 1, acc = 0;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    x = __builtin_amdgcn_readfirstlane(x * 1103515245 + 12345);
    switch ((x >> 16) & 3) { case 0: acc += x; break; case 1: acc ^= x; break; case 2: acc -= i; break; default: acc += 3; break; }
    if ((x >> 20) & 1) acc = acc * 3 + 1; else acc = acc + 7;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = acc;
}
__global__ void k_dpp(long long* out, int* sink) {        // DPP wave_shr chain
  int x = threadIdx.x;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, false) + i;
    x = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xF, 0xF, false) ^ 5;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = x;
}
__global__ void k_cmp64(long long* out, int* sink) {      // v_cmp_gt_u64 -> v_bcnt (VALU) chain
  unsigned long long k = ((unsigned long long)threadIdx.x << 40) | threadIdx.x * 2654435761u;
  unsigned acc = 0;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    unsigned long long nk = ((unsigned long long)acc << 38) + i;
    unsigned long long m = __ballot(k > nk);
    unsigned r;
    asm volatile("s_nop 1\n\tv_bcnt_u32_b32 %0, %1, %2\n\tv_bcnt_u32_b32 %0, %3, %0" : "=&v"(r) : "s"((unsigned)m), "v"(acc), "s"((unsigned)(m >> 32)));
    acc = r & 63;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = acc;
}
__global__ void k_cmp32(long long* out, int* sink) {      // v_cmp_gt_u32 -> v_bcnt (VALU) chain
  unsigned k = threadIdx.x * 2654435761u;
  unsigned acc = 0;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    unsigned nk = (acc << 26) + i;
    unsigned long long m = __ballot(k > nk);
    unsigned r;
    asm volatile("s_nop 1\n\tv_bcnt_u32_b32 %0, %1, %2\n\tv_bcnt_u32_b32 %0, %3, %0" : "=&v"(r) : "s"((unsigned)m), "v"(acc), "s"((unsigned)(m >> 32)));
    acc = r & 63;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = acc;
}
__global__ void k_rl_vmov(long long* out, int* sink) {    // v_readlane -> v_mov (VALU) -> v ops -> readlane (lane from loop)
  unsigned x = threadIdx.x * 7 + 1;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    unsigned s = __builtin_amdgcn_readlane(x, i & 63);
    unsigned v; asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    x = x + (v >> 3) + 1;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = x;
}
__global__ void k_cnd(long long* out, int* sink) {        // v_cmp + v_cndmask dependent chain (x4)
  unsigned x = threadIdx.x, y = threadIdx.x * 3;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    x = (y < x + i) ? y + 1 : x + 3;
    y = (x == y) ? x : y + 5;
    x = (y < x) ? y + 7 : x + 1;
    y = (x == y + 2) ? x : y + 1;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; } sink[threadIdx.x] = x + y;
}
int main() {
  long long* d; int* s; hipMalloc(&d, 64); hipMalloc(&s, 4096);
  long long h[2];
  const char* names[] = {"valu x4 dep/iter", "readlane pingpong (2 rl + 3 alu)/iter", "ballot+bcnt x2/iter", "scalar branches (switch+if)/iter", "dpp wave_shr x2/iter", "cmp_u64 + 2 v_bcnt /iter", "cmp_u32 + 2 v_bcnt /iter", "readlane -> v_mov -> 2 valu /iter", "4 x (v_cmp + v_cndmask + add) /iter"};
  for (int rep = 0; rep < 1; rep++)
  for (int k = 0; k < 9; k++) {
    int grid = rep == 0 ? 1 : 256;
    switch (k) {
      case 0: hipLaunchKernelGGL(k_valu, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 1: hipLaunchKernelGGL(k_pingpong, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 2: hipLaunchKernelGGL(k_ballot, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 3: hipLaunchKernelGGL(k_branch, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 4: hipLaunchKernelGGL(k_dpp, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 5: hipLaunchKernelGGL(k_cmp64, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 6: hipLaunchKernelGGL(k_cmp32, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 7: hipLaunchKernelGGL(k_rl_vmov, dim3(grid), dim3(64), 0, 0, d, s); break;
      case 8: hipLaunchKernelGGL(k_cnd, dim3(grid), dim3(64), 0, 0, d, s); break;
    }
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("grid %3d  %-40s  %.1f shader-cycles/iter  %.1f ns/iter (wall 100MHz)  => %.2f GHz\n", grid, names[k], (double)h[0] / N, (double)h[1] * 10.0 / N, (double)h[0] / ((double)h[1] * 10.0));
  }
  return 0;
}
