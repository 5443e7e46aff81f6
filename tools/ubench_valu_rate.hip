// Micro-benchmark: how long one SIMD is busy with an instruction, i.e. what W waves per SIMD pay for it when they all issue it
// (the serial-per-block kernels run 1 - 3 waves per SIMD: alone a wave is bound by its issue cadence, together by the pipe).
// One workgroup of 256 * W threads on one CU = W waves per SIMD; every wave runs the same unrolled stream of independent instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_rate.hip -o /tmp/ubench_valu_rate && /tmp/ubench_valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 20000
#define X16(a) a a a a a a a a a a a a a a a a
template <int K>
__global__ __launch_bounds__(1024) void k_rate(long long* out, int* sink) {
  double a = threadIdx.x * 1.5 + 3.0, b = threadIdx.x * 0.25 + 2.0;
  unsigned x = threadIdx.x * 2654435761u, y = x ^ 0x1234567u;
  __syncthreads();
  const long long w0 = wall_clock64();
  for (int i = 0; i < REP; i++) {
    if (K == 0) asm volatile(X16("v_add_u32 %0, 1, %0\n\tv_add_u32 %1, 3, %1\n\t") : "+v"(x), "+v"(y));                 // 32 plain VALU
    if (K == 1) asm volatile(X16("v_min_f64 %0, %0, %1\n\tv_max_f64 %1, %0, %1\n\t") : "+v"(a), "+v"(b));                 // 32 DP min / max (dependent pairs)
    if (K == 2) asm volatile(X16("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(x), "+v"(y));
    if (K == 3) asm volatile(X16("v_cmpx_ge_u32 vcc, 40, %0\n\ts_mov_b64 exec, -1\n\t") : "+v"(x) : : "vcc");            // 16 v_cmpx + 16 SALU
    if (K == 4) asm volatile(X16("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(x), "+v"(y) : : "vcc");
    if (K == 5) asm volatile(X16("v_readlane_b32 s40, %0, 5\n\tv_readlane_b32 s41, %1, 9\n\t") : : "v"(x), "v"(y) : "s40", "s41");
    if (K == 6) asm volatile(X16("v_add_f64 %0, %0, %1\n\tv_add_f64 %1, %0, %1\n\t") : "+v"(a), "+v"(b));
    if (K == 7) asm volatile(X16("s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 3\n\t") : : : "s40", "s41", "scc");      // 32 SALU
    if (K == 9) asm volatile(X16("v_bitop3_b32 %0, %0, %1, %0 bitop3:0xf6\n\tv_bitop3_b32 %1, %1, %0, %1 bitop3:0xf6\n\t") : "+v"(x), "+v"(y));   // gfx950's 3-input boolean
    if (K == 10) asm volatile(X16("v_bitop3_b32 %0, %0, s40, %1 bitop3:0xf6\n\tv_bitop3_b32 %1, %1, s41, %0 bitop3:0xf6\n\t") : "+v"(x), "+v"(y) : : "s40", "s41");
    if (K == 11) asm volatile(X16("v_cmp_ne_u32_e64 s[40:41], 0, %0\n\tv_add_u32 %1, 3, %1\n\t") : "+v"(x), "+v"(y) : : "s40", "s41");   // compare into an SGPR pair + plain
    if (K == 12) asm volatile(X16("v_cmp_ne_u32_e64 s[40:41], 0, %0\n\tv_add_u32 %1, 3, %1\n\tv_add_u32 %0, 5, %0\n\tv_xor_b32 %1, s40, %1\n\t") : "+v"(x), "+v"(y) : : "s40", "s41");   // 64 instructions: the SGPR read back by a VALU two slots later
    if (K == 8) asm volatile(X16("v_alignbit_b32 %0, %1, %0, 10\n\tv_bfi_b32 %1, %0, %1, %0\n\t") : "+v"(x), "+v"(y));  // VOP3 32-bit
  }
  const long long w1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out, (unsigned long long)(w1 - w0));   // the slowest wave (the arbiter favours the oldest)
  sink[threadIdx.x] = (int)(a + b) + (int)(x + y);
}
int main() {
  long long* d; int* s; hipMalloc(&d, 64); hipMalloc(&s, 8192);
  const char* names[] = {"v_add_u32", "v_min_f64 / v_max_f64", "v_mov_b32_dpp wave_shr", "v_cmpx + s_mov exec", "v_cmp + v_cndmask", "v_readlane", "v_add_f64", "s_add_u32", "v_alignbit / v_bfi", "v_bitop3 (VGPRs)", "v_bitop3 (one SGPR)", "v_cmp -> SGPR + v_add", "cmp,add,add,xor(SGPR) x2"};
  for (int k = 0; k < 13; k++) {
    printf("%-26s", names[k]);
    for (int W = 1; W <= 4; W *= 2) {
      long long h = 0;
      for (int rep = 0; rep < 2; rep++) {
        hipMemset(d, 0, 8);
        switch (k) {
          case 0: hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 1: hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 2: hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 3: hipLaunchKernelGGL(k_rate<3>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 4: hipLaunchKernelGGL(k_rate<4>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 5: hipLaunchKernelGGL(k_rate<5>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 6: hipLaunchKernelGGL(k_rate<6>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 7: hipLaunchKernelGGL(k_rate<7>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 8: hipLaunchKernelGGL(k_rate<8>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 9: hipLaunchKernelGGL(k_rate<9>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 10: hipLaunchKernelGGL(k_rate<10>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 11: hipLaunchKernelGGL(k_rate<11>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
          case 12: hipLaunchKernelGGL(k_rate<12>, dim3(1), dim3(256 * W), 0, 0, d, s); break;
        }
        hipDeviceSynchronize();
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      }
      printf("  W=%d: %6.2f ns/instr/wave", W, (double)h * 10.0 / ((double)REP * 32.0));
    }
    printf("\n");
  }
  return 0;
}
