// kz_datatype.h -- Global.DataType values and Global.detectSimpleType for a workgroup of 256 threads.
// K/Global.java:40-80 (enum), :556-605 (detectSimpleType).  Numbering as KZ_DT_* in include/kanzi_hip.h.
#pragma once
#include "kz_device.h"

#define DT_UNDEFINED 0
#define DT_DNA 1
#define DT_SMALL_ALPHABET 2
#define DT_TEXT 3
#define DT_MULTIMEDIA 4
#define DT_EXE 5
#define DT_NUMERIC 6
#define DT_BASE64 7
#define DT_BIN 8
#define DT_UTF8 9

// sum over the 256 threads of a workgroup (one value per thread); every thread gets the total; lds4 = 4 x 8 bytes
__device__ __forceinline__ long long kz_wg256_sum64(long long v, long long* lds4) {
  for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// f = freqs0[threadIdx.x]; every thread returns the verdict.  Must be called by all 256 threads.
__device__ __forceinline__ int kz_detect_simple_type_wg(int len, int f, int fEq, long long* lds4) {
  const int tid = (int)threadIdx.x;
  const bool isDna = tid == 'a' || tid == 'c' || tid == 'g' || tid == 'n' || tid == 't' || tid == 'u' ||
                     tid == 'A' || tid == 'C' || tid == 'G' || tid == 'N' || tid == 'T' || tid == 'U';
  const bool isDigit = tid >= '0' && tid <= '9';
  const bool isNum = isDigit || tid == '+' || tid == '-' || tid == '*' || tid == '/' || tid == '=' || tid == ',' || tid == '.' ||
                     tid == ':' || tid == ';' || tid == ' ';
  const bool isB64 = isDigit || (tid >= 'A' && tid <= 'Z') || (tid >= 'a' && tid <= 'z') || tid == '+' || tid == '/';
  const long long sDna = kz_wg256_sum64(isDna ? f : 0, lds4);
  const long long sNum = kz_wg256_sum64(isNum ? f : 0, lds4);
  const long long sB64 = kz_wg256_sum64(isB64 ? f : 0, lds4) + ((fEq == 1) ? 1 : 0);     // fEq = freqs0['=']: trailing padding
  const long long nSym = kz_wg256_sum64(f > 0 ? 1 : 0, lds4);
  if (len == 0) return DT_UNDEFINED;
  if (sDna > len - len / 12) return DT_DNA;
  if (sNum == len) return DT_NUMERIC;
  if (sB64 == len) return DT_BASE64;
  if (nSym == 256) return DT_BIN;
  if (nSym <= 4) return DT_SMALL_ALPHABET;
  return DT_UNDEFINED;
}
