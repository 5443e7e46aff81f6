/* ORACLE (test infrastructure only; see kzo.h).
 * Suffix array construction by induced sorting (SA-IS, Nong/Zhang/Chan 2009), written from the
 * published algorithm.  The reference sorts suffixes with DivSufSort
 * (K/transform/DivSufSort.java:330-494); the suffix array of a string is unique, so any correct
 * sorter yields the same BWT bytes.  Suffixes compare as plain strings, a proper
 * prefix being smaller (i.e. a virtual sentinel smaller than every byte).
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

#define CH(i) (cs == 1 ? (int32_t)((const uint8_t*)T)[i] : ((const int32_t*)T)[i])

static void get_counts(const void* T, int32_t* C, int32_t n, int32_t k, int cs) {
  for (int32_t i = 0; i < k; i++) C[i] = 0;
  for (int32_t i = 0; i < n; i++) C[CH(i)]++;
}
static void get_buckets(const int32_t* C, int32_t* B, int32_t k, int end) {
  int32_t sum = 0;
  if (end) for (int32_t i = 0; i < k; i++) { sum += C[i]; B[i] = sum; }
  else for (int32_t i = 0; i < k; i++) { sum += C[i]; B[i] = sum - C[i]; }
}

/* induce L then S suffixes from the LMS suffixes already placed in SA */
static void induce(const void* T, int32_t* SA, int32_t* C, int32_t* B, int32_t n, int32_t k, int cs) {
  int32_t i, j, c0, c1, *b;
  get_counts(T, C, n, k, cs);
  get_buckets(C, B, k, 0);
  j = n - 1;
  b = SA + B[c1 = CH(j)];
  *b++ = ((0 < j) && (CH(j - 1) < c1)) ? ~j : j;
  for (i = 0; i < n; i++) {
    j = SA[i]; SA[i] = ~j;
    if (0 < j) {
      j--;
      if ((c0 = CH(j)) != c1) { B[c1] = (int32_t)(b - SA); b = SA + B[c1 = c0]; }
      *b++ = ((0 < j) && (CH(j - 1) < c1)) ? ~j : j;
    }
  }
  get_counts(T, C, n, k, cs);
  get_buckets(C, B, k, 1);
  for (i = n - 1, b = SA + B[c1 = 0]; 0 <= i; i--) {
    if (0 < (j = SA[i])) {
      j--;
      if ((c0 = CH(j)) != c1) { B[c1] = (int32_t)(b - SA); b = SA + B[c1 = c0]; }
      *--b = ((j == 0) || (CH(j - 1) > c1)) ? ~j : j;
    } else {
      SA[i] = ~j;
    }
  }
}

static int sais_main(const void* T, int32_t* SA, int32_t fs, int32_t n, int32_t k, int cs) {
  int32_t *C, *B, *RA;
  int32_t i, j, c, m, p, q, plen, qlen, name;
  int32_t c0, c1;
  int diff;
  int own = 0;
  if (k <= fs) { C = SA + n; B = (k <= (fs - k)) ? C + k : C; }
  else { C = (int32_t*)malloc((size_t)k * 2 * sizeof(int32_t)); if (!C) return -2; B = C + k; own = 1; }

  /* stage 1: sort all LMS substrings */
  get_counts(T, C, n, k, cs);
  get_buckets(C, B, k, 1);
  for (i = 0; i < n; i++) SA[i] = 0;
  for (i = n - 2, c = 0, c1 = CH(n - 1); 0 <= i; i--, c1 = c0) {
    if ((c0 = CH(i)) < (c1 + c)) c = 1;
    else if (c != 0) { SA[--B[c1]] = i + 1; c = 0; }
  }
  induce(T, SA, C, B, n, k, cs);

  /* compact sorted LMS substrings into the first m items of SA */
  for (i = 0, m = 0; i < n; i++) {
    p = SA[i];
    if ((0 < p) && (CH(p - 1) > (c0 = CH(p)))) {
      for (j = p + 1; (j < n) && (c0 == (c1 = CH(j))); j++) {}
      if ((j < n) && (c0 < c1)) SA[m++] = p;
    }
  }
  j = m + (n >> 1);
  for (i = m; i < j; i++) SA[i] = 0;
  /* store the length of all substrings */
  for (i = n - 2, j = n, c = 0, c1 = CH(n - 1); 0 <= i; i--, c1 = c0) {
    if ((c0 = CH(i)) < (c1 + c)) c = 1;
    else if (c != 0) { SA[m + ((i + 1) >> 1)] = j - i - 1; j = i + 1; c = 0; }
  }
  /* find the lexicographic names of all substrings */
  for (i = 0, name = 0, q = n, qlen = 0; i < m; i++) {
    p = SA[i]; plen = SA[m + (p >> 1)]; diff = 1;
    if (plen == qlen) {
      for (j = 0; (j < plen) && (CH(p + j) == CH(q + j)); j++) {}
      if (j == plen) diff = 0;
    }
    if (diff) { name++; q = p; qlen = plen; }
    SA[m + (p >> 1)] = name;
  }

  /* stage 2: solve the reduced problem */
  if (name < m) {
    RA = SA + n + fs - m;
    for (i = m + (n >> 1) - 1, j = m - 1; m <= i; i--)
      if (SA[i] != 0) RA[j--] = SA[i] - 1;
    if (sais_main(RA, SA, fs + n - m * 2, m, name, 4) != 0) { if (own) free(C); return -2; }
    for (i = n - 2, j = m - 1, c = 0, c1 = CH(n - 1); 0 <= i; i--, c1 = c0) {
      if ((c0 = CH(i)) < (c1 + c)) c = 1;
      else if (c != 0) { RA[j--] = i + 1; c = 0; }
    }
    for (i = 0; i < m; i++) SA[i] = RA[SA[i]];
  }

  /* stage 3: induce the result for the original problem */
  get_counts(T, C, n, k, cs);
  get_buckets(C, B, k, 1);
  for (i = m; i < n; i++) SA[i] = 0;
  for (i = m - 1; 0 <= i; i--) { j = SA[i]; SA[i] = 0; SA[--B[CH(j)]] = j; }
  induce(T, SA, C, B, n, k, cs);
  if (own) free(C);
  return 0;
}

void kzo_suffix_array(const uint8_t* t, int32_t* sa, int n) {
  if (n <= 0) return;
  if (n == 1) { sa[0] = 0; return; }
  sais_main(t, sa, 0, n, 256, 1);
}
