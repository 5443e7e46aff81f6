/* ORACLE (test infrastructure only; parity unpinned -- see kzo.h).
 * Bit streams + EntropyUtils restated from the reference.
 *   K/bitstream/DefaultOutputBitStream.java:103-205 (writeBits: low `count` bits, MSB first)
 *   K/bitstream/DefaultInputBitStream.java:94-177
 *   K/entropy/EntropyUtils.java:38-300
 */
#include "kzo.h"
#include <stdlib.h>
#include <string.h>

void kzo_obs_init(kzo_obs* s, size_t cap) {
  if (cap < 64) cap = 64;
  s->buf = (uint8_t*)calloc(cap, 1); s->cap = cap; s->nbits = 0; s->owns = 1; s->overflow = 0;
}
void kzo_obs_wrap(kzo_obs* s, uint8_t* buf, size_t cap) {
  s->buf = buf; s->cap = cap; s->nbits = 0; s->owns = 0; s->overflow = 0;
  memset(buf, 0, cap);
}
void kzo_obs_free(kzo_obs* s) { if (s->owns) free(s->buf); s->buf = NULL; }

static int obs_reserve(kzo_obs* s, uint64_t moreBits) {
  size_t need = (size_t)((s->nbits + moreBits + 7) >> 3) + 8;
  if (need <= s->cap) return 1;
  if (!s->owns) { s->overflow = 1; return 0; }
  size_t ncap = s->cap * 2; if (ncap < need) ncap = need;
  uint8_t* nb = (uint8_t*)realloc(s->buf, ncap);
  if (!nb) { s->overflow = 1; return 0; }
  memset(nb + s->cap, 0, ncap - s->cap);
  s->buf = nb; s->cap = ncap; return 1;
}

/* Invariant: all bits at positions >= nbits in buf are zero. */
void kzo_obs_write(kzo_obs* s, uint64_t v, int count) {
  if (count <= 0) return;
  if (!obs_reserve(s, (uint64_t)count)) return;
  if (count < 64) v &= ((1ULL << count) - 1);
  uint64_t pos = s->nbits;
  int rem = count;
  while (rem > 0) {
    int bitoff = (int)(pos & 7);
    int room = 8 - bitoff;
    int take = rem < room ? rem : room;
    uint8_t bits = (uint8_t)((v >> (rem - take)) & ((1u << take) - 1));
    s->buf[pos >> 3] |= (uint8_t)(bits << (room - take));
    pos += take; rem -= take;
  }
  s->nbits = pos;
}

void kzo_obs_write_bytes(kzo_obs* s, const uint8_t* p, uint64_t nbits) {
  if (!nbits) return;
  if (!obs_reserve(s, nbits)) return;
  if ((s->nbits & 7) == 0) {
    size_t nb = (size_t)(nbits >> 3);
    memcpy(s->buf + (s->nbits >> 3), p, nb);
    s->nbits += (uint64_t)nb << 3;
    int r = (int)(nbits & 7);
    if (r) kzo_obs_write(s, (uint64_t)(p[nb] >> (8 - r)), r);
    else s->buf[s->nbits >> 3] = 0;
    return;
  }
  uint64_t full = nbits >> 3;
  for (uint64_t i = 0; i < full; i++) kzo_obs_write(s, p[i], 8);
  int r = (int)(nbits & 7);
  if (r) kzo_obs_write(s, (uint64_t)(p[full] >> (8 - r)), r);
}

void kzo_ibs_init(kzo_ibs* s, const uint8_t* buf, uint64_t nbits) {
  s->buf = buf; s->nbits = nbits; s->pos = 0; s->error = 0;
}

uint64_t kzo_ibs_read(kzo_ibs* s, int count) {
  if (count <= 0) return 0;
  if (s->pos + (uint64_t)count > s->nbits) { s->error = 1; s->pos = s->nbits; return 0; }
  uint64_t v = 0, pos = s->pos;
  int rem = count;
  while (rem > 0) {
    int bitoff = (int)(pos & 7);
    int room = 8 - bitoff;
    int take = rem < room ? rem : room;
    uint8_t b = s->buf[pos >> 3];
    uint64_t bits = (uint64_t)((b >> (room - take)) & ((1u << take) - 1));
    v = (v << take) | bits;
    pos += take; rem -= take;
  }
  s->pos = pos;
  return v;
}

void kzo_ibs_read_bytes(kzo_ibs* s, uint8_t* p, uint64_t nbits) {
  if (s->pos + nbits > s->nbits) { s->error = 1; s->pos = s->nbits; return; }
  if ((s->pos & 7) == 0) {
    size_t nb = (size_t)(nbits >> 3);
    memcpy(p, s->buf + (s->pos >> 3), nb);
    s->pos += (uint64_t)nb << 3;
    int r = (int)(nbits & 7);
    if (r) p[nb] = (uint8_t)(kzo_ibs_read(s, r) << (8 - r));
    return;
  }
  uint64_t full = nbits >> 3;
  for (uint64_t i = 0; i < full; i++) p[i] = (uint8_t)kzo_ibs_read(s, 8);
  int r = (int)(nbits & 7);
  if (r) p[full] = (uint8_t)(kzo_ibs_read(s, r) << (8 - r));
}

/* ---------------- EntropyUtils ---------------- */

/* K/entropy/EntropyUtils.java:38-75 */
int kzo_encode_alphabet(kzo_obs* s, const int* alphabet, int count) {
  if (count == 0) { kzo_obs_write(s, 0, 1); kzo_obs_write(s, 1, 1); return 0; }
  if (count == 256) { kzo_obs_write(s, 0, 1); kzo_obs_write(s, 0, 1); return 256; }
  kzo_obs_write(s, 1, 1);
  uint8_t masks[32]; memset(masks, 0, 32);
  for (int i = 0; i < count; i++) masks[alphabet[i] >> 3] |= (uint8_t)(1 << (alphabet[i] & 7));
  int lastMask = alphabet[count - 1] >> 3;
  kzo_obs_write(s, (uint64_t)lastMask, 5);
  for (int i = 0; i <= lastMask; i++) kzo_obs_write(s, masks[i], 8);
  return count;
}

/* K/entropy/EntropyUtils.java:86-122 */
int kzo_decode_alphabet(kzo_ibs* s, int* alphabet) {
  int type = (int)kzo_ibs_read(s, 1);
  if (type == 0) {
    if (kzo_ibs_read(s, 1) == 1) return 0;
    for (int i = 0; i < 256; i++) alphabet[i] = i;
    return 256;
  }
  int lastMask = (int)kzo_ibs_read(s, 5), count = 0;
  for (int i = 0; i <= lastMask; i++) {
    int mask = (int)kzo_ibs_read(s, 8);
    for (int j = 0; j < 8; j++)
      if (mask & (1 << j)) alphabet[count++] = (i << 3) + j;
  }
  return count;
}

/* K/entropy/EntropyUtils.java:141-250; alphabet has 256 slots. */
int kzo_normalize_freqs(int* freqs, int* alphabet, int totalFreq, int scale) {
  if (totalFreq == 0) return 0;
  int alphabetSize = 0;
  if (totalFreq == scale) {                                   /* :155-162 shortcut */
    for (int i = 0; i < 256; i++) if (freqs[i] != 0) alphabet[alphabetSize++] = i;
    return alphabetSize;
  }
  int sumScaledFreq = 0, sumFreq = 0, idxMax = 0;
  for (int i = 0; i < 256; i++) {                             /* :169-190 */
    alphabet[i] = 0;
    int f = freqs[i];
    if (f == 0) continue;
    int64_t sf = (int64_t)freqs[i] * scale;
    int scaledFreq = (sf <= totalFreq) ? 1 : (int)((sf + ((int64_t)totalFreq >> 1)) / (int64_t)totalFreq);
    alphabet[alphabetSize++] = i;
    sumScaledFreq += scaledFreq;
    freqs[i] = scaledFreq;
    sumFreq += f;
    if (scaledFreq > freqs[idxMax]) idxMax = i;
    if (sumFreq >= totalFreq) break;
  }
  if (alphabetSize == 0) return 0;
  if (alphabetSize == 1) { freqs[alphabet[0]] = scale; return 1; }
  if (sumScaledFreq == scale) return alphabetSize;
  int delta = sumScaledFreq - scale;
  int errThr = freqs[idxMax] >> 4;
  int ad = delta < 0 ? -delta : delta;
  if (ad <= errThr) { freqs[idxMax] -= delta; return alphabetSize; }  /* :204-208 fast path */
  if (delta < 0) { delta += errThr; freqs[idxMax] += errThr; }
  else { delta -= errThr; freqs[idxMax] -= errThr; }
  int inc = (delta > 0) ? -1 : 1;                              /* :219-246 slow path */
  delta = delta < 0 ? -delta : delta;
  int round = 0;
  while ((++round < 6) && (delta > 0)) {
    int adjustments = 0;
    for (int i = 0; i < alphabetSize; i++) {
      int idx = alphabet[i];
      if (freqs[idx] <= 2) continue;
      freqs[idx] += inc; adjustments++; delta--;
      if (delta == 0) break;
    }
    if (adjustments == 0) break;
  }
  int v = freqs[idxMax] - delta;
  freqs[idxMax] = v > 1 ? v : 1;
  return alphabetSize;
}

/* K/entropy/EntropyUtils.java:259-276 (value is a Java int; >>> logical) */
void kzo_write_varint(kzo_obs* s, uint32_t value) {
  if (value >= 128) {   /* (value >= 128) || (value < 0) as signed == unsigned >= 128 */
    kzo_obs_write(s, 0x80 | (value & 0x7F), 8);
    value >>= 7;
    while (value >= 128) { kzo_obs_write(s, 0x80 | (value & 0x7F), 8); value >>= 7; }
  }
  kzo_obs_write(s, value, 8);
}

/* K/entropy/EntropyUtils.java:284-300 */
uint32_t kzo_read_varint(kzo_ibs* s) {
  uint32_t value = (uint32_t)kzo_ibs_read(s, 8);
  uint32_t res = value & 0x7F;
  int shift = 7;
  while (value >= 128) {
    value = (uint32_t)kzo_ibs_read(s, 8);
    res |= ((value & 0x7F) << shift);
    if (shift == 28) break;
    shift += 7;
    if (s->error) break;
  }
  return res;
}

/* java.util.Random: 48-bit LCG */
void kzo_jrandom_init(kzo_jrandom* r, int64_t seed) {
  r->seed = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
}
static int32_t jnext(kzo_jrandom* r, int bits) {
  r->seed = (r->seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
  return (int32_t)((int64_t)r->seed >> (48 - bits));
}
int32_t kzo_jrandom_next_int(kzo_jrandom* r, int32_t bound) {
  if (bound <= 0) return jnext(r, 32);
  int32_t rr = jnext(r, 31);
  int32_t m = bound - 1;
  if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
  for (int32_t u = rr; (int32_t)((uint32_t)u - (uint32_t)(rr = u % bound) + (uint32_t)m) < 0; u = jnext(r, 31)) {}
  return rr;
}
