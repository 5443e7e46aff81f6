// kz_huffman.hip -- canonical length-limited (12-bit) Huffman chunk encoder / decoder for gfx950.
//
// Replaces K/entropy/HuffmanEncoder.java:380-416 (encode), :103-178 (updateFrequencies),
// :285-308 (computeCodeLengths) + :317-376 (Moffat-Katajainen in-place phases), :191-273
// (limitCodeLengths), :419-493 (encodeChunk); K/entropy/HuffmanCommon.java:71-111
// (generateCanonicalCodes); K/entropy/ExpGolombEncoder.java:123-131;
// K/entropy/HuffmanDecoder.java:115-154 (readLengths), :162-191 (tables), :353-390, :404-587.
//
// Statistics reset every 16 KiB chunk (HuffmanCommon.java:30-40): one wave64 per chunk.  Parallel
// parts: LDS histogram, the (freq,symbol) sort as a rank-by-counting over LDS, canonical code
// assignment from per-length ballots, and the bit packing itself (code-length prefix sums give every
// lane its bit offset inside each of the 4 fragments).  Serial on one lane (n <= 256 steps): the
// Moffat-Katajainen phases and the Exp-Golomb header.  Chunk bit strings are concatenated at bit
// granularity by the scan + funnel-shift kernels shared with ANS0 (kz_ans.hip).
#include "kz_device.h"
#include "kz_internal.h"
#include "kz_chunk.h"

typedef uint16_t u16;

#define HUF_MAXLEN 12

// ---- bit helpers ---------------------------------------------------------------------------------
struct HBitW { u8* p; u32 pos; };
__device__ __forceinline__ void hbw_put(HBitW& w, u32 v, int count) {      // MSB first into a zeroed buffer
  while (count > 0) {
    const int bitoff = w.pos & 7, room = 8 - bitoff;
    const int take = count < room ? count : room;
    const u32 bits = (v >> (count - take)) & ((1u << take) - 1u);
    w.p[w.pos >> 3] |= (u8)(bits << (room - take));
    w.pos += take; count -= take;
  }
}
// signed Exp-Golomb (ExpGolombEncoder.java:123-131, table CACHE_VALUES[1]): 0 -> '1'; else a=|v|,
// k=floor(log2(a+1)), m=a+1-2^k: k zeros, '1', m in k bits, sign bit
__device__ __forceinline__ void hbw_eg(HBitW& w, int v) {
  if (v == 0) { hbw_put(w, 1, 1); return; }
  const int a = v < 0 ? -v : v;
  const int k = kz_ilog2((u32)(a + 1));
  const int m = a + 1 - (1 << k);
  hbw_put(w, (1u << (k + 1)) | ((u32)m << 1) | (v < 0 ? 1u : 0u), 2 * k + 2);
}
// 32 bits of the bit string p[0..nbits) starting at bit r (may be negative); zero outside
__device__ __forceinline__ u32 huf_fetch32(const u8* p, int nbits, int r) {
  if (r >= nbits || r + 32 <= 0) return 0;
  const int byte0 = (r >= 0) ? (r >> 3) : -((-r + 7) >> 3);
  const int nbytes = (nbits + 7) >> 3;
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) { const int bi = byte0 + k; acc = (acc << 8) | (u64)((bi >= 0 && bi < nbytes) ? p[bi] : 0); }
  const int sh = r - byte0 * 8;
  u32 w = (u32)((acc << sh) >> 8);
  const int over = r + 32 - nbits;
  if (over > 0) w &= (over >= 32) ? 0u : (0xFFFFFFFFu << over);
  if (r < 0) w &= ((-r) >= 32) ? 0u : (0xFFFFFFFFu >> (-r));
  return w;
}
// wave-wide: append the nbits-long bit string src to dst (bytes, MSB-first stream) at bit dpos
__device__ __forceinline__ void huf_append(u8* dst, int dpos, const u8* src, int nbits) {
  if (nbits <= 0) return;
  u32* o = (u32*)dst;
  const int w0 = dpos >> 5, w1 = (dpos + nbits - 1) >> 5;
  for (int w = w0 + kz_lane(); w <= w1; w += 64) {
    const int r = w * 32 - dpos;
    const u32 v = __builtin_bswap32(huf_fetch32(src, nbits, r));
    if (r >= 0 && r + 32 <= nbits) o[w] = v; else atomicOr(&o[w], v);
  }
}

// canonical codes (HuffmanCommon.java:74-108): symbols ordered by (length, symbol); lane holds symbols q*64+lane
__device__ __forceinline__ void huf_canonical(const u32 sz[4], u32 code[4]) {
  u32 first = 0; int cur = 0;          // wave-uniform running canonical code
#pragma unroll
  for (int q = 0; q < 4; q++) code[q] = 0;
  for (int L = 1; L <= HUF_MAXLEN; L++) {
    u32 run = 0, rk[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint64_t bal = kz_ballot(sz[q] == (u32)L);
      rk[q] = run + (u32)__popcll(bal & kz_lanemask_lt());
      run += (u32)__popcll(bal);
    }
    if (run == 0) continue;
    // code = (code + count[prevLen]) << (len - prevLen), starting at 0 for the shortest length
    if (cur != 0) first <<= (L - cur);
    cur = L;
#pragma unroll
    for (int q = 0; q < 4; q++) if (sz[q] == (u32)L) code[q] = first + rk[q];
    first += run;
  }
}

// =================================================================================================
__global__ __launch_bounds__(64) void k_huf_enc_chunk(const u8* __restrict__ src, int64_t stride,
                                                       const int32_t* __restrict__ d_len, AnsEnc E) {
  const int b = blockIdx.y, ck = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  const int start = ck * ANS_CHUNK;
  if (start >= count) return;
  const int len = min(ANS_CHUNK, count - start);
  const int64_t ci = (int64_t)b * E.C + ck;
  const u8* blk = src + (int64_t)b * stride + start;
  u8* scr = E.scr + ci * ANS_SCRATCH;
  for (int i = lane; i < ANS_SCRATCH / 16; i += 64) ((uint4*)scr)[i] = make_uint4(0, 0, 0, 0);
  if (lane == 0) { E.hdrBits[ci] = 0; E.tailOff[ci] = 0; }
  if (len < 32) {                                               // HuffmanEncoder.java:400-402
    __syncthreads();
    if (lane < len) scr[lane] = blk[lane];
    if (lane == 0) E.tailBits[ci] = 8u * (u32)len;
    return;
  }
  __shared__ u32 hist[256];
  __shared__ u32 keys[256];
  __shared__ u32 sorted[256];
  __shared__ int mk[256];
  __shared__ u8 ranks[256];
  __shared__ u8 sizes[256];
  __shared__ u32 codes[256];            // (len << 24) | code
  __shared__ u8 ll[6][256];
  __shared__ u32 firstCode[16];
  __shared__ u32 hbuf[ANS_HDR_BYTES / 4];
  __shared__ u32 fragbuf[(4096 * HUF_MAXLEN) / 32 + 8];
  const u8* data = blk;                 // symbols are re-read from global memory (L1): a 16 KiB LDS copy halves the waves per CU
  __shared__ int sh_n, sh_maxlen;
  __shared__ u32 sh_nb[4];

  for (int i = lane; i < 256; i += 64) { hist[i] = 0; sizes[i] = 0; }
  for (int i = lane; i < ANS_HDR_BYTES / 4; i += 64) hbuf[i] = 0;
  __syncthreads();
  for (int i = lane * 4; i < len; i += 256) {
    u32 w = 0; const int nb = min(4, len - i);
    for (int k = 0; k < nb; k++) w |= (u32)blk[i + k] << (8 * k);
    for (int k = 0; k < 4; k++) {
      const bool valid = k < nb;
      const u32 c = (w >> (8 * k)) & 0xFF;
      const uint64_t peers = kz_match8(c, valid);
      if (valid && (peers & kz_lanemask_lt()) == 0) atomicAdd(&hist[c], (u32)__popcll(peers));
    }
  }
  __syncthreads();
  // ---- alphabet + sort keys (freq << 8 | symbol), rank by counting ----
  u32 f[4], key[4]; int n = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    f[q] = hist[q * 64 + lane];
    key[q] = f[q] ? ((f[q] << 8) | (u32)(q * 64 + lane)) : 0xFFFFFFFFu;
    keys[q * 64 + lane] = key[q];
    n += (int)__popcll(kz_ballot(f[q] != 0));
  }
  __syncthreads();
  if (n > 1) {
    u32 r[4] = {0, 0, 0, 0};
    for (int t = 0; t < 256; t++) {
      const u32 kt = keys[t];
#pragma unroll
      for (int q = 0; q < 4; q++) r[q] += (kt < key[q]) ? 1u : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) if (f[q]) sorted[r[q]] = key[q];     // Arrays.sort(ranks, 0, count) (:288)
  }
  __syncthreads();
  // ---- code lengths: serial Moffat-Katajainen on one lane ----
  if (lane == 0) {
    int maxlen = 0;
    if (n == 1) {
      for (int s = 0; s < 256; s++) if (hist[s]) sizes[s] = 1;       // :122-124
      maxlen = 1;
    } else {
      for (int i = 0; i < n; i++) { mk[i] = (int)(sorted[i] >> 8); ranks[i] = (u8)(sorted[i] & 0xFF); }
      // phase 1 (:317-340)
      for (int s = 0, r = 0, t = 0; t < n - 1; t++) {
        int sum = 0;
        for (int i = 0; i < 2; i++) {
          if ((s >= n) || ((r < t) && (mk[r] < mk[s]))) { sum += mk[r]; mk[r] = t; r++; continue; }
          sum += mk[s];
          if (s > t) mk[s] = 0;
          s++;
        }
        mk[t] = sum;
      }
      // phase 2 (:342-376)
      {
        int levelTop = n - 2, depth = 1, i = n, total = 2;
        while (i > 0) {
          int k = levelTop;
          while ((k > 0) && (mk[k - 1] >= levelTop)) k--;
          const int internal = levelTop - k;
          const int leaves = total - internal;
          for (int j = 0; j < leaves; j++) mk[--i] = depth;
          total = internal << 1;
          levelTop = k;
          depth++;
        }
        maxlen = depth - 1;
      }
      for (int i = 0; i < n; i++) sizes[ranks[i]] = (u8)mk[i];
      if (maxlen > HUF_MAXLEN) {
        // limitCodeLengths (:191-273); ranks[] = symbols by increasing frequency
        int nn = 0, debt = 0;
        while (nn < n && sizes[ranks[nn]] >= HUF_MAXLEN) { debt += sizes[ranks[nn]] - HUF_MAXLEN; sizes[ranks[nn]] = HUF_MAXLEN; nn++; }
        int head[6] = {0, 0, 0, 0, 0, 0}, tail[6] = {0, 0, 0, 0, 0, 0};
        while (nn < n) {
          const int idx = HUF_MAXLEN - 1 - sizes[ranks[nn]];
          if ((idx >= 6) || (debt < (1 << idx))) break;
          ll[idx][tail[idx]++] = ranks[nn];
          nn++;
        }
        int idx = 5;
        while ((debt > 0) && (idx >= 0)) {
          if ((head[idx] == tail[idx]) || (debt < (1 << idx))) { idx--; continue; }
          const int r = ll[idx][head[idx]++];
          sizes[r]++;
          debt -= (1 << idx);
        }
        idx = 0;
        while ((debt > 0) && (idx < 6)) {
          if (head[idx] == tail[idx]) { idx++; continue; }
          const int r = ll[idx][head[idx]++];
          sizes[r]++;
          debt -= (1 << idx);
        }
        maxlen = HUF_MAXLEN;
        if (debt > 0) {
          // slow fallback (:250-270): renormalise the frequencies (alphabet order) to 2048 and redo.
          // normalizeFrequencies on the compacted array; reuse mk[] as f[], keys[] as the symbol list
          int alen = 0, totalFreq = 0;
          for (int s = 0; s < 256; s++) if (hist[s]) { mk[alen] = (int)hist[s]; keys[alen] = (u32)s; totalFreq += (int)hist[s]; alen++; }
          const int scale = ANS_CHUNK >> 3;
          if (totalFreq != scale) {
            int sumScaled = 0, idxMax = 0, asz = 0;
            for (int i = 0; i < alen; i++) {
              const long long sf = (long long)mk[i] * scale;
              const int sc = (sf <= totalFreq) ? 1 : (int)((sf + (totalFreq >> 1)) / totalFreq);
              sumScaled += sc; mk[i] = sc; asz++;
              if (sc > mk[idxMax]) idxMax = i;
            }
            if (asz > 1 && sumScaled != scale) {
              int delta = sumScaled - scale;
              const int errThr = mk[idxMax] >> 4;
              if ((delta < 0 ? -delta : delta) <= errThr) mk[idxMax] -= delta;
              else {
                if (delta < 0) { delta += errThr; mk[idxMax] += errThr; } else { delta -= errThr; mk[idxMax] -= errThr; }
                const int inc = (delta > 0) ? -1 : 1;
                delta = delta < 0 ? -delta : delta;
                int round = 0;
                while ((++round < 6) && (delta > 0)) {
                  int adj = 0;
                  for (int i = 0; i < asz; i++) { if (mk[i] <= 2) continue; mk[i] += inc; adj++; delta--; if (delta == 0) break; }
                  if (adj == 0) break;
                }
                const int v = mk[idxMax] - delta;
                mk[idxMax] = v > 1 ? v : 1;
              }
            }
          }
          // ranks = (f << 8 | sym) sorted ascending: insertion sort (rare path)
          for (int i = 0; i < alen; i++) sorted[i] = ((u32)mk[i] << 8) | keys[i];
          for (int i = 1; i < alen; i++) { const u32 v = sorted[i]; int j = i - 1; while (j >= 0 && sorted[j] > v) { sorted[j + 1] = sorted[j]; j--; } sorted[j + 1] = v; }
          for (int i = 0; i < n; i++) { mk[i] = (int)(sorted[i] >> 8); ranks[i] = (u8)(sorted[i] & 0xFF); }
          for (int s = 0, r = 0, t = 0; t < n - 1; t++) {
            int sum = 0;
            for (int i = 0; i < 2; i++) {
              if ((s >= n) || ((r < t) && (mk[r] < mk[s]))) { sum += mk[r]; mk[r] = t; r++; continue; }
              sum += mk[s];
              if (s > t) mk[s] = 0;
              s++;
            }
            mk[t] = sum;
          }
          int levelTop = n - 2, depth = 1, i = n, total = 2;
          while (i > 0) {
            int k = levelTop;
            while ((k > 0) && (mk[k - 1] >= levelTop)) k--;
            const int internal = levelTop - k;
            const int leaves = total - internal;
            for (int j = 0; j < leaves; j++) mk[--i] = depth;
            total = internal << 1;
            levelTop = k;
            depth++;
          }
          maxlen = depth - 1;
          for (int i2 = 0; i2 < n; i2++) sizes[ranks[i2]] = (u8)mk[i2];
        }
      }
      if (maxlen > HUF_MAXLEN) {                                  // :146-155 fixed 8-bit codes
        int k = 0;
        for (int s = 0; s < 256; s++) if (hist[s]) { sizes[s] = 8; codes[s] = (8u << 24) | (u32)k; k++; }
      }
    }
    sh_n = n; sh_maxlen = maxlen;
  }
  __syncthreads();
  const int maxlen = sh_maxlen;
  u32 sz[4], cd[4];
#pragma unroll
  for (int q = 0; q < 4; q++) sz[q] = sizes[q * 64 + lane];
  if (n > 1 && maxlen <= HUF_MAXLEN) {
    huf_canonical(sz, cd);
#pragma unroll
    for (int q = 0; q < 4; q++) if (sz[q]) codes[q * 64 + lane] = (sz[q] << 24) | cd[q];
  }
  __syncthreads();
  // ---- header: alphabet (EntropyUtils.encodeAlphabet) + signed Exp-Golomb length deltas (:163-174) ----
  if (lane == 0) {
    HBitW w{(u8*)hbuf, 0};
    if (n == 256) { hbw_put(w, 0, 1); hbw_put(w, 0, 1); }
    else {
      hbw_put(w, 1, 1);
      int last = 255; while (last > 0 && !hist[last]) last--;
      const int lastMask = last >> 3;
      hbw_put(w, (u32)lastMask, 5);
      for (int i = 0; i <= lastMask; i++) { u32 m = 0; for (int j = 0; j < 8; j++) if (hist[i * 8 + j]) m |= 1u << j; hbw_put(w, m, 8); }
    }
    int prev = 2;
    for (int s = 0; s < 256; s++) if (hist[s]) { hbw_eg(w, (int)(signed char)((int)sizes[s] - prev)); prev = sizes[s]; }
    sh_nb[0] = w.pos;                                             // header bits (temp)
  }
  __syncthreads();
  int pos = (int)sh_nb[0];
  huf_append(scr, 0, (const u8*)hbuf, pos);
  if (n <= 1) { if (lane == 0) E.tailBits[ci] = (u32)pos; return; }       // :408 one symbol: no payload
  // ---- payload: 4 fragments of len/4 symbols (:426-474); pass 1 = bit counts ----
  const int szFrag = len >> 2;
  const int per = (szFrag + 63) >> 6;
  u32 myBits[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int s0 = j * szFrag + lane * per, s1 = min(j * szFrag + szFrag, s0 + per);
    u32 bits = 0;
    for (int i = s0; i < s1; i++) bits += codes[data[i]] >> 24;
    myBits[j] = bits;
    const u32 tot = kz_wave_sum(bits);
    if (lane == 0) sh_nb[j] = tot;
  }
  __syncthreads();
  // varint x4 (EntropyUtils.writeVarInt) assembled by lane 0 in hbuf
  __syncthreads();
  if (lane == 0) {
    u8* vb = (u8*)hbuf; int k = 0;
    for (int j = 0; j < 4; j++) { u32 v = sh_nb[j]; while (v >= 128) { vb[k++] = (u8)(0x80 | (v & 0x7F)); v >>= 7; } vb[k++] = (u8)v; }
    firstCode[0] = (u32)k;
  }
  __syncthreads();
  { const int vbytes = (int)firstCode[0]; huf_append(scr, pos, (const u8*)hbuf, vbytes * 8); pos += vbytes * 8; }
  // pass 2: pack each fragment in LDS (lanes OR their codes at their bit offsets), append to scr
  for (int j = 0; j < 4; j++) {
    const int nbj = (int)sh_nb[j];
    __syncthreads();
    for (int i = lane; i < (4096 * HUF_MAXLEN) / 32 + 8; i += 64) fragbuf[i] = 0;
    __syncthreads();
    const u32 inc = kz_wave_incl_sum(myBits[j]);
    u32 bp = inc - myBits[j];
    const int s0 = j * szFrag + lane * per, s1 = min(j * szFrag + szFrag, s0 + per);
    for (int i = s0; i < s1; i++) {
      const u32 c = codes[data[i]];
      const u32 cl = c >> 24, cv = c & 0xFFFFFF;
      // place cl bits at stream position bp (MSB first) inside big-endian 32-bit words
      const u32 w = bp >> 5, o = bp & 31;
      const u64 v = (u64)cv << (64 - cl - o);
      atomicOr(&fragbuf[w], (u32)(v >> 32));
      if (o + cl > 32) atomicOr(&fragbuf[w + 1], (u32)v);
      bp += cl;
    }
    __syncthreads();
    // fragbuf holds big-endian words: convert to byte stream order while appending
    {
      u32* o = (u32*)scr;
      const int w0 = pos >> 5, w1 = (pos + nbj - 1) >> 5;
      for (int w = w0 + lane; nbj > 0 && w <= w1; w += 64) {
        const int r = w * 32 - pos;                      // fragment-relative bit of this word's first bit
        // 32 bits of the fragment starting at bit r
        u32 v = 0;
        if (r >= 0) { const int wi = r >> 5, sh = r & 31; v = fragbuf[wi] << sh; if (sh) v |= fragbuf[wi + 1] >> (32 - sh); }
        else { v = fragbuf[0] >> (-r); }
        const int over = r + 32 - nbj;
        if (over > 0) v &= (over >= 32) ? 0u : (0xFFFFFFFFu << over);
        v = __builtin_bswap32(v);
        if (r >= 0 && r + 32 <= nbj) o[w] = v; else atomicOr(&o[w], v);
      }
    }
    pos += nbj;
  }
  // chunk last bytes (:486-492)
  __syncthreads();
  huf_append(scr, pos, data + 4 * szFrag, 8 * (len - 4 * szFrag));
  pos += 8 * (len - 4 * szFrag);
  if (lane == 0) E.tailBits[ci] = (u32)pos;
}

int kz_stage_huffman_encode(kz_ctx* ctx, kz_batch& bt, uint8_t* out, int64_t outStride, const int32_t* d_hdrBytes, int64_t* d_bits) {
  AnsEnc E; int chunks = 0;
  int rc = kz_chunk_enc_alloc(ctx, bt, E, &chunks);
  if (rc) return rc;
  if (chunks > 0) KZ_LAUNCH(ctx, KID_HUF_ENC_CHUNK, k_huf_enc_chunk, dim3(chunks, bt.B), dim3(64), bt.buf[bt.cur], bt.stride, bt.d_len, E);
  return kz_chunk_enc_finish(ctx, bt, E, chunks, out, outStride, d_hdrBytes, d_bits, 0);
}

// =================================================================================================
// decode
struct HufDec { u64* chunkBit; int32_t* status; int C; };

__device__ __forceinline__ u32 huf_peek(const u8* __restrict__ p, u64 pos, int count) {      // count <= 25
  const u64 by = pos >> 3;
  u32 acc = ((u32)p[by] << 24) | ((u32)p[by + 1] << 16) | ((u32)p[by + 2] << 8) | (u32)p[by + 3];
  acc <<= (pos & 7);
  return count ? (acc >> (32 - count)) : 0;
}
// ExpGolombDecoder.java:41-58 (signed) as Java computes it on ANY input: the zeros are counted until a 1 arrives (running off the
// block's bits throws), readBits takes 1..64 bits (more: IllegalArgumentException, DefaultInputBitStream.java:98-99), `1 << log2`
// is an int shift (count mod 32) and the result is cast to BYTE: a damaged header may decode to a small delta after 13 or 40 zeros,
// and the reference then goes on.  Only the low 9 bits of the (log2 + 1)-bit field reach the byte.
__device__ __forceinline__ int huf_eg_get(const u8* __restrict__ p, u64& pos, u64 endBits, bool& bad) {
  if (pos + 1 > endBits) { bad = true; return 0; }
  if (huf_peek(p, pos, 1) == 1) { pos += 1; return 0; }
  pos += 1;
  int log2 = 1;
  for (;;) {
    if (pos + 1 > endBits) { bad = true; return 0; }
    const u32 bit = huf_peek(p, pos, 1);
    pos += 1;
    if (bit) break;
    log2++;
  }
  if (log2 + 1 > 64 || pos + (u64)(log2 + 1) > endBits) { bad = true; return 0; }
  const int take = min(9, log2 + 1);
  const u32 v = huf_peek(p, pos + (u64)(log2 + 1 - take), take);      // the field's last bits
  pos += (u64)(log2 + 1);
  const u32 sgn = v & 1u;
  const u32 k8 = ((log2 & 31) < 8) ? (1u << (log2 & 31)) : 0u;          // low byte of (int)(1 << log2)
  const u32 t = (v >> 1) + k8 - 1u;
  const u32 r = sgn ? ~(t - 1u) : t;                                   // (res - sgn) ^ -sgn, low byte
  return (int)(int8_t)(u8)r;
}
__device__ __forceinline__ u32 huf_varint(const u8* __restrict__ p, u64& pos) {
  u32 v = huf_peek(p, pos, 8); pos += 8;
  u32 r = v & 0x7F; int shift = 7;
  while (v >= 128) { v = huf_peek(p, pos, 8); pos += 8; r |= (v & 0x7F) << shift; if (shift == 28) break; shift += 7; }
  return r;
}

// index pass: one lane per block walks the chunk headers
__global__ void k_huf_dec_index(const u8* __restrict__ in, int64_t inStride, const int64_t* __restrict__ d_bitOff,
                                const int64_t* __restrict__ d_bitEnd, const int32_t* __restrict__ d_len, HufDec D, int B, long long* __restrict__ endOut) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int count = d_len[b];
  const u8* p = in + (int64_t)b * inStride;
  u64 pos = (u64)d_bitOff[b];
  const u64 endBits = (u64)d_bitEnd[b];
  int status = 0;
  const int chunks = (count + ANS_CHUNK - 1) / ANS_CHUNK;
  for (int c = 0; c < chunks && !status; c++) {
    D.chunkBit[(int64_t)b * D.C + c] = pos;
    const int size = min(ANS_CHUNK, count - c * ANS_CHUNK);
    if (size < 32) {                                                // :364-366 bulk read of the raw bytes: past the block's bits it throws
      pos += 8ULL * size;
      if (pos > endBits) status = -KZ_ERR_PROCESS_BLOCK;
      continue;
    }
    if (pos + 2 > endBits) { status = -KZ_ERR_PROCESS_BLOCK; break; }
    int asz;
    if (huf_peek(p, pos, 1) == 0) { asz = (huf_peek(p, pos + 1, 1) == 1) ? 0 : 256; pos += 2; }
    else {
      const int lastMask = (int)huf_peek(p, pos + 1, 5); pos += 6;
      asz = 0;
      for (int i = 0; i <= lastMask; i++) { asz += __popc(huf_peek(p, pos, 8)); pos += 8; }
    }
    if (asz == 0) { status = -KZ_ERR_PROCESS_BLOCK; break; }
    bool bad = false;
    for (int i = 0; i < asz; i++) { (void)huf_eg_get(p, pos, endBits, bad); if (bad) break; }
    if (bad) { status = -KZ_ERR_PROCESS_BLOCK; break; }
    if (asz > 1) {
      u64 tot = 0;
      for (int j = 0; j < 4; j++) tot += huf_varint(p, pos);
      pos += tot + 8ULL * (size - 4 * (size >> 2));
    }
    if (pos > endBits) status = -KZ_ERR_PROCESS_BLOCK;
  }
  if (endOut) endOut[b] = (long long)pos;                          // bits consumed (EntropyDecoder contract)
  D.status[b] = status;
}

__global__ __launch_bounds__(64) void k_huf_dec_chunk(const u8* __restrict__ in, int64_t inStride, const int32_t* __restrict__ d_len,
                                                       HufDec D, u8* __restrict__ dst, int64_t stride, const int64_t* __restrict__ d_bitEnd) {
  const int b = blockIdx.y, ck = blockIdx.x;
  const int count = d_len[b];
  const int lane = kz_lane();
  const int start = ck * ANS_CHUNK;
  if (start >= count || D.status[b] != 0) return;
  const int size = min(ANS_CHUNK, count - start);
  const u8* p = in + (int64_t)b * inStride;
  u8* o = dst + (int64_t)b * stride + start;
  u64 pos = D.chunkBit[(int64_t)b * D.C + ck];
  if (size < 32) { if (lane < size) o[lane] = (u8)huf_peek(p, pos + 8ULL * lane, 8); return; }   // :363-365
  __shared__ u8 sizes[256];
  __shared__ u16 table[1 << HUF_MAXLEN];
  __shared__ int sh_asz, sh_bad, sh_one;
  __shared__ u64 sh_pos;
  for (int i = lane; i < 256; i += 64) sizes[i] = 0;
  __syncthreads();
  if (lane == 0) {                                               // readLengths :115-154
    const u64 endBits = (u64)d_bitEnd[b];
    bool bad = false; int asz = 0, one = 0;
    int cur = 2;
    if (huf_peek(p, pos, 1) == 0) {
      if (huf_peek(p, pos + 1, 1) == 1) asz = 0;
      else { asz = 256; }
      pos += 2;
      if (asz == 256) for (int s = 0; s < 256 && !bad; s++) { cur += huf_eg_get(p, pos, endBits, bad); if (cur <= 0 || cur > HUF_MAXLEN) bad = true; else sizes[s] = (u8)cur; one = s; }
    } else {
      const int lastMask = (int)huf_peek(p, pos + 1, 5); pos += 6;
      u64 mpos = pos;
      pos += 8ULL * (lastMask + 1);
      for (int i = 0; i <= lastMask && !bad; i++) {
        const u32 m = huf_peek(p, mpos, 8); mpos += 8;
        for (int j = 0; j < 8 && !bad; j++) if (m & (1u << j)) {
          cur += huf_eg_get(p, pos, endBits, bad);
          if (cur <= 0 || cur > HUF_MAXLEN) bad = true; else { sizes[(i << 3) + j] = (u8)cur; one = (i << 3) + j; asz++; }
        }
      }
    }
    sh_asz = asz; sh_bad = bad ? 1 : 0; sh_pos = pos; sh_one = one;
  }
  __syncthreads();
  if (sh_bad || sh_asz == 0) { if (lane == 0) atomicExch(&D.status[b], -KZ_ERR_PROCESS_BLOCK); return; }
  if (sh_asz == 1) { const u8 c = (u8)sh_one; for (int i = lane; i < size; i += 64) o[i] = c; return; }   // :374-377
  u32 sz[4], cd[4];
#pragma unroll
  for (int q = 0; q < 4; q++) sz[q] = sizes[q * 64 + lane];
  huf_canonical(sz, cd);
  for (int i = lane; i < (1 << HUF_MAXLEN); i += 64) table[i] = 7;      // :165-166 default
  __syncthreads();
  // Code lengths that over-subscribe the code space give canonical codes past the table: the reference dies on
  // table[idx] there (buildDecodingTables :183-186) -> the block fails; nothing is written out of range here.
  bool over = false;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (!sz[q]) continue;
    const u16 val = (u16)((sz[q] << 8) | (u32)(q * 64 + lane));
    const u32 idx = cd[q] << (HUF_MAXLEN - sz[q]);
    const u32 cnt = 1u << (HUF_MAXLEN - sz[q]);
    if (idx >= (1u << HUF_MAXLEN) || idx + cnt > (1u << HUF_MAXLEN)) { over = true; continue; }
    for (u32 k = 0; k < cnt; k++) table[idx + k] = val;
  }
  if (kz_ballot(over)) { if (lane == 0) atomicExch(&D.status[b], -KZ_ERR_PROCESS_BLOCK); return; }
  __syncthreads();
  pos = sh_pos;
  u32 nbq[4];
  for (int j = 0; j < 4; j++) nbq[j] = huf_varint(p, pos);
  const int szFrag = size >> 2;
  if (lane < 4) {
    u64 fs = pos;
    for (int j = 0; j < lane; j++) fs += nbq[j];
    const u64 fe = fs + nbq[lane];
    u64 bp = fs;
    u8* oo = o + lane * szFrag;
    for (int i = 0; i < szFrag; i++) {
      u32 v = (bp < fe) ? huf_peek(p, bp, HUF_MAXLEN) : 0;
      if (bp + HUF_MAXLEN > fe && bp < fe) { const int over = (int)(bp + HUF_MAXLEN - fe); v &= ~((1u << over) - 1u); }   // zero padded
      const u16 t = table[v];
      oo[i] = (u8)t;
      bp += (u64)(t >> 8);
    }
    // decodeChunk's verdict (HuffmanDecoder.java, the four-way `== szBits` at its end): a fragment's symbols take exactly its stated bits
    if (bp != fe) atomicExch(&D.status[b], -KZ_ERR_PROCESS_BLOCK);
  }
  const u64 tailPos = pos + nbq[0] + nbq[1] + nbq[2] + nbq[3];
  const int rem = size - 4 * szFrag;
  if (lane < rem) o[4 * szFrag + lane] = (u8)huf_peek(p, tailPos + 8ULL * lane, 8);
}

__global__ void k_huf_dec_fin(const int32_t* __restrict__ d_len, int32_t* __restrict__ d_len2, int32_t* __restrict__ d_flag, HufDec D, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  d_len2[b] = d_len[b];
  d_flag[b] = (D.status[b] == 0) ? 1 : 0;
}

int kz_stage_huffman_decode(kz_ctx* ctx, kz_batch& bt, const uint8_t* in, int64_t inStride, const int64_t* d_bitOff, const int64_t* d_bitEnd) {
  const int B = bt.B;
  int maxN = 0;
  for (int b = 0; b < B; b++) if (bt.h_len[b] > maxN) maxN = bt.h_len[b];
  HufDec D;
  D.C = (maxN + ANS_CHUNK - 1) / ANS_CHUNK + 1;
  D.chunkBit = (u64*)kz_arena_alloc(ctx, (size_t)B * D.C * 8);
  D.status = (int32_t*)kz_arena_alloc(ctx, (size_t)B * 4);
  if (!D.status || !D.chunkBit) { snprintf(ctx->err, sizeof(ctx->err), "huffman_decode: arena overflow"); return -KZ_ERR_DEVICE; }
  u8* dst = bt.buf[bt.cur ^ 1];
  KZ_LAUNCH(ctx, KID_HUF_DEC_INDEX, k_huf_dec_index, dim3((B + 63) / 64), dim3(64), in, inStride, d_bitOff, d_bitEnd, bt.d_len, D, B, ctx->d_endBits);
  const int chunks = (maxN + ANS_CHUNK - 1) / ANS_CHUNK;
  if (chunks > 0) KZ_LAUNCH(ctx, KID_HUF_DEC_CHUNK, k_huf_dec_chunk, dim3(chunks, B), dim3(64), in, inStride, bt.d_len, D, dst, bt.stride, d_bitEnd);
  KZ_LAUNCH(ctx, KID_HUF_DEC_FIN, k_huf_dec_fin, dim3((B + 255) / 256), dim3(256), bt.d_len, bt.d_len2, bt.d_flag, D, B);
  KZ_HIP(hipGetLastError());
  bt.cur ^= 1;
  { int32_t* t = bt.d_len; bt.d_len = bt.d_len2; bt.d_len2 = t; }
  return 0;
}
