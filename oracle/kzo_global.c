/*
 * kzo_global.c -- ORACLE (test infrastructure only; see kzo.h).  Helpers shared by the transforms:
 *   K/Global.java:222-235 (log2_1024), :440-456 (computeFirstOrderEntropy1024), :556-605 (detectSimpleType),
 *   K/Magic.java (getType / isCompressed / isMultimedia / isExecutable).
 */
#include "kzo.h"
#include <math.h>
#include <pthread.h>

/* Global.LOG2_4096[x] = round(4096 * log2(x)) for x in 1..256 (entry 0 is 0).  Generated, not transcribed; pinned by
 * tests/test_oracle.py against entries read off Global.java:104-127. */
static int g_log2_4096[257];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_tables(void) {
  g_log2_4096[0] = 0;
  for (int x = 1; x <= 256; x++) g_log2_4096[x] = (int)floor(4096.0 * log2((double)x) + 0.5);
}
int kzo_log2_4096(int x) { pthread_once(&g_once, init_tables); return g_log2_4096[x]; }

int kzo_log2_1024(int x) {                                   /* Global.java:222-235; x > 0 */
  pthread_once(&g_once, init_tables);
  if (x < 256) return (g_log2_4096[x] + 2) >> 2;
  const int lg = 31 - __builtin_clz((uint32_t)x);
  if ((x & (x - 1)) == 0) return lg << 10;
  return ((lg - 7) * 1024) + ((g_log2_4096[x >> (lg - 7)] + 2) >> 2);
}

int kzo_entropy1024(int length, const int* histo) {          /* Global.java:440-456 */
  if (length == 0) return 0;
  int64_t sum = 0;
  const int logLength1024 = kzo_log2_1024(length);
  for (int i = 0; i < 256; i++) {
    if (histo[i] == 0) continue;
    const int64_t count = histo[i];
    sum += (count * (int64_t)(logLength1024 - kzo_log2_1024(histo[i]))) >> 3;
  }
  return (int)(sum / length);
}

int kzo_detect_simple_type(int count, const int* freqs0) {   /* Global.java:556-605 */
  if (count == 0) return KZO_DT_UNDEFINED;
  static const char DNA[] = "acgntuACGNTU";
  static const char NUM[] = "0123456789+-*/=,.:; ";
  static const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  int sum = 0;
  for (int i = 0; i < 12; i++) sum += freqs0[(uint8_t)DNA[i]];
  if (sum > count - count / 12) return KZO_DT_DNA;
  sum = 0;
  for (int i = 0; i < 20; i++) sum += freqs0[(uint8_t)NUM[i]];
  if (sum == count) return KZO_DT_NUMERIC;
  sum = (freqs0[0x3D] == 1) ? 1 : 0;                         /* trailing '=' padding */
  for (int i = 0; i < 64; i++) sum += freqs0[(uint8_t)B64[i]];
  if (sum == count) return KZO_DT_BASE64;
  sum = 0;
  for (int i = 0; i < 256; i++) sum += (freqs0[i] > 0) ? 1 : 0;
  if (sum == 256) return KZO_DT_BIN;
  if (sum <= 4) return KZO_DT_SMALL_ALPHABET;
  return KZO_DT_UNDEFINED;
}

/* Magic.getType (K/Magic.java): the int it returns is a Java int, kept as int32_t so that the arithmetic shifts and
 * the (only exact-match) JPG test behave the same. */
int32_t kzo_magic_type(const uint8_t* src) {
  const int32_t key = (int32_t)(((uint32_t)src[0] << 24) | ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3]);
  if ((key & ~0x0F) == (int32_t)KZO_MAGIC_JPG) return key;
  if (((key >> 8) == KZO_MAGIC_BZIP2) || ((key >> 8) == KZO_MAGIC_MP3_ID3)) return key >> 8;
  static const uint32_t KEYS32[] = { 0x47494638u /*GIF*/, 0x25504446u /*PDF*/, 0x504B0304u /*ZIP*/, 0x377ABCAFu /*LZMA*/,
    0x89504E47u /*PNG*/, 0x7F454C46u /*ELF*/, 0xFEEDFACEu, 0xCEFAEDFEu, 0xFEEDFACFu, 0xCFFAEDFEu /*Mach-O*/,
    0x28B52FFDu /*ZSTD*/, 0x81CFB2CEu /*BROTLI*/, 0x4D534346u /*CAB*/, KZO_MAGIC_RIFF, 0x664C6143u /*FLAC*/,
    0xFD377A58u /*XZ*/, 0x4B414E5Au /*KNZ*/, 0x52617221u /*RAR*/ };
  for (unsigned i = 0; i < sizeof(KEYS32) / sizeof(KEYS32[0]); i++) if (key == (int32_t)KEYS32[i]) return key;
  const int32_t key16 = key >> 16;
  if (key16 == 0x1F8B /*GZIP*/ || key16 == KZO_MAGIC_BMP || key16 == 0x4D5A /*WIN*/) return key16;
  if (key16 == KZO_MAGIC_PBM || key16 == KZO_MAGIC_PGM || key16 == KZO_MAGIC_PPM) {
    const int subkey = (key >> 8) & 0xFF;
    if (subkey == 0x07 || subkey == 0x0A || subkey == 0x0D || subkey == 0x20) return key16;
  }
  return 0;                                                  /* NO_MAGIC */
}
int kzo_magic_is_compressed(int32_t m) {
  switch ((uint32_t)m) {
    case KZO_MAGIC_JPG: case 0x47494638u: case 0x89504E47u: case 0x377ABCAFu: case 0x28B52FFDu: case 0x81CFB2CEu:
    case 0x4D534346u: case 0x504B0304u: case 0x1F8Bu: case KZO_MAGIC_BZIP2: case 0x664C6143u: case KZO_MAGIC_MP3_ID3:
    case 0xFD377A58u: case 0x4B414E5Au: case 0x52617221u: return 1;
    default: return 0;
  }
}
int kzo_magic_is_multimedia(int32_t m) {
  switch ((uint32_t)m) {
    case KZO_MAGIC_JPG: case 0x47494638u: case 0x89504E47u: case KZO_MAGIC_RIFF: case 0x664C6143u: case KZO_MAGIC_MP3_ID3:
    case KZO_MAGIC_BMP: case KZO_MAGIC_PBM: case KZO_MAGIC_PGM: case KZO_MAGIC_PPM: return 1;
    default: return 0;
  }
}
int kzo_magic_is_executable(int32_t m) {
  switch ((uint32_t)m) {
    case 0x7F454C46u: case 0x4D5Au: case 0xFEEDFACEu: case 0xCEFAEDFEu: case 0xFEEDFACFu: case 0xCFFAEDFEu: return 1;
    default: return 0;
  }
}
/* the writer's per-block tag (CompressedOutputStream.java:795-804): only these three values are ever set here */
int kzo_block_data_type(const uint8_t* data, int n) {
  if (n < 4) return KZO_DT_UNDEFINED;
  const int32_t m = kzo_magic_type(data);
  if (kzo_magic_is_compressed(m)) return KZO_DT_BIN;
  if (kzo_magic_is_multimedia(m)) return KZO_DT_MULTIMEDIA;
  if (kzo_magic_is_executable(m)) return KZO_DT_EXE;
  return KZO_DT_UNDEFINED;
}
