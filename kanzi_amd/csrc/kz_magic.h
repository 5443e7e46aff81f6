// kz_magic.h -- Magic.getType (K/Magic.java:147-185) and its three classes, for device kernels and host stages alike.
// Java int semantics: arithmetic shifts, exact match for JPG.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__host__ __device__ __forceinline__ int32_t mm_magic_type(const uint8_t* p) {
  const int32_t key = (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
  if ((key & ~0x0F) == (int32_t)0xFFD8FFE0u) return key;                       // JPG
  if ((key >> 8) == 0x425A68 || (key >> 8) == 0x494433) return key >> 8;       // BZIP2, MP3 ID3
  const uint32_t k = (uint32_t)key;
  if (k == 0x47494638u || k == 0x25504446u || k == 0x504B0304u || k == 0x377ABCAFu || k == 0x89504E47u || k == 0x7F454C46u ||
      k == 0xFEEDFACEu || k == 0xCEFAEDFEu || k == 0xFEEDFACFu || k == 0xCFFAEDFEu || k == 0x28B52FFDu || k == 0x81CFB2CEu ||
      k == 0x4D534346u || k == 0x52494646u || k == 0x664C6143u || k == 0xFD377A58u || k == 0x4B414E5Au || k == 0x52617221u) return key;
  const int32_t key16 = key >> 16;
  if (key16 == 0x1F8B || key16 == 0x424D || key16 == 0x4D5A) return key16;       // GZIP, BMP, WIN
  if (key16 == 0x5034 || key16 == 0x5035 || key16 == 0x5036) {                   // PBM, PGM, PPM (binary flavours)
    const int sub = (key >> 8) & 0xFF;
    if (sub == 0x07 || sub == 0x0A || sub == 0x0D || sub == 0x20) return key16;
  }
  return 0;                                                                      // NO_MAGIC
}
__host__ __device__ __forceinline__ bool mm_is_compressed(int32_t m) {
  const uint32_t k = (uint32_t)m;
  return k == 0xFFD8FFE0u || k == 0x47494638u || k == 0x89504E47u || k == 0x377ABCAFu || k == 0x28B52FFDu || k == 0x81CFB2CEu ||
         k == 0x4D534346u || k == 0x504B0304u || k == 0x1F8Bu || k == 0x425A68u || k == 0x664C6143u || k == 0x494433u ||
         k == 0xFD377A58u || k == 0x4B414E5Au || k == 0x52617221u;
}
__host__ __device__ __forceinline__ bool mm_is_multimedia(int32_t m) {
  const uint32_t k = (uint32_t)m;
  return k == 0xFFD8FFE0u || k == 0x47494638u || k == 0x89504E47u || k == 0x52494646u || k == 0x664C6143u || k == 0x494433u ||
         k == 0x424Du || k == 0x5034u || k == 0x5035u || k == 0x5036u;
}
__host__ __device__ __forceinline__ bool mm_is_executable(int32_t m) {
  const uint32_t k = (uint32_t)m;
  return k == 0x7F454C46u || k == 0x4D5Au || k == 0xFEEDFACEu || k == 0xCEFAEDFEu || k == 0xFEEDFACFu || k == 0xCFFAEDFEu;
}
