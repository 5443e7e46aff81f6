/*
 * JNI shim between the Java adapter classes (integration/java/) and the C-ABI of libkanzi_hip.so
 * (include/kanzi_hip.h).  NOT compiled in this image (no JDK / jni.h): build on a host with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       kanzi_hip_jni.c -L../../kanzi_amd -lkanzi_hip -o libkanzi_hip_jni.so
 * Arrays are pinned with GetPrimitiveArrayCritical for the duration of one call only; the library
 * keeps no pointer after returning (ownership rule of SURVEY 8b).
 */
#include <jni.h>
#include <stdint.h>
#include "kanzi_hip.h"

#define CLS(name) Java_io_github_flanglet_kanzi_hip_KanziHip_##name

JNIEXPORT jlong JNICALL CLS(ctxCreate)(JNIEnv* env, jclass c, jint device) {
  (void)env; (void)c;
  return (jlong)(intptr_t)kz_ctx_create(device);
}
JNIEXPORT void JNICALL CLS(ctxDestroy)(JNIEnv* env, jclass c, jlong ctx) {
  (void)env; (void)c;
  kz_ctx_destroy((kz_ctx*)(intptr_t)ctx);
}
/* context map key "checksum": 0 / 32 / 64 (CompressedOutputStream.java:190-204) */
JNIEXPORT jint JNICALL CLS(ctxSetChecksum)(JNIEnv* env, jclass c, jlong ctx, jint bits) {
  (void)env; (void)c;
  return kz_ctx_set_checksum((kz_ctx*)(intptr_t)ctx, bits);
}
JNIEXPORT jint JNICALL CLS(ctxSetSkipBlocks)(JNIEnv* env, jclass c, jlong ctx, jboolean on) {
  (void)env; (void)c;
  return kz_ctx_set_skip_blocks((kz_ctx*)(intptr_t)ctx, on ? 1 : 0);
}
/* context map key "dataType" (Global.DataType <-> KZ_DT_*): read by MM and LZ/LZX forward, rewritten by MM */
JNIEXPORT jint JNICALL CLS(ctxSetDataType)(JNIEnv* env, jclass c, jlong ctx, jint dataType) {
  (void)env; (void)c;
  return kz_ctx_set_data_type((kz_ctx*)(intptr_t)ctx, dataType);
}
JNIEXPORT jint JNICALL CLS(ctxGetDataType)(JNIEnv* env, jclass c, jlong ctx) {
  (void)env; (void)c;
  return kz_ctx_get_data_type((kz_ctx*)(intptr_t)ctx);
}

JNIEXPORT jint JNICALL CLS(maxEncodedLength)(JNIEnv* env, jclass c, jint type, jint n) {
  (void)env; (void)c;
  return kz_transform_max_encoded_len((uint32_t)type, n);
}
/* returns produced length (>=0) when applied, -1000000 when declined, other negatives = -(Error code) */
JNIEXPORT jint JNICALL CLS(transform)(JNIEnv* env, jclass c, jlong ctx, jint type, jboolean forward,
                                      jbyteArray src, jint srcIdx, jint n, jbyteArray dst, jint dstIdx, jint dstCap) {
  (void)c;
  jbyte* s = (*env)->GetPrimitiveArrayCritical(env, src, NULL);
  jbyte* d = (*env)->GetPrimitiveArrayCritical(env, dst, NULL);
  int32_t produced = 0;
  int32_t rc = forward
      ? kz_transform_forward((kz_ctx*)(intptr_t)ctx, (uint32_t)type, (const uint8_t*)s + srcIdx, n, (uint8_t*)d + dstIdx, dstCap, &produced)
      : kz_transform_inverse((kz_ctx*)(intptr_t)ctx, (uint32_t)type, (const uint8_t*)s + srcIdx, n, (uint8_t*)d + dstIdx, dstCap, &produced);
  (*env)->ReleasePrimitiveArrayCritical(env, dst, d, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, src, s, JNI_ABORT);
  if (rc == 1) return produced;
  return rc == 0 ? -1000000 : rc;
}
/* returns the number of BITS written into out, or -(Error code) */
JNIEXPORT jlong JNICALL CLS(entropyEncode)(JNIEnv* env, jclass c, jlong ctx, jint type,
                                           jbyteArray block, jint blkptr, jint n, jbyteArray out) {
  (void)c;
  jsize cap = (*env)->GetArrayLength(env, out);
  jbyte* s = (*env)->GetPrimitiveArrayCritical(env, block, NULL);
  jbyte* d = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
  int64_t bits = kz_entropy_encode((kz_ctx*)(intptr_t)ctx, (uint32_t)type, (const uint8_t*)s + blkptr, n, (uint8_t*)d, cap);
  (*env)->ReleasePrimitiveArrayCritical(env, out, d, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, block, s, JNI_ABORT);
  return bits;
}
/* returns count or -(Error code); bitsUsed[0] = bits the codec consumed from in[inOff..] (EntropyDecoder contract) */
JNIEXPORT jint JNICALL CLS(entropyDecode)(JNIEnv* env, jclass c, jlong ctx, jint type, jbyteArray in, jint inOff, jlong inBits,
                                          jbyteArray block, jint blkptr, jint count, jlongArray bitsUsed) {
  (void)c;
  jbyte* s = (*env)->GetPrimitiveArrayCritical(env, in, NULL);
  jbyte* d = (*env)->GetPrimitiveArrayCritical(env, block, NULL);
  int64_t used = 0;
  int32_t rc = kz_entropy_decode((kz_ctx*)(intptr_t)ctx, (uint32_t)type, (const uint8_t*)s + inOff, inBits, (uint8_t*)d + blkptr, count, &used);
  (*env)->ReleasePrimitiveArrayCritical(env, block, d, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, in, s, JNI_ABORT);
  if (bitsUsed != NULL) { jlong u = (jlong)used; (*env)->SetLongArrayRegion(env, bitsUsed, 0, 1, &u); }
  return rc;
}
/* Fused batched path over direct ByteBuffers (pinned host memory owned by Java): the form that pays. */
JNIEXPORT jint JNICALL CLS(encodeBlocks)(JNIEnv* env, jclass c, jlong ctx, jlong transformType, jint entropyType,
                                         jobject in, jlong inStride, jintArray lengths, jint nBlocks,
                                         jobject out, jlong outStride, jlongArray bitsOut, jintArray postLenOut, jbyteArray skipFlagsOut) {
  (void)c;
  uint8_t* pin = (*env)->GetDirectBufferAddress(env, in);
  uint8_t* pout = (*env)->GetDirectBufferAddress(env, out);
  jint* len = (*env)->GetIntArrayElements(env, lengths, NULL);
  kz_block_result* res = (kz_block_result*)calloc((size_t)nBlocks, sizeof(kz_block_result));
  int32_t rc = kz_encode_blocks((kz_ctx*)(intptr_t)ctx, (uint64_t)transformType, (uint32_t)entropyType, pin, inStride,
                                (const int32_t*)len, nBlocks, pout, outStride, res, KZ_MEM_HOST);
  if (rc == 0) {
    for (jint i = 0; i < nBlocks; i++) {
      jlong b = res[i].bits; jint pl = res[i].status ? res[i].status : res[i].length; jbyte sf = (jbyte)res[i].skipFlags;
      (*env)->SetLongArrayRegion(env, bitsOut, i, 1, &b);
      (*env)->SetIntArrayRegion(env, postLenOut, i, 1, &pl);
      (*env)->SetByteArrayRegion(env, skipFlagsOut, i, 1, &sf);
    }
  }
  free(res);
  (*env)->ReleaseIntArrayElements(env, lengths, len, JNI_ABORT);
  return rc;
}
