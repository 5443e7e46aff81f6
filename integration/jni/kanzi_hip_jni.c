/*
 * JNI shim between the Java adapter classes (integration/java/) and the C-ABI of libkanzi_hip.so
 * (include/kanzi_hip.h).  Build on a host with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       kanzi_hip_jni.c -L../../kanzi_amd -lkanzi_hip -o libkanzi_hip_jni.so
 * The build image has no JDK: tests/test_abi.py checks this file with `gcc -fsyntax-only` against
 * integration/jni/stub/jni.h (types and JNIEnv functions per the JNI specification).
 *
 * Ownership: Java owns every array; an array is pinned with GetPrimitiveArrayCritical for the duration of one
 * single-block call only, the library keeps no pointer after returning.  The batched calls take DIRECT ByteBuffers (their
 * address is stable, nothing is pinned while the GPU works) and plain int[] / long[] descriptors copied with Get/Release.
 * Every JNI result is checked for NULL; a failed pin returns -ERR_UNKNOWN (127) or, for the transform call, "declined".
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "kanzi_hip.h"

#define CLS(name) Java_io_github_flanglet_kanzi_hip_KanziHip_##name
#define CTX(h) ((kz_ctx*)(intptr_t)(h))

JNIEXPORT jlong JNICALL CLS(ctxCreate)(JNIEnv* env, jclass c, jint device) {
  (void)env; (void)c;
  return (jlong)(intptr_t)kz_ctx_create(device);
}
JNIEXPORT void JNICALL CLS(ctxDestroy)(JNIEnv* env, jclass c, jlong ctx) {
  (void)env; (void)c;
  kz_ctx_destroy(CTX(ctx));
}
/* context map key "checksum": 0 / 32 / 64 (CompressedOutputStream.java:190-204) */
JNIEXPORT jint JNICALL CLS(ctxSetChecksum)(JNIEnv* env, jclass c, jlong ctx, jint bits) {
  (void)env; (void)c;
  return kz_ctx_set_checksum(CTX(ctx), bits);
}
JNIEXPORT jint JNICALL CLS(ctxSetSkipBlocks)(JNIEnv* env, jclass c, jlong ctx, jboolean on) {
  (void)env; (void)c;
  return kz_ctx_set_skip_blocks(CTX(ctx), on ? 1 : 0);
}
/* context map keys "blockSize" and "entropy" (TEXT reads them: TextCodec.java:561-575, TransformFactory.java:275-286) */
JNIEXPORT jint JNICALL CLS(ctxSetBlockSize)(JNIEnv* env, jclass c, jlong ctx, jint blockSize) {
  (void)env; (void)c;
  return kz_ctx_set_block_size(CTX(ctx), blockSize);
}
JNIEXPORT jint JNICALL CLS(ctxSetEntropy)(JNIEnv* env, jclass c, jlong ctx, jint entropyType) {
  (void)env; (void)c;
  return kz_ctx_set_entropy(CTX(ctx), (uint32_t)entropyType);
}
/* context map key "dataType" (Global.DataType <-> KZ_DT_*: HipByteTransform maps the enum) */
JNIEXPORT jint JNICALL CLS(ctxSetDataType)(JNIEnv* env, jclass c, jlong ctx, jint dataType) {
  (void)env; (void)c;
  return kz_ctx_set_data_type(CTX(ctx), dataType);
}
JNIEXPORT jint JNICALL CLS(ctxGetDataType)(JNIEnv* env, jclass c, jlong ctx) {
  (void)env; (void)c;
  return kz_ctx_get_data_type(CTX(ctx));
}

JNIEXPORT jint JNICALL CLS(ctxReset)(JNIEnv* env, jclass c, jlong ctx) {
  (void)env; (void)c;
  return kz_ctx_reset(CTX(ctx));
}

JNIEXPORT jint JNICALL CLS(maxEncodedLength)(JNIEnv* env, jclass c, jint type, jint n) {
  (void)env; (void)c;
  return kz_transform_max_encoded_len((uint32_t)type, n);
}
/* returns produced length (>=0) when applied, -1000000 when declined, other negatives = -(Error code) */
JNIEXPORT jint JNICALL CLS(transform)(JNIEnv* env, jclass c, jlong ctx, jint type, jboolean forward,
                                      jbyteArray src, jint srcIdx, jint n, jbyteArray dst, jint dstIdx, jint dstCap) {
  (void)c;
  if (src == NULL || dst == NULL || srcIdx < 0 || dstIdx < 0 || n < 0 || dstCap < 0) return -KZ_ERR_INVALID_PARAM;
  if ((jlong)srcIdx + n > (*env)->GetArrayLength(env, src) || (jlong)dstIdx + dstCap > (*env)->GetArrayLength(env, dst)) return -KZ_ERR_INVALID_PARAM;
  jbyte* s = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, src, NULL);
  if (s == NULL) return -KZ_ERR_UNKNOWN;
  jbyte* d = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, dst, NULL);
  if (d == NULL) { (*env)->ReleasePrimitiveArrayCritical(env, src, s, JNI_ABORT); return -KZ_ERR_UNKNOWN; }
  int32_t produced = 0;
  int32_t rc = forward
      ? kz_transform_forward(CTX(ctx), (uint32_t)type, (const uint8_t*)s + srcIdx, n, (uint8_t*)d + dstIdx, dstCap, &produced)
      : kz_transform_inverse(CTX(ctx), (uint32_t)type, (const uint8_t*)s + srcIdx, n, (uint8_t*)d + dstIdx, dstCap, &produced);
  (*env)->ReleasePrimitiveArrayCritical(env, dst, d, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, src, s, JNI_ABORT);
  if (rc == 1) return produced;
  return rc == 0 ? -1000000 : rc;
}
/* returns the number of BITS written into out, or -(Error code) */
JNIEXPORT jlong JNICALL CLS(entropyEncode)(JNIEnv* env, jclass c, jlong ctx, jint type,
                                           jbyteArray block, jint blkptr, jint n, jbyteArray out) {
  (void)c;
  if (block == NULL || out == NULL || blkptr < 0 || n < 0 || (jlong)blkptr + n > (*env)->GetArrayLength(env, block)) return -KZ_ERR_INVALID_PARAM;
  const jsize cap = (*env)->GetArrayLength(env, out);
  jbyte* s = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, block, NULL);
  if (s == NULL) return -KZ_ERR_UNKNOWN;
  jbyte* d = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, out, NULL);
  if (d == NULL) { (*env)->ReleasePrimitiveArrayCritical(env, block, s, JNI_ABORT); return -KZ_ERR_UNKNOWN; }
  int64_t bits = kz_entropy_encode(CTX(ctx), (uint32_t)type, (const uint8_t*)s + blkptr, n, (uint8_t*)d, cap);
  (*env)->ReleasePrimitiveArrayCritical(env, out, d, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, block, s, JNI_ABORT);
  return bits;
}
/* returns count or -(Error code); bitsUsed[0] = bits the codec consumed from in[inOff..] (EntropyDecoder contract) */
JNIEXPORT jint JNICALL CLS(entropyDecode)(JNIEnv* env, jclass c, jlong ctx, jint type, jbyteArray in, jint inOff, jlong inBits,
                                          jbyteArray block, jint blkptr, jint count, jlongArray bitsUsed) {
  (void)c;
  if (in == NULL || block == NULL || inOff < 0 || blkptr < 0 || count < 0 || inBits < 0) return -KZ_ERR_INVALID_PARAM;
  if ((jlong)inOff + ((inBits + 7) >> 3) > (*env)->GetArrayLength(env, in) || (jlong)blkptr + count > (*env)->GetArrayLength(env, block)) return -KZ_ERR_INVALID_PARAM;
  jbyte* s = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, in, NULL);
  if (s == NULL) return -KZ_ERR_UNKNOWN;
  jbyte* d = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, block, NULL);
  if (d == NULL) { (*env)->ReleasePrimitiveArrayCritical(env, in, s, JNI_ABORT); return -KZ_ERR_UNKNOWN; }
  int64_t used = 0;
  int32_t rc = kz_entropy_decode(CTX(ctx), (uint32_t)type, (const uint8_t*)s + inOff, inBits, (uint8_t*)d + blkptr, count, &used);
  (*env)->ReleasePrimitiveArrayCritical(env, block, d, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, in, s, JNI_ABORT);
  if (bitsUsed != NULL) { jlong u = (jlong)used; (*env)->SetLongArrayRegion(env, bitsUsed, 0, 1, &u); }
  return rc;
}

/* ---- fused batched path over direct ByteBuffers (the form that pays: one call per batch of blocks) ---- */
static void publish_results(JNIEnv* env, const kz_block_result* res, jint nBlocks, jlongArray bitsOut, jintArray lenOut, jbyteArray skipFlagsOut) {
  for (jint i = 0; i < nBlocks; i++) {
    const jlong b = res[i].bits;
    const jint pl = res[i].status ? res[i].status : res[i].length;          /* negative = -(Error code) of that block */
    const jbyte sf = (jbyte)res[i].skipFlags;
    if (bitsOut != NULL) (*env)->SetLongArrayRegion(env, bitsOut, i, 1, &b);
    if (lenOut != NULL) (*env)->SetIntArrayRegion(env, lenOut, i, 1, &pl);
    if (skipFlagsOut != NULL) (*env)->SetByteArrayRegion(env, skipFlagsOut, i, 1, &sf);
  }
}
/* the codec span of EncodingTask.encodeBlock for nBlocks blocks (CompressedOutputStream.java:792-985) */
JNIEXPORT jint JNICALL CLS(encodeBlocks)(JNIEnv* env, jclass c, jlong ctx, jlong transformType, jint entropyType,
                                         jobject in, jlong inStride, jintArray lengths, jint nBlocks,
                                         jobject out, jlong outStride, jlongArray bitsOut, jintArray postLenOut, jbyteArray skipFlagsOut) {
  (void)c;
  if (in == NULL || out == NULL || lengths == NULL || nBlocks < 0 || (*env)->GetArrayLength(env, lengths) < nBlocks) return -KZ_ERR_INVALID_PARAM;
  uint8_t* pin = (uint8_t*)(*env)->GetDirectBufferAddress(env, in);
  uint8_t* pout = (uint8_t*)(*env)->GetDirectBufferAddress(env, out);
  if (pin == NULL || pout == NULL) return -KZ_ERR_INVALID_PARAM;            /* not direct buffers */
  if ((*env)->GetDirectBufferCapacity(env, in) < inStride * nBlocks || (*env)->GetDirectBufferCapacity(env, out) < outStride * nBlocks) return -KZ_ERR_INVALID_PARAM;
  jint* len = (*env)->GetIntArrayElements(env, lengths, NULL);
  if (len == NULL) return -KZ_ERR_UNKNOWN;
  kz_block_result* res = (kz_block_result*)calloc((size_t)nBlocks + 1, sizeof(kz_block_result));
  int32_t rc = -KZ_ERR_UNKNOWN;
  if (res != NULL) {
    rc = kz_encode_blocks(CTX(ctx), (uint64_t)transformType, (uint32_t)entropyType, pin, inStride,
                          (const int32_t*)len, nBlocks, pout, outStride, res, KZ_MEM_HOST);
    if (rc == 0) publish_results(env, res, nBlocks, bitsOut, postLenOut, skipFlagsOut);
    free(res);
  }
  (*env)->ReleaseIntArrayElements(env, lengths, len, JNI_ABORT);
  return rc;
}
/* the codec span of DecodingTask.decodeBlock for nBlocks blocks (CompressedInputStream.java:1106-1378): stream b = the W =
   bitLengths[b] bits the reader copied out of the shared stream after walking the length prefixes (:1127-1129), header included;
   decodedLenOut[b] = decoded length, or -(Error code) of that block */
JNIEXPORT jint JNICALL CLS(decodeBlocks)(JNIEnv* env, jclass c, jlong ctx, jlong transformType, jint entropyType, jint blockSize,
                                         jobject in, jlong inStride, jlongArray bitLengths, jint nBlocks,
                                         jobject out, jlong outStride, jintArray decodedLenOut, jbyteArray skipFlagsOut) {
  (void)c;
  if (in == NULL || out == NULL || bitLengths == NULL || nBlocks < 0 || (*env)->GetArrayLength(env, bitLengths) < nBlocks) return -KZ_ERR_INVALID_PARAM;
  uint8_t* pin = (uint8_t*)(*env)->GetDirectBufferAddress(env, in);
  uint8_t* pout = (uint8_t*)(*env)->GetDirectBufferAddress(env, out);
  if (pin == NULL || pout == NULL) return -KZ_ERR_INVALID_PARAM;
  if ((*env)->GetDirectBufferCapacity(env, in) < inStride * nBlocks || (*env)->GetDirectBufferCapacity(env, out) < outStride * nBlocks) return -KZ_ERR_INVALID_PARAM;
  jlong* bits = (*env)->GetLongArrayElements(env, bitLengths, NULL);
  if (bits == NULL) return -KZ_ERR_UNKNOWN;
  kz_block_result* res = (kz_block_result*)calloc((size_t)nBlocks + 1, sizeof(kz_block_result));
  int32_t rc = -KZ_ERR_UNKNOWN;
  if (res != NULL) {
    rc = kz_decode_blocks(CTX(ctx), (uint64_t)transformType, (uint32_t)entropyType, blockSize, pin, inStride,
                          (const int64_t*)bits, nBlocks, pout, outStride, res, KZ_MEM_HOST);
    if (rc == 0) publish_results(env, res, nBlocks, NULL, decodedLenOut, skipFlagsOut);
    free(res);
  }
  (*env)->ReleaseLongArrayElements(env, bitLengths, bits, JNI_ABORT);
  return rc;
}
