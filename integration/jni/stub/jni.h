/* STUB of <jni.h> for syntax checking integration/jni/kanzi_hip_jni.c in an image without a JDK
 * (tests/test_abi.py compiles the shim against it with -fsyntax-only so that it cannot rot).  It declares only the
 * JNI types and JNIEnv functions the shim uses, with the signatures of the JNI specification (Java SE 8+, jni.h).
 * NEVER link against this: build the real shim with $JAVA_HOME/include. */
#ifndef KZ_STUB_JNI_H
#define KZ_STUB_JNI_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jsize (JNICALL* GetArrayLength)(JNIEnv* env, jarray array);
  void* (JNICALL* GetPrimitiveArrayCritical)(JNIEnv* env, jarray array, jboolean* isCopy);
  void (JNICALL* ReleasePrimitiveArrayCritical)(JNIEnv* env, jarray array, void* carray, jint mode);
  void* (JNICALL* GetDirectBufferAddress)(JNIEnv* env, jobject buf);
  jlong (JNICALL* GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
  jint* (JNICALL* GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
  void (JNICALL* ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
  jlong* (JNICALL* GetLongArrayElements)(JNIEnv* env, jlongArray array, jboolean* isCopy);
  void (JNICALL* ReleaseLongArrayElements)(JNIEnv* env, jlongArray array, jlong* elems, jint mode);
  void (JNICALL* SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
  void (JNICALL* SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
  void (JNICALL* SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
};
#endif
