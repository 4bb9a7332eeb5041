/* Declarations-only stand-in for the JDK's <jni.h>, just enough to let `gcc -fsyntax-only` look at java/jni/gsgpu_jni.c
 * in an image without a JDK (tests/test_java_glue_cpu.py).  TEST INFRASTRUCTURE: nothing links against it. */
#ifndef GS_JNI_STUB_H
#define GS_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;
typedef void *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jlongArray;
typedef jarray jintArray;
typedef jarray jobjectArray;
typedef jobject jthrowable;
#define JNIEXPORT
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    void *(*GetDirectBufferAddress)(JNIEnv *, jobject);
    jobject (*NewDirectByteBuffer)(JNIEnv *, void *, jlong);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
    jobject (*GetObjectArrayElement)(JNIEnv *, jobjectArray, jsize);
    void (*DeleteLocalRef)(JNIEnv *, jobject);
    jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
    void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
    jstring (*NewStringUTF)(JNIEnv *, const char *);
};
#endif
