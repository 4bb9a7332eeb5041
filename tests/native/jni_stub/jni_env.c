/* A small FUNCTIONAL JNIEnv over the declarations of jni.h in this directory: enough of the JNI for java/jni/gsgpu_jni.c to run without
 * a JVM, so that tests/test_gpu_jni.py can drive the shim's file-level entry points on the GPU box (the image has no JDK).  Objects are
 * tagged C structs; ThrowNew records the message.  TEST INFRASTRUCTURE: the product never links against it. */
#include "jni.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { K_STRING = 1, K_BUFFER, K_LONGS, K_INTS, K_OBJECTS, K_CLASS };
typedef struct {
    int kind;
    jsize n;
    void *p; /* char* / buffer address / jlong* / jint* / jobject* */
} Obj;

static char g_exception[2048];
static int g_thrown;

static jclass s_FindClass(JNIEnv *e, const char *name) {
    static Obj cls = {K_CLASS, 0, NULL};
    (void)e;
    (void)name;
    return &cls;
}
static jint s_ThrowNew(JNIEnv *e, jclass c, const char *msg) {
    (void)e;
    (void)c;
    snprintf(g_exception, sizeof g_exception, "%s", msg ? msg : "");
    g_thrown = 1;
    return 0;
}
static void *s_GetDirectBufferAddress(JNIEnv *e, jobject b) {
    (void)e;
    return b ? ((Obj *)b)->p : NULL;
}
static jobject s_NewDirectByteBuffer(JNIEnv *e, void *addr, jlong cap) {
    (void)e;
    Obj *o = (Obj *)calloc(1, sizeof(Obj));
    o->kind = K_BUFFER;
    o->n = (jsize)cap;
    o->p = addr;
    return o;
}
static jsize s_GetArrayLength(JNIEnv *e, jarray a) {
    (void)e;
    return ((Obj *)a)->n;
}
static jlong *s_GetLongArrayElements(JNIEnv *e, jlongArray a, jboolean *copy) {
    (void)e;
    if (copy) *copy = 0;
    return (jlong *)((Obj *)a)->p;
}
static void s_ReleaseLongArrayElements(JNIEnv *e, jlongArray a, jlong *p, jint mode) {
    (void)e;
    (void)a;
    (void)p;
    (void)mode; /* (the elements are the array: nothing to copy back) */
}
static void s_SetLongArrayRegion(JNIEnv *e, jlongArray a, jsize at, jsize n, const jlong *v) {
    (void)e;
    memcpy((jlong *)((Obj *)a)->p + at, v, sizeof(jlong) * (size_t)n);
}
static const char *s_GetStringUTFChars(JNIEnv *e, jstring s, jboolean *copy) {
    (void)e;
    if (copy) *copy = 0;
    return (const char *)((Obj *)s)->p;
}
static void s_ReleaseStringUTFChars(JNIEnv *e, jstring s, const char *p) {
    (void)e;
    (void)s;
    (void)p;
}
static jobject s_GetObjectArrayElement(JNIEnv *e, jobjectArray a, jsize i) {
    (void)e;
    return ((jobject *)((Obj *)a)->p)[i];
}
static void s_DeleteLocalRef(JNIEnv *e, jobject o) {
    (void)e;
    (void)o; /* (the test owns its objects) */
}
static jint *s_GetIntArrayElements(JNIEnv *e, jintArray a, jboolean *copy) {
    (void)e;
    if (copy) *copy = 0;
    return (jint *)((Obj *)a)->p;
}
static void s_ReleaseIntArrayElements(JNIEnv *e, jintArray a, jint *p, jint mode) {
    (void)e;
    (void)a;
    (void)p;
    (void)mode;
}
static jstring s_NewStringUTF(JNIEnv *e, const char *s) {
    (void)e;
    Obj *o = (Obj *)calloc(1, sizeof(Obj));
    o->kind = K_STRING;
    o->p = strdup(s ? s : "");
    o->n = (jsize)strlen((const char *)o->p);
    return o;
}

static const struct JNINativeInterface_ g_table = {
    s_FindClass,          s_ThrowNew,           s_GetDirectBufferAddress, s_NewDirectByteBuffer,    s_GetArrayLength,
    s_GetLongArrayElements, s_ReleaseLongArrayElements, s_SetLongArrayRegion, s_GetStringUTFChars, s_ReleaseStringUTFChars,
    s_GetObjectArrayElement, s_DeleteLocalRef,  s_GetIntArrayElements,   s_ReleaseIntArrayElements, s_NewStringUTF,
};
static JNIEnv g_env = &g_table;

/* ---- what the test uses to make and read the "Java" objects ---- */
JNIEnv *stub_env(void) { return &g_env; }
void *stub_string(const char *s) { return s_NewStringUTF(&g_env, s); }
const char *stub_string_chars(void *s) { return s ? (const char *)((Obj *)s)->p : NULL; }
void *stub_buffer(void *addr, jlong cap) { return s_NewDirectByteBuffer(&g_env, addr, cap); }
void *stub_long_array(jsize n) {
    Obj *o = (Obj *)calloc(1, sizeof(Obj));
    o->kind = K_LONGS;
    o->n = n;
    o->p = calloc((size_t)n + 1, sizeof(jlong));
    return o;
}
jlong stub_long_array_get(void *a, jsize i) { return ((jlong *)((Obj *)a)->p)[i]; }
void stub_long_array_set(void *a, jsize i, jlong v) { ((jlong *)((Obj *)a)->p)[i] = v; }
void *stub_int_array(jsize n) {
    Obj *o = (Obj *)calloc(1, sizeof(Obj));
    o->kind = K_INTS;
    o->n = n;
    o->p = calloc((size_t)n + 1, sizeof(jint));
    return o;
}
void stub_int_array_set(void *a, jsize i, jint v) { ((jint *)((Obj *)a)->p)[i] = v; }
void *stub_object_array(jsize n) {
    Obj *o = (Obj *)calloc(1, sizeof(Obj));
    o->kind = K_OBJECTS;
    o->n = n;
    o->p = calloc((size_t)n + 1, sizeof(jobject));
    return o;
}
void stub_object_array_set(void *a, jsize i, void *v) { ((jobject *)((Obj *)a)->p)[i] = v; }
/* the pending "exception": its message, or NULL; cleared by the call */
const char *stub_take_exception(void) {
    if (!g_thrown) return NULL;
    g_thrown = 0;
    return g_exception;
}
