/*
 * gsgpu_jni.c -- JNI shim between org.metagene.genestrip.gpu.GsGpuNative and the C ABI (include/gsgpu.h).
 * Not built in the build container (no JDK / jni.h); tests/test_java_glue_cpu.py syntax-checks it against a minimal
 * stand-in for jni.h (tests/native/jni_stub/jni.h: declarations only).  Build on a host with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       -o libgsgpu_jni.so gsgpu_jni.c -L../../genestrip_amd -lgsgpu
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>

#include "gsgpu.h"

#define JNAME(n) Java_org_metagene_genestrip_gpu_GsGpuNative_##n

static void throw_gs(JNIEnv *env, int rc) {
    char msg[512];
    snprintf(msg, sizeof(msg), "gsgpu error %d (%s): %s", rc, gs_strerror(rc), gs_last_error());
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), msg);
}

static void *addr(JNIEnv *env, jobject buf) { return buf ? (*env)->GetDirectBufferAddress(env, buf) : NULL; }

JNIEXPORT jint JNICALL JNAME(deviceCount)(JNIEnv *env, jclass c) {
    int n = 0;
    gs_device_count(&n);
    return n;
}

JNIEXPORT jlong JNICALL JNAME(dbCreate)(JNIEnv *env, jclass c, jint device, jint k, jlong n, jobject kmers, jobject vidx,
                                        jint nValues, jobject parentVi) {
    gs_db *db = NULL;
    int rc = gs_db_create(&db, device, k, n, (const int64_t *)addr(env, kmers), (const int32_t *)addr(env, vidx), nValues,
                          (const int32_t *)addr(env, parentVi));
    if (rc) throw_gs(env, rc);
    return (jlong)(intptr_t)db;
}

JNIEXPORT void JNICALL JNAME(dbCreateStriped)(JNIEnv *env, jclass c, jlongArray devicesThenHandles, jint k, jlong n, jobject kmers,
                                              jobject vidx, jint nValues, jobject parentVi) {
    jsize ns = (*env)->GetArrayLength(env, devicesThenHandles);
    jlong *h = (*env)->GetLongArrayElements(env, devicesThenHandles, NULL);
    int devices[GS_MAX_STRIPES];
    gs_db *dbs[GS_MAX_STRIPES];
    int rc = (ns < 2 || ns > GS_MAX_STRIPES) ? GS_E_INVALID : GS_OK;
    for (jsize i = 0; i < ns && rc == GS_OK; i++) devices[i] = (int)h[i];
    if (rc == GS_OK)
        rc = gs_db_create_striped(dbs, devices, (int)ns, k, n, (const int64_t *)addr(env, kmers), (const int32_t *)addr(env, vidx),
                                  nValues, (const int32_t *)addr(env, parentVi));
    for (jsize i = 0; i < ns && rc == GS_OK; i++) h[i] = (jlong)(intptr_t)dbs[i];
    (*env)->ReleaseLongArrayElements(env, devicesThenHandles, h, rc == GS_OK ? 0 : JNI_ABORT);
    if (rc) throw_gs(env, rc);
}

JNIEXPORT jlong JNICALL JNAME(dbBuildBegin)(JNIEnv *env, jclass c, jint device, jint k, jint nValues, jobject parentVi,
                                            jboolean lowerCaseBases, jint maxDust, jint stepSize) {
    gs_dbbuild *b = NULL;
    int rc = gs_dbbuild_begin(&b, device, k, nValues, (const int32_t *)addr(env, parentVi), lowerCaseBases ? 1 : 0, maxDust, stepSize);
    if (rc) throw_gs(env, rc);
    return (jlong)(intptr_t)b;
}

JNIEXPORT void JNICALL JNAME(dbBuildAdd)(JNIEnv *env, jclass c, jlong builder, jobject bases, jobject offsets, jobject nodeVi,
                                         jlong nRegions, jboolean update) {
    int rc = gs_dbbuild_add((gs_dbbuild *)(intptr_t)builder, (const uint8_t *)addr(env, bases), (const uint64_t *)addr(env, offsets),
                            (const int32_t *)addr(env, nodeVi), nRegions, GS_MEM_HOST, update ? 1 : 0);
    if (rc) throw_gs(env, rc);
}

JNIEXPORT jlong JNICALL JNAME(dbBuildFinish)(JNIEnv *env, jclass c, jlong builder) {
    int64_t n = 0;
    int rc = gs_dbbuild_finish((gs_dbbuild *)(intptr_t)builder, &n);
    if (rc) throw_gs(env, rc);
    return (jlong)n;
}

JNIEXPORT void JNICALL JNAME(dbBuildFetch)(JNIEnv *env, jclass c, jlong builder, jobject kmers, jobject valueIdx) {
    int rc = gs_dbbuild_fetch((gs_dbbuild *)(intptr_t)builder, (int64_t *)addr(env, kmers), (int32_t *)addr(env, valueIdx));
    if (rc) throw_gs(env, rc);
}

JNIEXPORT jlong JNICALL JNAME(dbBuildToDb)(JNIEnv *env, jclass c, jlong builder) {
    gs_db *db = NULL;
    int rc = gs_dbbuild_to_db((gs_dbbuild *)(intptr_t)builder, &db);
    if (rc) throw_gs(env, rc);
    return (jlong)(intptr_t)db;
}

JNIEXPORT void JNICALL JNAME(dbBuildDestroy)(JNIEnv *env, jclass c, jlong builder) { gs_dbbuild_destroy((gs_dbbuild *)(intptr_t)builder); }

JNIEXPORT void JNICALL JNAME(dbDestroy)(JNIEnv *env, jclass c, jlong db) { gs_db_destroy((gs_db *)(intptr_t)db); }

JNIEXPORT void JNICALL JNAME(dbSave)(JNIEnv *env, jclass c, jlong db, jstring path) {
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    int rc = gs_db_save((gs_db *)(intptr_t)db, p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc) throw_gs(env, rc);
}

JNIEXPORT jlong JNICALL JNAME(dbLoad)(JNIEnv *env, jclass c, jint device, jstring path) {
    gs_db *db = NULL;
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    int rc = gs_db_load(&db, device, p);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc) throw_gs(env, rc);
    return (jlong)(intptr_t)db;
}

JNIEXPORT jlong JNICALL JNAME(matchBegin)(JNIEnv *env, jclass c, jlong db, jboolean classify, jboolean countUnique,
                                          jint maxPaths, jint threshold, jdouble taxErr, jdouble classErr, jint maxKmerResCounts) {
    gs_match_cfg cfg = {classify ? 1 : 0, countUnique ? 1 : 0, maxPaths, threshold, taxErr, classErr, 0, maxKmerResCounts};
    gs_run *run = NULL;
    int rc = gs_match_begin(&run, (gs_db *)(intptr_t)db, &cfg);
    if (rc) throw_gs(env, rc);
    return (jlong)(intptr_t)run;
}

JNIEXPORT void JNICALL JNAME(matchSubmit)(JNIEnv *env, jclass c, jlong run, jobject seq, jobject offsets, jlong nReads,
                                          jlong firstReadNo, jobject classVi, jobject flags) {
    int rc = gs_match_submit((gs_run *)(intptr_t)run, (const uint8_t *)addr(env, seq), (const uint64_t *)addr(env, offsets),
                             nReads, firstReadNo, GS_MEM_HOST, (int32_t *)addr(env, classVi), (uint8_t *)addr(env, flags));
    if (rc) throw_gs(env, rc);
}

JNIEXPORT jlong JNICALL JNAME(matchSubmitAsync)(JNIEnv *env, jclass c, jlong run, jobject seq, jobject offsets, jlong nReads,
                                                jlong firstReadNo, jobject classVi, jobject flags) {
    int64_t ticket = -1;
    int rc = gs_match_submit_async((gs_run *)(intptr_t)run, (const uint8_t *)addr(env, seq), (const uint64_t *)addr(env, offsets),
                                   nReads, firstReadNo, (int32_t *)addr(env, classVi), (uint8_t *)addr(env, flags), &ticket);
    if (rc) throw_gs(env, rc);
    return ticket;
}

JNIEXPORT void JNICALL JNAME(matchWait)(JNIEnv *env, jclass c, jlong run, jlong ticket) {
    int rc = gs_match_wait((gs_run *)(intptr_t)run, ticket);
    if (rc) throw_gs(env, rc);
}

JNIEXPORT jlong JNICALL JNAME(matchSubmitText)(JNIEnv *env, jclass c, jlong run, jobject text, jlong nBytes, jlong nLines,
                                               jlong firstReadNo) {
    int64_t ticket = -1;
    int rc = gs_match_submit_text((gs_run *)(intptr_t)run, (const uint8_t *)addr(env, text), nBytes, nLines, GS_MEM_HOST,
                                  firstReadNo, NULL, NULL, &ticket);
    if (rc) throw_gs(env, rc);
    return (jlong)ticket;
}

JNIEXPORT void JNICALL JNAME(matchTextWaitCopy)(JNIEnv *env, jclass c, jlong run, jlong ticket) {
    int rc = gs_match_text_wait_copy((gs_run *)(intptr_t)run, ticket);
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchTextStatus)(JNIEnv *env, jclass c, jlong run, jlongArray out) {
    int64_t v[5] = {-1, -1, 0, 0, 0};
    int rc = gs_match_text_status((gs_run *)(intptr_t)run, &v[0], &v[1], &v[2]);
    if (rc) throw_gs(env, rc);
    (*env)->SetLongArrayRegion(env, out, 0, 5, (const jlong *)v);
}

JNIEXPORT void JNICALL JNAME(matchTextClearError)(JNIEnv *env, jclass c, jlong run) {
    int rc = gs_match_text_clear_error((gs_run *)(intptr_t)run);
    if (rc) throw_gs(env, rc);
}

/* page-locked staging memory as a direct ByteBuffer: copies from it run at the full host-to-device rate and
 * asynchronously, copies from ordinary (pageable) direct buffers are staged by the runtime */
JNIEXPORT jobject JNICALL JNAME(pinnedAlloc)(JNIEnv *env, jclass c, jlong bytes) {
    void *p = NULL;
    int rc = gs_pinned_alloc(&p, (size_t)bytes);
    if (rc) {
        throw_gs(env, rc);
        return NULL;
    }
    return (*env)->NewDirectByteBuffer(env, p, bytes);
}

JNIEXPORT void JNICALL JNAME(pinnedFree)(JNIEnv *env, jclass c, jobject buf) {
    int rc = gs_pinned_free(addr(env, buf));
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchSegments)(JNIEnv *env, jclass c, jlong run, jobject seq, jobject offsets, jlong nReads,
                                            jobject segOff) {
    int rc = gs_match_segments((gs_run *)(intptr_t)run, (const uint8_t *)addr(env, seq), (const uint64_t *)addr(env, offsets),
                               nReads, GS_MEM_HOST, (uint64_t *)addr(env, segOff));
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchSegmentsFetch)(JNIEnv *env, jclass c, jlong run, jobject codes, jobject starts) {
    int rc = gs_match_segments_fetch((gs_run *)(intptr_t)run, (int32_t *)addr(env, codes), (int32_t *)addr(env, starts));
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchMaxContigReads)(JNIEnv *env, jclass c, jlong run, jobject readNo) {
    int rc = gs_match_max_contig_reads((gs_run *)(intptr_t)run, (int64_t *)addr(env, readNo));
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchMaxCounts)(JNIEnv *env, jclass c, jlong run, jobject out) {
    int rc = gs_match_max_counts((gs_run *)(intptr_t)run, (int16_t *)addr(env, out));
    if (rc) throw_gs(env, rc);
}

/* the runs of this JVM (one per GPU) into a global state held by each (RCCL between devices) */
JNIEXPORT void JNICALL JNAME(matchMerge)(JNIEnv *env, jclass c, jlongArray runs) {
    jsize n = (*env)->GetArrayLength(env, runs);
    jlong *h = (*env)->GetLongArrayElements(env, runs, NULL);
    gs_run *r[64];
    int rc = n > 64 ? GS_E_INVALID : GS_OK;
    for (jsize i = 0; i < n && rc == GS_OK; i++) r[i] = (gs_run *)(intptr_t)h[i];
    if (rc == GS_OK) rc = gs_match_merge(r, (int)n);
    (*env)->ReleaseLongArrayElements(env, runs, h, JNI_ABORT);
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchFinish)(JNIEnv *env, jclass c, jlong run, jobject table, jobject dtable) {
    int rc = gs_match_finish((gs_run *)(intptr_t)run, (int64_t *)addr(env, table), (double *)addr(env, dtable));
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchReset)(JNIEnv *env, jclass c, jlong run) {
    int rc = gs_match_reset((gs_run *)(intptr_t)run);
    if (rc) throw_gs(env, rc);
}

JNIEXPORT void JNICALL JNAME(matchDestroy)(JNIEnv *env, jclass c, jlong run) { gs_match_destroy((gs_run *)(intptr_t)run); }

JNIEXPORT jlong JNICALL JNAME(bloomBuild)(JNIEnv *env, jclass c, jint device, jobject kmers, jlong nKmers, jlong expectedInsertions,
                                          jdouble fpp) {
    gs_bloom *b = NULL;
    int rc = gs_bloom_build(&b, device, GS_BLOOM_XOR, (const int64_t *)addr(env, kmers), nKmers, GS_MEM_HOST, expectedInsertions, fpp);
    if (rc) throw_gs(env, rc);
    return (jlong)(intptr_t)b;
}

JNIEXPORT jlong JNICALL JNAME(bloomCreate)(JNIEnv *env, jclass c, jint device, jint kind, jlong bits, jint nHashes,
                                           jlongArray factors, jobject words, jlong nWords) {
    gs_bloom *b = NULL;
    jlong *f = (*env)->GetLongArrayElements(env, factors, NULL);
    int rc = gs_bloom_create(&b, device, kind, bits, nHashes, (const int64_t *)f, (const uint64_t *)addr(env, words), nWords);
    (*env)->ReleaseLongArrayElements(env, factors, f, JNI_ABORT);
    if (rc) throw_gs(env, rc);
    return (jlong)(intptr_t)b;
}

JNIEXPORT void JNICALL JNAME(bloomDestroy)(JNIEnv *env, jclass c, jlong b) { gs_bloom_destroy((gs_bloom *)(intptr_t)b); }

JNIEXPORT void JNICALL JNAME(filterSubmit)(JNIEnv *env, jclass c, jlong b, jint k, jint minPos, jdouble ratio, jobject seq,
                                           jobject offsets, jlong nReads, jobject accept) {
    int rc = gs_filter_submit((gs_bloom *)(intptr_t)b, k, minPos, ratio, (const uint8_t *)addr(env, seq),
                              (const uint64_t *)addr(env, offsets), nReads, GS_MEM_HOST, (uint8_t *)addr(env, accept), 0);
    if (rc) throw_gs(env, rc);
}
