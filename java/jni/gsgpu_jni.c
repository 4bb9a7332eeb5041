/*
 * gsgpu_jni.c -- JNI shim between org.metagene.genestrip.gpu.GsGpuNative and the C ABI (include/gsgpu.h).
 * Not built in the build container (no JDK / jni.h); tests/test_java_glue_cpu.py syntax-checks it against a minimal
 * stand-in for jni.h (tests/native/jni_stub/jni.h: declarations only).  Build on a host with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       -o libgsgpu_jni.so gsgpu_jni.c -L../../genestrip_amd -lgshost -lgsgpu
 * tests/test_gpu_jni.py builds it against the stand-in with a small functional JNIEnv (tests/native/jni_stub/jni_env.c) and drives
 * the file-level entry points (hostMatchFiles / hostMatchRun / hostFilterFiles) on the GPU box.
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsgpu.h"
#include "gshost.h"

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

/* ---------------------------------------------------------------------------------------------------
 * The file-level entry points of include/gshost.h: what runMatcher / runFilter call when every resource of the
 * StreamingResourceStream is a local file (FastqKMerMatcher.java:181-235, FastqBloomFilter.java:80-89).  The whole pipeline --
 * file, (device) gunzip, record scan, kernels, per-read outputs (gathered and compressed on the device) -- then runs below the
 * JVM: 11-19 Gbp/s instead of the parser thread's 0.2-0.7.
 * ------------------------------------------------------------------------------------------------- */
static void throw_host(JNIEnv *env, int rc) {
    char msg[640];
    const char *h = gs_host_last_error();
    snprintf(msg, sizeof(msg), "gsgpu error %d (%s): %s", rc, gs_strerror(rc), (h && h[0]) ? h : gs_last_error());
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), msg);
}

/* a Java String[] as a C array of UTF-8 strings (NULL elements stay NULL); released with free_strings */
typedef struct {
    jsize n;
    const char **c;
    jstring *j;
} StrArray;

static int get_strings(JNIEnv *env, jobjectArray a, StrArray *out) {
    out->n = a ? (*env)->GetArrayLength(env, a) : 0;
    out->c = (const char **)calloc((size_t)out->n + 1, sizeof(char *));
    out->j = (jstring *)calloc((size_t)out->n + 1, sizeof(jstring));
    if (!out->c || !out->j) return GS_E_NOMEM;
    for (jsize i = 0; i < out->n; i++) {
        out->j[i] = (jstring)(*env)->GetObjectArrayElement(env, a, i);
        out->c[i] = out->j[i] ? (*env)->GetStringUTFChars(env, out->j[i], NULL) : NULL;
    }
    return GS_OK;
}

static void free_strings(JNIEnv *env, StrArray *s) {
    for (jsize i = 0; i < s->n; i++) {
        if (s->j && s->j[i]) {
            if (s->c && s->c[i]) (*env)->ReleaseStringUTFChars(env, s->j[i], s->c[i]);
            (*env)->DeleteLocalRef(env, s->j[i]);
        }
    }
    free((void *)s->c);
    free(s->j);
}

static void put_totals(JNIEnv *env, jlongArray totals, const gs_host_totals *t) {
    if (!totals) return;
    const jlong v[4] = {t->reads, t->kmers, t->bps, t->filtered_reads};
    (*env)->SetLongArrayRegion(env, totals, 0, 4, v);
}

static void fill_opts(gs_host_match_opts *o, const char *flt, const char *kr, jboolean writeAll, const StrArray *tax, jboolean withProbs, void *desc, jint stride) {
    memset(o, 0, sizeof(*o));
    o->filtered_path = flt;
    o->kraken_out_path = kr;
    o->write_all = writeAll ? 1 : 0;
    o->taxids = tax->n ? tax->c : NULL;
    o->with_probs = withProbs ? 1 : 0;
    o->max_contig_desc = (uint8_t *)desc;
    o->max_contig_desc_stride = stride;
}

/* gs_host_match_files: one call = one runMatcher over local files with a run of its own; totals = {reads, kmers, bps, filtered reads} */
JNIEXPORT void JNICALL JNAME(hostMatchFiles)(JNIEnv *env, jclass c, jlong db, jboolean classify, jboolean countUnique, jint maxPaths, jint threshold,
                                             jdouble taxErr, jdouble classErr, jint maxKmerResCounts, jobjectArray paths, jstring filteredPath,
                                             jstring krakenOutPath, jboolean writeAll, jobjectArray taxids, jboolean withProbs, jobject table,
                                             jobject dtable, jobject maxContigDesc, jint descStride, jlongArray totals) {
    gs_match_cfg cfg = {classify ? 1 : 0, countUnique ? 1 : 0, maxPaths, threshold, taxErr, classErr, 0, maxKmerResCounts};
    StrArray p = {0, NULL, NULL}, tax = {0, NULL, NULL};
    int rc = get_strings(env, paths, &p);
    if (!rc) rc = get_strings(env, taxids, &tax);
    const char *flt = filteredPath ? (*env)->GetStringUTFChars(env, filteredPath, NULL) : NULL;
    const char *kr = krakenOutPath ? (*env)->GetStringUTFChars(env, krakenOutPath, NULL) : NULL;
    gs_host_totals t;
    memset(&t, 0, sizeof(t));
    if (!rc) {
        gs_host_match_opts o;
        fill_opts(&o, flt, kr, writeAll, &tax, withProbs, addr(env, maxContigDesc), descStride);
        rc = gs_host_match_files((gs_db *)(intptr_t)db, &cfg, p.c, (int)p.n, &o, (int64_t *)addr(env, table), (double *)addr(env, dtable), &t);
    }
    if (flt) (*env)->ReleaseStringUTFChars(env, filteredPath, flt);
    if (kr) (*env)->ReleaseStringUTFChars(env, krakenOutPath, kr);
    free_strings(env, &p);
    free_strings(env, &tax);
    if (rc)
        throw_host(env, rc);
    else
        put_totals(env, totals, &t);
}

/* gs_host_match_run: the same into the matcher's own run (matchReset before, matchFinish after) */
JNIEXPORT void JNICALL JNAME(hostMatchRun)(JNIEnv *env, jclass c, jlong run, jlong db, jobjectArray paths, jstring filteredPath, jstring krakenOutPath,
                                           jboolean writeAll, jobjectArray taxids, jboolean withProbs, jobject maxContigDesc, jint descStride,
                                           jlongArray totals) {
    StrArray p = {0, NULL, NULL}, tax = {0, NULL, NULL};
    int rc = get_strings(env, paths, &p);
    if (!rc) rc = get_strings(env, taxids, &tax);
    const char *flt = filteredPath ? (*env)->GetStringUTFChars(env, filteredPath, NULL) : NULL;
    const char *kr = krakenOutPath ? (*env)->GetStringUTFChars(env, krakenOutPath, NULL) : NULL;
    gs_host_totals t;
    memset(&t, 0, sizeof(t));
    if (!rc) {
        gs_host_match_opts o;
        fill_opts(&o, flt, kr, writeAll, &tax, withProbs, addr(env, maxContigDesc), descStride);
        rc = gs_host_match_run((gs_run *)(intptr_t)run, (gs_db *)(intptr_t)db, p.c, (int)p.n, &o, &t);
    }
    if (flt) (*env)->ReleaseStringUTFChars(env, filteredPath, flt);
    if (kr) (*env)->ReleaseStringUTFChars(env, krakenOutPath, kr);
    free_strings(env, &p);
    free_strings(env, &tax);
    if (rc)
        throw_host(env, rc);
    else
        put_totals(env, totals, &t);
}

/* gs_host_match_into: some of the files of a sample into a run that is merged with others (matchMerge) before matchFinish */
JNIEXPORT void JNICALL JNAME(hostMatchInto)(JNIEnv *env, jclass c, jlong run, jlong db, jobjectArray paths, jintArray fileIndex, jlongArray readsOfFile,
                                            jlongArray totals) {
    StrArray p = {0, NULL, NULL};
    int rc = get_strings(env, paths, &p);
    jint *idx = fileIndex ? (*env)->GetIntArrayElements(env, fileIndex, NULL) : NULL;
    jlong *rof = readsOfFile ? (*env)->GetLongArrayElements(env, readsOfFile, NULL) : NULL;
    gs_host_totals t;
    memset(&t, 0, sizeof(t));
    if (!rc && (!idx || !rof || (*env)->GetArrayLength(env, fileIndex) < p.n || (*env)->GetArrayLength(env, readsOfFile) < p.n)) rc = GS_E_INVALID;
    if (!rc) rc = gs_host_match_into((gs_run *)(intptr_t)run, (gs_db *)(intptr_t)db, p.c, (int)p.n, (const int32_t *)idx, (int64_t *)rof, &t);
    if (idx) (*env)->ReleaseIntArrayElements(env, fileIndex, idx, JNI_ABORT);
    if (rof) (*env)->ReleaseLongArrayElements(env, readsOfFile, rof, rc ? JNI_ABORT : 0);
    free_strings(env, &p);
    if (rc)
        throw_host(env, rc);
    else
        put_totals(env, totals, &t);
}

/* gs_host_filter_files: one call = one runFilter over local files */
JNIEXPORT void JNICALL JNAME(hostFilterFiles)(JNIEnv *env, jclass c, jlong bloom, jint k, jint minPosCount, jdouble positiveRatio, jobjectArray paths,
                                              jstring filteredPath, jstring restPath, jboolean withProbs, jlongArray totals) {
    StrArray p = {0, NULL, NULL};
    int rc = get_strings(env, paths, &p);
    const char *flt = filteredPath ? (*env)->GetStringUTFChars(env, filteredPath, NULL) : NULL;
    const char *rest = restPath ? (*env)->GetStringUTFChars(env, restPath, NULL) : NULL;
    gs_host_totals t;
    memset(&t, 0, sizeof(t));
    if (!rc) rc = gs_host_filter_files((gs_bloom *)(intptr_t)bloom, k, minPosCount, positiveRatio, p.c, (int)p.n, flt, rest, withProbs ? 1 : 0, &t);
    if (flt) (*env)->ReleaseStringUTFChars(env, filteredPath, flt);
    if (rest) (*env)->ReleaseStringUTFChars(env, restPath, rest);
    free_strings(env, &p);
    if (rc)
        throw_host(env, rc);
    else
        put_totals(env, totals, &t);
}

/* gs_host_last_error */
JNIEXPORT jstring JNICALL JNAME(hostLastError)(JNIEnv *env, jclass c) { return (*env)->NewStringUTF(env, gs_host_last_error()); }

/* gs_host_release_pools */
JNIEXPORT void JNICALL JNAME(hostReleasePools)(JNIEnv *env, jclass c) {
    int rc = gs_host_release_pools();
    if (rc) throw_host(env, rc);
}
