/*
 * Reference-side binding of include/gsgpu.h.  SOURCE ONLY: the build container has no JDK (no javac, no jni.h),
 * so this file and java/jni/gsgpu_jni.c are not compiled here; tools/check_java_glue.py checks every use of a reference
 * member in java/src against the reference's sources instead (see INTEGRATION.md).
 *
 * One static native method per C entry point.  Handles travel as long; batches as direct ByteBuffers
 * (address + length are taken with GetDirectBufferAddress, no copies on the Java side).
 * A non-zero status from the C ABI is turned into a RuntimeException carrying gs_last_error().
 */
package org.metagene.genestrip.gpu;

import java.nio.ByteBuffer;

public final class GsGpuNative {
	static {
		System.loadLibrary("gsgpu_jni"); // links against libgsgpu.so
	}

	private GsGpuNative() {
	}

	/** gs_device_count */
	public static native int deviceCount();

	/** gs_db_create: kmers = n_entries x int64 in KMerStore.visit order (reference encoding), valueIdx = n_entries x
	 *  int32, parentVi = n_values x int32 (-1 root, -2 no node) or null.  Returns the gs_db handle. */
	public static native long dbCreate(int device, int k, long nEntries, ByteBuffer kmers, ByteBuffer valueIdx,
			int nValues, ByteBuffer parentVi);

	/** gs_db_create_striped: ONE store whose record table is split over several GPUs (a store that does not fit one of
	 *  them).  devicesThenHandles holds the device of every stripe on entry and the gs_db handle of every stripe on return;
	 *  begin one run per handle (matchBegin), deal the reads to them, matchMerge, matchFinish on any. */
	public static native void dbCreateStriped(long[] devicesThenHandles, int k, long nEntries, ByteBuffer kmers,
			ByteBuffer valueIdx, int nValues, ByteBuffer parentVi);

	/** gs_dbbuild_begin: DB construction on the device (FillDBGoal + DBGoal); parentVi as for dbCreate, exactly one root */
	public static native long dbBuildBegin(int device, int k, int nValues, ByteBuffer parentVi, boolean lowerCaseBases,
			int maxDust, int stepSize);

	/** gs_dbbuild_add: nRegions regions (bases without headers and line ends, offsets = nRegions + 1 x int64 from 0,
	 *  nodeVi = nRegions x int32, all direct buffers in native order); update = a DBGoal region */
	public static native void dbBuildAdd(long builder, ByteBuffer bases, ByteBuffer offsets, ByteBuffer nodeVi, long nRegions,
			boolean update);

	/** gs_dbbuild_finish: sort + LCA fold; returns the number of stored k-mers */
	public static native long dbBuildFinish(long builder);

	/** gs_dbbuild_fetch: kmers (n x int64 ascending, the reference's encoding) and value indices (n x int32) */
	public static native void dbBuildFetch(long builder, ByteBuffer kmers, ByteBuffer valueIdx);

	/** gs_dbbuild_to_db: the store over the built arrays, laid out on the device (returns a gs_db handle as dbCreate does) */
	public static native long dbBuildToDb(long builder);

	public static native void dbBuildDestroy(long builder);

	/** gs_db_save / gs_db_load: the native image of the device store */
	public static native void dbSave(long db, String path);

	public static native long dbLoad(int device, String path);

	public static native void dbDestroy(long db);

	/** gs_match_begin */
	public static native long matchBegin(long db, boolean classify, boolean countUnique, int maxPaths, int threshold,
			double maxReadTaxErr, double maxReadClassErr, int maxKmerResCounts);

	/** gs_match_submit with GS_MEM_HOST: seq = concatenated read bytes, offsets = (nReads+1) x uint64,
	 *  classVi = nReads x int32 (or null), flags = nReads x uint8 (or null). */
	public static native void matchSubmit(long run, ByteBuffer seq, ByteBuffer offsets, long nReads, long firstReadNo,
			ByteBuffer classVi, ByteBuffer flags);

	/** gs_match_submit_async: queues a host batch (best in buffers from pinnedAlloc) and returns its ticket; the buffers
	 *  belong to the library until matchWait(ticket).  Two batches can be under way: the copy of one runs under the
	 *  kernel of the other. */
	public static native long matchSubmitAsync(long run, ByteBuffer seq, ByteBuffer offsets, long nReads, long firstReadNo,
			ByteBuffer classVi, ByteBuffer flags);

	public static native void matchWait(long run, long ticket);

	/** gs_match_submit_text with GS_MEM_HOST: text = a direct buffer holding whole four-line FASTQ records, nLines =
	 *  number of '\n' in it (a multiple of 4).  Returns the ticket; the buffer may be refilled after matchTextWaitCopy. */
	public static native long matchSubmitText(long run, ByteBuffer text, long nBytes, long nLines, long firstReadNo);

	public static native void matchTextWaitCopy(long run, long ticket);

	/** gs_match_text_status: out[0] = ticket of the first refused chunk or -1, out[1] = first bad record or -1,
	 *  out[2..4] = reads, k-mers, bases of the accepted chunks. */
	public static native void matchTextStatus(long run, long[] out);

	public static native void matchTextClearError(long run);

	/** gs_pinned_alloc as a direct buffer (native byte order is up to the caller): the place to stage batches and text
	 *  chunks -- copies from page-locked memory run at the full host-to-device rate.  Free it with pinnedFree, never
	 *  let it be garbage collected while a submit may still read it. */
	public static native ByteBuffer pinnedAlloc(long bytes);

	public static native void pinnedFree(ByteBuffer buf);

	/** gs_match_segments with GS_MEM_HOST: the Kraken-style runs of the reads of a batch; segOff = (nReads+1) x uint64 */
	public static native void matchSegments(long run, ByteBuffer seq, ByteBuffer offsets, long nReads, ByteBuffer segOff);

	/** gs_match_segments_fetch: codes / starts = n_segments x int32 (n_segments = segOff[nReads]) */
	public static native void matchSegmentsFetch(long run, ByteBuffer codes, ByteBuffer starts);

	/** gs_match_max_contig_reads: readNo = n_values x int64 */
	public static native void matchMaxContigReads(long run, ByteBuffer readNo);

	/** gs_match_max_counts: out = (n_values + 1) x maxKmerResCounts int16 */
	public static native void matchMaxCounts(long run, ByteBuffer out);

	/** gs_match_merge: the runs of this process (one per GPU) into a global state held by each */
	public static native void matchMerge(long[] runs);

	/** gs_match_finish: table = n_values x GS_N_COLS int64, dtable = n_values x GS_N_DCOLS double. */
	public static native void matchFinish(long run, ByteBuffer table, ByteBuffer dtable);

	public static native void matchReset(long run);

	public static native void matchDestroy(long run);

	/** gs_bloom_create */
	public static native long bloomCreate(int device, int kind, long bits, int nHashes, long[] hashFactors,
			ByteBuffer words, long nWords);

	/** gs_bloom_build: the XOR index filter of BloomIndexGoal built on the device from nKmers k-mers (direct buffer, int64 in
	 *  native order), sized for expectedInsertions at fpp; returns the gs_bloom handle (as bloomCreate does) */
	public static native long bloomBuild(int device, ByteBuffer kmers, long nKmers, long expectedInsertions, double fpp);

	public static native void bloomDestroy(long bloom);

	/** gs_filter_submit with GS_MEM_HOST: accept = nReads x uint8 */
	public static native void filterSubmit(long bloom, int k, int minPosCount, double positiveRatio, ByteBuffer seq,
			ByteBuffer offsets, long nReads, ByteBuffer accept);

	/**
	 * gs_host_match_files (include/gshost.h): one runMatcher over LOCAL FILES below the JVM -- file, gunzip (on the device), record
	 * scan, kernels, per-read outputs (gathered and gzip-compressed on the device) -- with a run of its own.  paths: the FASTQ /
	 * FASTA files in order; filteredPath / krakenOutPath: null = off, a name ending in .gz / .gzip is written gzip (BGZF members);
	 * taxids: tax id per value index (needed for Kraken-style lines, else null); table / dtable as matchFinish; maxContigDesc: null,
	 * or nValues x descStride bytes that receive CountsPerTaxid.maxContigDescriptor per value index (NUL-terminated);
	 * totals[4] = totalReads, totalKMers, totalBPs, reads written to filteredPath.
	 */
	public static native void hostMatchFiles(long db, boolean classify, boolean countUnique, int maxPaths, int threshold,
			double maxReadTaxErr, double maxReadClassErr, int maxKmerResCounts, String[] paths, String filteredPath,
			String krakenOutPath, boolean writeAll, String[] taxids, boolean withProbs, ByteBuffer table, ByteBuffer dtable,
			ByteBuffer maxContigDesc, int descStride, long[] totals);

	/** gs_host_match_run: the same into the matcher's own run -- matchReset before, matchFinish (and matchMaxCounts) after */
	public static native void hostMatchRun(long run, long db, String[] paths, String filteredPath, String krakenOutPath,
			boolean writeAll, String[] taxids, boolean withProbs, ByteBuffer maxContigDesc, int descStride, long[] totals);

	/** gs_host_match_into: some of a sample's files into a run that is merged with the runs of other GPUs (matchMerge) before
	 *  matchFinish; fileIndex[i] = position of paths[i] in the sample's file order, readsOfFile[i] receives its read count */
	public static native void hostMatchInto(long run, long db, String[] paths, int[] fileIndex, long[] readsOfFile, long[] totals);

	/** gs_host_filter_files: one runFilter over local files; filteredPath / restPath: null = off, .gz = gzip; totals as above
	 *  (totals[3] = accepted reads) */
	public static native void hostFilterFiles(long bloom, int k, int minPosCount, double positiveRatio, String[] paths,
			String filteredPath, String restPath, boolean withProbs, long[] totals);

	/** gs_host_last_error: the message of the last failure inside the host layer on this thread */
	public static native String hostLastError();

	/** gs_host_release_pools: page-locked blocks and device decoders the host layer keeps from call to call go back to the system */
	public static native void hostReleasePools();

	// column indices of the integer table (include/gsgpu.h, GS_C_*)
	public static final int C_READS = 0, C_READS_KMERS = 1, C_KMERS = 2, C_UNIQUE_KMERS = 3, C_CONTIGS = 4,
			C_CONTIG_LEN_SQ_SUM = 5, C_MAX_CONTIG_LEN = 6, C_READS_1KMER = 7, C_READS_BPS = 8,
			C_MAX_CONTIG_READ_NO = 9, N_COLS = 10;
	public static final int D_ERR_SUM = 0, D_ERR_SQ_SUM = 1, D_CLASS_ERR_SUM = 2, D_CLASS_ERR_SQ_SUM = 3, N_DCOLS = 4;
	public static final int F_FOUND = 1, F_RETURNED = 2, F_COUNTED = 4;
	public static final int BLOOM_XOR = 0, BLOOM_MURMUR = 1, BLOOM_BLOCKED = 2;
}
