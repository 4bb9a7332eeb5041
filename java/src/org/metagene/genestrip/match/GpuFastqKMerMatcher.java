/*
 * GPU-backed drop-in for FastqKMerMatcher (reference: core/src/main/java/org/metagene/genestrip/match/
 * FastqKMerMatcher.java).  SOURCE ONLY -- not compiled in the build container (no JDK); tools/check_java_glue.py checks
 * every reference member used here against the reference's sources.  See INTEGRATION.md.
 *
 * It lives in the reference's own package because it fills the protected CountsPerTaxid fields
 * (CountsPerTaxid.java:127-159) and the protected statsIndex array (FastqKMerMatcher.java:76).
 *
 * How it hooks in (all seams are the reference's own virtual methods).  Two flows, chosen per runMatcher call:
 *   FILES -- every resource of the StreamingResourceStream is a local file (StreamingFileResource: the normal case, `-f reads.fastq.gz`):
 *     processFastqStreams() hands the paths to gs_host_match_run (GsGpuNative.hostMatchRun) and the whole pipeline runs below the JVM:
 *     file -> gunzip on the device -> record scan -> match kernel -> filtered FASTQ gathered and gzip-compressed on the device,
 *     Kraken-style lines formatted by the host layer's threads; 11-19 Gbp/s.  The reference's parser never runs.
 *   STREAMS -- anything else (URL resources): the parser stays the reference's (AbstractFastqReader.doReadFastq, one thread behind
 *     GZIPInputStream: 0.2-0.7 Gbp/s, which is what this flow delivers whatever the GPU does); with a zero-consumer execution
 *     context it calls nextEntry -> matchRead(entry, 0) on the producer thread (AbstractFastqReader.java:350-352);
 *   - matchRead() only appends the read to one of TWO page-locked batches; a full batch goes to gs_match_submit_async, the parser
 *     fills the other one meanwhile, and the per-read outcome (class, flags, Kraken-style runs) of the batch before drives the
 *     same writeback afterMatch does (FastqKMerMatcher.java:304-315), in input order;
 *   - processFastqStreams() flushes the last batch and turns the device table into CountsPerTaxid objects before
 *     runMatcher() collects them (FastqKMerMatcher.java:199-204);
 *   - unique k-mer counts come from the device, so runMatcher() is called without a KMerUniqueCounterBits and the
 *     counts are patched into the result afterwards (FastqKMerMatcher.java:207-213).
 */
package org.metagene.genestrip.match;

import java.io.File;
import java.io.IOException;
import java.io.InputStream;
import java.io.OutputStream;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.charset.StandardCharsets;
import java.util.Arrays;

import org.metagene.genestrip.DefaultExecutionContext;
import org.metagene.genestrip.ExecutionContext;
import org.metagene.genestrip.gpu.GsGpuNative;
import org.metagene.genestrip.io.StreamingFileResource;
import org.metagene.genestrip.io.StreamingResource;
import org.metagene.genestrip.io.StreamingResourceStream;
import org.metagene.genestrip.store.KMerStore;
import org.metagene.genestrip.store.KMerUniqueCounterBits;
import org.metagene.genestrip.tax.SmallTaxTree;
import org.metagene.genestrip.tax.SmallTaxTree.SmallTaxIdNode;

public class GpuFastqKMerMatcher extends FastqKMerMatcher {
	private static final int BATCH_READS = 1 << 20;
	private static final int BATCH_BYTES = 256 << 20;
	private static final long UPLOAD_SLICE = 1L << 26; // entries per direct-buffer slice of the store hand-over
	private static final int DESC_STRIDE = 256;        // bytes per value index for the names of the max-contig reads (FILES flow)
	private static final String[] FASTA_SUFFIXES = { "fasta", "fa", "fna", "fas" }; // (+ .gz / .gzip: FastqMapGoal.java:64)

	/** everything one batch of the STREAMS flow carries: page-locked device-facing buffers + the host-side copies for the writeback */
	private static final class Batch {
		final ByteBuffer seq = GsGpuNative.pinnedAlloc(BATCH_BYTES).order(ByteOrder.nativeOrder());
		final ByteBuffer offsets = GsGpuNative.pinnedAlloc(8L * (BATCH_READS + 1)).order(ByteOrder.nativeOrder());
		final ByteBuffer classVi = GsGpuNative.pinnedAlloc(4L * BATCH_READS).order(ByteOrder.nativeOrder());
		final ByteBuffer flags = GsGpuNative.pinnedAlloc(BATCH_READS);
		// descriptors (and qualities, if they are written) of the batch, for the writeback after the submit
		byte[] descs = new byte[64 * 1024 * 1024];
		byte[] quals = new byte[0];
		final int[] descOff = new int[BATCH_READS + 1];
		final int[] qualOff = new int[BATCH_READS + 1];
		int reads;
		long firstReadNo;
		long ticket = -1; // of the submit that is under way (-1: none)

		void free() {
			GsGpuNative.pinnedFree(seq);
			GsGpuNative.pinnedFree(offsets);
			GsGpuNative.pinnedFree(classVi);
			GsGpuNative.pinnedFree(flags);
		}
	}

	private final long db;
	private final long run;
	private final int nValues;
	private final boolean keepQualities;
	private final SmallTaxIdNode[] nodeOfValue; // value index -> tree node (null: value without node)
	private final byte[][] taxidBytes;          // value index -> tax id as bytes (Kraken-style lines)
	private final String[] taxidStrings;        // the same as strings (gs_host_match_opts.taxids)

	private Batch[] batches;  // STREAMS flow only (allocated on first use)
	private int filling;      // the batch matchRead appends to
	private final ByteBuffer segOff = direct(8L * (BATCH_READS + 1));
	private final ByteBuffer maxReadNo;
	private ByteBuffer segCodes = direct(4L << 20), segStarts = direct(4L << 20);
	private long globalReadNo; // file-order read number over all files of this runMatcher call
	private long[] uniqueCounts;
	private short[][] maxCounts;
	private final byte[][] maxContigDescriptor; // per value index: descriptor of the read that holds the longest contig
	// FILES flow: the output files of the current runMatcher call (written below the JVM)
	private File nativeFiltered, nativeKraken;

	public GpuFastqKMerMatcher(KMerStore<SmallTaxIdNode> kmerStore, int initialReadSize, int maxQueueSize,
			ExecutionContext bundle, boolean withProbs, int maxKmerResCounts, SmallTaxTree taxTree, int maxPaths,
			double maxReadTaxErrorCount, double maxReadClassErrorCount, boolean writeAll, int threshold, String dbMD5,
			int device) {
		// zero consumers: the producer thread calls matchRead directly, the GPU is the parallel part
		super(kmerStore, initialReadSize, maxQueueSize,
				new DefaultExecutionContext(Thread.currentThread(), 0, bundle.getLogUpdateCycle()), withProbs,
				maxKmerResCounts, taxTree, maxPaths, maxReadTaxErrorCount, maxReadClassErrorCount, writeAll, threshold,
				dbMD5);
		nValues = kmerStore.getNValues();
		keepQualities = withProbs;
		nodeOfValue = new SmallTaxIdNode[nValues];
		taxidBytes = new byte[nValues][];
		taxidStrings = new String[nValues];
		maxContigDescriptor = new byte[nValues][];
		maxReadNo = direct(8L * nValues);
		db = upload(kmerStore, taxTree, device);
		run = GsGpuNative.matchBegin(db, taxTree != null, true, maxPaths, threshold, maxReadTaxErrorCount,
				maxReadClassErrorCount, maxKmerResCounts);
	}

	private static ByteBuffer direct(long bytes) {
		if (bytes > Integer.MAX_VALUE - 8) {
			throw new IllegalArgumentException("direct buffer of " + bytes + " bytes");
		}
		return ByteBuffer.allocateDirect((int) bytes).order(ByteOrder.nativeOrder());
	}

	/**
	 * KMerStore.visit yields (kmer, valueIndex, pos) for positions 0 .. entries-1: ascending k-mers for the
	 * KMerSortedArray (KMerSortedArray.java:426-439), bucket by bucket for the RadixKMerStore
	 * (RadixKMerStore.java:714-729); gs_db_create takes either order.  Stores beyond one direct buffer are handed over
	 * through two file-backed mappings instead (not shown: FileChannel.map in slices of UPLOAD_SLICE entries).
	 */
	private long upload(KMerStore<SmallTaxIdNode> store, SmallTaxTree tree, int device) {
		long n = store.getEntries();
		if (n > UPLOAD_SLICE * 4) {
			throw new UnsupportedOperationException("store of " + n + " k-mers: hand it over through mapped slices");
		}
		ByteBuffer kmers = direct(8L * n);
		ByteBuffer vidx = direct(4L * n);
		store.visit((s, kmer, index, pos) -> {
			kmers.putLong(kmer);
			vidx.putInt(index);
		});
		for (int v = 0; v < nValues; v++) {
			nodeOfValue[v] = store.getValueForIndex(v);
			taxidStrings[v] = nodeOfValue[v] == null ? "" : nodeOfValue[v].getTaxId();
			taxidBytes[v] = nodeOfValue[v] == null ? null : taxidStrings[v].getBytes(StandardCharsets.UTF_8);
		}
		ByteBuffer parent = null;
		if (tree != null) {
			parent = direct(4L * nValues);
			for (int v = 0; v < nValues; v++) {
				SmallTaxIdNode node = nodeOfValue[v];
				// every tree node has a store index (Database.initStoreIndices, Database.java:107-128)
				parent.putInt(node == null ? -2 : node.getParent() == null ? -1 : node.getParent().getStoreIndex());
			}
		}
		return GsGpuNative.dbCreate(device, store.getK(), n, kmers, vidx, nValues, parent);
	}

	/**
	 * The paths of the resources if EVERY one of them is a local file whose type the host layer decides as the reference does
	 * (FASTA by suffix: FastqMapGoal.java:64, 188-201; gzip by content), else null: the STREAMS flow.
	 */
	private String[] localFiles(StreamingResourceStream fastqs) {
		java.util.List<String> paths = new java.util.ArrayList<>();
		for (StreamingResource r : fastqs) {
			if (!(r instanceof StreamingFileResource)) {
				return null;
			}
			File f = ((StreamingFileResource) r).getFile();
			if (isFastaStream(r) != hasFastaName(f.getName())) {
				return null;
			}
			paths.add(f.getPath());
		}
		return paths.toArray(new String[0]);
	}

	private static boolean hasFastaName(String name) {
		String n = name.toLowerCase();
		for (String gz : new String[] { ".gzip", ".gz" }) {
			if (n.endsWith(gz)) {
				n = n.substring(0, n.length() - gz.length());
				break;
			}
		}
		for (String suffix : FASTA_SUFFIXES) {
			if (n.endsWith("." + suffix)) {
				return true;
			}
		}
		return false;
	}

	@Override
	public MatchingResult runMatcher(StreamingResourceStream fastqs, File filteredFile, File krakenOutStyleFile,
			KMerUniqueCounterBits uniqueCounter) throws IOException {
		GsGpuNative.matchReset(run); // stats + unique bitmap start cleared per key (FastqKMerMatcher.java:192-193)
		globalReadNo = 0;
		uniqueCounts = null;
		maxCounts = null;
		Arrays.fill(maxContigDescriptor, null);
		final boolean files = localFiles(fastqs) != null;
		// FILES: the output files are written below the JVM, so the base class must not open them; STREAMS: the base class opens
		// `indexed` / `out`.  Either way it runs processFastqStreams (overridden below) and collects statsIndex.
		nativeFiltered = files ? filteredFile : null;
		nativeKraken = files ? krakenOutStyleFile : null;
		MatchingResult res = super.runMatcher(fastqs, files ? null : filteredFile, files ? null : krakenOutStyleFile, null);
		nativeFiltered = null;
		nativeKraken = null;
		for (int vi = 0; vi < nValues; vi++) {
			CountsPerTaxid stats = statsIndex[vi];
			if (stats != null) {
				stats.uniqueKmers = uniqueCounter == null ? -1 : uniqueCounts[vi];
				if (maxCounts != null) {
					stats.maxKMerCounts = maxCounts[vi];
				}
			}
		}
		return res;
	}

	@Override
	public void processFastqStreams(StreamingResourceStream fastqs) throws IOException {
		String[] files = localFiles(fastqs);
		if (files != null) {
			// FILES flow: one native call does what the loop of AbstractLoggingFastqStreamer.processFastqStreams (:95-131) does
			ByteBuffer descs = direct((long) nValues * DESC_STRIDE);
			long[] totals = new long[4];
			GsGpuNative.hostMatchRun(run, db, files, nativeFiltered == null ? null : nativeFiltered.getPath(),
					nativeKraken == null ? null : nativeKraken.getPath(), writeAll, taxidStrings, keepQualities, descs,
					DESC_STRIDE, totals);
			totalReads = totals[0];
			totalKMers = totals[1];
			totalBPs = totals[2];
			byte[] row = new byte[DESC_STRIDE];
			for (int vi = 0; vi < nValues; vi++) {
				descs.position(vi * DESC_STRIDE);
				descs.get(row);
				int n = 0;
				while (n < DESC_STRIDE && row[n] != 0) {
					n++;
				}
				maxContigDescriptor[vi] = n == 0 ? null : Arrays.copyOf(row, n);
			}
		} else {
			super.processFastqStreams(fastqs);
			drain();
		}
		fillStatsFromDevice();
	}

	@Override
	protected void readFastq(InputStream inputStream, boolean fasta) throws IOException {
		super.readFastq(inputStream, fasta);
		drain(); // the writeback of a file's last reads belongs before the next file starts
	}

	private Batch batch() {
		if (batches == null) {
			batches = new Batch[] { new Batch(), new Batch() };
		}
		return batches[filling];
	}

	/** Called by the (final) nextEntry for every parsed read; only batches the read. */
	@Override
	protected boolean matchRead(final MatcherReadEntry entry, final int index) {
		final boolean outputs = indexed != null || out != null;
		Batch b = batch();
		if (b.reads == BATCH_READS || b.seq.remaining() < entry.readSize
				|| b.descOff[b.reads] + entry.readDescriptorSize > b.descs.length) {
			try {
				submit();
			} catch (IOException e) {
				throw new RuntimeException(e);
			}
			b = batch();
		}
		if (b.reads == 0) {
			b.firstReadNo = globalReadNo;
			b.seq.clear();
			b.offsets.clear();
			b.offsets.putLong(0);
			b.descOff[0] = 0;
			b.qualOff[0] = 0;
		}
		b.seq.put(entry.read, 0, entry.readSize);
		b.offsets.putLong(b.seq.position());
		// descriptors are kept for every batch: the longest contig of a tax id may turn up in any read
		System.arraycopy(entry.readDescriptor, 0, b.descs, b.descOff[b.reads], entry.readDescriptorSize);
		b.descOff[b.reads + 1] = b.descOff[b.reads] + entry.readDescriptorSize;
		int q = b.qualOff[b.reads];
		if (outputs && keepQualities && entry.readProbs != null && entry.readProbsSize >= 0) {
			if (q + entry.readProbsSize > b.quals.length) {
				b.quals = Arrays.copyOf(b.quals, Math.max(2 * b.quals.length, q + entry.readProbsSize + (1 << 20)));
			}
			System.arraycopy(entry.readProbs, 0, b.quals, q, entry.readProbsSize);
			q += entry.readProbsSize;
		}
		b.qualOff[b.reads + 1] = q;
		b.reads++;
		globalReadNo++;
		return false; // the per-read outcome arrives with the batch: afterMatch below has nothing to do per read
	}

	@Override
	protected void afterMatch(MatcherReadEntry myEntry, boolean found) throws IOException {
		// the writeback happens per batch in complete(): it needs the outcome of the device
	}

	/**
	 * The batch that was being filled goes to the device (gs_match_submit_async: returns at once, its copy runs under the kernel
	 * of the batch before); the OTHER batch -- submitted one call earlier -- is waited for and written back; the parser then
	 * refills it while this one is on the device.
	 */
	private void submit() throws IOException {
		Batch b = batch();
		if (b.reads == 0) {
			return;
		}
		b.ticket = GsGpuNative.matchSubmitAsync(run, b.seq, b.offsets, b.reads, b.firstReadNo, b.classVi, b.flags);
		filling ^= 1;
		complete(batches[filling]);
	}

	/** both batches through: at the end of a file and of the run */
	private void drain() throws IOException {
		if (batches == null) {
			return;
		}
		submit();                     // what is being filled (its predecessor is completed inside)
		complete(batches[filling ^ 1]); // ... and the one that was just submitted
	}

	/** waits for the batch's outcome, then does for its reads what afterMatch does per read */
	private void complete(Batch b) throws IOException {
		if (b.ticket < 0) {
			return;
		}
		GsGpuNative.matchWait(run, b.ticket);
		b.ticket = -1;
		if (out != null) {
			GsGpuNative.matchSegments(run, b.seq, b.offsets, b.reads, segOff);
			long nSeg = segOff.getLong(8 * b.reads);
			if (4 * nSeg > segCodes.capacity()) {
				segCodes = direct(8 * nSeg);
				segStarts = direct(8 * nSeg);
			}
			GsGpuNative.matchSegmentsFetch(run, segCodes, segStarts);
		}
		// maxContigDescriptor (FastqKMerMatcher.java:401-407): the read that holds a tax id's longest contig right now
		GsGpuNative.matchMaxContigReads(run, maxReadNo);
		for (int vi = 0; vi < nValues; vi++) {
			long r = maxReadNo.getLong(8 * vi) - b.firstReadNo;
			if (r >= 0 && r < b.reads) {
				int i = (int) r;
				int end = indexOfBlank(b.descs, b.descOff[i] + 1, b.descOff[i + 1]);
				maxContigDescriptor[vi] = Arrays.copyOfRange(b.descs, b.descOff[i] + 1, end); // chars after '@' up to the first blank
			}
		}
		for (int i = 0; i < b.reads; i++) {
			int f = b.flags.get(i);
			int cls = b.classVi.getInt(4 * i);
			if ((f & GsGpuNative.F_RETURNED) != 0 && indexed != null) {
				writeRead(indexed, b, i); // rewriteInput(myEntry, indexed) of afterMatch
				updateWriteStats();
			}
			if (out != null && (writeAll || cls >= 0)) {
				writeKrakenLine(out, b, i, cls);
			}
		}
		b.reads = 0;
	}

	private static int indexOfBlank(byte[] a, int from, int to) {
		for (int i = from; i < to; i++) {
			if (a[i] == ' ') {
				return i;
			}
		}
		return to;
	}

	/** ReadEntry.write (AbstractFastqReader.java:570-584) for read i of the batch */
	private void writeRead(OutputStream o, Batch b, int i) throws IOException {
		int s0 = (int) b.offsets.getLong(8 * i), s1 = (int) b.offsets.getLong(8 * (i + 1));
		byte[] line = new byte[s1 - s0];
		o.write(b.descs, b.descOff[i], b.descOff[i + 1] - b.descOff[i]);
		o.write('\n');
		for (int j = 0; j < line.length; j++) {
			line[j] = b.seq.get(s0 + j);
		}
		o.write(line);
		o.write('\n');
		o.write('+');
		o.write('\n');
		if (b.qualOff[i + 1] > b.qualOff[i]) {
			o.write(b.quals, b.qualOff[i], b.qualOff[i + 1] - b.qualOff[i]);
		} else {
			Arrays.fill(line, (byte) '~');
			o.write(line);
		}
		o.write('\n');
	}

	/** MatcherReadEntry.writeMatchDetails (FastqKMerMatcher.java:723-756) from the device's runs of read i */
	private void writeKrakenLine(OutputStream o, Batch b, int i, int cls) throws IOException {
		long s0 = segOff.getLong(8 * i), s1 = segOff.getLong(8 * (i + 1));
		if (s1 == s0) {
			return; // no k-mer position: the reference has no buffer for this read
		}
		int len = (int) (b.offsets.getLong(8 * (i + 1)) - b.offsets.getLong(8 * i));
		int max = len - k + 1;
		StringBuilder sb = new StringBuilder();
		sb.append(cls >= 0 ? 'C' : 'U').append('\t');
		int end = indexOfBlank(b.descs, b.descOff[i] + 1, b.descOff[i + 1]);
		sb.append(new String(b.descs, b.descOff[i] + 1, end - b.descOff[i] - 1, StandardCharsets.ISO_8859_1)).append('\t');
		sb.append(cls >= 0 ? new String(taxidBytes[cls], StandardCharsets.UTF_8) : "0").append('\t').append(len).append('\t');
		for (long s = s0; s < s1; s++) {
			int code = segCodes.getInt((int) (4 * s));
			int start = segStarts.getInt((int) (4 * s));
			int stop = s + 1 < s1 ? segStarts.getInt((int) (4 * (s + 1))) : max;
			if (s > s0) {
				sb.append(' ');
			}
			sb.append(code >= 0 ? new String(taxidBytes[code], StandardCharsets.UTF_8) : code == -1 ? "0" : "A");
			sb.append(':').append(stop - start);
		}
		sb.append('\n');
		o.write(sb.toString().getBytes(StandardCharsets.ISO_8859_1));
	}

	/** device table -> CountsPerTaxid objects in statsIndex (what matchRead would have accumulated). */
	private void fillStatsFromDevice() {
		ByteBuffer table = direct(8L * GsGpuNative.N_COLS * nValues);
		ByteBuffer dtable = direct(8L * GsGpuNative.N_DCOLS * nValues);
		GsGpuNative.matchFinish(run, table, dtable);
		uniqueCounts = new long[nValues];
		if (maxKmerResCounts > 0) {
			ByteBuffer mc = direct(2L * maxKmerResCounts * (nValues + 1));
			GsGpuNative.matchMaxCounts(run, mc);
			maxCounts = new short[nValues + 1][maxKmerResCounts];
			for (int v = 0; v <= nValues; v++) {
				for (int j = 0; j < maxKmerResCounts; j++) {
					maxCounts[v][j] = mc.getShort(2 * (v * maxKmerResCounts + j));
				}
			}
		}
		for (int vi = 0; vi < nValues; vi++) {
			int row = 8 * GsGpuNative.N_COLS * vi;
			long reads = table.getLong(row + 8 * GsGpuNative.C_READS);
			long reads1KMer = table.getLong(row + 8 * GsGpuNative.C_READS_1KMER);
			uniqueCounts[vi] = table.getLong(row + 8 * GsGpuNative.C_UNIQUE_KMERS);
			if ((reads == 0 && reads1KMer == 0) || nodeOfValue[vi] == null) {
				continue; // the reference creates a CountsPerTaxid on the first hit k-mer or classified read only
			}
			CountsPerTaxid stats = getCountsPerTaxid(nodeOfValue[vi], vi);
			stats.reads = reads;
			stats.reads1KMer = reads1KMer;
			stats.readsKmers = table.getLong(row + 8 * GsGpuNative.C_READS_KMERS);
			stats.kmers = table.getLong(row + 8 * GsGpuNative.C_KMERS);
			stats.contigs = (int) table.getLong(row + 8 * GsGpuNative.C_CONTIGS); // Java field is int
			stats.contigLenSquaredSum = table.getLong(row + 8 * GsGpuNative.C_CONTIG_LEN_SQ_SUM);
			stats.maxContigLen = (int) table.getLong(row + 8 * GsGpuNative.C_MAX_CONTIG_LEN);
			stats.readsBPs = table.getLong(row + 8 * GsGpuNative.C_READS_BPS);
			int drow = 8 * GsGpuNative.N_DCOLS * vi;
			stats.errorSum = dtable.getDouble(drow + 8 * GsGpuNative.D_ERR_SUM);
			stats.errorSquaredSum = dtable.getDouble(drow + 8 * GsGpuNative.D_ERR_SQ_SUM);
			stats.classErrorSum = dtable.getDouble(drow + 8 * GsGpuNative.D_CLASS_ERR_SUM);
			stats.classErrorSquaredSum = dtable.getDouble(drow + 8 * GsGpuNative.D_CLASS_ERR_SQ_SUM);
			byte[] d = maxContigDescriptor[vi];
			if (d != null && stats.maxContigDescriptor.length > 0) { // copied and NUL-terminated as in :404-408
				int n = Math.min(d.length, stats.maxContigDescriptor.length - 1);
				System.arraycopy(d, 0, stats.maxContigDescriptor, 0, n);
				stats.maxContigDescriptor[n] = 0;
			}
		}
	}

	public void close() {
		if (batches != null) {
			for (Batch b : batches) {
				b.free();
			}
			batches = null;
		}
		GsGpuNative.matchDestroy(run);
		GsGpuNative.dbDestroy(db);
	}
}
