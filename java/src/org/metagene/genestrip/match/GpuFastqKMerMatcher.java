/*
 * GPU-backed drop-in for FastqKMerMatcher (reference: core/src/main/java/org/metagene/genestrip/match/
 * FastqKMerMatcher.java).  SOURCE ONLY -- not compiled in the build container (no JDK); see INTEGRATION.md.
 *
 * It lives in the reference's own package because it fills the protected CountsPerTaxid fields
 * (CountsPerTaxid.java:127-159) and the protected statsIndex array (FastqKMerMatcher.java:76).
 *
 * How it hooks in (all seams are the reference's own virtual methods):
 *   - the parser stays the reference's (AbstractFastqReader.doReadFastq); with a zero-consumer execution context it
 *     calls nextEntry -> matchRead(entry, 0) on the producer thread (AbstractFastqReader.java:350-352);
 *   - matchRead() only appends the read to a direct-buffer batch; a full batch goes to gs_match_submit;
 *   - processFastqStreams() flushes the last batch and turns the device table into CountsPerTaxid objects before
 *     runMatcher() collects them (FastqKMerMatcher.java:199-204);
 *   - unique k-mer counts come from the device bitmap, so runMatcher() is called without a KMerUniqueCounterBits and
 *     the counts are patched into the result afterwards (FastqKMerMatcher.java:207-213).
 */
package org.metagene.genestrip.match;

import java.io.File;
import java.io.IOException;
import java.io.InputStream;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;

import org.metagene.genestrip.DefaultExecutionContext;
import org.metagene.genestrip.ExecutionContext;
import org.metagene.genestrip.gpu.GsGpuNative;
import org.metagene.genestrip.io.StreamingResourceStream;
import org.metagene.genestrip.store.KMerStore;
import org.metagene.genestrip.store.KMerUniqueCounterBits;
import org.metagene.genestrip.tax.SmallTaxTree;
import org.metagene.genestrip.tax.SmallTaxTree.SmallTaxIdNode;

public class GpuFastqKMerMatcher extends FastqKMerMatcher {
	private static final int BATCH_READS = 1 << 20;
	private static final int BATCH_BYTES = 256 << 20;

	private final long db;
	private final long run;
	private final int nValues;
	private final SmallTaxIdNode[] nodeOfValue; // value index -> tree node (null: value without node)

	private final ByteBuffer seq = direct(BATCH_BYTES);
	private final ByteBuffer offsets = direct(8 * (BATCH_READS + 1));
	private final ByteBuffer classVi = direct(4 * BATCH_READS);
	private final ByteBuffer flags = direct(BATCH_READS);
	private int batchReads;
	private long batchFirstReadNo;
	private long globalReadNo; // file-order read number over all files of this runMatcher call
	private long[] uniqueCounts;

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
		nodeOfValue = new SmallTaxIdNode[nValues];
		db = upload(kmerStore, taxTree, device);
		run = GsGpuNative.matchBegin(db, taxTree != null, true, maxPaths, threshold, maxReadTaxErrorCount,
				maxReadClassErrorCount);
	}

	private static ByteBuffer direct(int bytes) {
		return ByteBuffer.allocateDirect(bytes).order(ByteOrder.nativeOrder());
	}

	/** KMerStore.visit (KMerSortedArray.java:426-439) yields (kmer, valueIndex) in ascending kmer order. */
	private long upload(KMerStore<SmallTaxIdNode> store, SmallTaxTree tree, int device) {
		long n = store.getEntries();
		// for stores beyond 2^27 entries the buffers would be filled and handed over in slices (not shown)
		ByteBuffer kmers = direct((int) (8 * n));
		ByteBuffer vidx = direct((int) (4 * n));
		store.visit((s, kmer, index, pos) -> {
			kmers.putLong(kmer);
			vidx.putInt(index);
		});
		for (int v = 0; v < nValues; v++) {
			nodeOfValue[v] = store.getValueForIndex(v);
		}
		ByteBuffer parent = null;
		if (tree != null) {
			parent = direct(4 * nValues);
			for (int v = 0; v < nValues; v++) {
				SmallTaxIdNode node = nodeOfValue[v];
				// every tree node has a store index (Database.initStoreIndices, Database.java:107-128)
				parent.putInt(node == null ? -2 : node.getParent() == null ? -1 : node.getParent().getStoreIndex());
			}
		}
		return GsGpuNative.dbCreate(device, store.getK(), n, kmers, vidx, nValues, parent);
	}

	@Override
	public MatchingResult runMatcher(StreamingResourceStream fastqs, File filteredFile, File krakenOutStyleFile,
			KMerUniqueCounterBits uniqueCounter) throws IOException {
		GsGpuNative.matchReset(run); // stats + unique bitmap start cleared per key (FastqKMerMatcher.java:192-193)
		globalReadNo = 0;
		uniqueCounts = null;
		MatchingResult res = super.runMatcher(fastqs, filteredFile, krakenOutStyleFile, null);
		for (CountsPerTaxid stats : res.getTaxid2Stats().values()) {
			int vi = kmerStore.getIndexForValue(taxTreeNode(stats.getTaxid()));
			stats.uniqueKmers = uniqueCounter == null || vi < 0 ? -1 : uniqueCounts[vi];
		}
		return res;
	}

	private SmallTaxIdNode taxTreeNode(String taxid) {
		for (SmallTaxIdNode n : nodeOfValue) {
			if (n != null && taxid.equals(n.getTaxId())) {
				return n;
			}
		}
		return null;
	}

	@Override
	public void processFastqStreams(StreamingResourceStream fastqs) throws IOException {
		super.processFastqStreams(fastqs);
		fillStatsFromDevice();
	}

	@Override
	protected void readFastq(InputStream inputStream, boolean fasta) throws IOException {
		super.readFastq(inputStream, fasta);
		flush(); // per-file read numbers restart (AbstractFastqReader.java:226-228); keep batches inside one file
	}

	/** Called by the (final) nextEntry for every parsed read; only batches the read. */
	@Override
	protected boolean matchRead(final MatcherReadEntry entry, final int index) {
		if (batchReads == BATCH_READS || seq.remaining() < entry.readSize) {
			flush();
		}
		if (batchReads == 0) {
			batchFirstReadNo = globalReadNo;
			offsets.clear();
			offsets.putLong(0);
		}
		seq.put(entry.read, 0, entry.readSize);
		offsets.putLong(seq.position());
		batchReads++;
		globalReadNo++;
		return false; // the per-read outcome arrives with the batch
	}

	@Override
	protected void afterMatch(MatcherReadEntry myEntry, boolean found) throws IOException {
		// filtered-FASTQ / Kraken-style writeback is driven from flush() with the per-read class + flags of the
		// batch (classVi, flags); it needs the descriptors and qualities of the batch kept beside `seq` (omitted
		// here: this sketch covers the CSV path, which is the bit-exact contract).
	}

	private void flush() {
		if (batchReads == 0) {
			return;
		}
		GsGpuNative.matchSubmit(run, seq, offsets, batchReads, batchFirstReadNo, classVi, flags);
		seq.clear();
		batchReads = 0;
	}

	/** device table -> CountsPerTaxid objects in statsIndex (what matchRead would have accumulated). */
	private void fillStatsFromDevice() {
		ByteBuffer table = direct(8 * GsGpuNative.N_COLS * nValues);
		ByteBuffer dtable = direct(8 * GsGpuNative.N_DCOLS * nValues);
		GsGpuNative.matchFinish(run, table, dtable);
		uniqueCounts = new long[nValues];
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
			// max contig descriptor: C_MAX_CONTIG_READ_NO is the global file-order read number of the first read with
			// the maximum; the descriptor is looked up from the batch that contained it (kept by flush(), omitted).
		}
	}

	public void close() {
		GsGpuNative.matchDestroy(run);
		GsGpuNative.dbDestroy(db);
	}
}
