/*
 * GPU-backed drop-in for FastqBloomFilter (reference: core/src/main/java/org/metagene/genestrip/bloom/
 * FastqBloomFilter.java).  SOURCE ONLY -- not compiled in the build container (no JDK); tools/check_java_glue.py checks
 * every reference member used here against the reference's sources.  See INTEGRATION.md.
 *
 * It lives in the reference's bloom package because the index filter's state is held in protected fields
 * (AbstractKMerBloomFilter.java:52-62: bits, bitVector, hashes, hashFactors; LargeBitVector.bits/largeBits are public).
 * The device copy replicates the bit array and the hash factors exactly, so false positives are identical.
 *
 * Hook: GpuFilterGoal.makeFile constructs this class where FilterGoal.makeFile (goals/FilterGoal.java:80-108) constructs
 * the FastqBloomFilter.  Two flows, chosen per runFilter call:
 *   FILES -- every resource is a local file (StreamingFileResource): runFilter hands the paths to gs_host_filter_files
 *     (GsGpuNative.hostFilterFiles) and the whole goal runs below the JVM: gzip input inflated on the device, the filter on the
 *     device text, accepted / dumped records gathered AND gzip-compressed on the device (gzipFastqOutput is the default,
 *     GSConfigKey.java:155), only compressed bytes cross PCIe: 11 Gbp/s gz -> gz at 100 M reads.
 *   STREAMS -- anything else: reads are batched in nextEntry(); the accept flags come back per batch and the reads are rewritten
 *     in input order exactly like the reference's nextEntry (FastqBloomFilter.java:92-105) -- at the speed of the reference's
 *     parser thread (0.2-0.7 Gbp/s).  The two output streams of the reference are private, so runFilter is overridden with streams
 *     of its own.
 */
package org.metagene.genestrip.bloom;

import java.io.File;
import java.io.IOException;
import java.io.OutputStream;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.List;

import org.metagene.genestrip.DefaultExecutionContext;
import org.metagene.genestrip.ExecutionContext;
import org.metagene.genestrip.gpu.GsGpuNative;
import org.metagene.genestrip.io.StreamProvider;
import org.metagene.genestrip.io.StreamingFileResource;
import org.metagene.genestrip.io.StreamingResource;
import org.metagene.genestrip.io.StreamingResourceStream;

public class GpuFastqBloomFilter extends FastqBloomFilter {
	private static final int BATCH_READS = 1 << 20;
	private static final int BATCH_BYTES = 256 << 20;

	private final long bloom;
	private final int kk;
	private final int minPos;
	private final double ratio;
	private final boolean keepQualities;
	private final ByteBuffer seq = ByteBuffer.allocateDirect(BATCH_BYTES).order(ByteOrder.nativeOrder());
	private final ByteBuffer offsets = ByteBuffer.allocateDirect(8 * (BATCH_READS + 1)).order(ByteOrder.nativeOrder());
	private final ByteBuffer accept = ByteBuffer.allocateDirect(BATCH_READS);
	// descriptor / read / quality copies of the batch, for the rewrite in input order
	private final List<byte[][]> pending = new ArrayList<>();
	private OutputStream accepted, rejected;

	public GpuFastqBloomFilter(int k, AbstractKMerBloomFilter filter, int minPosCount, double positiveRatio,
			int initialReadSize, int maxQueueSize, ExecutionContext bundle, boolean withProbs, int device) {
		super(k, filter, minPosCount, positiveRatio, initialReadSize, maxQueueSize,
				new DefaultExecutionContext(Thread.currentThread(), 0, bundle.getLogUpdateCycle()), withProbs);
		this.kk = k;
		this.minPos = minPosCount;
		this.ratio = positiveRatio;
		this.keepQualities = withProbs;
		if (filter.bitVector.isLarge()) {
			throw new UnsupportedOperationException("index filters beyond 2^31 words: hand largeBits over segment by segment");
		}
		long[] words = filter.bitVector.bits;
		ByteBuffer w = ByteBuffer.allocateDirect(8 * words.length).order(ByteOrder.nativeOrder());
		w.asLongBuffer().put(words);
		int kind = filter instanceof XORKMerBloomFilter ? GsGpuNative.BLOOM_XOR : GsGpuNative.BLOOM_MURMUR;
		bloom = GsGpuNative.bloomCreate(device, kind, filter.bits, filter.hashes, filter.hashFactors, w, words.length);
	}

	/** the paths of the resources if every one of them is a local FASTQ file, else null (FASTA resources take the STREAMS flow too) */
	private String[] localFiles(StreamingResourceStream fastqs) {
		List<String> paths = new ArrayList<>();
		for (StreamingResource r : fastqs) {
			if (!(r instanceof StreamingFileResource) || isFastaStream(r)) {
				return null;
			}
			paths.add(((StreamingFileResource) r).getFile().getPath());
		}
		return paths.toArray(new String[0]);
	}

	@Override
	public void runFilter(StreamingResourceStream fastqs, File filteredFile, File restFile) throws IOException {
		String[] files = localFiles(fastqs);
		if (files != null) { // FILES flow: FastqBloomFilter.runFilter (:80-89) in one native call
			long[] totals = new long[4];
			GsGpuNative.hostFilterFiles(bloom, kk, minPos, ratio, files, filteredFile == null ? null : filteredFile.getPath(),
					restFile == null ? null : restFile.getPath(), keepQualities, totals);
			totalReads = totals[0];
			totalKMers = totals[1];
			totalBPs = totals[2];
			return;
		}
		try (OutputStream a = filteredFile != null ? StreamProvider.getOutputStreamForFile(filteredFile) : null;
				OutputStream r = restFile != null ? StreamProvider.getOutputStreamForFile(restFile) : null) {
			accepted = a;
			rejected = r;
			processFastqStreams(fastqs);
			flush();
		}
		accepted = null;
		rejected = null;
	}

	@Override
	protected void nextEntry(ReadEntry readStruct, int index) throws IOException {
		if (pending.size() == BATCH_READS || seq.remaining() < readStruct.readSize) {
			flush();
		}
		if (pending.isEmpty()) {
			offsets.clear();
			offsets.putLong(0);
		}
		seq.put(readStruct.read, 0, readStruct.readSize);
		offsets.putLong(seq.position());
		pending.add(new byte[][] { Arrays.copyOf(readStruct.readDescriptor, readStruct.readDescriptorSize),
				Arrays.copyOf(readStruct.read, readStruct.readSize),
				readStruct.readProbs == null || readStruct.readProbsSize < 0 ? null
						: Arrays.copyOf(readStruct.readProbs, readStruct.readProbsSize) });
	}

	/** submit the batch, then write every read to the accepted or the rejected stream in input order */
	public void flush() throws IOException {
		if (pending.isEmpty()) {
			return;
		}
		GsGpuNative.filterSubmit(bloom, kk, minPos, ratio, seq, offsets, pending.size(), accept);
		for (int i = 0; i < pending.size(); i++) {
			writeRecord(pending.get(i), accept.get(i) != 0); // ReadEntry.write layout (AbstractFastqReader.java:570-584)
		}
		pending.clear();
		seq.clear();
	}

	private void writeRecord(byte[][] rec, boolean ok) throws IOException {
		OutputStream out = ok ? accepted : rejected;
		if (out == null) {
			return;
		}
		out.write(rec[0]);
		out.write('\n');
		out.write(rec[1]);
		out.write('\n');
		out.write('+');
		out.write('\n');
		if (rec[2] != null) {
			out.write(rec[2]);
		} else {
			for (int i = 0; i < rec[1].length; i++) {
				out.write('~');
			}
		}
		out.write('\n');
		updateWriteStats();
	}

	public void close() {
		GsGpuNative.bloomDestroy(bloom);
	}
}
