/*
 * GPU-backed drop-in for FastqBloomFilter (reference: core/src/main/java/org/metagene/genestrip/bloom/
 * FastqBloomFilter.java).  SOURCE ONLY -- not compiled in the build container (no JDK); see INTEGRATION.md.
 *
 * It lives in the reference's bloom package because the index filter's state is held in protected fields
 * (AbstractKMerBloomFilter.java:52-62: bits, bitVector, hashes, hashFactors; LargeBitVector.bits/largeBits are public).
 * The device copy replicates the bit array and the hash factors exactly, so false positives are identical.
 *
 * Hook: FilterGoal.makeFile (goals/FilterGoal.java:80-108) constructs the FastqBloomFilter; a subclass of FilterGoal
 * constructs this class instead.  Reads are batched in nextEntry(); the accept flags come back per batch and the
 * reads are rewritten in input order exactly like the reference's nextEntry (FastqBloomFilter.java:92-105).
 */
package org.metagene.genestrip.bloom;

import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.List;

import org.metagene.genestrip.DefaultExecutionContext;
import org.metagene.genestrip.ExecutionContext;
import org.metagene.genestrip.gpu.GsGpuNative;

public class GpuFastqBloomFilter extends FastqBloomFilter {
	private static final int BATCH_READS = 1 << 20;
	private static final int BATCH_BYTES = 256 << 20;

	private final long bloom;
	private final int kk;
	private final int minPos;
	private final double ratio;
	private final ByteBuffer seq = ByteBuffer.allocateDirect(BATCH_BYTES).order(ByteOrder.nativeOrder());
	private final ByteBuffer offsets = ByteBuffer.allocateDirect(8 * (BATCH_READS + 1)).order(ByteOrder.nativeOrder());
	private final ByteBuffer accept = ByteBuffer.allocateDirect(BATCH_READS);
	// descriptor / read / quality copies of the batch, for the rewrite in input order
	private final List<byte[][]> pending = new ArrayList<>();

	public GpuFastqBloomFilter(int k, AbstractKMerBloomFilter filter, int minPosCount, double positiveRatio,
			int initialReadSize, int maxQueueSize, ExecutionContext bundle, boolean withProbs, int device) {
		super(k, filter, minPosCount, positiveRatio, initialReadSize, maxQueueSize,
				new DefaultExecutionContext(Thread.currentThread(), 0, bundle.getLogUpdateCycle()), withProbs);
		this.kk = k;
		this.minPos = minPosCount;
		this.ratio = positiveRatio;
		long[] words = filter.bitVector.bits; // small variant; largeBits is handed over segment by segment
		ByteBuffer w = ByteBuffer.allocateDirect(8 * words.length).order(ByteOrder.nativeOrder());
		w.asLongBuffer().put(words);
		int kind = filter instanceof XORKMerBloomFilter ? 0 : 1; // GS_BLOOM_XOR / GS_BLOOM_MURMUR
		bloom = GsGpuNative.bloomCreate(device, kind, filter.bits, filter.hashes, filter.hashFactors, w, words.length);
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
		pending.add(new byte[][] { java.util.Arrays.copyOf(readStruct.readDescriptor, readStruct.readDescriptorSize),
				java.util.Arrays.copyOf(readStruct.read, readStruct.readSize),
				readStruct.readProbs == null || readStruct.readProbsSize < 0 ? null
						: java.util.Arrays.copyOf(readStruct.readProbs, readStruct.readProbsSize) });
	}

	/** submit the batch, then write every read to `indexed` or `notIndexed` in input order */
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

	private void writeRecord(byte[][] rec, boolean accepted) throws IOException {
		java.io.OutputStream out = accepted ? indexed : notIndexed;
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
	}

	public void close() {
		GsGpuNative.bloomDestroy(bloom);
	}
}
