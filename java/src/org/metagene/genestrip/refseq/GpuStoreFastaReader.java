/*
 * Reference-side glue for gs_dbbuild (include/gsgpu.h): the FASTA reader of FillDBGoal / DBGoal with the k-mer work moved to
 * the GPU.  Source only -- there is no JDK in the build image; tools/check_java_glue.py checks every member this class
 * touches against the reference's sources.
 *
 * The reference's readers (FillDBGoal.MyFastaReader, DBGoal.MyFastaReader; both AbstractStoreFastaReader) push every base
 * through a CGATLongBuffer and call store.put / store.update per k-mer (refseq/AbstractStoreFastaReader.java:87-115).  This
 * reader keeps everything that decides WHICH regions count and under WHICH node -- infoLine(), the accession map, the
 * per-taxid limits, reworkNode(): all inherited -- and only collects the bases of the included regions; a full buffer goes
 * to the device as one batch of regions (GsGpuNative.dbBuildAdd).  After the last file of the fill pass and of the update pass
 * the goal calls finish(), which returns the sorted k-mers and value indices a KMerSortedArray / gs_db_create takes.
 */
package org.metagene.genestrip.refseq;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.Map;
import java.util.Set;

import org.metagene.genestrip.gpu.GsGpuNative;
import org.metagene.genestrip.tax.Rank;
import org.metagene.genestrip.tax.TaxTree.TaxIdNode;

public class GpuStoreFastaReader extends AbstractRefSeqFastaReader {
	private final long builder;
	private final boolean update;
	private final boolean allRegions;
	private final Map<TaxIdNode, Integer> valueIndexOfNode;
	private final ByteBuffer bases;
	private final ByteBuffer offsets;
	private final ByteBuffer nodes;
	private final int maxRegions;
	private int regions;
	private boolean open;

	/**
	 * @param builder          handle of GsGpuNative.dbBuildBegin (shared by the fill and the update reader)
	 * @param update           false: FillDBGoal (regions of the requested taxa are stored), true: DBGoal (LCA updates)
	 * @param allRegions       DBGoal without minUpdate: every region with a node counts (goals/refseq/DBGoal.java:267-283)
	 * @param valueIndexOfNode the value index every tree node got when the tree was handed to dbBuildBegin
	 * @param batchBytes       bases collected before a batch goes to the device
	 */
	public GpuStoreFastaReader(long builder, boolean update, boolean allRegions, Map<TaxIdNode, Integer> valueIndexOfNode,
			int batchBytes, int bufferSize, Set<TaxIdNode> taxNodes, AccessionMap accessionMap, int k, int maxGenomesPerTaxId,
			Rank maxGenomesPerTaxIdRank, long maxKmersPerTaxId, int stepSize, boolean completeGenomesOnly,
			StringLong2DigitTrie regionsPerTaxid) {
		super(bufferSize, taxNodes, accessionMap, k, maxGenomesPerTaxId, maxGenomesPerTaxIdRank, maxKmersPerTaxId, stepSize,
				completeGenomesOnly, regionsPerTaxid);
		this.builder = builder;
		this.update = update;
		this.allRegions = allRegions;
		this.valueIndexOfNode = valueIndexOfNode;
		this.maxRegions = Math.max(1024, batchBytes / 256);
		this.bases = ByteBuffer.allocateDirect(batchBytes);
		this.offsets = ByteBuffer.allocateDirect(8 * (maxRegions + 1)).order(ByteOrder.nativeOrder());
		this.nodes = ByteBuffer.allocateDirect(4 * maxRegions).order(ByteOrder.nativeOrder());
		this.offsets.putLong(0, 0L);
	}

	@Override
	protected void infoLine() {
		super.infoLine();
		if (update && allRegions) {
			// DBGoal.MyFastaReader.infoLine: without minUpdate every region is walked, under its own node if it has one
			includeRegion = true;
			if (node != null) {
				node = reworkNode();
			}
		}
		open = includeRegion && node != null && valueIndexOfNode.containsKey(node);
		if (open && regions == maxRegions) {
			flush();
		}
	}

	@Override
	protected void dataLine() {
		if (!open || !isAllowMoreKmers()) {
			return;
		}
		// the line without its terminator(s), as AbstractStoreFastaReader.dataLine strips them
		int end = size;
		while (end > 0 && (target[end - 1] == '\n' || target[end - 1] == '\r')) {
			end--;
		}
		if (end > bases.remaining()) {
			// the region goes on in the next batch: a region border resets the k-mer window, so the last k - 1 bases are
			// handed in again in front of the rest (they form no k-mer by themselves).  (With stepSize > 1 the positions
			// that are sampled are counted from the region start: give such builds a batch that holds the longest region.)
			int keep = Math.min(k - 1, bases.position() - (int) offsets.getLong(8 * regions));
			byte[] tail = new byte[keep];
			for (int i = 0; i < keep; i++) {
				tail[i] = bases.get(bases.position() - keep + i);
			}
			closeRegion();
			flush();
			bases.put(tail);
		}
		bases.put(target, 0, end);
		bpsInRegion += end;
	}

	@Override
	protected void endRegion() {
		if (open) {
			closeRegion();
			open = false;
		}
		super.endRegion();
	}

	private void closeRegion() {
		nodes.putInt(4 * regions, valueIndexOfNode.get(node));
		regions++;
		offsets.putLong(8 * regions, bases.position());
	}

	/** hands the collected regions to the device (also called by the goal after the last file of a pass) */
	public void flush() {
		if (regions > 0) {
			GsGpuNative.dbBuildAdd(builder, bases, offsets, nodes, regions, update);
		}
		bases.clear();
		regions = 0;
		offsets.putLong(0, 0L);
	}
}
