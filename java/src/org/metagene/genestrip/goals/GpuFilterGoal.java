/*
 * The filter goal with the GPU filter behind it: FilterGoal.makeFile (reference: core/src/main/java/org/metagene/
 * genestrip/goals/FilterGoal.java:80-108) constructs the FastqBloomFilter inline, so the override repeats that method
 * with GpuFastqBloomFilter in its place.  Index filters the device does not replicate (anything that is not an
 * AbstractKMerBloomFilter) keep the reference's CPU path.  SOURCE ONLY (no JDK in the build container).
 */
package org.metagene.genestrip.goals;

import java.io.File;
import java.io.IOException;
import java.util.Map;

import org.metagene.genestrip.ExecutionContext;
import org.metagene.genestrip.GSConfigKey;
import org.metagene.genestrip.GSProject;
import org.metagene.genestrip.GSProject.GSFileType;
import org.metagene.genestrip.bloom.AbstractKMerBloomFilter;
import org.metagene.genestrip.bloom.GpuFastqBloomFilter;
import org.metagene.genestrip.bloom.KMerProbFilter;
import org.metagene.genestrip.io.StreamingResourceStream;
import org.metagene.genestrip.make.Goal;
import org.metagene.genestrip.make.ObjectGoal;

public class GpuFilterGoal<P extends GSProject> extends FilterGoal<P> {
	// (FilterGoal keeps its own copies private)
	private final LoadIndexGoal<P> index;
	private final ExecutionContext context;
	private final int device;

	@SafeVarargs
	public GpuFilterGoal(P project, ObjectGoal<Map<String, StreamingResourceStream>, P> fastqMapGoal,
			LoadIndexGoal<P> indexedGoal, ExecutionContext executorServiceBundle, int device, Goal<P>... deps) {
		super(project, fastqMapGoal, indexedGoal, executorServiceBundle, deps);
		this.index = indexedGoal;
		this.context = executorServiceBundle;
		this.device = device;
	}

	@Override
	protected void makeFile(File file) throws IOException {
		KMerProbFilter filter = index.get();
		if (!(filter instanceof AbstractKMerBloomFilter)) {
			super.makeFile(file);
			return;
		}
		GpuFastqBloomFilter f = null;
		try {
			P project = getProject();
			StreamingResourceStream resources = fileToFastqs.get(file);
			File dumpFile = booleanConfigValue(GSConfigKey.WRITE_DUMPED_FASTQ)
					? project.getOutputFile("dumped", null, file.getName(), GSFileType.FASTQ_RES, isUseGZip())
					: null;
			f = new GpuFastqBloomFilter(intConfigValue(GSConfigKey.KMER_SIZE), (AbstractKMerBloomFilter) filter,
					intConfigValue(GSConfigKey.MIN_POS_COUNT_FILTER), doubleConfigValue(GSConfigKey.POS_RATIO_FILTER),
					intConfigValue(GSConfigKey.INITIAL_READ_SIZE_BYTES), intConfigValue(GSConfigKey.THREAD_QUEUE_SIZE),
					context, booleanConfigValue(GSConfigKey.WITH_PROBS), device) {
				@Override
				protected boolean isProgressBar() {
					return booleanConfigValue(GSConfigKey.PROGRESS_BAR);
				}

				@Override
				protected String getProgressBarTaskName() {
					return getKey().getName();
				}
			};
			f.runFilter(resources, file, dumpFile);
		} finally {
			if (f != null) {
				f.dump();
				f.close();
			}
		}
	}
}
