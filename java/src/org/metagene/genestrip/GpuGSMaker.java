/*
 * The one place where the GPU goals enter the goal graph: GSMaker builds the chains for `match` and `filter` in two
 * protected factory methods (reference: core/src/main/java/org/metagene/genestrip/GSMaker.java:560-583 and :601-625);
 * the overrides repeat them with GpuMatchResultGoal / GpuFilterGoal as the last link.  Use it wherever the reference
 * creates a GSMaker (Main, API users): `new GpuGSMaker<>(project, device)`.  SOURCE ONLY (no JDK in the build container).
 */
package org.metagene.genestrip;

import java.util.Map;

import org.metagene.genestrip.goals.FastqDownloadsGoal;
import org.metagene.genestrip.goals.FastqMapGoal;
import org.metagene.genestrip.goals.FastqMapTransformGoal;
import org.metagene.genestrip.goals.FilterGoal;
import org.metagene.genestrip.goals.GpuFilterGoal;
import org.metagene.genestrip.goals.GpuMatchResultGoal;
import org.metagene.genestrip.goals.LoadDBGoal;
import org.metagene.genestrip.goals.LoadIndexGoal;
import org.metagene.genestrip.goals.MatchResultGoal;
import org.metagene.genestrip.io.StreamingResourceStream;
import org.metagene.genestrip.make.ObjectGoal;

public class GpuGSMaker<P extends GSProject> extends GSMaker<P> {
	private final int device;

	public GpuGSMaker(P project, int device) {
		super(project);
		this.device = device;
	}

	@Override
	@SuppressWarnings({ "unchecked", "rawtypes" })
	protected MatchResultGoal<P> createGoalChainForMatchResult(boolean lr, String key, String... pathsOrURLs) {
		ObjectGoal<Map<String, StreamingResourceStream>, P> fastqMapGoal = new FastqMapGoal(getProject(), true,
				getGoal(GSGoalKey.SETUP)) {
			@Override
			protected void doMakeThis() {
				Map<String, StreamingResourceStream> map = createFastqMap(key, pathsOrURLs, null, null, null);
				set(map);
			}
		};
		ObjectGoal<Map<String, StreamingResourceStream>, P> fastqMapTransfGoal = new FastqMapTransformGoal(getProject(),
				true, fastqMapGoal, getGoal(GSGoalKey.SETUP));
		FastqDownloadsGoal<P> fastqDownloadsGoal = new FastqDownloadsGoal(getProject(), true, fastqMapGoal,
				fastqMapTransfGoal, getGoal(GSGoalKey.SETUP));
		LoadDBGoal<P> loadDBGoal = getLoadDBGoal();
		return new GpuMatchResultGoal<P>(getProject(), (lr ? GSGoalKey.MATCHRESLR : GSGoalKey.MATCHRES),
				fastqMapTransfGoal, loadDBGoal, getExecutionContext(getProject()), device, getGoal(GSGoalKey.SETUP),
				fastqDownloadsGoal);
	}

	@Override
	@SuppressWarnings({ "unchecked", "rawtypes" })
	protected FilterGoal<P> createGoalChainForFilter(String key, String... pathsOrURLs) {
		ObjectGoal<Map<String, StreamingResourceStream>, P> fastqMapGoal = new FastqMapGoal(getProject(), true,
				getGoal(GSGoalKey.SETUP)) {
			@Override
			protected void doMakeThis() {
				Map<String, StreamingResourceStream> map = createFastqMap(key, pathsOrURLs, null, null, null);
				set(map);
			}
		};
		ObjectGoal<Map<String, StreamingResourceStream>, P> fastqMapTransfGoal = new FastqMapTransformGoal(getProject(),
				true, fastqMapGoal, getGoal(GSGoalKey.SETUP));
		FastqDownloadsGoal<P> fastqDownloadsGoal = new FastqDownloadsGoal(getProject(), true, fastqMapGoal,
				fastqMapTransfGoal, getGoal(GSGoalKey.SETUP));
		LoadIndexGoal<P> bloomIndexedGoal = (LoadIndexGoal) getGoal(GSGoalKey.LOAD_INDEX);
		return new GpuFilterGoal<P>(getProject(), fastqMapTransfGoal, bloomIndexedGoal, getExecutionContext(getProject()),
				device, getGoal(GSGoalKey.SETUP), fastqDownloadsGoal);
	}
}
