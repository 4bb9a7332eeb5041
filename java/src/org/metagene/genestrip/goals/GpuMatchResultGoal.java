/*
 * The factory hook: MatchResultGoal.createMatcher (reference: core/src/main/java/org/metagene/genestrip/goals/
 * MatchResultGoal.java:174-194) is protected and exists exactly for this kind of override -- the reference's own
 * tests override it the same way (core/src/test/java/org/metagene/genestrip/goals/refseq/
 * ComprehensiveFilterTest.java:114-150).  SOURCE ONLY (no JDK in the build container); see INTEGRATION.md.
 *
 * GpuGSMaker.createGoalChainForMatchResult constructs this goal instead of MatchResultGoal; everything else
 * (Database.load, CSV reporting, goal graph) stays untouched.
 */
package org.metagene.genestrip.goals;

import java.util.Map;

import org.metagene.genestrip.ExecutionContext;
import org.metagene.genestrip.GSConfigKey;
import org.metagene.genestrip.GSProject;
import org.metagene.genestrip.io.StreamingResourceStream;
import org.metagene.genestrip.make.Goal;
import org.metagene.genestrip.make.GoalKey;
import org.metagene.genestrip.make.ObjectGoal;
import org.metagene.genestrip.match.FastqKMerMatcher;
import org.metagene.genestrip.match.GpuFastqKMerMatcher;
import org.metagene.genestrip.store.Database;
import org.metagene.genestrip.store.KMerStore;
import org.metagene.genestrip.tax.SmallTaxTree;
import org.metagene.genestrip.tax.SmallTaxTree.SmallTaxIdNode;

public class GpuMatchResultGoal<P extends GSProject> extends MatchResultGoal<P> {
	private final int device;

	@SafeVarargs
	public GpuMatchResultGoal(P project, GoalKey key, ObjectGoal<Map<String, StreamingResourceStream>, P> fastqMapGoal,
			ObjectGoal<Database, P> storeGoal, ExecutionContext bundle, int device, Goal<P>... deps) {
		super(project, key, fastqMapGoal, storeGoal, bundle, deps);
		this.device = device;
	}

	@Override
	protected FastqKMerMatcher createMatcher(KMerStore<SmallTaxIdNode> store, SmallTaxTree taxTree,
			ExecutionContext bundle, boolean withProbs, String dbMD5) {
		return new GpuFastqKMerMatcher(store, intConfigValue(GSConfigKey.INITIAL_READ_SIZE_BYTES),
				intConfigValue(GSConfigKey.THREAD_QUEUE_SIZE), bundle, withProbs,
				intConfigValue(GSConfigKey.MAX_KMER_RES_COUNTS), taxTree,
				intConfigValue(GSConfigKey.MAX_CLASSIFICATION_PATHS),
				doubleConfigValue(GSConfigKey.MAX_READ_TAX_ERROR_COUNT),
				doubleConfigValue(GSConfigKey.MAX_READ_CLASS_ERROR_COUNT), booleanConfigValue(GSConfigKey.WRITE_ALL),
				intConfigValue(GSConfigKey.MIN_KMERS_FOR_CLASS), dbMD5, device) {
			@Override
			protected boolean isProgressBar() {
				return booleanConfigValue(GSConfigKey.PROGRESS_BAR);
			}

			@Override
			protected String getProgressBarTaskName() {
				return getKey().getName();
			}
		};
	}
}
