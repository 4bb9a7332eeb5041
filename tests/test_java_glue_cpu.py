"""The reference-side Java glue (java/) cannot be compiled here -- no JDK in the image.  What can be checked is checked:
* tools/check_java_glue.py: every reference class / member / constructor the glue uses exists in the reference's sources
  with that name, arity and a visibility the glue can reach (needs /root/reference: build container only);
* the checker itself flags the mistakes it is meant to catch (a deliberately broken class);
* java/jni/gsgpu_jni.c passes `gcc -fsyntax-only` against a declarations-only jni.h and implements exactly the native
  methods GsGpuNative.java declares, with matching argument counts, calling only functions include/gsgpu.h declares."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"
CHECK = os.path.join(ROOT, "tools", "check_java_glue.py")
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "core", "src", "main", "java")),
                               reason="the reference's sources exist in the build container only")


@needs_ref
def test_glue_uses_only_what_the_reference_declares():
    r = subprocess.run([sys.executable, CHECK], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    assert re.search(r"checked (\d+) glue classes", r.stdout) and int(re.search(r"checked (\d+) glue", r.stdout).group(1)) >= 6


BROKEN = '''
package org.metagene.genestrip.match;

import java.io.IOException;
import org.metagene.genestrip.ExecutionContext;
import org.metagene.genestrip.GSConfigKey;
import org.metagene.genestrip.store.KMerStore;
import org.metagene.genestrip.tax.SmallTaxTree;
import org.metagene.genestrip.tax.SmallTaxTree.SmallTaxIdNode;

public class Broken extends FastqKMerMatcher {
	public Broken(KMerStore<SmallTaxIdNode> store, ExecutionContext bundle, SmallTaxTree tree) {
		super(store, 1, 2, bundle);                              // no such constructor arity
	}

	@Override
	protected void nextEntry(ReadEntry entry, int index) throws IOException {   // final in FastqKMerMatcher
	}

	@Override
	protected void noSuchHook(int x) {                           // nothing to override
	}

	void use(CountsPerTaxid stats, MatcherReadEntry e) {
		stats.noSuchField = 1;                                   // unknown member
		Object o = afterMatchCallback;                           // private field of the superclass
		int n = kmerStore.getNValues(1, 2);                      // wrong arity
		Object key = GSConfigKey.NO_SUCH_KEY;                    // unknown constant
		e.growPrintBufferTypo(3);                                // unknown method on a reference type
		FastqKMerMatcher m = new FastqKMerMatcher(kmerStore) {   // constructor arity + bogus override in an anonymous class
			@Override
			protected boolean matchReadTypo(MatcherReadEntry entry, int index) {
				return false;
			}
		};
	}
}
'''


@needs_ref
def test_checker_flags_what_it_is_meant_to_catch(tmp_path):
    d = tmp_path / "org" / "metagene" / "genestrip" / "match"
    d.mkdir(parents=True)
    (d / "Broken.java").write_text(BROKEN)
    r = subprocess.run([sys.executable, CHECK, "--glue", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1
    out = r.stdout
    for needle in ("super(...) with 4 arguments", "@Override nextEntry", "is final", "@Override noSuchHook", "noSuchField",
                   "afterMatchCallback is private", "getNValues(...): no overload takes 2", "NO_SUCH_KEY",
                   "growPrintBufferTypo", "new FastqKMerMatcher(...) with 1 arguments", "@Override matchReadTypo"):
        assert needle in out, (needle, out)


def test_jni_shim_syntax_and_coverage():
    src = os.path.join(ROOT, "java", "jni", "gsgpu_jni.c")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror", "-fsyntax-only",
                        "-I" + os.path.join(ROOT, "tests", "native", "jni_stub"), "-I" + os.path.join(ROOT, "include"), src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    java = open(os.path.join(ROOT, "java", "src", "org", "metagene", "genestrip", "gpu", "GsGpuNative.java")).read()
    natives = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
               for m in re.finditer(r"static\s+native\s+[\w\[\]]+\s+(\w+)\s*\(([^)]*)\)", java, re.S)}
    c = open(src).read()
    impl = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()]) - 2  # JNIEnv *, jclass
            for m in re.finditer(r"JNAME\((\w+)\)\s*\(([^)]*)\)", c, re.S)}
    assert natives == impl, (set(natives) ^ set(impl), {k: (natives[k], impl[k]) for k in natives if k in impl and natives[k] != impl[k]})
    assert len(natives) >= 26
    header = open(os.path.join(ROOT, "include", "gsgpu.h")).read() + open(os.path.join(ROOT, "include", "gshost.h")).read()
    declared = set(re.findall(r"\b(gs_[a-z_0-9]+)\s*\(", header))
    called = set(re.findall(r"\b(gs_[a-z_0-9]+)\s*\(", c))
    assert called <= declared, called - declared
    # the file-level flows of the reference's seam (runMatcher / runFilter over local files) are bound (missing 2)
    for name in ("hostMatchFiles", "hostMatchRun", "hostMatchInto", "hostFilterFiles", "hostLastError"):
        assert name in natives, name
    glue = open(os.path.join(ROOT, "java", "src", "org", "metagene", "genestrip", "match", "GpuFastqKMerMatcher.java")).read()
    assert "GsGpuNative.hostMatchRun(" in glue and "GsGpuNative.matchSubmitAsync(" in glue and "GsGpuNative.matchSubmit(" not in glue
    glue = open(os.path.join(ROOT, "java", "src", "org", "metagene", "genestrip", "bloom", "GpuFastqBloomFilter.java")).read()
    assert "GsGpuNative.hostFilterFiles(" in glue
