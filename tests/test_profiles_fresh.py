"""the evidence under profiles/ must belong to the code.  `scripts/final_evidence.sh` records the digest
of csrc/ + the C-ABI header (build.py --digest; the GPU box has no .git) in `<tag>_evidence_meta.json` next to the files it
produced; profiles/README.md names that file under "Final build".  This test fails when the tree's digest has moved on -
i.e. a kernel source was edited after the profiles were taken - and when a listed file is missing."""
import importlib.util
import json
import re

from conftest import ROOT


def _digest():
    spec = importlib.util.spec_from_file_location("sslam_build", ROOT / "opencv-simpleslam_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_digest()


def test_final_build_profiles_carry_the_digest_of_the_sources_in_the_tree():
    readme = (ROOT / "profiles" / "README.md").read_text()
    m = re.search(r"## Final build[^\n]*\n.*?`(r\d+\w*_evidence_meta\.json)`", readme, re.S)
    if not m:
        import pytest
        pytest.skip("profiles/README.md has no '## Final build' section naming an evidence_meta.json yet")
    meta = json.loads((ROOT / "profiles" / m.group(1)).read_text())
    assert meta["csrc_digest"] == _digest(), (
        f"profiles/{m.group(1)} was taken on sources {meta['csrc_digest']} (head {meta.get('git_head')}), the tree is "
        f"{_digest()}: re-run scripts/final_evidence.sh on the GPU box and copy its files into profiles/")
    missing = [f for f in meta["files"] if not (ROOT / "profiles" / f).exists()]
    assert not missing, f"listed in {m.group(1)} but not under profiles/: {missing}"
    # the bench line among them names the same sources
    bench = [f for f in meta["files"] if f.endswith("_bench_n1.json")]
    assert bench, "no bench line among the final-build files"
    line = json.loads((ROOT / "profiles" / bench[0]).read_text().strip().splitlines()[-1])
    assert line["build"]["csrc_digest"] == meta["csrc_digest"]
