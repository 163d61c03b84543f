"""tools/exp/attn_gemm_experiments.patch (the alternative attention bodies and every timing-only switch of rounds 1-3) must
keep applying to the product tree: the A/B scripts under tools/ run in a scratch copy made by tools/exp/with_experiments.sh, and a
kernel change that silently breaks the patch would take all of them down (it happened once in round 4).  CPU only."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "tools", "exp", "attn_gemm_experiments.patch")


@pytest.mark.skipif(shutil.which("patch") is None, reason="no `patch` binary on this host")
def test_experiments_patch_applies_to_the_product_tree(tmp_path):
    files = sorted(set(re.findall(r"^\+\+\+ b/(\S+)", open(PATCH).read(), flags=re.M)))
    assert "trajectorycrafter_amd/csrc/attn_fwd.hip" in files and len(files) >= 8
    for f in files:
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(ROOT, f), dst)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
    # and the product sources themselves carry none of the switches the patch re-introduces
    for f in files:
        if f.endswith(".hip"):
            src = open(os.path.join(ROOT, f)).read()
            assert not re.search(r"#\s*if(n?def)?\s+.*\bTCX_(EXP_|A4_|GEMM_EXP_|CONV_EXP_|NORM_EXP_|ATTN_STAMP|ATTN_FINE_SUM)", src), f
