"""Host-side pieces of bench.py that need no GPU: the FLOP model the roofline uses, and the rule that the committed PMC
figures are quoted only for the build they were collected on (they used to go stale silently)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_flops_model_matches_the_survey_counts():
    # 39.51 GF per sample at T=1024, L=256; 9.48 at T=256 (L=256); 87.55 at T=2048
    assert abs(bench.flops_model(1, 1024, 256) / 1e9 - 39.51) < 0.15
    assert abs(bench.flops_model(1, 256, 256) / 1e9 - 9.48) < 0.15
    assert abs(bench.flops_model(1, 2048, 256) / 1e9 - 87.55) < 0.3
    assert abs(bench.flops_model(8, 1024, 256) / 1e9 - 316.1) < 0.5


def test_pmc_figures_are_tied_to_the_build():
    from diff_vits_amd import _lib
    ver = _lib.lib().dv_version().decode()
    assert "src=" in ver and not ver.endswith("src=unknown")          # the Makefile hashed csrc/ into the library
    assert bench.pmc_identity_ok({"build": {"dv_version": ver, "git_head": "x"}})
    assert not bench.pmc_identity_ok({"build": {"dv_version": ver.replace("src=", "src=0")}})
    assert not bench.pmc_identity_ok({"gemm": {}})                      # round-2 style JSON without a stamp
    assert not bench.pmc_identity_ok(None)


def test_committed_pmc_json_is_stamped():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_roofline.json")))
    files = [f for f in files if os.path.basename(f) >= "r03"]          # (round 2's file predates the stamp)
    assert files
    for p in files:
        d = json.load(open(p))
        assert d["build"]["dv_version"].startswith("dvits_hip") and d["build"]["git_head"]
        # (the schedule of round 4's final build has no k_gn_apply launch left: that family is present up to round 3 only)
        for fam in ("gemm", "chain", "attention") + (("gn_apply",) if "gn_apply" in d else ()):
            assert d[fam]["launches"] > 0 and d[fam]["hbm_bytes_per_launch"] > 0
