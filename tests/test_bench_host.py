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


def test_roofline_families_sum_to_the_engine_count_and_the_headline_follows_the_most_time():
    """`roofline.families` must add up to `forward.engine_counted_gflop`, and `roofline.kernel / frac /
    frac_rocprofv3` must describe the family with the most kernel time (it was the GEMM kind by habit while the chain kind held
    49 % of the time); every family carries its rocprofv3 fraction and its share of the time."""
    import pytest
    peak = bench.PEAK_TFLOPS["bf16x3"]
    # (launches per forward, live us per launch, GFLOP per forward) - round 5's figures
    live = {"gemm": (62, 12.7, 114.6), "chain": (46, 24.6, 155.7), "attn": (22, 13.5, 41.0)}
    pmc = {"gemm": {"launches": 744, "avg_us_kernel_trace": 13.9, "hbm_bytes_per_launch": 21.8e6, "hbm_gbps": 1567.0, "mfma_util": 0.153},
           "chain": {"launches": 420, "avg_us_kernel_trace": 22.4, "hbm_bytes_per_launch": 42.5e6, "hbm_gbps": 1901.0, "mfma_util": 0.14},
           "ff_split": {"launches": 132, "avg_us_kernel_trace": 36.5, "hbm_bytes_per_launch": 87.8e6, "hbm_gbps": 2402.0, "mfma_util": 0.241},
           "attention": {"launches": 264, "avg_us_kernel_trace": 14.8, "hbm_bytes_per_launch": 17.4e6, "hbm_gbps": 1174.0, "mfma_util": 0.155},
           "build": {"dv_version": "x", "git_head": "y"}}
    total = sum(v[2] for v in live.values())
    fam, dom = bench.roofline_families(live, pmc, peak, total)
    assert dom == "chain"                                               # 46 x 24.6 us > 62 x 12.7 us
    assert abs(sum(f["gflop_per_forward"] for f in fam.values()) - total) < 1e-9
    assert abs(sum(f["time_share"] for f in fam.values()) - 1.0) < 1e-9 and abs(sum(f["flop_share"] for f in fam.values()) - 1.0) < 1e-9
    for k, f in fam.items():
        n, us, gf = live[k]
        assert f["frac"] == pytest.approx(gf * 1e9 / (n * us * 1e-6) / 1e12 / peak)
        assert f["frac_rocprofv3"] is not None and f["rocprofv3"]["avg_launch_us"] > 0
    # the chain kind's rocprofv3 average weighs k_chain* and k_ff_split by their launches
    assert fam["chain"]["rocprofv3"]["avg_launch_us"] == pytest.approx((420 * 22.4 + 132 * 36.5) / 552)
    assert fam["chain"]["frac_rocprofv3"] == pytest.approx(155.7e9 / (46 * fam["chain"]["rocprofv3"]["avg_launch_us"] * 1e-6) / 1e12 / peak)
    # a FLOP count that does not add up is an error, not a line
    with pytest.raises(AssertionError):
        bench.roofline_families(live, pmc, peak, total + 14.8)
    # without a PMC file of this build the live figures stand alone
    fam2, dom2 = bench.roofline_families(live, None, peak, total)
    assert dom2 == "chain" and all(f["frac_rocprofv3"] is None for f in fam2.values())
