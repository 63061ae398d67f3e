"""No kernel of the shipped library may use scratch (private-segment) memory: a spill inside a k-loop is a
round trip through the slowest memory path of the chip for every lane.  Parsed from the code objects' metadata
(`.private_segment_fixed_size`, tools/kernel_resources.py) - no GPU needed.  The only exceptions are named below: code
paths that are off by default and measured slower (docs/HISTORY.md section 4), kept for their parity tests."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

# off-by-default experiment paths that are allowed to spill (substring of the demangled kernel name)
ALLOWED_SCRATCH = (
    "k_persist(",                              # persistent per-XCD schedule (DVITS_PERSIST=1)
)


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(KR.DEFAULT_LIB):
        pytest.skip("libdvits_hip.so is not built")
    ks = KR.kernel_resources(KR.DEFAULT_LIB)
    assert len(ks) > 50, "metadata parse found only %d kernels" % len(ks)
    return ks


def test_no_scratch_in_default_schedule_kernels(kernels):
    bad = [(k["name"], k["private_segment_fixed_size"]) for k in kernels
           if k["private_segment_fixed_size"] > 0 and not any(a in k["name"] for a in ALLOWED_SCRATCH)]
    assert not bad, "kernels with scratch memory:\n" + "\n".join("%6d B/lane  %s" % (b, n) for n, b in bad)


def test_config2_schedule_kernels_are_present_and_spill_free(kernels):
    """The kernels the BASELINE config-2 schedule launches (profiles/r02_rocprofv3_kernel_stats.csv), by name."""
    want = ["k_gemm<64, 64, 64, 2, 2, 3, 2>", "k_gemm<128, 128, 32, 2, 2, 3, 2>", "k_attention_frag<16, 8, 3, 1>",
            "k_attention_frag<32, 4, 3, 1>", "k_attention_frag<48, 4, 3, 2>", "k_attention_frag<64, 4, 3, 2>",
            "k_ff_split<256, 4, 2>", "k_ff_split<384, 8, 2>", "k_ff_split<512, 8, 1>",             # (round 5)
            "k_conv3<128>", "k_conv3<256>", "k_conv3<384>", "k_conv3<512>", "k_conv3s(", "k_conv3u<256>", "k_conv3u<384>", "k_conv3u<512>",
            "k_chain2<1, 0, true, false, false>", "k_chain2<2, 0, true, false, false>", "k_chain2<1, 1, false, true, false>",
            "k_chain2<2, 1, false, true, false>", "k_chain2<3, 1, false, true, false>", "k_chain2<3, 0, false, false, true>",
            "k_qkv_split<256, 0>", "k_qkv_split<384, 0>", "k_qkv_split<384, 1>", "k_qkv_split<256, 2>", "k_qkv_split<128, 0>",   # (round 6)
            "k_chain_ff(", "k_gn_apply(", "k_pack_input(", "k_lincomb("]
    for w in want:
        hit = [k for k in kernels if w in k["name"]]
        assert hit, "kernel %r not found in the library" % w
        for k in hit:
            assert k["private_segment_fixed_size"] == 0, (k["name"], k["private_segment_fixed_size"])
            assert k.get("vgpr_spill_count", 0) == 0, k      # (SGPR spills go to VGPR lanes, not to memory)


def test_vgpr_budget_matches_workgroup_size(kernels):
    """A 512-thread workgroup may use 256 VGPRs per lane, a 576-thread one (the chain kernels' ninth wave) 168."""
    for k in kernels:
        if k.get("max_flat_workgroup_size", 0) > 512:
            assert k["vgpr_count"] <= 168, k
        assert k["vgpr_count"] + k.get("agpr_count", 0) <= 512, k


def test_gemm_k_loops_keep_their_dma_queue():
    """No `s_waitcnt vmcnt(0)` inside any steady-state k-loop of k_gemm (tools/kloop_waits.py: kernels_gemm.hip compiled to
    assembly here, ~1 min).  The loop keeps three k-tiles of LDS-DMA in flight with hand-counted waits; a compiler-inserted
    full drain per k-tile - a register-allocation accident hipcc's wait insertion is free to commit whenever a pending
    compiler-visible load can reach the loop in the control-flow graph - cost 10 % end to end in round 4 and no test saw it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kloop_waits
    loops = kloop_waits.kloops(kloop_waits.compile_asm())
    assert len(loops) >= 20, "k_gemm instantiations not found in the assembly"
    assert all(v for v in loops.values()), "a k_gemm instantiation without a recognisable k-loop: %s" % [k for k, v in loops.items() if not v]
    bad = {k: v for k, v in loops.items() if any(l[4] for l in v)}
    assert not bad, "k-loops that drain the LDS-DMA queue: %s" % bad
