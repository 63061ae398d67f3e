"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/dvits_hip.h declares (no compute calls here — those are the -m gpu tests).  The host-only
entry points (sampler plans) are exercised since they never touch the device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from diff_vits_amd import _lib, synth


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dvits_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dv_[a-z0-9_]+)\s*\(", text)) - {"dv_model_fn"})


def test_header_symbols_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libdvits_hip.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.dv_version().decode().startswith("dvits_hip")


def test_create_validates_config(lib):
    c = _lib.UNetCfg()
    c.in_channels, c.out_channels, c.n_levels = 208, 80, 4
    for i, ch in enumerate((128, 256, 384, 500)):          # 500 is not a multiple of 32
        c.block_out_channels[i] = ch
    c.layers_per_block, c.num_heads, c.cross_attention_dim, c.norm_num_groups, c.add_embed_heads = 2, 8, 128, 8, 64
    h = C.c_void_p()
    assert lib.dv_unet_create(C.byref(c), C.byref(h)) == -1
    assert b"multiple of 32" in lib.dv_last_error()


def test_plan_is_host_only_and_counts_nfe(lib):
    betas = synth.make_betas()
    for solver, steps, order in [(0, 50, 2), (2, 30, 2), (1, 20, 3), (0, 8, 2)]:
        h = C.c_void_p()
        _lib.check(lib.dv_sampler_plan(solver, betas.ctypes.data_as(C.c_void_p), len(betas), steps, order, 0, 1, C.byref(h)))
        nfe = C.c_int32()
        ts = np.zeros(steps + 1)
        tin = np.zeros(steps)
        _lib.check(lib.dv_plan_info(h, C.byref(nfe), tin.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p)))
        assert nfe.value == steps                       # NFE == steps for both solvers
        assert ts[0] == 1.0 and abs(ts[-1] - 1e-3) < 1e-9 and np.all(np.diff(ts) < 0)
        assert abs(tin[0] - 999.0) < 1e-3
        lib.dv_plan_destroy(h)
    h = C.c_void_p()
    assert lib.dv_sampler_plan(0, betas.ctypes.data_as(C.c_void_p), len(betas), 1, 2, 0, 1, C.byref(h)) == -1
    assert lib.dv_sampler_plan(0, betas.ctypes.data_as(C.c_void_p), len(betas), 10, 4, 0, 1, C.byref(h)) == -1
