"""ctypes binding of libdvits_hip.so (C ABI: include/dvits_hip.h).

The library is built in-tree by `build()` (hipcc, gfx950) and loaded lazily.  There is no
fallback: if the shared object is missing or fails to load, `lib()` raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DVITS_LIB_FILE: development aid - load another in-tree build of the same library, e.g. for same-box A/B runs)
LIB_PATH = os.environ.get("DVITS_LIB_FILE") or os.path.join(_HERE, "libdvits_hip.so")
CSRC = os.path.join(_HERE, "csrc")

PREC_BF16X3, PREC_BF16 = 0, 1
SOLVER_DPMPP, SOLVER_UNIPC_BH1, SOLVER_UNIPC_BH2, SOLVER_UNIPC_VARY = 0, 1, 2, 3
SKIP = {"time_uniform": 0, "time_quadratic": 1, "logSNR": 2}
SCHEDULE = {"discrete": 0, "linear": 1, "cosine": 2}
METHOD = {"multistep": 0, "singlestep": 1, "singlestep_fixed": 2}


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("n_levels", C.c_int32),
                ("block_out_channels", C.c_int32 * 6), ("layers_per_block", C.c_int32), ("num_heads", C.c_int32),
                ("cross_attention_dim", C.c_int32), ("norm_num_groups", C.c_int32), ("add_embed_heads", C.c_int32),
                ("norm_eps", C.c_float)]


class PencCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("hidden_channels", C.c_int32), ("out_channels", C.c_int32),
                ("n_layers", C.c_int32), ("num_heads", C.c_int32), ("ffn_kernel", C.c_int32)]


MODEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p)

# every symbol include/dvits_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "dv_last_error": (C.c_char_p, []),
    "dv_version": (C.c_char_p, []),
    "dv_unet_create": (C.c_int, [C.POINTER(UNetCfg), C.POINTER(C.c_void_p)]),
    "dv_unet_destroy": (None, [C.c_void_p]),
    "dv_unet_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "dv_unet_prepare": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "dv_unet_set_cond": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_unet_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "dv_unet_forward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int32]),
    "dv_unet_op_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "dv_unet_op_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.POINTER(C.c_double), C.c_char_p]),
    "dv_penc_create": (C.c_int, [C.POINTER(PencCfg), C.POINTER(C.c_void_p)]),
    "dv_penc_destroy": (None, [C.c_void_p]),
    "dv_penc_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "dv_penc_prepare": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "dv_penc_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_penc_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "dv_unet_time_family": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.POINTER(C.c_float),
                                      C.POINTER(C.c_int32)]),
    "dv_unet_persist_ticks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_int32]),
    "dv_unet_persist_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dv_unet_handover_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dv_unet_set_exclusive": (C.c_int, [C.c_void_p, C.c_int32]),
    "dv_unet_handover_reset": (C.c_int, [C.c_void_p]),
    "dv_unet_probe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "dv_sampler_plan": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(C.c_void_p)]),
    "dv_sampler_plan_ex": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_void_p)]),
    "dv_sampler_plan_sched": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_void_p)]),
    "dv_sampler_plan_method": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_void_p)]),
    "dv_plan_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "dv_plan_destroy": (None, [C.c_void_p]),
    "dv_plan_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "dv_plan_coefs": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "dv_plan_events": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32)]),
    "dv_sampler_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_sampler_run_custom": (C.c_int, [C.c_void_p, MODEL_FN, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dv_op_conv1d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p]),
    "dv_op_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "dv_op_linear_planes": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 6 + [C.c_void_p]),
    "dv_op_group_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_float, C.c_void_p]),
    "dv_op_attention": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p]),
}

_lib = None


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into libdvits_hip.so (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    proc = subprocess.run(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
    if proc.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libdvits_hip.so failed:\n" + proc.stdout[-4000:])
    return LIB_PATH


def lib():
    """The loaded library with argtypes set.  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libdvits_hip.so not found at %s — run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (or `make -C diff-vits_amd/csrc`) first; there is no fallback path"
                               % LIB_PATH)
        # torch's bundled HIP runtime must be THE runtime of the process (same SONAME as /opt/rocm's):
        # import torch first and pin its libamdhip64 so the loader resolves our NEEDED entry to it
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().dv_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what or "dvits_hip call", rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device/host pointer of a contiguous torch tensor (or None) as c_void_p."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor passed to the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
