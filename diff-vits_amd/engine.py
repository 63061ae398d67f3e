"""Python handle on the native denoiser engine (libdvits_hip.so) for one
`UNet1DConditionModel` instance: pushes the module's parameters through the C ABI,
prepares the kernel schedule for the current (B, T, L) and runs forwards on the current
HIP stream.  torch is used only for device memory and streams.
"""
import atexit
import collections
import ctypes as C
import os
import sys
import time
import warnings
import weakref

import torch

from . import _lib

_PRECISIONS = {"bf16x3": _lib.PREC_BF16X3, "bf16": _lib.PREC_BF16}


def default_precision():
    return os.environ.get("DVITS_PRECISION", "bf16x3")


def verify_handover_default():
    """How results of a schedule with in-launch hand-overs (GroupNorm exchange, split feed-forward, split-K) are verified before
    they leave the engine (DVITS_HANDOVER_VERIFY):
      "lazy"  (default) - no host synchronisation in the steady state.  The first result of a freshly planned schedule is checked
                the old way (wait for the stream, read the flag, repeat on the fallback schedule if a wait timed out: a GPU that
                is shared from the start never hands out a wrong tensor); after that a time-out - flagged in host-mapped memory by
                the kernel that gave up - is noticed by the NEXT call on the engine, which recovers and raises ("repeat the run"),
                or by an explicit `engine.wait()`, which returns False.  This package's own loops (model3.NaturalSpeech2.sample,
                tts_infer.synthesize) call wait() once per utterance and repeat it themselves.
      "eager" / "1"    - every forward / sampler run waits for the stream and is repeated transparently (round 4's behaviour:
                one host synchronisation per denoiser evaluation in a Python-driven solver loop).
      "off" / "0"      - never waits, never repeats (measurements)."""
    v = os.environ.get("DVITS_HANDOVER_VERIFY", "lazy").lower()
    return {"0": "off", "off": "off", "false": "off", "1": "eager", "eager": "eager", "true": "eager"}.get(v, "lazy")


class HandoverLost(RuntimeError):
    """Raised by the call that NOTICES an earlier run's timed-out in-launch hand-over (lazy verification): that earlier result
    was invalid; the engine has already recovered (fallback schedule) - repeat what was computed since the last verified point."""


_LIVE_ENGINES = weakref.WeakSet()      # engines with results that may still be unverified at interpreter exit


def _verify_at_exit():
    """Interpreter exit: the LAST results of a process (an unmodified reference loop never calls wait()) are verified here -
    nothing can be repeated any more, but a timed-out hand-over is reported on stderr instead of passing silently."""
    for eng in list(_LIVE_ENGINES):
        try:
            eng._verify_final("at interpreter exit")
        except Exception:
            pass


atexit.register(_verify_at_exit)


class UNetEngine:
    def __init__(self, module):
        self.module = module
        cfg = module.config
        chans = tuple(cfg["block_out_channels"])
        c = _lib.UNetCfg()
        c.in_channels = cfg["in_channels"]
        c.out_channels = cfg["out_channels"]
        c.n_levels = len(chans)
        for i, ch in enumerate(chans):
            c.block_out_channels[i] = ch
        c.layers_per_block = cfg["layers_per_block"]
        c.num_heads = cfg["attention_head_dim"]
        c.cross_attention_dim = cfg["cross_attention_dim"]
        c.norm_num_groups = cfg["norm_num_groups"]
        c.add_embed_heads = cfg["addition_embed_type_num_heads"]
        c.norm_eps = cfg["norm_eps"]
        self.in_channels, self.out_channels = c.in_channels, c.out_channels
        self.cross_dim = c.cross_attention_dim
        self.n_up = len(chans) - 1
        self._cfg = c
        # Prepared schedules are kept per shape (LRU of native handles, DVITS_PLAN_CACHE entries, default 4): utterances of
        # a length seen before replay their schedule - and the sampler its captured hipGraph - instead of re-planning
        # (~8 ms) and re-capturing (~100 ms for a 30-step loop).  Every handle owns its packed weights (~260 MB for the
        # 64.7 M-parameter denoiser in split-bf16 + fragment-major copies: nothing against 288 GB).
        self._plans = collections.OrderedDict()     # (B, T, L, precision, force_up) -> _Slot
        self._cap = max(1, int(os.environ.get("DVITS_PLAN_CACHE", "4")))
        self._cur = self._new_slot()
        self._sig = None                # current signature of the module's parameters / buffers
        self.precision = None
        self.cond_serial = 0            # bumped by every set_cond (callers that cache conditioning compare it)
        self.prepare_serial = 0         # bumped whenever another schedule becomes current (the caller must set_cond again)
        self.plan_builds = 0            # native prepares really run (a cached shape does not count)
        self._fwd_cond = None           # forward(): the (enc, mask) tensors the engine is currently conditioned on
        self.handover_downgraded = False  # an in-launch hand-over timed out: this engine runs the fallback schedule (until it is retried)
        self.verify_handover = verify_handover_default()   # "lazy" | "eager" | "off" (True / False are accepted: eager / off)
        self._probation = 1             # results still verified eagerly in lazy mode (reset whenever a schedule is planned)
        self.unverified_results = 0     # results handed out since the last verification point (lazy mode)
        self.host_syncs = 0             # host synchronisations this engine has issued for verification (tests, bench)
        # a downgraded engine tries the fused schedule again after this many clean results (doubled after every failed retry;
        # DVITS_HANDOVER_RETRY=0: never - round 4's permanent downgrade)
        self._retry_after = max(0, int(os.environ.get("DVITS_HANDOVER_RETRY", "64")))
        self._clean_since_downgrade = 0
        self._retry_pending = False     # the clean results are in: the fused schedule is planned again at the next utterance boundary
        self.handover_retries = 0
        # lazy verification, the windows nobody's next call closes:
        #   * unverified results + a host that was idle for DVITS_VERIFY_IDLE_MS: the stream has drained long ago - the next call
        #     verifies (one wait that returns at once) before it enqueues anything;
        #   * more unverified results than DVITS_UNVERIFIED_WARN (a few utterances' evaluations): warn once, naming wait();
        #   * the engine is destroyed / the interpreter exits with unverified results: verified there, reported on stderr.
        self._idle_ms = float(os.environ.get("DVITS_VERIFY_IDLE_MS", "50"))
        self._warn_unverified = int(os.environ.get("DVITS_UNVERIFIED_WARN", "512"))
        self._warned_unverified = False
        self._last_call = time.monotonic()
        self.final_check_failed = False  # set by the destruction / exit check when it found a timed-out hand-over
        _LIVE_ENGINES.add(self)

    class _Slot:
        __slots__ = ("h", "weight_sig", "prepared", "cond_keepalive")

        def __init__(self, h):
            self.h, self.weight_sig, self.prepared, self.cond_keepalive = h, None, None, None

    def _new_slot(self):
        h = C.c_void_p()
        _lib.check(_lib.lib().dv_unet_create(C.byref(self._cfg), C.byref(h)), "dv_unet_create")
        if not getattr(self, "_exclusive", True):
            _lib.check(_lib.lib().dv_unet_set_exclusive(h, 0), "dv_unet_set_exclusive")
        return UNetEngine._Slot(h)

    def set_exclusive(self, exclusive):
        """exclusive=False: this engine's kernels may run beside kernels of other streams (e.g. several engines driven
        concurrently on one device) - the in-launch GroupNorm hand-over, which needs every workgroup of a launch resident
        at once, is then not used (see dv_unet_set_exclusive).  Takes effect at the next prepare."""
        self._exclusive = bool(exclusive)
        # every cached schedule is stale: the handles of the non-current slots are destroyed (each holds its packed
        # weights and slab), the current one is flagged and re-planned by the next prepare
        self._prepared = None
        _lib.check(_lib.lib().dv_unet_set_exclusive(self._cur.h, int(self._exclusive)), "dv_unet_set_exclusive")
        self._fwd_cond = None

    # the current schedule's native handle / key (read by the samplers, the bench and the tests)
    @property
    def _h(self):
        return self._cur.h

    @property
    def _prepared(self):
        return self._cur.prepared

    @_prepared.setter
    def _prepared(self, v):
        self._cur.prepared = v
        if v is None:                    # forget every cached schedule (weights / precision changed, or a test asks)
            for sl in self._plans.values():
                if sl is not self._cur and sl.h and sl.h.value:
                    _lib.lib().dv_unet_destroy(sl.h)
                    sl.h = C.c_void_p()
            self._plans.clear()

    def _slots(self):
        """Every live native handle of this engine (the current schedule and the cached ones)."""
        seen = {}
        for sl in list(self._plans.values()) + [self._cur]:
            if sl is not None and sl.h and sl.h.value:
                seen[id(sl)] = sl
        return list(seen.values())

    def _verify_final(self, when):
        """Last chance: results were handed out unverified and nobody will call the engine again."""
        if getattr(self, "unverified_results", 0) <= 0 or not torch.cuda.is_initialized():
            return
        torch.cuda.synchronize()
        self.unverified_results = 0
        if any(self._slot_status(sl)[1] for sl in self._slots()):
            self.final_check_failed = True
            msg = ("diff_vits_amd: an in-kernel hand-over timed out in one of the LAST denoiser runs of this engine (noticed %s: "
                   "the GPU was shared with other kernels) - results returned since the last verified point are INVALID; "
                   "call engine.wait() after each utterance and repeat the run when it returns False" % when)
            sys.stderr.write(msg + "\n")
            warnings.warn(msg, RuntimeWarning, stacklevel=2)

    def __del__(self):
        try:
            self._verify_final("when the engine was destroyed")
        except Exception:
            pass
        try:
            slots = {id(sl): sl for sl in list(getattr(self, "_plans", {}).values()) + [getattr(self, "_cur", None)] if sl is not None}
            for sl in slots.values():
                if sl.h and sl.h.value:
                    _lib.lib().dv_unet_destroy(sl.h)
                    sl.h = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def sync_weights(self, precision=None):
        """Hand the module's current parameters to the engine if they changed since the last
        call (load_state_dict / .to() / in-place updates bump the version counters)."""
        precision = precision or self.precision or default_precision()
        if precision not in _PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
        # cheap change detector first (no state_dict() with its 701 prefixed names per call): storage address + version
        # counter of every parameter / buffer; load_state_dict, .to() and in-place updates all move one of them
        m = self.module
        sig = tuple((v.data_ptr(), v._version) for v in m.parameters()) + tuple((v.data_ptr(), v._version) for v in m.buffers())
        if sig != self._sig:             # new parameters: every cached schedule is stale
            self._sig = sig
            self._prepared = None
        if precision != self.precision:
            self.precision = precision
            self._prepared = None
        self._upload(self._cur)

    def _upload(self, slot):
        """Give `slot`'s native handle the module's current parameters if it does not have them yet."""
        if slot.weight_sig == self._sig:
            return
        L = _lib.lib()
        for name, t in self.module.state_dict().items():
            if not t.is_cuda:
                raise RuntimeError("backend='hip' needs the module on a GPU (parameter %s is on %s); "
                                   "call .to('cuda') or construct with backend='torch'" % (name, t.device))
            t32 = t.detach().to(torch.float32).contiguous()
            shape = (C.c_int64 * t32.dim())(*t32.shape)
            _lib.check(L.dv_unet_set_weight(slot.h, name.encode(), _lib.ptr(t32), shape, t32.dim()),
                       "dv_unet_set_weight(%s)" % name)
        torch.cuda.synchronize()
        slot.weight_sig = self._sig
        slot.prepared = None

    # ------------------------------------------------------------------ schedule
    def prepare(self, B, T, L, force_upsample_size=None):
        if force_upsample_size is None:
            # reference unet_1d_condition.py:789-797: tests BOTH the channel and the frame axis
            up = 2 ** self.n_up
            force_upsample_size = (self.in_channels % up != 0) or (T % up != 0)
        key = (B, T, L, self.precision, bool(force_upsample_size))
        if key == self._cur.prepared:
            return key
        # another shape becomes current: a cached schedule if there is one, else a free / the least recently used handle
        slot = self._plans.get(key)
        if slot is None:
            if self._cur.prepared is None and self._cur not in self._plans.values():
                slot = self._cur                                  # the very first schedule of this engine
            elif len(self._plans) < self._cap:
                slot = self._new_slot()
            else:
                _, slot = self._plans.popitem(last=False)         # evict: its handle is re-planned (packed weights kept)
                slot.prepared = None
        self._upload(slot)
        if slot.prepared != key:
            _lib.check(_lib.lib().dv_unet_prepare(slot.h, B, T, L, _PRECISIONS[self.precision],
                                                  int(bool(force_upsample_size))), "dv_unet_prepare")
            slot.prepared = key
            self.plan_builds += 1
            self._probation = 1          # (lazy verification: the first result of a new schedule is checked before it leaves)
        self._plans[key] = slot
        self._plans.move_to_end(key)
        self._cur = slot
        self.prepare_serial += 1         # (a cached schedule still holds ITS last conditioning: callers must set_cond again)
        self._fwd_cond = None
        return key

    def set_cond(self, enc, bias=None):
        """Hoisted, step-invariant conditioning: enc [B, L, D]; bias additive [B,1,L] / [B,L] or None."""
        enc = enc.detach().to(torch.float32).contiguous()
        if bias is not None:
            bias = bias.detach().to(torch.float32).reshape(enc.shape[0], enc.shape[1]).contiguous()
        if self._retry_pending and self._cur.prepared is not None:
            # a new conditioning = a new utterance: the safe place to go back to the fused schedule (re-plan, same shape)
            B, T, L, _, fu = self._cur.prepared
            self._apply_retry()
            self.prepare(B, T, L, fu)
        self._cur.cond_keepalive = (enc, bias)
        self.cond_serial += 1
        self.check_or_recover(_lib.lib().dv_unet_set_cond(self._h, _lib.ptr(enc), _lib.ptr(bias), _lib.stream_ptr()),
                              "dv_unet_set_cond")

    def check_or_recover(self, rc, what):
        """`_lib.check` for the calls that start with the native health check (set_cond, forward, dv_sampler_run): if the
        failure is an EARLIER run's timed-out in-launch hand-over (noticed now: runs are asynchronous), the engine recovers -
        downgraded to the separate-GroupNorm schedule, it needs prepare + set_cond again - and the caller is told to repeat."""
        if rc != 0 and getattr(self, "_exclusive", True) and self._cur.prepared is not None and self._any_timed_out():
            self.recover_handover()
            raise HandoverLost("diff_vits_amd: the previous denoiser run was invalid (in-kernel GroupNorm hand-over timed out "
                               "on a shared GPU); the engine has switched to the separate-GroupNorm schedule - repeat the run")
        _lib.check(rc, what)

    def eval(self, x, cond, t, out=None):
        """One denoiser evaluation with the conditioning set by set_cond.
        x [B, cx, T] (+ cond [B, in_channels - cx, T]) channels-first; t [B] float32."""
        B, cx, T = x.shape
        if out is None:
            out = torch.empty((B, self.out_channels, T), device=x.device, dtype=torch.float32)
        self.before_enqueue()
        rc = _lib.lib().dv_unet_forward(self._h, _lib.ptr(x), cx, _lib.ptr(cond), _lib.ptr(t), _lib.ptr(out), _lib.stream_ptr())
        self.check_or_recover(rc, "dv_unet_forward")
        return out

    # ------------------------------------------------------------------ module-level forward
    def forward(self, sample, timesteps, enc, bias, mask_src=None):
        """UNet1DConditionModel.forward on GPU tensors (reference unet_1d_condition.py:743-1037).

        The step-invariant conditioning (pooled-text embedding, the 16 cross-attention K/V projections, the mask bias:
        `set_cond`) is re-run only when the caller passes different `encoder_hidden_states` / mask tensors (or edits them
        in place, or the schedule / weights changed): an unmodified reference caller - the solver's lambda calling
        `unet(x, t, enc, encoder_attention_mask=mask)` every step - pays for it once per utterance, not once per step.
        The cache entry owns the two tensors, so a recycled address cannot alias it.  `mask_src` is the caller's mask
        object (`bias` is derived from it afresh on every call and cannot serve as identity)."""
        if not sample.is_cuda:
            raise RuntimeError("backend='hip' needs GPU tensors; got sample on %s (use backend='torch' for CPU)"
                               % sample.device)
        B, cin, T = sample.shape
        if cin != self.in_channels:
            raise RuntimeError("expected %d input channels, got %d" % (self.in_channels, cin))
        if enc.shape[0] != B or enc.shape[2] != self.cross_dim:
            raise RuntimeError("encoder_hidden_states must be [B=%d, L, %d], got %s" % (B, self.cross_dim, tuple(enc.shape)))
        self.prepare(B, T, enc.shape[1])
        c = self._fwd_cond
        keyed = bias is None or mask_src is not None
        hit = (keyed and c is not None and c[0] is enc and c[1] == enc._version and c[2] is mask_src
               and (mask_src is None or c[3] == mask_src._version) and c[4] == (self.cond_serial, self.prepare_serial))
        if not hit:
            self.set_cond(enc.to(sample.device), None if bias is None else bias.to(sample.device))
            self._fwd_cond = (enc, enc._version, mask_src, None if mask_src is None else mask_src._version,
                              (self.cond_serial, self.prepare_serial)) if keyed else None
        x = sample.detach().to(torch.float32).contiguous()
        t = timesteps.detach().to(device=sample.device, dtype=torch.float32).contiguous()
        y = self.eval(x, None, t)
        # The result leaves the engine here (an unmodified reference caller, a Python-driven solver loop, the bench's parity
        # forward).  Verification policy: verify_handover_default() - in the steady state of the default ("lazy") no host
        # synchronisation happens here; a timed-out hand-over is noticed by the next call / by wait().
        def again():
            self.prepare(B, T, enc.shape[1])
            self.set_cond(enc.to(sample.device), None if bias is None else bias.to(sample.device))
            self._fwd_cond = None
            return self.eval(x, None, t)
        y2 = self.result_leaves(again)
        if y2 is not None:
            y = y2
        return y if sample.dtype == torch.float32 else y.to(sample.dtype)

    # ------------------------------------------------------------------ verification of in-launch hand-overs
    def _verify_mode(self):
        v = self.verify_handover
        if v is True:
            return "eager"
        if v is False or v is None:
            return "off"
        return v

    def before_enqueue(self):
        """Called before a forward / sampler run is enqueued.  Lazy verification leaves results unverified until "the next
        call"; if that call comes after the host was idle for a while (the mel went to the vocoder, the process waited for the
        next request) the stream has drained long ago: verify NOW - a wait that returns at once - so the time-out of the run
        before the pause is reported here ("repeat the run") whichever schedule slot it ran on."""
        now = time.monotonic()
        idle = 1e3 * (now - self._last_call)
        self._last_call = now
        if (self.unverified_results > 0 and idle >= self._idle_ms and self._verify_mode() == "lazy"
                and not torch.cuda.is_current_stream_capturing()):
            if not self.wait(_boundary=False):
                raise HandoverLost("diff_vits_amd: a denoiser run before the last pause was invalid (in-kernel hand-over timed "
                                   "out on a shared GPU); the engine has switched to the fallback schedule - repeat the run")

    def _count_unverified(self):
        self.unverified_results += 1
        if self.unverified_results > self._warn_unverified and not self._warned_unverified and self._verify_mode() == "lazy":
            self._warned_unverified = True
            warnings.warn("diff_vits_amd: %d denoiser results have left the engine without a verification point; call "
                          "`unet.hip_engine().wait()` once per utterance (True: the results since the last call are valid; "
                          "False: repeat them) or set DVITS_HANDOVER_VERIFY=eager" % self.unverified_results,
                          RuntimeWarning, stacklevel=4)

    def result_leaves(self, again):
        """Called by forward() / NativeUNetModel.run_plan once a result has been enqueued and before it is returned.
        `again()` re-enqueues the same work on the re-prepared engine and returns its result.  Returns None if the result
        stands (verified, or handed out unverified: lazy mode in the steady state), else the repeated result."""
        if not self.handover_active():
            self._note_clean()
            return None
        mode = self._verify_mode()
        if mode == "off" or torch.cuda.is_current_stream_capturing():   # (a host wait would invalidate a capture in progress)
            self._count_unverified()
            return None
        if mode == "lazy" and self._probation <= 0:
            self._count_unverified()
            return None
        self.host_syncs += 1
        torch.cuda.current_stream().synchronize()
        self.unverified_results = 0
        if self.recover_handover():
            out = again()
            self.host_syncs += 1
            torch.cuda.current_stream().synchronize()
            if self._any_timed_out():
                raise RuntimeError("in-kernel hand-over timed out on the fallback schedule (it has none): internal error")
            return out
        self._probation -= 1
        self._note_clean()
        return None

    def wait(self, _boundary=True):
        """Wait for everything enqueued on the current stream and verify the in-launch hand-overs of the results handed out since
        the last verification point.  True: they are valid.  False: a hand-over timed out somewhere among them (a foreign
        kernel shared the GPU); the engine has recovered - it is on the fallback schedule and needs prepare + set_cond, which
        forward() / a sampler run do by themselves - and the caller repeats what it computed since the last wait()."""
        self.host_syncs += 1
        torch.cuda.current_stream().synchronize()
        self.unverified_results = 0
        self._last_call = time.monotonic()
        if self.recover_handover():      # (looks at EVERY cached schedule: the flag lives in the handle the run used)
            return False
        if self._retry_pending and _boundary:          # an utterance boundary: the safe place to plan the fused schedule again
            self._apply_retry()
        return True

    def _note_clean(self):
        """One more result without a time-out.  A downgraded engine goes back to the fused schedule after `_retry_after` of them -
        not here, in the middle of somebody's solver loop (re-planning destroys the cached schedules and their graphs), but at the
        next utterance boundary: `wait()` or the next `set_cond` (its first result is verified eagerly; a second failure doubles
        the distance)."""
        if not self.handover_downgraded or self._retry_after <= 0 or self._retry_pending:
            return
        self._clean_since_downgrade += 1
        if self._clean_since_downgrade >= self._retry_after:
            self._clean_since_downgrade = 0
            self._retry_pending = True

    def _apply_retry(self):
        self._retry_pending = False
        self.handover_downgraded = False
        self.handover_retries += 1
        self.set_exclusive(True)

    def stats(self):
        n, f = C.c_int64(), C.c_double()
        _lib.check(_lib.lib().dv_unet_stats(self._h, C.byref(n), C.byref(f)), "dv_unet_stats")
        return n.value, f.value

    def profile_forward(self, x, cond, t):
        """One eager forward with HIP events around every operation -> list of (kind, flops, ms, desc)."""
        n_ops = C.c_int32()
        _lib.check(_lib.lib().dv_unet_op_count(self._h, C.byref(n_ops)), "dv_unet_op_count")
        n = n_ops.value
        ms = (C.c_float * n)()
        out = torch.empty((x.shape[0], self.out_channels, x.shape[2]), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().dv_unet_forward_timed(self._h, _lib.ptr(x), x.shape[1], _lib.ptr(cond), _lib.ptr(t),
                                                    _lib.ptr(out), _lib.stream_ptr(), ms, n), "dv_unet_forward_timed")
        rows = []
        kind = C.create_string_buffer(16)
        desc = C.create_string_buffer(128)
        fl = C.c_double()
        for i in range(n):
            _lib.check(_lib.lib().dv_unet_op_info(self._h, i, kind, C.byref(fl), desc), "dv_unet_op_info")
            rows.append((kind.value.decode(), fl.value, float(ms[i]), desc.value.decode()))
        return rows

    def persist_status(self):
        """(operations inside the persistent launch, error flag) - see dv_unet_persist_status."""
        n, err = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().dv_unet_persist_status(self._h, C.byref(n), C.byref(err)), "dv_unet_persist_status")
        return n.value, err.value

    def handover_status(self):
        """(GEMMs that finish their consumer's GroupNorm in the epilogue, timed-out flag) of the CURRENT schedule - see
        dv_unet_handover_status.  Meaningful after the stream has drained; a set flag also fails every later forward / sampler
        run on that schedule."""
        return self._slot_status(self._cur)

    @staticmethod
    def _slot_status(slot):
        n, bad = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().dv_unet_handover_status(slot.h, C.byref(n), C.byref(bad)), "dv_unet_handover_status")
        return n.value, bad.value

    def _any_timed_out(self):
        """The time-out flag lives in the native handle a run used, and the engine keeps several (one per cached shape): a
        time-out on shape A must not be missed because shape B is current now."""
        return any(self._slot_status(sl)[1] for sl in self._slots())

    def handover_active(self):
        """True while the current schedule finishes GroupNorms inside producer GEMMs (in-launch hand-over: needs the device
        to itself for the duration of each such launch)."""
        return getattr(self, "_exclusive", True) and not self.handover_downgraded and self.handover_status()[0] > 0

    def recover_handover(self):
        """Deal with a timed-out in-launch hand-over (a foreign kernel - another stream, another process - kept workgroups
        of a GroupNorm-finishing GEMM off the CUs past the bounded wait; the results of that run are invalid).  Clears the
        native flag, switches this engine to the fallback schedule (GroupNorm by k_gn_apply launches, the feed-forward blocks as
        two GEMMs: no in-launch waits; the fused schedule is tried again after DVITS_HANDOVER_RETRY clean results), and reports it once.  Returns True if a time-out had happened: the caller repeats the lost run (the sampler
        path does: sampler/_plan.py) - the engine must be prepared and conditioned again first."""
        bad = [sl for sl in self._slots() if self._slot_status(sl)[1]]
        if not bad:
            return False
        for sl in bad:
            _lib.check(_lib.lib().dv_unet_handover_reset(sl.h), "dv_unet_handover_reset")
        self._retry_pending = False
        if self.handover_retries == 0:
            warnings.warn("diff_vits_amd: an in-kernel GroupNorm hand-over timed out (the GPU is shared with other kernels); "
                          "this engine now runs GroupNorm as separate launches (DVITS_GNX=0 schedule) and the lost run is repeated",
                          RuntimeWarning, stacklevel=3)
        else:                                       # a retry of the fused schedule failed as well: try again later
            self._retry_after = min(self._retry_after * 2, 1 << 20)
        self.handover_downgraded = True
        self._clean_since_downgrade = 0
        self.set_exclusive(False)
        return True

    def time_family(self, kind, reps=5):
        """Average launch duration (us) of one kernel family: its launches of the schedule replayed back to back
        between one HIP event pair (after a completed forward).  Returns (us_per_launch, launches_per_forward)."""
        ms, n = C.c_float(), C.c_int32()
        _lib.check(_lib.lib().dv_unet_time_family(self._h, kind.encode(), reps, _lib.stream_ptr(), C.byref(ms),
                                                  C.byref(n)), "dv_unet_time_family")
        return 1e3 * ms.value / n.value, n.value // reps

    def probe(self, name):
        """Named intermediate [B, T, C] of the last forward (needs DVITS_KEEP_INTERMEDIATES=1)."""
        dims = (C.c_int64 * 3)()
        _lib.check(_lib.lib().dv_unet_probe(self._h, name.encode(), None, 0, dims), "dv_unet_probe")
        out = torch.empty(tuple(dims), dtype=torch.float32)
        _lib.check(_lib.lib().dv_unet_probe(self._h, name.encode(), C.c_void_p(out.data_ptr()), out.numel(), dims),
                   "dv_unet_probe")
        return out

    @property
    def handle(self):
        return self._h


class PromptEncoderEngine:
    """Native PromptEncoder (dv_penc_*, include/dvits_hip.h) for one mirror module (diff_vits_amd.model3.PromptEncoder)."""

    def __init__(self, module):
        self.module = module
        c = _lib.PencCfg()
        c.in_channels, c.hidden_channels, c.out_channels = module.in_channels, module.hidden_size, module.out_channels
        c.n_layers, c.num_heads, c.ffn_kernel = module.num_layers, 8, 9
        self.out_channels = c.out_channels
        self._h = C.c_void_p()
        _lib.check(_lib.lib().dv_penc_create(C.byref(c), C.byref(self._h)), "dv_penc_create")
        self._weight_sig = None
        self._prepared = None
        self.precision = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().dv_penc_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def sync_weights(self, precision=None):
        precision = precision or self.precision or default_precision()
        if precision not in _PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
        # cheap change detector first (no state_dict() with its 701 prefixed names per call): storage address + version
        # counter of every parameter / buffer; load_state_dict, .to() and in-place updates all move one of them
        m = self.module
        sig = tuple((v.data_ptr(), v._version) for v in m.parameters()) + tuple((v.data_ptr(), v._version) for v in m.buffers())
        if sig != self._weight_sig:
            sd = m.state_dict()
            L = _lib.lib()
            for name, t in sd.items():
                if name.startswith("g_proj."):
                    continue                       # speaker projection: applied by the mirror before the call
                if not t.is_cuda:
                    raise RuntimeError("backend='hip' needs the module on a GPU (parameter %s is on %s)" % (name, t.device))
                t32 = t.detach().to(torch.float32)
                if name.endswith("conv.weight") and t32.dim() == 3:
                    if t32.shape[0] != 1:
                        raise ValueError("ConvTBC kernel_size %d is not used by PromptEncoder" % t32.shape[0])
                    t32 = t32[0].t()               # [1, C_in, C_out] -> [C_out, C_in] (include/dvits_hip.h)
                t32 = t32.contiguous()
                shape = (C.c_int64 * t32.dim())(*t32.shape)
                _lib.check(L.dv_penc_set_weight(self._h, name.encode(), _lib.ptr(t32), shape, t32.dim()),
                           "dv_penc_set_weight(%s)" % name)
            torch.cuda.synchronize()
            self._weight_sig = sig
            self._prepared = None
        if precision != self.precision:
            self.precision = precision
            self._prepared = None

    def forward(self, prompt, keep):
        """prompt [B, C_in, L] float32 GPU, keep [B, L] float32 (1 = valid frame) -> [B, L, C_out] float32."""
        if not prompt.is_cuda:
            raise RuntimeError("backend='hip' needs GPU tensors; got prompt on %s" % prompt.device)
        self.sync_weights()
        B, _, L = prompt.shape
        key = (B, L, self.precision)
        if key != self._prepared:
            _lib.check(_lib.lib().dv_penc_prepare(self._h, B, L, _PRECISIONS[self.precision]), "dv_penc_prepare")
            self._prepared = key
        x = prompt.detach().to(torch.float32).contiguous()
        k = keep.detach().to(device=prompt.device, dtype=torch.float32).contiguous()
        out = torch.empty((B, L, self.out_channels), device=prompt.device, dtype=torch.float32)
        _lib.check(_lib.lib().dv_penc_forward(self._h, _lib.ptr(x), _lib.ptr(k), _lib.ptr(out), _lib.stream_ptr()),
                   "dv_penc_forward")
        return out

    def stats(self):
        n, f = C.c_int64(), C.c_double()
        _lib.check(_lib.lib().dv_penc_stats(self._h, C.byref(n), C.byref(f)), "dv_penc_stats")
        return n.value, f.value
