"""ctypes binding of libpfa_hip.so (include/pfa_hip.h).

The library is mandatory for anything that touches a GPU tensor: `lib()` raises if it is missing
or fails to load - there is no eager-PyTorch or CPU fallback for the HIP path.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "lib", "libpfa_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
CONV_S2, CONV_S1, TCONV_S2, CONV_S1_DGRAD = 0, 1, 2, 3
FUSE_DEFER = 4


class GatherArgs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("mode", c_int32), ("dtype", c_int32),
        ("nb", c_int32), ("hq", c_int32), ("wq", c_int32), ("hs", c_int32), ("ws", c_int32),
        ("c1", c_int32), ("c2", c_int32), ("n", c_int32), ("n1", c_int32), ("ld1", c_int32), ("ld2", c_int32),
        ("act", c_int32), ("slope", c_float), ("splitk", c_int32), ("ntiles_stat", c_int32),
        ("src1", c_void_p), ("src2", c_void_p), ("w", c_void_p), ("bias", c_void_p),
        ("out1", c_void_p), ("out2", c_void_p), ("psum", c_void_p), ("psq", c_void_p), ("slab", c_void_p),
        ("fuse_out", c_int32), ("fuse_act1", c_int32), ("fuse_act2", c_int32), ("c_real", c_int32),
        ("fuse_raw", c_void_p), ("fuse_g2", c_void_p), ("fuse_scale", c_void_p), ("fuse_shift", c_void_p),
        ("fuse_mean", c_void_p), ("fuse_invstd", c_void_p),
    ]


class WgradArgs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("dtype", c_int32), ("stride", c_int32),
        ("nb", c_int32), ("hq", c_int32), ("wq", c_int32), ("hg", c_int32), ("wg", c_int32),
        ("d1p", c_int32), ("d1r", c_int32), ("d2p", c_int32), ("d2r", c_int32),
        ("g1p", c_int32), ("g1r", c_int32), ("g2p", c_int32), ("g2r", c_int32),
        ("msplit", c_int32), ("fresh", c_int32),
        ("dir1", c_void_p), ("dir2", c_void_p), ("gat1", c_void_p), ("gat2", c_void_p), ("dw", c_void_p),
        ("slab", c_void_p),
    ]


class ScatterArgs(Structure):
    _fields_ = [
        ("struct_size", c_int32), ("dtype", c_int32), ("nb", c_int32), ("hs", c_int32), ("ws", c_int32),
        ("c1", c_int32), ("c2", c_int32), ("n", c_int32), ("ld", c_int32), ("act", c_int32), ("slope", c_float),
        ("relu_src1", c_int32),
        ("src1", c_void_p), ("src2", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p),
        ("l1_fake", c_void_p), ("l1_real", c_void_p), ("l1_lambda", c_float), ("l1_creal", c_int32),
    ]


class ShadowDesc(Structure):
    _fields_ = [("master", c_void_p), ("same", c_void_p), ("tr", c_void_p),
                ("rows", c_int32), ("c1r", c_int32), ("c1p", c_int32), ("c2r", c_int32), ("c2p", c_int32),
                ("rows_p", c_int32), ("same_blk0", c_int32), ("tr_blk0", c_int32)]


# name -> (restype, argtypes); every symbol include/pfa_hip.h declares
PROTOTYPES = {
    "pfa_version": (c_int, []),
    "pfa_last_error_string": (c_char_p, []),
    "pfa_launch_count": (c_uint64, []),
    "pfa_set_knob": (c_int, [c_int, c_int]),
    "pfa_set_debug_buffer": (c_int, [c_void_p, c_size_t]),
    "pfa_last_gather_kernel": (c_char_p, []),
    "pfa_gather_timing": (c_int, [c_int]),
    "pfa_gather_timing_count": (c_int, []),
    "pfa_gather_timing_get": (c_int, [c_int, POINTER(c_float), POINTER(c_char_p), POINTER(c_double), POINTER(c_double),
                                      POINTER(c_int)]),
    "pfa_gather_timing_start": (c_int, [c_int, c_int, POINTER(c_float)]),
    "pfa_gather_timing_grid": (c_int, [c_int, POINTER(c_int)]),
    "pfa_gather_stat_tiles": (c_int, [POINTER(GatherArgs)]),
    "pfa_gather_slab_floats": (c_size_t, [POINTER(GatherArgs)]),
    "pfa_gather_gemm": (c_int, [POINTER(GatherArgs), c_void_p]),
    "pfa_splitk_finish": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p,
                                  c_int, c_void_p, c_int, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "pfa_wgrad": (c_int, [POINTER(WgradArgs), c_void_p]),
    "pfa_tconv_scatter_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pfa_tconv_scatter": (c_int, [POINTER(ScatterArgs), c_void_p]),
    "pfa_wgrad_slab_floats": (c_size_t, [POINTER(WgradArgs)]),
    "pfa_wgrad_slices": (c_int, [POINTER(WgradArgs)]),
    "pfa_logit_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "pfa_logit_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pfa_logit_fwd_nparts": (c_int, [c_int, c_int, c_int]),
    "pfa_logit_fwd_bce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                  c_float, c_void_p, c_void_p, c_void_p]),
    "pfa_logit_dgrad_tiles": (c_int, [c_int, c_int, c_int]),
    "pfa_logit_dgrad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pfa_logit_wgrad_slab_floats": (c_size_t, [c_int, c_int]),
    "pfa_logit_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pfa_bn_finalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_float, c_float,
                                c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_eval_coeffs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_act": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_stats_fused_supported": (c_int, [c_int, c_int, c_int]),
    "pfa_bn_act_stats": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                 c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_float, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_bwd_apply_stats": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_double, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pfa_bn_bwd_reduce": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "pfa_bn_bwd_tiles": (c_int, [c_int64, c_int, c_int]),
    "pfa_bn_bwd_finalize": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "pfa_bn_bwd_apply": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_bce_logits": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "pfa_l1_tanh_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                c_void_p, c_void_p, c_void_p]),
    "pfa_bce_logits_nparts": (c_int, [c_int64]),
    "pfa_bce_logits_parts": (c_int, [c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "pfa_l1_tanh_bwd_nparts": (c_int, [c_int64, c_int, c_int]),
    "pfa_l1_tanh_bwd_parts": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                      c_void_p, c_void_p, c_void_p]),
    "pfa_loss_finish": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pfa_colsum_ws": (c_size_t, [c_int64, c_int]),
    "pfa_colsum": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int,
                         c_void_p, c_float, c_void_p]),
    "pfa_adam_tick": (c_int, [c_void_p, c_void_p]),
    "pfa_weight_shadow": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_shadow_blocks": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "pfa_weight_shadow_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pfa_adam_shadow_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_float, c_float, c_float, c_float, c_int, c_void_p, c_float, c_void_p]),
    "pfa_pack_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pfa_unpack_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pfa_unpack_u8": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    "pfa_fill_zero": (c_int, [c_void_p, c_size_t, c_void_p]),
    "pfa_axpy": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "pfa_finish_bn_supported": (c_int, [c_int, c_int, c_int]),
    "pfa_finish_bn_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p,
                                  c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "pfa_finish_bn_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p]),
    "pfa_gather_plan_splitk": (c_int, [POINTER(GatherArgs)]),
    "pfa_finish_coop_supported": (c_int, [c_int, c_int, c_int]),
    "pfa_finish_coop_part_floats": (c_size_t, [c_int, c_int]),
    "pfa_finish_coop_bar_words": (c_size_t, [c_int]),
    "pfa_finish_coop_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "pfa_finish_coop_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                    c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "pfa_comm_unique_id": (c_int, [c_void_p, c_int]),
    "pfa_comm_create": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "pfa_comm_destroy": (c_int, [c_void_p]),
    "pfa_comm_size": (c_int, [c_void_p]),
    "pfa_comm_rank": (c_int, [c_void_p]),
    "pfa_allreduce": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "pfa_replay_index": (c_int, [c_char_p]),
    "pfa_replay": (c_int, [c_void_p, c_int, POINTER(c_int)]),
}


class ReplayOp(Structure):
    _fields_ = [("kind", c_int32), ("fn", c_int32), ("a", c_void_p), ("b", c_void_p)]


OP_CALL, OP_RECORD, OP_WAIT = 0, 1, 2

_LIB = None


class PfaError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if libpfa_hip.so has not been built (run __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("PFA_LIB") or LIB_PATH      # PFA_LIB: another BUILD of this library (same-box A/B of two builds)
        if not os.path.exists(path):
            raise PfaError(f"{path} is missing: the HIP extension is mandatory, build it with "
                           f"`make -C {PKG_DIR}` (or __graft_entry__.build()); there is no fallback path")
        handle = ctypes.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
        for item in filter(None, os.environ.get("PFA_KNOBS", "").split(",")):   # e.g. PFA_KNOBS=0=5 (A/B runs)
            k, v = item.split("=")
            handle.pfa_set_knob(int(k), int(v))
    if _RECORDER is not None:
        return _RECORDER.proxy
    return _LIB


# ------------------------------------------------------------------------------------ launch lists (pfa_replay)
_RECORDER = None


class LaunchList:
    """A recorded sequence of library calls, event records and stream waits; replay() re-issues it with one C call."""

    def __init__(self, ops, keep, patch=()):
        self.n = len(ops)
        self.ops = (ReplayOp * max(self.n, 1))(*ops)
        self._keep = keep               # argument blocks, struct copies, events, streams: alive as long as the list
        self.calls = sum(1 for o in ops if o.kind == OP_CALL)
        # pointer substitution: every pointer-typed field of the recorded argument blocks that held one of the `patch`
        # addresses (the step's input tensors) when the list was recorded; replay(subst=) rewrites them, so ONE list serves
        # any buffer of the same shape (no copy into static inputs in front of each replay)
        self._sites = {int(p): [] for p in patch}
        self._current = {}              # patch address -> the address its sites hold now (absent: as recorded)
        if self._sites:
            for obj in keep:
                if isinstance(obj, Structure):
                    for field in obj._fields_:
                        if field[1] is c_void_p and getattr(obj, field[0]) in self._sites:
                            self._sites[getattr(obj, field[0])].append((obj, field[0]))

    def sites(self, ptr):
        """Number of recorded pointer arguments that held `ptr` (a `patch` address of Recorder.finish)."""
        return len(self._sites.get(int(ptr), ()))

    def replay(self, subst=None):
        """subst: {recorded address: address this replay uses} for addresses given to Recorder.finish(patch=...).  Every
        patch address that subst does not name goes back to the address it was recorded with: a replay never inherits the
        buffers of an earlier one (a batch that had to be copied into the static inputs behind one that was substituted)."""
        subst = subst or {}
        for orig, sites in self._sites.items():
            new = int(subst.get(orig, orig))
            if self._current.get(orig, orig) != new:
                for obj, name in sites:
                    setattr(obj, name, new)
                self._current[orig] = new
        failed = c_int(-1)
        rc = _LIB.pfa_replay(ctypes.addressof(self.ops), self.n, ctypes.byref(failed))
        if rc != 0:
            msg = _LIB.pfa_last_error_string().decode(errors="replace")
            raise PfaError(f"pfa_replay failed at op {failed.value} (rc={rc}): {msg}")


class _Proxy:
    """Stands in for the library while a Recorder is active: every launching entry point is called AND noted."""

    def __init__(self, rec):
        self._rec = rec

    def __getattr__(self, name):
        fn = getattr(_LIB, name)
        idx = self._rec.index(name)
        if idx < 0:
            return fn
        rec = self._rec

        def call(*args):
            # by-reference argument blocks are copied BEFORE the call: entry points write results into them (pfa_gather_gemm:
            # the planes actually written, the number of statistics tiles), and a replay must pass the REQUEST again, not
            # the previous call's answers
            noted = rec.snapshot(name, args)
            rc = fn(*args)
            if rc == 0:
                rec.note_call(name, idx, noted)
            return rc
        return call


class Recorder:
    """with Recorder() as r: ...launches...; r.finish() -> LaunchList.  The launches run normally while they are recorded.
    Streams are taken from the calls themselves; event records / stream waits between streams must go through
    kernels.record_event / wait_event / wait_stream, which note them here."""

    _structs = {}

    def __init__(self):
        self.ops, self.keep = [], []
        self._idx = {}
        self.proxy = _Proxy(self)

    def __enter__(self):
        global _RECORDER
        lib()
        if _RECORDER is not None:
            raise PfaError("a launch list is already being recorded")
        _RECORDER = self
        return self

    def __exit__(self, *exc):
        global _RECORDER
        _RECORDER = None
        return False

    def index(self, name):
        if name not in self._idx:
            self._idx[name] = _LIB.pfa_replay_index(name.encode()) if name in PROTOTYPES else -1
        return self._idx[name]

    @classmethod
    def _arg_struct(cls, name):
        st = cls._structs.get(name)
        if st is None:
            fields = []
            for i, t in enumerate(PROTOTYPES[name][1]):
                is_ptr = isinstance(t, type) and issubclass(t, ctypes._Pointer)
                fields.append((f"a{i}", c_void_p if is_ptr else t))
            st = type(f"Args_{name}", (Structure,), {"_fields_": fields})
            cls._structs[name] = st
        return st

    def snapshot(self, name, args):
        """The call's arguments as they will be replayed: by-reference blocks (ctypes.byref(x)) copied as they are NOW."""
        st = self._arg_struct(name)()
        keep = [st]
        for i, (t, v) in enumerate(zip(PROTOTYPES[name][1], args)):
            if isinstance(t, type) and issubclass(t, ctypes._Pointer):
                if v is not None:
                    obj = getattr(v, "_obj", None)          # ctypes.byref(x)
                    if obj is None:
                        raise PfaError(f"{name}: argument {i} cannot be recorded (pass ctypes.byref of a ctypes object)")
                    copy = type(obj).from_buffer_copy(obj)
                    keep.append(copy)
                    v = ctypes.addressof(copy)
            setattr(st, f"a{i}", v)
        return st, keep

    def note_call(self, name, idx, noted):
        st, keep = noted
        self.keep += keep
        self.ops.append(ReplayOp(OP_CALL, idx, ctypes.addressof(st), None))

    def note_record(self, event, stream):
        self.keep += [event, stream]
        self.ops.append(ReplayOp(OP_RECORD, 0, event.cuda_event, stream.cuda_stream))

    def note_wait(self, stream, event):
        self.keep += [event, stream]
        self.ops.append(ReplayOp(OP_WAIT, 0, event.cuda_event, stream.cuda_stream))

    def finish(self, patch=()):
        """patch: device addresses (the step's input tensors) that replay(subst=) may replace."""
        return LaunchList(self.ops, self.keep, patch)


def recorder():
    return _RECORDER


def check(rc, what=""):
    if rc != 0:
        msg = lib().pfa_last_error_string().decode(errors="replace")
        raise PfaError(f"{what or 'libpfa_hip'} failed (rc={rc}): {msg}")


def launch_count():
    return int(lib().pfa_launch_count())
