"""CPU: the C-ABI library loads and exports exactly what include/pfa_hip.h declares (no compute calls)."""
import os
import re

from pfa_gan import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pfa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(pfa_[a-z0-9_]+)\s*\(", text))


def test_header_and_binding_agree():
    assert declared_functions() == set(_lib.PROTOTYPES), "include/pfa_hip.h and pfa_gan/_lib.py list different entry points"


def test_library_exports_every_symbol():
    lib = _lib.lib()                      # raises if the .so is missing or a symbol cannot be bound
    for name in declared_functions():
        assert getattr(lib, name) is not None
    assert lib.pfa_version() >= 100
    assert lib.pfa_last_error_string() is not None
    assert lib.pfa_launch_count() == 0 or lib.pfa_launch_count() > 0


def test_argument_validation_needs_no_gpu():
    a = _lib.GatherArgs()                 # struct_size 0 -> rejected before any HIP call
    assert _lib.lib().pfa_gather_gemm(a, None) < 0
    assert b"struct_size" in _lib.lib().pfa_last_error_string()
    w = _lib.WgradArgs()
    assert _lib.lib().pfa_wgrad(w, None) < 0
    # the exchange step's entry points reject bad arguments before RCCL is even looked for
    import ctypes
    out = ctypes.c_void_p()
    assert _lib.lib().pfa_comm_create(None, 128, 2, 0, ctypes.byref(out)) < 0
    assert _lib.lib().pfa_comm_create(b"\0" * 128, 128, 2, 2, ctypes.byref(out)) < 0      # rank outside [0, nranks)
    assert _lib.lib().pfa_comm_unique_id(None, 128) < 0
    assert _lib.lib().pfa_allreduce(None, None, 16, _lib.F32, None) < 0
    assert _lib.lib().pfa_comm_size(None) == 0 and _lib.lib().pfa_comm_rank(None) == -1
    assert _lib.lib().pfa_comm_destroy(None) == 0


def test_struct_layout_matches_the_header():
    import ctypes
    # 18 int32/float fields then 9 pointers; 16 x 4 + 2 x 4 = 72 bytes of scalars, 8-byte aligned pointers after
    assert ctypes.sizeof(_lib.GatherArgs) == 72 + 9 * 8 + 16 + 6 * 8
    assert ctypes.sizeof(_lib.WgradArgs) == 18 * 4 + 6 * 8
    assert ctypes.sizeof(_lib.ScatterArgs) == 12 * 4 + 7 * 8 + 2 * 4   # pfa_scatter_args


def test_asm_read_kernels_use_no_scratch():
    """The patch kernels read LDS through inline asm into registers they name: a spilled register that is the target of an
    in-flight asm read is saved before the data lands (silent corruption).  Every build must keep them
    out of scratch: read from the code objects' notes (no GPU needed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.kernel_resources()
    assert len(res) > 50, "kernel notes not found in libpfa_hip.so"
    asm_families = ("k_conv_patch2", "k_tconv4", "k_conv_s1_inplace", "k_conv_s1_patch32", "k_wgrad_patch<", "k_wgrad_image<",
                    "k_conv_image", "k_gemm_small")
    checked = 0
    for name, r in res.items():
        if any(f in name for f in asm_families):
            checked += 1
            assert r.get("private_segment_fixed_size", 0) == 0 and r.get("vgpr_spill_count", 0) == 0, (name, r)
    assert checked >= 10
