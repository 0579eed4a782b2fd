"""The drop-in boundary is a C ABI: libvisper_hip.so must load WITHOUT a GPU and export every entry point include/visper_hip.h declares
(no compute calls here), the ctypes binding must cover the same set, and argument-checking paths must fail with an error code and a
message instead of crashing."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "visper_hip.h")


DEBUG_HEADER = os.path.join(ROOT, "include", "visper_hip_debug.h")


def _declared(header=HEADER):
    src = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"^(?:const char\*|int|long)\s+(vp_\w+)\s*\(", src, flags=re.M)))


def test_library_loads_and_exports_every_declared_symbol():
    from visper_lm_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 55 and "vp_gemm_bf16" in names and "vp_comm_allreduce_async" in names and "vp_emb_loss_fwd" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.vp_version() >= 100


def test_ctypes_binding_covers_the_header():
    from visper_lm_amd import _lib
    declared, bound = set(_declared()), set(_lib.EXPORTS)
    assert bound <= declared, sorted(bound - declared)
    assert declared == bound, sorted(declared ^ bound)


def test_debug_entry_points_are_outside_the_product_abi():
    """vp_debug_* live in include/visper_hip_debug.h (built with -DVP_DEBUG), never in the product header; the binding keeps them apart too."""
    from visper_lm_amd import _lib
    assert not [n for n in _declared() if n.startswith("vp_debug")]
    dbg = set(_declared(DEBUG_HEADER))
    assert dbg == set(_lib._DEBUG_SIGS) and all(n.startswith("vp_debug_") for n in dbg)
    assert not (dbg & set(_lib.EXPORTS))
    lib = _lib.load()
    if not os.environ.get("VP_LIB_PATH"):
        assert not any(hasattr(lib, n) for n in dbg), "the product library must be the sealed build (make: no -DVP_DEBUG)"
        with _lib.debug_library() as dl:                                   # the -DVP_DEBUG build of the same sources carries all of them
            assert all(hasattr(dl, n) for n in dbg) and all(hasattr(dl, n) for n in _lib.EXPORTS)
        assert _lib.load() is lib                                          # ... and the product library is back afterwards


def test_bad_arguments_return_codes_not_crashes():
    from visper_lm_amd import _lib
    lib = _lib.load()
    assert lib.vp_emb_loss_workspace(8, 64, 884736) > 0 and lib.vp_emb_loss_workspace(0, 0, 0) == 0
    rc = lib.vp_emb_loss_fwd(0, 0, 0, 0, None, None, None, None, ctypes.c_float(0.3), None, None, None, None, None)
    assert rc == -2 and b"vp_emb_loss_fwd" in lib.vp_last_error_string()
    rc = lib.vp_comm_allreduce_async(None, None, 0, 0, None)
    assert rc == -1 and b"vp_comm_allreduce_async" in lib.vp_last_error_string()
    assert lib.vp_comm_unique_id_bytes() == 128
    assert lib.vp_gemm_sched_workspace_bytes() == 64 and lib.vp_emb_loss_counter_bytes() == 8 * 1025 * 4          # caller-owned counter blocks
    rc = lib.vp_gemm_bf16(0, 0, 0, None, 0, None, 0, None, 0, None, None, 0, 0, 0, 0, None, None)
    assert rc < 0
