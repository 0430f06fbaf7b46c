"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol declared in
include/jckgan.h; the ctypes table covers the header (no compute calls here - no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "jckgan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jck_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from hipgan import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    dll = _lib.load_library()
    syms = _header_symbols()
    assert len(syms) > 30
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in include/jckgan.h but not exported"
        assert s in _lib.PROTOS, f"{s} has no ctypes prototype"
    # ... and nothing else: the unmangled jck_* symbols of the dynamic table are exactly the header's
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[2] for l in nm.splitlines() if len(l.split()) == 3 and l.split()[1] == "T" and l.split()[2].startswith("jck_")})
    assert exported == syms, (sorted(set(exported) - set(syms)), sorted(set(syms) - set(exported)))
    assert set(_lib.PROTOS) == set(syms), sorted(set(_lib.PROTOS) ^ set(syms))
    from hipgan import build as b
    assert dll.jck_version() == b.source_id() >= 100          # the binary was built from the sources beside it
    assert dll.jck_pad_rows(3) == 16 and dll.jck_pad_rows(64) == 64 and dll.jck_pad_rows(200) == 256
    assert dll.jck_pad_chan(3) == 4


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hipgan import JckError, lib
    with pytest.raises(JckError):
        lib.jck_adam(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 4, 1e-3, 0.5, 0.999, 1e-8, 1, 1.0, None)


def test_comm_entry_points_validate_their_arguments_on_the_host():
    """jck_comm_* (RCCL behind the C ABI): argument errors come back as JCK_E_ARG with a message before any device or RCCL call;
    the data-parallel binding keeps torch.distributed unless asked (no GPU, no communicator here)."""
    import ctypes as C
    from hipgan import _lib
    dll = _lib.load_library()
    out = C.c_void_p()
    ident = (C.c_ubyte * 128)()
    assert dll.jck_comm_create(C.byref(out), ident, 0, 0) == -1 and b"world" in dll.jck_last_error()       # JCK_E_ARG
    assert dll.jck_comm_create(C.byref(out), ident, 2, 2) == -1
    assert dll.jck_comm_create(None, ident, 1, 0) == -1
    assert dll.jck_comm_unique_id(None) == -1
    assert dll.jck_comm_wait(None, 0, None) == -1 and dll.jck_comm_wait(None, 99, None) == -1
    tk = C.c_int(-1)
    assert dll.jck_comm_allreduce_enqueue(None, None, 0, None, C.byref(tk)) == -1
    assert dll.jck_comm_world(None) == 0 and dll.jck_comm_destroy(None) == 0
    from hipgan.dist import GradReducer
    os.environ.pop("JCK_RCCL_NATIVE", None)
    r = GradReducer(1)
    assert r.native is False and r._comm is None and r.start(None) is None      # one rank, nothing to exchange
