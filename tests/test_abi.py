"""CPU: libmla_hip.so builds for gfx950, loads, and exports every symbol include/mla_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mla_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mla_[a-z0-9_]+)\s*\(", txt)) - {"mla_stream_t"})


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from mla_amd import hip
    lib = ctypes.CDLL(hip._LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 45
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in include/mla_hip.h but not exported: {missing}"
    # the ctypes binding and the header agree on the symbol set
    assert set(hip.exported_symbols()) == set(syms), set(hip.exported_symbols()) ^ set(syms)


def test_query_and_error_plumbing():
    from mla_amd import hip
    lib = hip.lib()
    assert lib.mla_query(0) == 1 and lib.mla_query(1) == 950 and lib.mla_query(2) == 64 and lib.mla_query(99) == -1
    # argument validation happens on the host before any launch -> safe without a GPU
    rc = lib.mla_gemm_bf16(None, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, 0, None)
    assert rc < 0 and b"null operand" in lib.mla_last_error()
    rc = lib.mla_attn_fwd(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None,
                          1, 8, 1, 64, 192, 64, 0.125, None)
    assert rc < 0 and b"head_dim must be 128" in lib.mla_last_error()


def test_no_cpu_fallback():
    """The product path fails loudly without a GPU instead of silently computing on the CPU."""
    import pytest
    import torch
    from mla_amd import ops
    with pytest.raises((RuntimeError, TypeError)):
        ops.linear(torch.zeros(4, 8, dtype=torch.bfloat16), (torch.zeros(8, 8, dtype=torch.bfloat16),))
    with pytest.raises((RuntimeError, TypeError)):
        ops.rmsnorm(torch.zeros(4, 8), torch.ones(8), 1e-5)
