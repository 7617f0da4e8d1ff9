"""not gpu: the C-ABI library loads and exports every symbol include/r3m_hip.h declares, with the argument counts the ctypes
binding (r3m_amd/_lib.py) uses. No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "r3m_hip.h")


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(r3m_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[name] = n
    return out


def test_library_is_built_and_loads():
    from r3m_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    h = _lib.lib()
    assert h.r3m_abi_version() == 1
    assert h.r3m_last_error() is not None


def test_every_declared_symbol_is_exported_with_matching_arity():
    from r3m_amd import _lib
    decl = _header_functions()
    assert len(decl) >= 35
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(raw, name), f"{name} declared in include/r3m_hip.h but not exported by libr3m_hip.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in r3m_amd/_lib.py"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args, binding {len(_lib.SIGNATURES[name][1])}"
    for name in _lib.SIGNATURES:
        assert name in decl, f"{name} bound in _lib.py but not declared in the header"


def test_plan_queries_need_no_gpu():
    """Plan construction is host-only: sizes of the three encoders match the reference."""
    from r3m_amd import _lib
    L = _lib.lib()
    expect = {18: (11176512, 512, 20), 34: (21284672, 512, 36), 50: (23508032, 2048, 53)}
    for size, (nparam, dim, nconv) in expect.items():
        h = L.r3m_resnet_create(size, 4)
        assert h
        assert L.r3m_resnet_num_params(h) == nparam
        assert L.r3m_resnet_out_dim(h) == dim
        assert L.r3m_resnet_num_tensors(h) == nconv * 5
        assert L.r3m_resnet_arena_bytes(h) > 0
        tot = 0
        for st in range(4):
            off, cnt = ctypes.c_longlong(), ctypes.c_longlong()
            assert L.r3m_resnet_stage_range(h, st, ctypes.byref(off), ctypes.byref(cnt)) == 0
            tot += cnt.value
        assert tot == nparam
        L.r3m_resnet_destroy(h)
    assert not L.r3m_resnet_create(101, 4)
    assert b"unsupported" in L.r3m_last_error()
    assert not L.r3m_resnet_create(50, 0)                      # empty batch: refused at plan time, with a message
    assert b"F=0" in L.r3m_last_error()
    assert not L.r3m_resnet_create_dt(50, 4, 7)                # unknown activation dtype
    assert b"dtype" in L.r3m_last_error()


def test_bf16_plan_is_smaller_and_same_parameters():
    """bf16 plans (BASELINE configs[2], [4]) keep the fp32 parameter / gradient layout and roughly halve the activation arena."""
    from r3m_amd import _lib
    L = _lib.lib()
    for size in (18, 34, 50):
        h32, h16 = L.r3m_resnet_create_dt(size, 16, 0), L.r3m_resnet_create_dt(size, 16, 1)
        assert h32 and h16 and L.r3m_resnet_dtype(h32) == 0 and L.r3m_resnet_dtype(h16) == 1
        assert L.r3m_resnet_num_params(h32) == L.r3m_resnet_num_params(h16)
        assert L.r3m_resnet_num_tensors(h32) == L.r3m_resnet_num_tensors(h16)
        a32, a16 = L.r3m_resnet_arena_bytes(h32), L.r3m_resnet_arena_bytes(h16)
        assert 0.4 * a32 < a16 < 0.75 * a32, (size, a32, a16)
        L.r3m_resnet_destroy(h32)
        L.r3m_resnet_destroy(h16)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from r3m_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_shipped_library_reads_no_environment_switches():
    """SURVEY.md §8(b) / VERDICT r1 weak #11: experiment switches and the WRONG-result timing probes exist only in -DR3M_PROBES
    builds (tools/build_ab.sh). The shipped .so carries none of their names and does not import getenv."""
    import subprocess
    from r3m_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"R3M_GG_DEBUG", b"R3M_BF16_", b"R3M_WG", b"R3M_SIDE_STREAM", b"R3M_TRACE", b"R3M_BN_ITEMS", b"R3M_GG_GLDS"):
        assert name not in blob, name
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        assert "getenv" not in nm.stdout
