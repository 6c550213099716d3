"""CPU-only: librvit.so loads and exports every symbol include/rvit.h declares (no compute calls),
and the ctypes signature table matches the header's parameter counts."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "rvit.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(rvit_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_header_declares_entry_points():
    d = _header_decls()
    assert {"rvit_version", "rvit_last_error", "rvit_gemm_bf16", "rvit_conv2d_fwd_bf16", "rvit_layernorm_fwd",
            "rvit_attn_fwd"} <= set(d)


def test_library_exports_every_declared_symbol():
    from vit_amd import _lib
    lib = _lib.load()
    for name in _header_decls():
        assert hasattr(lib, name), f"librvit.so lacks {name}"
    assert lib.rvit_version() == _lib.ABI_VERSION


def test_ctypes_table_matches_header():
    from vit_amd import _lib
    d = _header_decls()
    assert set(d) == set(_lib.SIGNATURES), (set(d) ^ set(_lib.SIGNATURES))
    for name, n in d.items():
        assert len(_lib.SIGNATURES[name][0]) == n, f"{name}: header has {n} params, ctypes table " \
                                                   f"{len(_lib.SIGNATURES[name][0])}"


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from vit_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(ImportError):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from vit_amd import ops
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(a, a, torch.zeros(8, 8))
