"""CPU: the C-ABI library builds, loads and exports every symbol include/vrfhip.h declares;
without a GPU it refuses to create a context (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from ark_ec_vrfs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "vrfhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vrfhip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    decl = _declared_symbols()
    assert len(decl) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in vrfhip.h but not exported"
    assert sorted(_lib.SYMBOLS) == decl


def test_every_entry_point_has_a_declared_signature():
    """A call through ctypes without argtypes passes a size_t as a 32-bit int: every symbol that takes arguments has them
    declared, with the arity of its C declaration."""
    import re
    hdr = open(os.path.join(ROOT, "include", "vrfhip.h")).read()
    lib = _lib.load()
    for name in _lib.SYMBOLS:
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, hdr, re.S)
        assert m, name
        params = m.group(1).strip()
        arity = 0 if params == "void" else params.count(",") + 1
        at = getattr(lib, name).argtypes
        assert (at is None and arity == 0) or (at is not None and len(at) == arity), (name, arity, at)


def test_abi_version_and_no_cpu_fallback():
    lib = _lib.load()
    assert lib.vrfhip_abi_version() == _lib.ABI_VERSION == 144
    import torch
    if not torch.cuda.is_available():
        from ark_ec_vrfs_amd import Context, VrfHipError
        with pytest.raises(VrfHipError):
            Context(0)


def test_product_does_not_import_the_oracle():
    """The product package must never route through oracle/ (or the host simulation)."""
    pkg = os.path.join(ROOT, "ark_ec_vrfs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")) and not f.endswith(".gen.h"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt and "hostsim" not in txt.replace("tests/hostsim", ""), f
