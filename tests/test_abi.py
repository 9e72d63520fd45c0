"""The C-ABI library loads without a GPU and exports every symbol that
include/cutfemx_amd.h declares; compute entry points fail loudly without a device."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "cutfemx_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cfx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cutfemx_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"{n} declared in cutfemx_amd.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_no_torch_types_in_abi():
    text = (ROOT / "include" / "cutfemx_amd.h").read_text()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # comments cite the reference's C++ types
    assert "torch" not in code and "at::" not in code and "std::" not in code and "#include <hip" not in code


def test_product_never_touches_the_oracle():
    for p in (ROOT / "cutfemx_amd").rglob("*"):
        if p.suffix in {".py", ".h", ".hip", ".cpp"}:
            txt = p.read_text()
            assert "pyoracle" not in txt and "liboracle" not in txt and "cfx_oracle" not in txt, p


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cutfemx_amd import _lib
    lib = _lib.load()
    assert lib.cfx_init(0) != 0
    assert b"no HIP device" in lib.cfx_last_error()
    import cutfemx_amd as cfx
    with pytest.raises(RuntimeError):
        cfx.Mesh.create_box(3, 2)


def test_frozen_level_set_names_follow_the_reference():
    # cpp/cutfemx/cut/cut.cpp:82-138, python/tests/test_cut_api.py:750-773
    from cutfemx_amd.cut import frozen_level_set_names
    assert frozen_level_set_names(["f"]) == ("phi",)
    assert frozen_level_set_names(["", "u", "f"]) == ("phi", "phi1", "phi2")
    assert frozen_level_set_names(["fluid", "f"]) == ("fluid", "phi1")
    assert frozen_level_set_names(["f", "phi"]) == ("phi1", "phi")          # the default steps aside
    assert frozen_level_set_names(["phi1", "f", "f"]) == ("phi1", "phi2", "phi3")
    with pytest.raises(ValueError, match="Duplicate level-set function name"):
        frozen_level_set_names(["fluid", "fluid"])
    with pytest.raises(ValueError, match="not a valid selector identifier"):
        frozen_level_set_names(["2phase"])
