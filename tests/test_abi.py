"""The C-ABI library loads here (no GPU) and exports every symbol include/lisec_hip.h declares."""
import os
import re

from conftest import ROOT


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "lisec_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(lisec_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from lisec_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 5
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.lisec_abi_version() >= 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from lisec_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.LisecError):
        _lib.load()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under lisec_amd/ may import it."""
    pkg = os.path.join(ROOT, "lisec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
