"""CPU: the product path never touches the oracle (or any CPU fallback).

  * no file under mistral.rs_amd/ (Python, C++, HIP) imports, includes, dlopens or names anything under oracle/;
  * the built product libraries do not link against, or reference the symbols of, the oracle libraries;
  * the loader refuses to work without the HIP libraries (no silent fallback).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mistral.rs_amd")


def _sources():
    for root, dirs, files in os.walk(PKG):
        dirs[:] = [d for d in dirs if d not in ("build", "lib", "__pycache__")]
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".cpp", ".h", ".hpp")):
                yield os.path.join(root, f)


def test_package_sources_never_reference_the_oracle():
    pat = re.compile(r"(import\s+oracle|from\s+oracle|oracle/|libggml_oracle|libref_|orc_[a-z_0-9]+\s*\()")
    bad = []
    for p in _sources():
        for i, line in enumerate(open(p, errors="replace"), 1):
            code = line.split("#")[0] if p.endswith(".py") else line.split("//")[0]
            if pat.search(code):
                bad.append(f"{os.path.relpath(p, ROOT)}:{i}: {line.strip()}")
    assert not bad, "product code references the oracle:\n" + "\n".join(bad)


def test_product_libraries_do_not_link_the_oracle():
    libdir = os.path.join(PKG, "lib")
    if not os.path.isdir(libdir):
        pytest.skip("libraries not built")
    for f in os.listdir(libdir):
        if f.endswith(".so"):
            dyn = subprocess.check_output(["readelf", "-d", os.path.join(libdir, f)], text=True)
            assert "oracle" not in dyn and "libref_" not in dyn, f
            syms = subprocess.check_output(["nm", "-D", os.path.join(libdir, f)], text=True)
            assert " orc_" not in syms and " ref_mmvq" not in syms, f


def test_loader_fails_loudly_without_the_hip_library(tmp_path, monkeypatch):
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd import _lib
    monkeypatch.setattr(_lib, "LIB_DIR", str(tmp_path))
    monkeypatch.setattr(_lib, "_cache", {})
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load("quant")
