"""CPU: every symbol declared in include/*.h is exported by the in-tree C-ABI library built by
hipcc for gfx950 (no compute calls -- this runs without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER_LIB = {
    "mistralrs_quant.h": "libmistralrsquant.so",
    "mistralrs_paged_attn.h": "libmistralrspagedattention.so",
    "mistralrs_core.h": "libmistralrscuda.so",
    "mrs_hip_ext.h": "libmrs_hip_ext.so",
}


def declared_symbols(header: str):
    src = subprocess.check_output(["gcc", "-E", "-P", "-x", "c", os.path.join(ROOT, "include", header)], text=True)
    src = re.sub(r"typedef[^;]*;", "", src)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", re.sub(r"\([^()]*\)\s*\(", "(", src)))
    return sorted(n for n in names if not n.startswith("__"))  # __attribute__/__aligned__ come from system headers


@pytest.fixture(scope="module")
def built():
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd import build as _  # noqa: F401
    import importlib.util as u
    spec = u.spec_from_file_location("mrs_build", os.path.join(ROOT, "mistral.rs_amd", "build.py"))
    m = u.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.build(verbose=False)


@pytest.mark.parametrize("header", sorted(HEADER_LIB))
def test_header_symbols_exported(built, header):
    path = os.path.join(ROOT, "include", header)
    if not os.path.exists(path):
        pytest.skip(f"{header} not written yet")
    lib = built.get(HEADER_LIB[header])
    assert lib and os.path.exists(lib), f"{HEADER_LIB[header]} was not built"
    syms = declared_symbols(header)
    assert len(syms) > 0
    nm = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    exported = {line.split()[-1] for line in nm.splitlines() if " T " in line}
    missing = [s for s in syms if s not in exported]
    assert not missing, f"{header}: {len(missing)} declared symbols not exported, e.g. {missing[:5]}"
    h = ctypes.CDLL(lib)  # loads without a GPU (no compute calls)
    assert all(hasattr(h, s) for s in syms[:10])


def test_quant_header_counts():
    syms = declared_symbols("mistralrs_quant.h")
    assert sum(s.startswith("launch_mmvq_gguf_") for s in syms) >= 93
    # the MoE launchers exist for the 10 MMVQ formats and for Q8_1 weights (gguf/ffi.rs:268,424,601,800)
    for t in ("q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q8_1", "q2k", "q3k", "q4k", "q5k", "q6k"):
        for name in (f"launch_indexed_moe_forward_{t}_q8_1", f"launch_moe_grouped_gemm_{t}", f"launch_moe_gemv_fused_gate_up_{t}_q8_1",
                     f"launch_moe_gemv_down_aggregate_{t}_q8_1"):
            assert name in syms, name
