"""ctypes loader for the in-tree HIP libraries.  Fails loudly: there is no CPU/eager fallback."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")
NAMES = {
    "quant": "libmistralrsquant.so",
    "paged_attn": "libmistralrspagedattention.so",
    "core": "libmistralrscuda.so",
    "ext": "libmrs_hip_ext.so",
}
_cache: dict[str, C.CDLL] = {}


class HipLibraryMissing(RuntimeError):
    pass


def path(key: str) -> str:
    # MRS_EXT_LIB: an alternative BUILD of libmrs_hip_ext.so for A/B kernel experiments on the GPU box (e.g. lib/libmrs_hip_ext_occ4.so built by
    # profiles/experiments/build_occ4.sh).  Still an in-tree HIP library of the same sources; a missing file fails loudly like the default.
    if key == "ext" and os.environ.get("MRS_EXT_LIB"):
        p = os.environ["MRS_EXT_LIB"]
        return p if os.path.isabs(p) else os.path.join(LIB_DIR, p)
    return os.path.join(LIB_DIR, NAMES[key])


def load(key: str) -> C.CDLL:
    """Load one of the C-ABI libraries (RTLD_GLOBAL so libmrs_hip_ext can resolve the others)."""
    if key not in _cache:
        p = path(key)
        if not os.path.exists(p):
            raise HipLibraryMissing(
                f"{p} not built: run `python mistral.rs_amd/build.py` (hipcc, gfx950). "
                "There is deliberately no fallback path.")
        _cache[key] = C.CDLL(p, mode=C.RTLD_GLOBAL)
    return _cache[key]


def sym(key: str, name: str, argtypes, restype=None):
    f = getattr(load(key), name)
    f.argtypes = argtypes
    f.restype = restype
    return f
