"""No kernel may read past a weight tensor.  The weights are placed so that they END at a PROT_NONE guard page (host emulation of the product kernels,
in a subprocess: a stray read is a segfault).  Regression: waves of the last workgroup that own no rows used to issue their first prefetch at
first_row >= N -- megabytes past an lm_head-sized tensor; on the MI355X that was a page fault for a [128256, 8192] Q6_K head (70B shapes)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, mmap, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import oracle as O
O.build()
from tests.abi_backends import HostBackend
be = HostBackend()
libc = C.CDLL(None, use_errno=True)
libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
n, K, sym = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
t = O.Q6_K if sym == "norm_proj" else O.Q4_K
w = O.random_blocks(t, n, K, seed=1, d_scale=0.02).reshape(-1)
page = mmap.PAGESIZE
size = (w.size + page - 1) // page * page + page
mm = mmap.mmap(-1, size)
base = C.addressof(C.c_char.from_buffer(mm))
assert libc.mprotect(base + size - page, page, 0) == 0  # PROT_NONE guard page right behind the tensor
dst = base + size - page - w.size
C.memmove(dst, w.ctypes.data, w.size)
x = np.random.default_rng(0).standard_normal((1, K)).astype(np.float32)
nw = np.ones(K, np.float32)
xb, nb, ob = be.buf(x), be.buf(nw), be.buf(np.full((1, n), np.nan, np.float32))
if sym == "norm_proj":
    fn = be.sym("mrs_decode_norm_proj", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int)
    assert fn(dst, t, n, K, xb.ptr, nb.ptr, 1e-5, ob.ptr, n, 1, be.stream) == 0
    want = O.matmul_q8_1(t, w.reshape(n, -1), n, K, O.quantize_q8_1(O.rms_norm(x, nw, 1e-5)))
else:
    kp = O.pad512(K)
    y = be.buf(O.quantize_q8_1(x))
    fn = be.sym("launch_mmvq_gguf_q4_k_f32_plain", [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p])
    fn(dst, y.ptr, ob.ptr, K, n, kp // 32, n, 1, be.stream)
    want = O.matmul_q8_1(t, w.reshape(n, -1), n, K, O.quantize_q8_1(x))
got = ob.numpy()
assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
print("guard page intact")
'''


@pytest.mark.parametrize("n,k,sym,env", [(600, 512, "norm_proj", {"MRS_PROJ_WGS": "2"}), (129, 1024, "mmvq", {}), (2049, 256, "mmvq", {})])
def test_row_less_waves_do_not_read_past_the_tensor(n, k, sym, env):
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, str(n), str(k), sym], capture_output=True, text=True, timeout=600,
                       env={**os.environ, **env})
    assert r.returncode == 0 and "guard page intact" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
