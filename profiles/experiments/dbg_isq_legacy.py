"""Where do the Q4_0 / Q5_0 device ISQ blocks differ from the oracle?  (GPU box debugging aid.)"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from tests.abi_backends import GpuBackend
from tests.test_zz_isq_kquants import _weights
O.build()
be = GpuBackend(torch.device("cuda:0"))
for tname, ts in (("Q4_0", 18), ("Q5_0", 22)):
    t = getattr(O, tname)
    w = _weights("f32", seed=t)
    n, k = w.shape
    want = O.quantize(t, w)
    wb = be.buf(w)
    out = be.buf(np.full(want.shape, 0xAA, dtype=np.uint8))
    fn = be.sym("mrs_isq_quantize", [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p], C.c_int)
    assert fn(wb.ptr, 0, out.ptr, n * k, t, be.stream) == 0
    got = out.numpy().reshape(n, -1)
    want = want.reshape(n, -1)
    bad = np.argwhere(got != want)
    print(tname, "mismatches", len(bad))
    seen = set()
    for r, c in bad[:400]:
        b = c // ts
        if (r, b) in seen: continue
        seen.add((r, b))
        if len(seen) > 6: break
        print(" row", r, "block", b, "byte-in-block", c % ts, "got", got[r, b*ts:(b+1)*ts].tolist(), "want", want[r, b*ts:(b+1)*ts].tolist(), "x", w[r, b*32:b*32+4])
