#!/usr/bin/env python
"""Timeline of one mrs_dec2_gemv launch (dec_core2.cuh MRS_TL2 stamps): medians over the waves, us from the earliest entry."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import mistralrs_amd  # noqa: F401
from mistralrs_amd import _lib
from mistralrs_amd.gguf import GgmlDType
from mistralrs_amd.llama import random_qtensor


class Mat(C.Structure):
    _fields_ = [("planes", C.c_void_p), ("type", C.c_int), ("n", C.c_longlong), ("k", C.c_longlong)]


ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="gate_up,down4,o,qkv")
a = ap.parse_args()
dev = torch.device("cuda:0")
L = _lib.load("ext"); _lib.load("quant")
L.mrs_dec2_repack_bytes.restype = C.c_size_t
L.mrs_dec2_repack_bytes.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
L.mrs_dec2_repack.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
L.mrs_dec2_gemv.argtypes = [C.POINTER(Mat), C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.mrs_dec2_timeline.argtypes = [C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
Q4, Q6 = GgmlDType.Q4K, GgmlDType.Q6K
shapes = {"gate_up": (Q4, 28672, 4096, True), "down4": (Q4, 4096, 14336, False), "down6": (Q6, 4096, 14336, False), "o": (Q4, 4096, 4096, False), "qkv": (Q4, 6144, 4096, True)}
for name in a.cases.split(","):
    dt, n, k, norm = shapes[name]
    ws = []
    for i in range(6):  # several tensors so that the timed one is cold
        w = random_qtensor(dt, n, k, dev, 900 + i)
        p = torch.empty(L.mrs_dec2_repack_bytes(dt.id, n, k), dtype=torch.uint8, device=dev)
        assert L.mrs_dec2_repack(w.data.data_ptr(), dt.id, n, k, p.data_ptr(), st) == 0
        torch.cuda.synchronize(); del w
        ws.append((p, Mat(p.data_ptr(), dt.id, n, k)))
    x = torch.randn(1, k, device=dev); nw = torch.ones(k, device=dev); out = torch.empty(1, n, device=dev)
    tl = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
    filler = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev)
    for i, (p, m) in enumerate(ws):
        L.mrs_dec2_timeline(tl.data_ptr() if i == len(ws) - 1 else None)
        if i == len(ws) - 1:
            filler.fill_(1); torch.cuda.synchronize()
        assert L.mrs_dec2_gemv(C.byref(m), x.data_ptr(), k, nw.data_ptr() if norm else None, 1e-5, out.data_ptr(), n, 1, st) == 0
        torch.cuda.synchronize()
    L.mrs_dec2_timeline(None)
    t = tl.cpu().numpy().reshape(256, 8, 16).astype(np.float64)
    used = t[:, :, 0] > 0
    t0 = t[:, :, 0][used].min()
    rel = (t - t0) / 100.0  # 100 MHz -> us
    def med(sel, i):
        v = rel[:, sel, i][t[:, sel, i] > 0]
        return (round(float(np.median(v)), 2), round(float(v.max()), 2)) if v.size else None
    print(name, "waves 0-3 (median, max us): entry", med(slice(0, 4), 0), "issued", med(slice(0, 4), 1), "prologue", med(slice(0, 4), 2), "barrier", med(slice(0, 4), 3),
          "rec1", med(slice(0, 4), 4), "rec2", med(slice(0, 4), 5), "rec4", med(slice(0, 4), 7), "end", med(slice(0, 4), 14))
    print(name, "waves 4-7 (median, max us): entry", med(slice(4, 8), 0), "issued", med(slice(4, 8), 1), "prologue", med(slice(4, 8), 2), "barrier", med(slice(4, 8), 3),
          "rec1", med(slice(4, 8), 4), "rec2", med(slice(4, 8), 5), "rec4", med(slice(4, 8), 7), "end", med(slice(4, 8), 14))
    nrec = ((t[:, :, 4:14] > 0).sum(axis=2))
    print(name, "records per wave: waves 0-3 mean", round(float(nrec[:, :4].mean()), 2), "waves 4-7 mean", round(float(nrec[:, 4:].mean()), 2), "(capped at 10)")
    del ws, filler
    torch.cuda.empty_cache()
