#!/usr/bin/env python
"""Decode-engine phase microbenchmark (mrs_dec_*): us per launch and TB/s of GGUF weight bytes at Llama-3-8B shapes, weights rotated over
>= 1.2 GB so nothing is Infinity-Cache resident.  `--old` times the round-1 fused kernels (mrs_decode_*) on the same tensors."""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Mat(C.Structure):
    _fields_ = [("planes", C.c_void_p), ("type", C.c_int), ("n", C.c_longlong), ("k", C.c_longlong)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--old", action="store_true")
    ap.add_argument("--phases", default="qkv,o,gate_up,down4,down6,lm_head")
    ap.add_argument("--v2", action="store_true", help="time the round-4 core's plain launcher (mrs_dec2_gemv) on one tensor of each phase's bytes")
    ap.add_argument("--timeline", action="store_true", help="instead of timing: one cold launch per phase with the s_memrealtime stamps of dec_core2.cuh (MRS_TL2), medians per wave group")
    ap.add_argument("--hot", action="store_true", help="two rotating buffers per phase, replayed 8 x inside the graph: the weights stay in the 256 MiB Infinity Cache")
    ap.add_argument("--img", action="store_true", help="batched steps: the GEMV phases on an activation image built ONCE by mrs_dec_act_image (not timed here), as the runner does for b >= 2")
    ap.add_argument("--mm", action="store_true", help="with --img: the matrix-core route of batched steps (mrs_dec_mm_*, csrc/ext_dec_mm.hip) on the MFMA-order copy of the same tensors")
    a = ap.parse_args()
    if a.mm:
        a.img = True
    import torch
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd import _lib
    from mistralrs_amd.gguf import GgmlDType
    from mistralrs_amd.llama import random_qtensor
    dev = torch.device("cuda:0")
    L = _lib.load("ext")
    _lib.load("quant")
    L.mrs_dec_repack_bytes.restype = C.c_size_t
    L.mrs_dec_repack_bytes.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
    L.mrs_dec_repack.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    b = a.b
    L.mrs_gemm_qi_repack_bytes.restype = C.c_size_t
    L.mrs_gemm_qi_repack_bytes.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
    L.mrs_gemm_qi_repack.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
    L.mrs_dec_mm_proj.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
    L.mrs_dec_mm_gate_up.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mrs_dec_mm_qkv.argtypes = [C.c_void_p, C.c_int, C.c_int] * 3 + [C.c_int, C.c_void_p] + [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]

    def make(dt, n, k, seed):
        w = random_qtensor(dt, n, k, dev, seed)
        nb = L.mrs_dec_repack_bytes(dt.id, n, k)
        p = torch.empty(nb, dtype=torch.uint8, device=dev)
        assert L.mrs_dec_repack(w.data.data_ptr(), dt.id, n, k, p.data_ptr(), st) == 0
        qi = None
        if a.mm:
            qi = torch.empty(L.mrs_gemm_qi_repack_bytes(dt.id, n, k), dtype=torch.uint8, device=dev)
            assert L.mrs_gemm_qi_repack(w.data.data_ptr(), dt.id, n, k, qi.data_ptr(), st) == 0
            p = torch.empty(16, dtype=torch.uint8, device=dev)  # the decode-layout copy is not read: keep HBM for the rotation
        return w, p, Mat(p.data_ptr(), dt.id, n, k), qi

    d, ff, nq, nkv, hd = 4096, 14336, 4096, 1024, 128
    h = torch.randn(b, d, device=dev)
    nw = torch.ones(d, device=dev)
    act = torch.randn(b, ff, device=dev)
    attn = torch.randn(b, nq, device=dev)
    q_out = torch.empty(b, nq, device=dev)
    kc = torch.zeros(8, 8, hd // 8, 32, 8, dtype=torch.bfloat16, device=dev)
    vc = torch.zeros(8, 8, hd, 32, dtype=torch.bfloat16, device=dev)
    slots = torch.arange(b, dtype=torch.int64, device=dev)
    pos = torch.arange(b, dtype=torch.int32, device=dev)
    cos = torch.ones(64, hd // 2, device=dev); sin = torch.zeros(64, hd // 2, device=dev)
    logits = torch.empty(b, 128256, device=dev)
    ya = torch.zeros(b * (4096 // 32) * 36, dtype=torch.uint8, device=dev)
    yb = torch.zeros(b * (14336 // 32) * 36, dtype=torch.uint8, device=dev)
    Q4, Q6 = GgmlDType.Q4K, GgmlDType.Q6K
    MP = C.POINTER(Mat)
    L.mrs_dec_qkv.argtypes = [MP, MP, MP, C.c_void_p, C.c_int, C.c_void_p, C.c_float] + [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]
    L.mrs_dec_gate_up.argtypes = [MP, MP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mrs_dec_proj.argtypes = [MP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    L.mrs_decode_qkv.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p]
    L.mrs_decode_gate_up.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mrs_decode_proj.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mrs_decode_norm_proj.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]

    def nbytes(*ws):
        return sum(w.data.numel() for w in ws)

    phases = {}
    L.mrs_dec_act_image_bytes.restype = C.c_size_t
    L.mrs_dec_act_image_bytes.argtypes = [C.c_int, C.c_int]
    L.mrs_dec_act_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.mrs_dec_qkv_img.argtypes = [MP, MP, MP] + [C.c_void_p] * 8 + [C.c_int] * 7 + [C.c_void_p]
    L.mrs_dec_gate_up_img.argtypes = [MP, MP, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mrs_dec_proj_img.argtypes = [MP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]

    def image(x, ldx, norm, k, dt):
        im = torch.empty(L.mrs_dec_act_image_bytes(k, b), dtype=torch.uint8, device=dev)
        assert L.mrs_dec_act_image(x.data_ptr(), ldx, nw.data_ptr() if norm else None, 1e-5, k, dt.id, b, im.data_ptr(), st) == 0
        torch.cuda.synchronize()
        return im
    img_h = image(h, d, True, d, Q4) if a.img else None
    img_act = image(act, ff, False, ff, Q4) if a.img else None
    img_attn = image(attn, nq, False, nq, Q4) if a.img else None

    def ph_qkv(i):
        ws = (make(Q4, nq, d, 3 * i), make(Q4, nkv, d, 3 * i + 1), make(Q6, nkv, d, 3 * i + 2))
        new = lambda st: L.mrs_dec_qkv(C.byref(ws[0][2]), C.byref(ws[1][2]), C.byref(ws[2][2]), h.data_ptr(), d, nw.data_ptr(), 1e-5, q_out.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                    slots.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), hd, hd // 2, 8, 32, 1, b, st)
        if a.img:
            new = lambda st: L.mrs_dec_qkv_img(C.byref(ws[0][2]), C.byref(ws[1][2]), C.byref(ws[2][2]), img_h.data_ptr(), q_out.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                               slots.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), hd, hd // 2, 8, 32, 1, b, 0, st)
        if a.mm:
            new = lambda st: L.mrs_dec_mm_qkv(ws[0][3].data_ptr(), 12, nq, ws[1][3].data_ptr(), 12, nkv, ws[2][3].data_ptr(), 14, nkv, d, img_h.data_ptr(), q_out.data_ptr(), kc.data_ptr(),
                                              vc.data_ptr(), slots.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), hd, hd // 2, 8, 32, 1, b, st)
        old = lambda st: L.mrs_decode_qkv(ws[0][0].data.data_ptr(), ws[1][0].data.data_ptr(), ws[2][0].data.data_ptr(), 12, 12, 14, nq, nkv, nkv, d, h.data_ptr(), nw.data_ptr(), 1e-5,
                                       q_out.data_ptr(), kc.data_ptr(), vc.data_ptr(), slots.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), hd, hd // 2, 8, 32, b, st)
        return ws, new, old, nbytes(*[w[0] for w in ws])

    def ph_proj(dt, n, k, x, ldx, ybuf, stride):
        def f(i):
            ws = (make(dt, n, k, 100 + i),)
            new = lambda st: L.mrs_dec_proj(C.byref(ws[0][2]), n, None, x.data_ptr(), ldx, None, 0.0, h.data_ptr(), d, 1, 1.0, None, b, st)
            if a.img:
                im = img_act if k == ff else img_attn
                new = lambda st: L.mrs_dec_proj_img(C.byref(ws[0][2]), n, im.data_ptr(), h.data_ptr(), d, 1, 1.0, b, st)
                if a.mm:
                    new = lambda st: L.mrs_dec_mm_proj(ws[0][3].data_ptr(), dt.id, n, k, im.data_ptr(), h.data_ptr(), d, 1, 1.0, b, st)
            old = lambda st: L.mrs_decode_proj(ws[0][0].data.data_ptr(), dt.id, n, k, ybuf.data_ptr(), stride, h.data_ptr(), d, 1, b, st)
            return ws, new, old, nbytes(ws[0][0])
        return f

    def ph_gate_up(i):
        ws = (make(Q4, ff, d, 200 + 2 * i), make(Q4, ff, d, 201 + 2 * i))
        new = lambda st: L.mrs_dec_gate_up(C.byref(ws[0][2]), C.byref(ws[1][2]), ff, None, h.data_ptr(), d, nw.data_ptr(), 1e-5, 0, act.data_ptr(), ff, b, st)
        if a.img:
            new = lambda st: L.mrs_dec_gate_up_img(C.byref(ws[0][2]), C.byref(ws[1][2]), ff, img_h.data_ptr(), 0, act.data_ptr(), ff, b, st)
        if a.mm:
            new = lambda st: L.mrs_dec_mm_gate_up(ws[0][3].data_ptr(), ws[1][3].data_ptr(), 12, ff, d, img_h.data_ptr(), 0, act.data_ptr(), ff, b, st)
        old = lambda st: L.mrs_decode_gate_up(ws[0][0].data.data_ptr(), ws[1][0].data.data_ptr(), 12, ff, d, h.data_ptr(), nw.data_ptr(), 1e-5, 0, yb.data_ptr(), 14336 // 32, b, st)
        return ws, new, old, nbytes(ws[0][0], ws[1][0])

    def ph_lm(i):
        ws = (make(Q6, 128256, d, 300 + i),)
        new = lambda st: L.mrs_dec_proj(C.byref(ws[0][2]), 128256, None, h.data_ptr(), d, nw.data_ptr(), 1e-5, logits.data_ptr(), 128256, 0, 1.0, None, b, st)
        if a.img:
            new = lambda st: L.mrs_dec_proj_img(C.byref(ws[0][2]), 128256, img_h.data_ptr(), logits.data_ptr(), 128256, 0, 1.0, b, st)
        if a.mm:
            new = lambda st: L.mrs_dec_mm_proj(ws[0][3].data_ptr(), 14, 128256, d, img_h.data_ptr(), logits.data_ptr(), 128256, 0, 1.0, b, st)
        old = lambda st: L.mrs_decode_norm_proj(ws[0][0].data.data_ptr(), 14, 128256, d, h.data_ptr(), nw.data_ptr(), 1e-5, logits.data_ptr(), 128256, b, st)
        return ws, new, old, nbytes(ws[0][0])

    L.mrs_dec2_repack_bytes.restype = C.c_size_t
    L.mrs_dec2_repack_bytes.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
    L.mrs_dec2_repack.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
    L.mrs_dec2_gemv.argtypes = [MP, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    big_out = torch.empty(b, 128256, device=dev)

    def ph_v2(dt, n, k, x, norm):
        def f(i):
            w = random_qtensor(dt, n, k, dev, 500 + i)
            nb = L.mrs_dec2_repack_bytes(dt.id, n, k)
            p = torch.empty(nb, dtype=torch.uint8, device=dev)
            assert L.mrs_dec2_repack(w.data.data_ptr(), dt.id, n, k, p.data_ptr(), st) == 0
            torch.cuda.synchronize()
            m = Mat(p.data_ptr(), dt.id, n, k)
            new = lambda st: L.mrs_dec2_gemv(C.byref(m), x.data_ptr(), k, nw.data_ptr() if norm else None, 1e-5, big_out.data_ptr(), 128256, b, st)
            return (p, m), new, new, w.data.numel()
        return f

    table = {"qkv": ph_qkv, "o": ph_proj(Q4, d, nq, attn, nq, ya, 4096 // 32), "gate_up": ph_gate_up, "down4": ph_proj(Q4, d, ff, act, ff, yb, 14336 // 32),
             "down6": ph_proj(Q6, d, ff, act, ff, yb, 14336 // 32), "lm_head": ph_lm}
    if a.v2:
        table = {"qkv": ph_v2(Q4, nq + 2 * nkv, d, h, True), "o": ph_v2(Q4, d, nq, attn, False), "gate_up": ph_v2(Q4, 2 * ff, d, h, True),
                 "down4": ph_v2(Q4, d, ff, act, False), "down6": ph_v2(Q6, d, ff, act, False), "lm_head": ph_v2(Q6, 128256, d, h, True)}
    for name in a.phases.split(","):
        mk = table[name]
        insts, total = [], 0
        while (total < 1.2e9 and len(insts) < 64) if not a.hot else len(insts) < (1 if name == "lm_head" else 2):
            inst = mk(len(insts))
            insts.append(inst)
            total += inst[3]
        if a.timeline:
            import numpy as np
            L.mrs_dec_timeline.argtypes = [C.c_void_p, C.c_int]
            tl = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
            stc = torch.cuda.current_stream().cuda_stream
            for i, inst in enumerate(insts[:6]):
                L.mrs_dec_timeline(tl.data_ptr() if i == len(insts[:6]) - 1 else None, 1)
                assert inst[1](stc) == 0
                torch.cuda.synchronize()
            L.mrs_dec_timeline(None, 0)
            if a.mm:
                tm = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
                L.mrs_dec_mm_timeline.argtypes = [C.c_void_p]
                L.mrs_dec_mm_timeline(tm.data_ptr())
                assert insts[6 % len(insts)][1](stc) == 0
                torch.cuda.synchronize()
                L.mrs_dec_mm_timeline(None)
                t = tm.cpu().numpy().reshape(512, 4, 8).astype(np.float64)
                used = t[:, :, 0] > 0
                t0 = t[:, :, 0][used].min()
                rel = (t - t0) / 100.0
                def mm_med(i):
                    v = rel[:, :, i][t[:, :, i] > 0]
                    return (round(float(np.median(v)), 2), round(float(v.max()), 2)) if v.size else None
                print(name, "mm (median, max us over waves): entry", mm_med(0), "ring issued", mm_med(1), "image stored", mm_med(2), "barrier", mm_med(3), "records done (last unit)", mm_med(4),
                      "sums published", mm_med(5), "end", mm_med(6), "workgroups", int(used.any(axis=1).sum()), flush=True)
                del insts
                torch.cuda.empty_cache()
                continue
            t = tl.cpu().numpy().reshape(256, 8, 16).astype(np.float64)
            used = t[:, :, 0] > 0
            t0 = t[:, :, 0][used].min()
            rel = (t - t0) / 100.0

            def med(sel, i):
                v = rel[:, sel, i][t[:, sel, i] > 0]
                return (round(float(np.median(v)), 2), round(float(v.max()), 2)) if v.size else None
            for nm, sel in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
                print(name, nm, "(median, max us): entry", med(sel, 0), "ring issued", med(sel, 1), "prologue done", med(sel, 2), "barrier", med(sel, 3), "end", med(sel, 14), flush=True)
            del insts
            torch.cuda.empty_cache()
            continue
        for which in (("new", 1),) + ((("old", 2),) if a.old else ()):
            fns = [inst[which[1]] for inst in insts] * (8 if a.hot else 1)
            for f in fns:
                assert f(torch.cuda.current_stream().cuda_stream) == 0
            torch.cuda.synchronize()
            # one HIP graph over all buffers: the figure is GPU time (kernel + boundary), not the host's launch rate
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for f in fns:
                    f(side.cuda_stream)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for f in fns:
                    f(torch.cuda.current_stream().cuda_stream)
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (a.reps * len(fns))
            print(json.dumps({"phase": name, "impl": which[0], "b": b, "MB": round(insts[0][3] / 1e6, 2), "us": round(us, 2), "TBps": round(insts[0][3] / us / 1e6, 3),
                              "frac_8TBps": round(insts[0][3] / us / 1e6 / 8.0, 3), "buffers": len(insts), "hot": bool(a.hot), "v2": bool(a.v2)}), flush=True)
        del insts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
