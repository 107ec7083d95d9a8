#!/usr/bin/env python
"""Does streaming the NEXT launches' weights into the Infinity Cache on a side stream shorten a decode layer's chain of GEMV launches?
Layer = qkv, o_proj, gate/up, down (decode engine kernels, Llama-3-8B shapes, 123 MB of Q4_K / Q6_K per layer), L layers of distinct weights
(>= 1.2 GB: nothing is cache resident between replays).  Variants, all one captured HIP graph:
  serial        the chain alone
  pf_kernel     before launch k, a side stream starts mrs_l3_prefetch of launch k+1's weights; launch k+1 waits for it
  pf_layer      at the head of layer l, a side stream starts prefetching ALL of layer l+1; layer l+1's first launch waits for it
  pf_only       the prefetch kernels alone (their HBM rate)
Prints us per layer."""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


class Mat(C.Structure):
    _fields_ = [("planes", C.c_void_p), ("type", C.c_int), ("n", C.c_longlong), ("k", C.c_longlong)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--wgs", default="64,128,256")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
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
    L.mrs_l3_prefetch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    MP = C.POINTER(Mat)
    L.mrs_dec_qkv.argtypes = [MP, MP, MP, C.c_void_p, C.c_int, C.c_void_p, C.c_float] + [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]
    L.mrs_dec_gate_up.argtypes = [MP, MP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mrs_dec_proj.argtypes = [MP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    st0 = torch.cuda.current_stream().cuda_stream
    d, ff, nq, nkv, hd, b = 4096, 14336, 4096, 1024, 128, 1
    Q4, Q6 = GgmlDType.Q4K, GgmlDType.Q6K

    def make(dt, n, k, seed):
        w = random_qtensor(dt, n, k, dev, seed)
        nb = L.mrs_dec_repack_bytes(dt.id, n, k)
        p = torch.empty(nb, dtype=torch.uint8, device=dev)
        assert L.mrs_dec_repack(w.data.data_ptr(), dt.id, n, k, p.data_ptr(), st0) == 0
        torch.cuda.synchronize()
        del w
        return p, Mat(p.data_ptr(), dt.id, n, k)

    h = torch.randn(b, d, device=dev); nw = torch.ones(d, device=dev)
    act = torch.randn(b, ff, device=dev); attn = torch.randn(b, nq, device=dev); q_out = torch.empty(b, nq, device=dev)
    kc = torch.zeros(8, 8, hd // 8, 32, 8, dtype=torch.bfloat16, device=dev); vc = torch.zeros(8, 8, hd, 32, dtype=torch.bfloat16, device=dev)
    slots = torch.arange(b, dtype=torch.int64, device=dev); pos = torch.arange(b, dtype=torch.int32, device=dev)
    cos = torch.ones(64, hd // 2, device=dev); sin = torch.zeros(64, hd // 2, device=dev)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    layers = []
    for l in range(a.layers):
        s = 10 * l
        wq, wk, wv = make(Q4, nq, d, s), make(Q4, nkv, d, s + 1), make(Q6, nkv, d, s + 2)
        wo = make(Q4, d, nq, s + 3)
        wg, wu = make(Q4, ff, d, s + 4), make(Q4, ff, d, s + 5)
        wd = make(Q6 if l % 2 else Q4, d, ff, s + 6)
        k_qkv = lambda st, wq=wq, wk=wk, wv=wv: L.mrs_dec_qkv(C.byref(wq[1]), C.byref(wk[1]), C.byref(wv[1]), h.data_ptr(), d, nw.data_ptr(), 1e-5, q_out.data_ptr(), kc.data_ptr(),
                                                            vc.data_ptr(), slots.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), hd, hd // 2, 8, 32, 1, b, st)
        k_o = lambda st, wo=wo: L.mrs_dec_proj(C.byref(wo[1]), d, None, attn.data_ptr(), nq, None, 0.0, h.data_ptr(), d, 1, 1.0, None, b, st)
        k_gu = lambda st, wg=wg, wu=wu: L.mrs_dec_gate_up(C.byref(wg[1]), C.byref(wu[1]), ff, None, h.data_ptr(), d, nw.data_ptr(), 1e-5, 0, act.data_ptr(), ff, b, st)
        k_dn = lambda st, wd=wd: L.mrs_dec_proj(C.byref(wd[1]), d, None, act.data_ptr(), ff, None, 0.0, h.data_ptr(), d, 1, 1.0, None, b, st)
        layers.append([(k_qkv, [wq[0], wk[0], wv[0]]), (k_o, [wo[0]]), (k_gu, [wg[0], wu[0]]), (k_dn, [wd[0]])])
    launches = [x for lay in layers for x in lay]
    layer_mb = sum(t.numel() for _, ts in layers[0] for t in ts) / 1e6

    def prefetch(tensors, wgs, st):
        for t in tensors:
            assert L.mrs_l3_prefetch(t.data_ptr(), t.numel(), wgs, sink.data_ptr(), st) == 0

    def build(variant, wgs):
        main_s, side = torch.cuda.Stream(), torch.cuda.Stream()

        def body():
            ms = main_s.cuda_stream
            if variant == "serial":
                for k, _ in launches:
                    assert k(ms) == 0
            elif variant == "pf_only":
                for _, ts in launches:
                    prefetch(ts, wgs, ms)
            elif variant == "pf_kernel":
                for i, (k, _) in enumerate(launches):
                    if i + 1 < len(launches):
                        side.wait_stream(main_s)  # fork at the point launch i is enqueued
                        with torch.cuda.stream(side):
                            prefetch(launches[i + 1][1], wgs, side.cuda_stream)
                    assert k(ms) == 0
                    if i + 1 < len(launches):
                        main_s.wait_stream(side)  # launch i+1 starts after its weights were requested
            elif variant == "pf_layer":
                for l, lay in enumerate(layers):
                    if l + 1 < len(layers):
                        side.wait_stream(main_s)
                        with torch.cuda.stream(side):
                            prefetch([t for _, ts in layers[l + 1] for t in ts], wgs, side.cuda_stream)
                    for k, _ in lay:
                        assert k(ms) == 0
                    if l + 1 < len(layers):
                        main_s.wait_stream(side)
        main_s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main_s):
            body()
        main_s.synchronize(); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(main_s):
            with torch.cuda.graph(g, stream=main_s):
                body()
        torch.cuda.current_stream().wait_stream(main_s)
        return g

    def time_graph(g):
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(a.reps):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / a.layers)
        return best

    print(json.dumps({"layer_MB": round(layer_mb, 2), "layers": a.layers}), flush=True)
    us = time_graph(build("serial", 0))
    print(json.dumps({"variant": "serial", "us_per_layer": round(us, 2), "TBps": round(layer_mb / us, 3)}), flush=True)
    for wgs in [int(x) for x in a.wgs.split(",")]:
        for variant in ("pf_only", "pf_kernel", "pf_layer"):
            try:
                us = time_graph(build(variant, wgs))
                print(json.dumps({"variant": variant, "wgs": wgs, "us_per_layer": round(us, 2), "TBps": round(layer_mb / us, 3)}), flush=True)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"variant": variant, "wgs": wgs, "error": str(e)[:200]}), flush=True)


if __name__ == "__main__":
    main()
