#!/usr/bin/env python
"""Prefill GEMM microbenchmark: TFLOP/s of mrs_gemm_q_f32 on the Llama-3-8B shapes at T tokens."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=512)
    ap.add_argument("--types", default="q4_k,q6_k")
    ap.add_argument("--big", action="store_true", help="256-row-tile kernel over bf16 activations (mrs_gemm_q_bf16_multi)")
    ap.add_argument("--qi", action="store_true", help="the exact-integer prompt GEMM in the decode engine's arithmetic (mrs_gemm_qi, csrc/ext_gemm_qi.hip); also times the activation quantizer")
    ap.add_argument("--mmq", action="store_true", help="the reference-ABI route instead: launch_mmq_quantize_q8_1_* + launch_mmq_gguf_<t> (fast_mmq.plain)")
    a = ap.parse_args()
    import torch
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.gguf import GgmlDType, fast_gemm, fast_mmq
    from mistralrs_amd.llama import random_qtensor
    dev = torch.device("cuda:0")
    tags = {d.tag: d for d in GgmlDType}
    for tag in a.types.split(","):
        for name, n, k in (("q", 4096, 4096), ("qkv", 6144, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
            w = random_qtensor(tags[tag], n, k, dev, 5)
            x = torch.randn(a.t, k, device=dev)
            out = torch.empty(a.t, n, device=dev)
            ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
            xb = fast_gemm.to_slabs(x) if a.big else None
            if a.qi:
                import ctypes as C
                from mistralrs_amd import _lib
                L = _lib.load("ext")
                L.mrs_gemm_qi_repack_bytes.restype = C.c_size_t; L.mrs_gemm_qi_repack_bytes.argtypes = [C.c_int, C.c_longlong, C.c_longlong]
                L.mrs_qi_act_bytes.restype = C.c_size_t; L.mrs_qi_act_bytes.argtypes = [C.c_int, C.c_int]
                L.mrs_gemm_qi_repack.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]
                L.mrs_qi_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
                L.mrs_gemm_qi_ws.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
                st = torch.cuda.current_stream().cuda_stream
                lay = torch.empty(L.mrs_gemm_qi_repack_bytes(w.dtype.id, n, k), dtype=torch.uint8, device=dev)
                assert L.mrs_gemm_qi_repack(w.data.data_ptr(), w.dtype.id, n, k, lay.data_ptr(), st) == 0
                actb = torch.empty(L.mrs_qi_act_bytes(a.t, k), dtype=torch.uint8, device=dev)
                assert L.mrs_qi_quantize(x.data_ptr(), None, k, None, 0.0, a.t, k, actb.data_ptr(), None, st) == 0
                wsq = torch.empty(4 * a.t * n * 4, dtype=torch.uint8, device=dev)
                run = lambda: L.mrs_gemm_qi_ws(lay.data_ptr(), w.dtype.id, n, k, actb.data_ptr(), a.t, out.data_ptr(), n, 0, wsq.data_ptr(), wsq.numel(), st)
            elif a.mmq:
                run = lambda: fast_mmq.plain(w, x)
            elif a.big:
                run = lambda: fast_gemm.plain_bf16(w, xb, out=out, workspace=ws)
            else:
                run = lambda: fast_gemm.plain(w, x, out=out)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 1e3 / reps
            fl = 2.0 * a.t * n * k
            print(json.dumps({"type": tag, "shape": name, "T": a.t, "N": n, "K": k, "us": round(t * 1e6, 1), "TFLOPs": round(fl / t / 1e12, 1),
                              "frac_2.5PF": round(fl / t / 2.5e15, 3)}), flush=True)


if __name__ == "__main__":
    main()
