#!/usr/bin/env python
"""Fused HQQ dequant-GEMV (mrs_hqq_gemv) vs dequantize + dense matmul: us per call and packed-bytes rate."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mistralrs_amd  # noqa: F401
from mistralrs_amd.hqq import HqqConfig, HqqLayer

dev = torch.device("cuda:0")
for bits, n, k in ((4, 4096, 4096), (4, 14336, 4096), (4, 4096, 14336), (8, 4096, 4096)):
    for dt in (torch.bfloat16, torch.float32):
        w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
        layer = HqqLayer.quantize(w, HqqConfig(bits=bits, group_size=64))
        x = torch.randn(1, k, device=dev).to(dt)

        def timeit_eager(f, reps=20):
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps

        def timeit(f, reps=20):  # one HIP graph of `reps` calls: device time, not the Python launch path (~15 us per eager call)
            f(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                with torch.cuda.graph(g, stream=st):
                    for _ in range(reps):
                        f()
            torch.cuda.current_stream().wait_stream(st)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        fused = timeit(lambda: layer.forward(x))
        unfused = timeit_eager(lambda: x.to(layer.dequantize().dtype) @ layer.dequantize().t())
        nbytes = n * k * bits // 8
        print(json.dumps({"bits": bits, "N": n, "K": k, "dtype": str(dt).split(".")[-1], "fused_us": round(fused, 1), "dequant_matmul_us": round(unfused, 1),
                          "packed_TBps": round(nbytes / fused / 1e6, 3)}), flush=True)
