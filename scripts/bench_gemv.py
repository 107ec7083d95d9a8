#!/usr/bin/env python
"""Per-kernel decode-GEMV microbenchmark (GPU): GB/s of algorithmic weight bytes per launch.

Each case rotates over enough distinct weight buffers (>= 1 GiB total) that nothing is served from the
256 MiB Infinity Cache -- in the real decode step every weight byte is touched once per token with
~4.7 GB of other traffic in between.  Timed with HIP events on the launch stream, launches back-to-back.

    python scripts/bench_gemv.py [--types q4_k,q6_k] [--b 1] [--ext]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # Llama-3-8B: (name, N, K)
    ("q", 4096, 4096), ("kv", 1024, 4096), ("gate", 14336, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--types", default="q4_k,q6_k,q8_0,q5_k")
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ext", action="store_true", help="also time the fused ext_decode kernels")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--nbuf", type=int, default=0, help="force the number of rotating weight buffers (2 => Infinity-Cache resident)")
    a = ap.parse_args()
    import torch
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd import _lib
    from mistralrs_amd.gguf import GgmlDType
    from mistralrs_amd.gguf import fast_mmvq
    from mistralrs_amd.llama import random_qtensor
    dev = torch.device("cuda:0")
    stream = [torch.cuda.current_stream().cuda_stream]
    vp, ci = C.c_void_p, C.c_int
    tags = {d.tag: d for d in GgmlDType}
    shapes = [s for s in SHAPES if not a.shapes or s[0] in a.shapes.split(",")]
    rows = []
    ext = _lib.load("ext") if a.ext else None
    _lib.load("quant")
    for tag in a.types.split(","):
        dt = tags[tag]
        for name, n, k in shapes:
            nbytes = n * dt.row_bytes(k)
            nbuf = max(2, min(96, (1 << 30) // nbytes + 1))
            ws = [random_qtensor(dt, n, k, dev, 17 + i) for i in range(a.nbuf or nbuf)]
            if a.nbuf:
                ws = (ws * (nbuf // len(ws) + 1))[:nbuf]  # same launch count, but only a.nbuf distinct buffers (Infinity-Cache resident)
            x = torch.randn(a.b, k, device=dev)
            y, stride = fast_mmvq.quantize_q8_1(x, k, a.b)
            y = y.clone()
            out = torch.empty(a.b, n, device=dev)
            fn = _lib.sym("quant", f"launch_mmvq_gguf_{tag}_f32_plain", [vp, vp, vp, ci, ci, ci, ci, ci, vp])

            def launch_abi(i):
                fn(ws[i % nbuf].data.data_ptr(), y.data_ptr(), out.data_ptr(), k, n, stride, n, a.b, stream[0])
            cases = [("abi_plain", launch_abi)]
            if ext is not None and ext.mrs_decode_gemv_supported(dt.id):
                ext.mrs_decode_proj.argtypes = [vp, ci, ci, ci, vp, ci, vp, ci, ci, ci, vp]

                def launch_ext(i):
                    ext.mrs_decode_proj(ws[i % nbuf].data.data_ptr(), dt.id, n, k, y.data_ptr(), stride, out.data_ptr(), n, 0, a.b, stream[0])
                cases.append(("ext_proj", launch_ext))
            for cname, launch in cases:
                for i in range(nbuf):
                    launch(i)
                torch.cuda.synchronize()
                # HIP graph of the nbuf launches: device-side back-to-back issue, no host launch cost in the timing
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    saved = stream[0]
                    with torch.cuda.graph(g, stream=side):
                        stream[0] = torch.cuda.current_stream().cuda_stream
                        for i in range(nbuf):
                            launch(i)
                    stream[0] = saved
                torch.cuda.current_stream().wait_stream(side)
                g.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(a.reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    g.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 1e3 / nbuf)
                rows.append({"kernel": cname, "type": tag, "shape": name, "N": n, "K": k, "b": a.b, "MB": round(nbytes / 1e6, 2),
                             "us": round(best * 1e6, 2), "GBps": round(nbytes / best / 1e9, 1), "frac_8TBps": round(nbytes / best / 8e12, 3)})
                print(json.dumps(rows[-1]), flush=True)
            del ws
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
