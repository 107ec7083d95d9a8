"""Does alternating between different (large) kernels cost extra per launch (instruction-cache refetch)?"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mistralrs_amd
from mistralrs_amd import _lib
from mistralrs_amd.gguf import GgmlDType, fast_mmvq
from mistralrs_amd.llama import random_qtensor
dev = torch.device("cuda:0")
vp, ci = C.c_void_p, C.c_int
ext = _lib.load("ext"); _lib.load("quant")
ext.mrs_decode_proj.argtypes = [vp, ci, ci, ci, vp, ci, vp, ci, ci, ci, vp]
n = k = 4096
kinds = [("q4_k", GgmlDType.Q4K), ("q6_k", GgmlDType.Q6K), ("q5_k", GgmlDType.Q5K), ("q8_0", GgmlDType.Q8_0)]
W = {tag: [random_qtensor(dt, n, k, dev, 3 + i) for i in range(24)] for tag, dt in kinds}
x = torch.randn(1, k, device=dev)
y, stride = fast_mmvq.quantize_q8_1(x, k, 1); y = y.clone()
out = torch.empty(1, n, device=dev)
abi = {tag: _lib.sym("quant", f"launch_mmvq_gguf_{tag}_f32_plain", [vp, vp, vp, ci, ci, ci, ci, ci, vp]) for tag, _ in kinds}
def seq(order):
    def run(st):
        for i, (kind, tag) in enumerate(order * 6):
            w = W[tag][i % 24]
            if kind == "abi": abi[tag](w.data.data_ptr(), y.data_ptr(), out.data_ptr(), k, n, stride, n, 1, st)
            else: ext.mrs_decode_proj(w.data.data_ptr(), w.dtype.id, n, k, y.data_ptr(), stride, out.data_ptr(), n, 0, 1, st)
    return run, len(order) * 6
def time_graph(run, cnt):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(side.cuda_stream); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3 / cnt)
    return best
for name, order in (("same abi q4_k", [("abi", "q4_k")] * 4), ("same abi q6_k", [("abi", "q6_k")] * 4),
                    ("alternating 4 abi kernels", [("abi", "q4_k"), ("abi", "q6_k"), ("abi", "q5_k"), ("abi", "q8_0")]),
                    ("same ext q4_k", [("ext", "q4_k")] * 4), ("ext alternating types (same kernel, different paths)", [("ext", "q4_k"), ("ext", "q6_k"), ("ext", "q5_k"), ("ext", "q8_0")]),
                    ("abi q4_k / ext q4_k alternating", [("abi", "q4_k"), ("ext", "q4_k")] * 2)):
    run, cnt = seq(order)
    print(f"{name:60s} {time_graph(run, cnt):6.2f} us per launch", flush=True)
