"""Time variants of ext_gemm.hip (compiled with -DGV=n into profiles/experiments/gemm_var_n.so) on the Llama-3-8B prefill shapes (Q4_K)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mistralrs_amd
from mistralrs_amd.gguf import GgmlDType
from mistralrs_amd.llama import random_qtensor
dev = torch.device("cuda:0")
T = 512
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
shapes = (("q", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336))
ten = {n: (random_qtensor(GgmlDType.Q4K, N, K, dev, 5), torch.randn(16 * K // 64, T, 64, device=dev).to(torch.bfloat16), torch.empty(T, N, device=dev)) for n, N, K in shapes}
for v in sys.argv[1:]:
    L = C.CDLL(os.path.join(ROOT, "scripts/expbin", f"gemm_var_{v}.so"))
    fn = L.mrs_gemm_q_bf16_multi
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    for name, N, K in shapes:
        w, xb, out = ten[name]
        wp, np_, op, ld = (C.c_void_p * 1)(w.data.data_ptr()), (C.c_int * 1)(N), (C.c_void_p * 1)(out.data_ptr()), (C.c_int * 1)(N)
        run = lambda: fn(1, wp, np_, op, ld, 12, K, xb.data_ptr(), T, 0, ws.data_ptr(), ws.numel(), None)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"variant {v} {name:8s} {us:8.1f} us  {2.0 * T * N * K / us / 1e6:7.1f} TF/s", flush=True)
