"""One prefill GEMM shape, a few launches: the target of rocprofv3 counter passes.  usage: gemm_one.py T N K [type]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mistralrs_amd  # noqa: F401
from mistralrs_amd.gguf import GgmlDType, fast_gemm
from mistralrs_amd.llama import random_qtensor
T, N, K = (int(v) for v in sys.argv[1:4])
tag = sys.argv[4] if len(sys.argv) > 4 else "q4_k"
dev = torch.device("cuda:0")
w = random_qtensor({d.tag: d for d in GgmlDType}[tag], N, K, dev, 5)
x = torch.randn(T, K, device=dev)
out = torch.empty(T, N, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
xb = fast_gemm.to_slabs(x)
for _ in range(6):
    fast_gemm.plain_bf16(w, xb, out=out, workspace=ws)
torch.cuda.synchronize()
