#!/bin/bash
# call 25: ablations of the wave-specialised prefill GEMM (gate/up shape, T = 512 and 2048)
O=gpurun_out/r2c25; mkdir -p $O
cd /root/repo
cat > /tmp/one.py <<'PY'
import os, sys, torch, json
sys.path.insert(0, '/root/repo')
import mistralrs_amd
from mistralrs_amd.gguf import GgmlDType, fast_gemm
from mistralrs_amd.llama import random_qtensor
dev = torch.device("cuda:0")
w = random_qtensor({d.tag: d for d in GgmlDType}["q4_k"], 28672, 4096, dev, 5)
for T in (512, 2048):
    x = torch.randn(T, 4096, device=dev); out = torch.empty(T, 28672, device=dev); ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    xb = fast_gemm.to_slabs(x)
    for _ in range(3): fast_gemm.plain_bf16(w, xb, out=out, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fast_gemm.plain_bf16(w, xb, out=out, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("MRS_EXT_LIB", "shipped"), "T", T, "us %.1f" % (e0.elapsed_time(e1) * 100), flush=True)
PY
for n in "" 1 16 2 4 8 6; do
  if [ -z "$n" ]; then timeout 120 python /tmp/one.py; else MRS_EXT_LIB=libmrs_hip_ext_abl$n.so timeout 120 python /tmp/one.py; fi
done 2>&1 | grep -v amdgpu.ids | tee $O/ablate.log
