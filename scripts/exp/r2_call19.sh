timeout 600 python -m pytest tests/test_hqq.py -m gpu -q -p no:cacheprovider -k "fused or 3bit" 2>&1 | tail -5
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3
python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
import mistralrs_amd
from mistralrs_amd.hqq import HqqConfig, HqqLayer
dev = torch.device('cuda:0')
for bits in (4, 8):
    for n, k in ((4096, 4096), (14336, 4096), (4096, 14336)):
        w = torch.randn(n, k, device=dev) * 0.05
        layer = HqqLayer.quantize(w, HqqConfig(bits=bits, group_size=64)).to_dtype(torch.bfloat16)
        x = torch.randn(1, k, device=dev, dtype=torch.bfloat16)
        for fused in (True, False):
            f = (lambda: layer.forward(x)) if fused else (lambda: x @ layer.dequantize().t())
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            by = n * k * bits / 8
            print(f"hqq{bits} [{n},{k}] {'fused gemv' if fused else 'dequantize+matmul'}: {us:8.1f} us  {by / us / 1e6:6.3f} TB/s of packed bytes")
PY
