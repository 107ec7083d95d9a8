"""round 6 measurement (not product code): what the library bf16 GEMM (hipBLASLt behind torch.matmul) reaches at the prompt shapes of Llama-3-8B / 70B -- the A/B the
VERDICT asks for before a bf16 SHADOW COPY of the weights (dequantized once at load) replaces the fused block-dequant kernel on the selectable bf16 prompt path.
C [T, N] = X [T, K] . W [N, K]^T, bf16 operands, f32 accumulate; TFLOP/s = 2 T N K / time, weights rotated over >= 1 GB (nothing cache-resident)."""
import torch
dev = torch.device("cuda:0")
shapes = {"8b": [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)],
          "70b": [("qkv", 10240, 8192), ("o", 8192, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)]}
for model, T in (("8b", 512), ("8b", 2048), ("70b", 2048)):
    tot_f, tot_t = 0.0, 0.0
    for name, N, K in shapes[model]:
        nbuf = max(2, int(1.2e9 // (N * K * 2)) + 1)
        ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) for _ in range(min(nbuf, 24))]
        x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
        for w in ws[:2]:
            torch.matmul(x, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for w in ws:
                torch.matmul(x, w.t())
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 1e3 / (3 * len(ws))
        f = 2.0 * T * N * K
        tot_f += f; tot_t += t
        print(f"{model} T={T} {name:8s} N={N} K={K}: {t * 1e6:8.1f} us  {f / t / 1e12:7.1f} TFLOP/s", flush=True)
        del ws
        torch.cuda.empty_cache()
    print(f"{model} T={T} one layer: {tot_t * 1e6:.1f} us = {tot_f / tot_t / 1e12:.1f} TFLOP/s = {tot_f / tot_t / 2.5e15:.3f} of the bf16 MFMA peak", flush=True)
