"""Ablation of the decode GEMV (experiment): builds libmistralrsquant variants with parts disabled and times q4_k shapes."""
import ctypes as C, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
vp, ci = C.c_void_p, C.c_int
def bench(lib, tag, n, k, ts, blk):
    fn = getattr(lib, f"launch_mmvq_gguf_{tag}_f32_plain"); fn.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    nb = n * (k // blk) * ts
    nbuf = max(2, (1 << 30) // nb + 1)
    ws = [torch.randint(0, 255, (nb,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    y = torch.randint(0, 255, ((k + 511) // 512 * 512 // 32 * 36,), dtype=torch.uint8, device=dev)
    out = torch.empty(n, device=dev)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for w in ws: fn(w.data_ptr(), y.data_ptr(), out.data_ptr(), k, n, y.numel() // 36, n, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().synchronize()
        with torch.cuda.graph(g, stream=side):
            for w in ws: fn(w.data_ptr(), y.data_ptr(), out.data_ptr(), k, n, y.numel() // 36, n, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 1e3 / nbuf)
    return best, nb
for variant in sys.argv[1:]:
    lib = C.CDLL(os.path.join(ROOT, "scripts", "exp", f"abl_{variant}.so"))
    for tag, ts in (("q4_k", 144), ("q6_k", 210)):
        for name, n, k in (("q", 4096, 4096), ("gate", 14336, 4096), ("down", 4096, 14336), ("lm_head", 128256, 4096)):
            t, nb = bench(lib, tag, n, k, ts, 256)
            print(f"{variant:10s} {tag} {name:8s} {t*1e6:7.2f} us  {nb/t/1e12:.2f} TB/s", flush=True)
