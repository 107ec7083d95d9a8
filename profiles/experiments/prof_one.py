"""Run one GEMV shape a few times (eager) so rocprofv3 can attach counters to it."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mistralrs_amd
from mistralrs_amd import _lib
tag, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ts = {"q4_k": 144, "q6_k": 210, "q8_0": 34 * 8, "q5_k": 176}[tag]
dev = torch.device("cuda:0")
vp, ci = C.c_void_p, C.c_int
fn = _lib.sym("quant", f"launch_mmvq_gguf_{tag}_f32_plain", [vp, vp, vp, ci, ci, ci, ci, ci, vp])
nb = n * (k // 256) * ts
nbuf = max(2, (1 << 29) // nb + 1)
ws = [torch.randint(0, 255, (nb,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
y = torch.randint(0, 255, ((k + 511) // 512 * 512 // 32 * 36,), dtype=torch.uint8, device=dev)
out = torch.empty(n, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    for w in ws: fn(w.data_ptr(), y.data_ptr(), out.data_ptr(), k, n, y.numel() // 36, n, 1, st)
torch.cuda.synchronize()
