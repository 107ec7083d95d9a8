"""bench.py's 8B model: engine logits vs the two CPU evaluation orders of the oracle (teacher-forced on the exact-order run's greedy tokens)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from mistralrs_amd.llama import LlamaConfig, rope_tables
from oracle import llama_ref, oracle as O
O.build(); O.set_threads(min(64, os.cpu_count() or 1))
dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = LlamaConfig.llama3_8b(max_batch=1, max_context_len=832, max_position_embeddings=8192)
if layers != 32:
    import dataclasses
    cfg = dataclasses.replace(cfg, num_layers=layers)
m = bench.build_model(cfg, dev, seed=0, max_new_tokens=16)
w = {}
for name, t in m._keep.items():
    if "#" in name: continue
    if hasattr(t, "dtype") and hasattr(t, "shape") and not hasattr(t, "data_ptr"):
        w[name] = (t.dtype.id, t.data.cpu().numpy().reshape(t.shape[0], -1))
    else:
        w[name] = t.cpu().numpy()
cos, sin = rope_tables(cfg)
a = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu", kv_dtype="bf16")
b = llama_ref.LlamaRef(cfg, w, cos, sin, mode="cpu_fast", kv_dtype="bf16")
tok = 1000 % cfg.vocab_size
for pos in range(4):
    x, y = a.step(tok, pos), b.step(tok, pos)
    m.set_state([tok], [pos])
    g = m.forward_logits(1)[0].float().cpu().numpy()
    s = float(np.abs(x).max())
    print(f"layers {layers} pos {pos}: engine-cpu {np.abs(g - x).max() / s:.3e}  engine-cpu_fast {np.abs(g - y).max() / s:.3e}  cpu-cpu_fast {np.abs(x - y).max() / s:.3e}  max|logit| {s:.3f} std {x.std():.3f}", flush=True)
    tok = int(x.argmax())
