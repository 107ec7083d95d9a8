#!/usr/bin/env python
"""Per-workgroup timeline of the decode engine's GEMV launches inside the captured decode graph (needs MRS_EXT_LIB=libmrs_hip_ext_tl.so,
built by profiles/experiments/build_variant.sh tl ext_dec.hip -DMRS_DEC_TIMELINE).  Stamps (100 MHz constant clock, lane 0 of wave 0 / wave 7):
0 kernel entry, 1 ring issued (prologue starts), 2 prologue done, 3 wave 0 finished its rows, 4 wave 7 finished."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import mistralrs_amd  # noqa
from mistralrs_amd import _lib
from mistralrs_amd.llama import LlamaConfig

dev = torch.device("cuda:0")
cfg = LlamaConfig.llama3_8b(max_batch=8, max_context_len=1024, max_position_embeddings=8192)
model = bench.build_model(cfg, dev, seed=0, max_new_tokens=64)
prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(512)]
last = model.prefill(prompt, 0)
ext = _lib.load("ext")
CAP = 256
buf = torch.zeros(CAP * 256 * 32, dtype=torch.int64, device=dev)
ext.mrs_dec_timeline.argtypes = [C.c_void_p, C.c_int]
ext.mrs_dec_timeline_log.argtypes = [C.c_void_p, C.c_int]
ext.mrs_dec_timeline_log.restype = C.c_int
model.set_state([int(last.argmax())], [512])
model.step_counter.zero_()
ext.mrs_dec_timeline(buf.data_ptr(), CAP)
model.capture_decode_graph(1)
n_launch_total = ext.mrs_dec_timeline_log(None, 0)
for _ in range(6):
    model.replay()
torch.cuda.synchronize()
kinds = (C.c_int * CAP)()
n = ext.mrs_dec_timeline_log(kinds, CAP)
t = buf.cpu().numpy().reshape(CAP, 256, 32).astype(np.int64)
names = {0: "store", 1: "resid", 2: "glu", 3: "qkv", 4: "resid2"}
print("launches recorded", n)
rows, prev_end = [], None
for i in range(min(n, CAP)):
    r = t[i]
    live = r[:, 0] > 0
    if not live.any():
        continue
    r = r[live]
    base = r[:, 0].min()
    rel = lambda x: x - r[:, 0]          # per workgroup, relative to its own entry
    t1 = r[:, 1:9]
    end = r[:, 11:19]
    med = lambda x: float(np.median(x))
    row = dict(i=i, kind=names.get(kinds[i], "?"), wgs=len(r), entry_spread=int(r[:, 0].max() - base),
               ring_first=med(t1.min(1) - r[:, 0]), ring_last=med(t1.max(1) - r[:, 0]),
               sq=med(rel(r[:, 19])) if r[:, 19].max() > 0 else -1, normbar=med(rel(r[:, 9])) if r[:, 9].max() > 0 else -1,
               quant=med(rel(r[:, 20])), pro_done=med(rel(r[:, 10])), end_first=med(end.min(1) - r[:, 0]), end_last=med(end.max(1) - r[:, 0]),
               iss=[int(np.median(rel(r[:, 21 + k]))) for k in range(8)] if r[:, 21].max() > 0 else None,
               dur=int(end.max() - base), gap=(int(base - prev_end) if prev_end is not None else -1))
    prev_end = end.max()
    rows.append(row)
print("units: 10 ns ticks, medians over workgroups of (stamp - own kernel entry); ring_* = first/last wave done issuing act loads + ring; sq = wave 0 has its squares; normbar = after the norm barrier;")
print("quant = wave 0 quantized its share; pro_done = after the last prologue barrier; end_* = first/last wave finished; gap = previous GEMV end -> this entry (attention sits in the gap before resid #1)")
for row in rows[6 * 4: 6 * 4 + 8]:
    print(row)
agg = {}
seen = {}
for row in rows[8:-2]:
    k = row["kind"]
    if k == "resid":  # o_proj (first after qkv) vs down
        k = "o_proj" if seen.get("last") == "qkv" else "down"
    seen["last"] = row["kind"]
    agg.setdefault(k, []).append(row)
for k, a in agg.items():
    print(k, "n", len(a), {f: round(float(np.mean([r[f] for r in a])), 1) for f in ("entry_spread", "ring_first", "ring_last", "sq", "normbar", "quant", "pro_done", "end_first", "end_last", "dur", "gap")})
