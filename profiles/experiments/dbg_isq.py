import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mistralrs_amd
from mistralrs_amd.llama import Llama, LlamaConfig, rope_tables
from mistralrs_amd.gguf import GgmlDType, QTensor
from oracle import oracle as O, llama_ref
from tests.util import round_through
O.build()
dev = torch.device("cuda:0")
for tname, hd, heads, kvh in (("Q8_0", 128, 4, 2), ("Q8_0", 64, 8, 4), ("Q4_K", 128, 4, 2)):
    t = getattr(O, tname)
    res = {}
    for fused in (True, False):
        cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_layers=2, num_heads=heads, num_kv_heads=kvh, vocab_size=512, head_dim=hd,
                          rope_theta=10000.0, max_position_embeddings=256, max_batch=2, max_context_len=192, use_fused=fused)
        rng = np.random.default_rng(11)
        d, ff, nq, nkv = 512, 1024, heads * hd, kvh * hd
        shapes = {"token_embd.weight": (512, d), "output.weight": (512, d)}
        for i in range(2):
            for nm, sh in (("attn_q", (nq, d)), ("attn_k", (nkv, d)), ("attn_v", (nkv, d)), ("attn_output", (d, nq)), ("ffn_gate", (ff, d)), ("ffn_up", (ff, d)), ("ffn_down", (d, ff))):
                shapes[f"blk.{i}.{nm}.weight"] = sh
        m = Llama(cfg, dev, max_new_tokens=16)
        w = {}
        for name, sh in shapes.items():
            dense = round_through((rng.standard_normal(sh) * 0.04).astype(np.float32), "bf16")
            packed = O.quantize(t, dense)
            m.set_tensor(name, QTensor.from_numpy(GgmlDType.from_id(t), sh, packed, dev))
            w[name] = (t, packed)
        for name in [f"blk.{i}.{n}.weight" for i in range(2) for n in ("attn_norm", "ffn_norm")] + ["output_norm.weight"]:
            nw = (1.0 + 0.01 * rng.standard_normal(d)).astype(np.float32)
            m.set_tensor(name, torch.from_numpy(nw)); w[name] = nw
        cos, sin = rope_tables(cfg)
        ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="q8_1", kv_dtype="bf16")
        rels = []
        for pos, tok in enumerate([(1000 + i) % 512 for i in range(4)]):
            want = ref.step(tok, pos)
            m.set_state([tok], [pos])
            got = m.forward_logits(1)[0].cpu().numpy()
            rels.append(float(np.abs(got - want).max() / np.abs(want).max()))
        print(tname, "hd", hd, "fused" if fused else "unfused", ["%.2e" % r for r in rels], flush=True)
