import sys, numpy as np, torch
sys.path.insert(0, '.')
import mistralrs_amd
from tests.test_llama_runner import _mk, Q4KM, Q8, _tokens
from oracle import oracle as O, llama_ref
dev = torch.device('cuda:0')
for rep in range(2):
  for mix, types in (("q4km", Q4KM(O)),):
    cfg, w, mf, cos, sin = _mk(O, dev, True, types)
    _, _, mu, _, _ = _mk(O, dev, False, types)
    ref = llama_ref.LlamaRef(cfg, w, cos, sin, mode="q8_1", kv_dtype="bf16")
    for pos, t in enumerate(_tokens(16)):
        want = ref.step(t, pos)
        outs = []
        for m in (mf, mu):
            m.set_state([t], [pos]); outs.append(m.forward_logits(1)[0].clone())
        torch.cuda.synchronize()
        got = outs[0].cpu().numpy(); gu = outs[1].cpu().numpy()
        print(rep, mix, pos, "fused err %.2e" % np.abs(got - want).max(), "unfused err %.2e" % np.abs(gu - want).max(), "equal", bool(torch.equal(outs[0], outs[1])))
