"""Frozen golden vectors (tests/golden/*.npz, produced by scripts/gen_golden.py from the reference's own device functions
compiled for the host): they pin
  * the oracle (CPU, always run):  block decode bit-exact, Q8_1 quantizer bit-exact, Q8_1 matvec to f32 term rounding;
  * the HIP kernels (-m gpu):      launch_mmvq_gguf_quantize_q8_1_f32 bit-exact, launch_mmvq_gguf_<t>_f32_plain within the
                                   f32-accumulation bound of the reference value, for all 10 MMVQ formats.
"""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "mmvq_*.npz")))


def test_golden_present():
    assert len(FILES) == 10, "tests/golden is incomplete: run scripts/gen_golden.py in the build container"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    t, n, k = int(g["type"]), int(g["n"]), int(g["k"])
    np.testing.assert_array_equal(oracle.quantize_q8_1(g["x"]), g["y_q8_1"])
    deq = oracle.dequantize(t, g["w"], k)
    np.testing.assert_array_equal(deq, g["dequant_ref"])
    got, mag = oracle.matmul_q8_1_mag(t, g["w"], n, k, g["y_q8_1"])
    tol = 2.0 ** -21 * mag.astype(np.float64) + 2.0 ** -23 * np.abs(g["out_ref"]) + 1e-30
    assert (np.abs(got - g["out_ref"]) <= tol).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_kernels_reproduce_golden(dev, path):
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_mmvq
    g = np.load(path)
    t, n, k = int(g["type"]), int(g["n"]), int(g["k"])
    x = torch.from_numpy(g["x"]).to(dev)
    ws, stride = fast_mmvq.quantize_q8_1(x, k, x.shape[0])
    torch.cuda.synchronize()
    got_y = ws[: x.shape[0] * stride * 36].cpu().numpy().reshape(x.shape[0], -1)
    np.testing.assert_array_equal(got_y, g["y_q8_1"])  # activation quantizer: bit-exact
    w = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), g["w"], dev)
    got = fast_mmvq.plain(w, x).cpu().numpy().astype(np.float64)
    # f32 accumulation of ~k/16 terms in an arbitrary order: 8 eps sqrt(terms) * sum|terms|
    tol = 8 * 2.0 ** -23 * np.sqrt(k / 16) * g["mag"].astype(np.float64) + 2.0 ** -23 * np.abs(g["out_ref"]) + 1e-30
    err = np.abs(got - g["out_ref"])
    assert (err <= tol).all(), float((err / tol).max())


# ---- RoPE + paged-cache data movement: outputs of the reference's own kernels (scripts/gen_golden.py via oracle/_ref/libref_cache.so)
CACHE_GOLD = os.path.join(GOLD, "cache_ops.npz")


def test_oracle_reproduces_cache_ops_golden(oracle):
    g = np.load(CACHE_GOLD)
    for neox in (0, 1):
        np.testing.assert_array_equal(oracle.rope(g["q"], g["cos"], g["sin"], g["pos"].astype(np.int32), bool(neox)).view(np.uint32), g[f"q_out_{neox}"].view(np.uint32))
        np.testing.assert_array_equal(oracle.rope(g["k"], g["cos"], g["sin"], g["pos"].astype(np.int32), bool(neox)).view(np.uint32), g[f"k_out_{neox}"].view(np.uint32))
    kc, vc = np.zeros_like(g["kc_ref"]), np.zeros_like(g["vc_ref"])
    oracle.kv_cache_write(kc, vc, g["key"], g["val"], g["slots"])
    np.testing.assert_array_equal(kc, g["kc_ref"])
    np.testing.assert_array_equal(vc, g["vc_ref"])
    cu = g["cu"]
    for s in range(2):
        kg, vg = oracle.kv_cache_gather(g["kc_ref"], g["vc_ref"], g["tables"][s], int(cu[s + 1] - cu[s]))
        np.testing.assert_array_equal(kg, g["k_gather_ref"][cu[s]:cu[s + 1]])
        np.testing.assert_array_equal(vg, g["v_gather_ref"][cu[s]:cu[s + 1]])


@pytest.mark.gpu
def test_hip_rope_and_cache_ops_reproduce_reference_golden(dev):
    """f32 instantiations of rotary_embedding_positions / reshape_and_cache / gather_kv_cache == the reference kernels, bit for bit."""
    import torch
    from mistralrs_amd import ops, paged_attn
    g = np.load(CACHE_GOLD)
    cos, sin = torch.from_numpy(g["cos"]).to(dev), torch.from_numpy(g["sin"]).to(dev)
    pos = torch.from_numpy(g["pos"].astype(np.int32)).to(dev)
    for neox in (0, 1):
        q, k = torch.from_numpy(g["q"]).to(dev), torch.from_numpy(g["k"]).to(dev)
        ops.apply_rotary_qk(q, k, cos, sin, bool(neox), pos)
        np.testing.assert_array_equal(q.cpu().numpy().view(np.uint32), g[f"q_out_{neox}"].view(np.uint32))
        np.testing.assert_array_equal(k.cpu().numpy().view(np.uint32), g[f"k_out_{neox}"].view(np.uint32))
        q2, k2 = torch.from_numpy(g["q"]).to(dev), torch.from_numpy(g["k"]).to(dev)
        ops.apply_rotary_qk(q2, k2, cos[pos.long()].contiguous(), sin[pos.long()].contiguous(), bool(neox))  # per-token rows, no positions
        assert torch.equal(q2, q) and torch.equal(k2, k)
    kc, vc = torch.zeros(g["kc_ref"].shape, device=dev), torch.zeros(g["vc_ref"].shape, device=dev)
    paged_attn.reshape_and_cache(torch.from_numpy(g["key"]).to(dev), torch.from_numpy(g["val"]).to(dev), kc, vc, torch.from_numpy(g["slots"]).to(dev))
    np.testing.assert_array_equal(kc.cpu().numpy(), g["kc_ref"])
    np.testing.assert_array_equal(vc.cpu().numpy(), g["vc_ref"])
    k_out, v_out = paged_attn.gather_kv_cache(kc, vc, torch.from_numpy(g["tables"]).to(dev), torch.from_numpy(g["cu"]).to(dev), torch.float32)
    np.testing.assert_array_equal(k_out.cpu().numpy(), g["k_gather_ref"])
    np.testing.assert_array_equal(v_out.cpu().numpy(), g["v_gather_ref"])


# ---- paged attention: outputs of the reference's own v1 / v2 + reduce kernels (f32) run on host fibers (oracle/_ref/libref_pa.so);
#      inputs are the index formulas of scripts/gen_golden.py:paged_attn_case, so only the reference outputs are stored
def _pa_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(os.path.dirname(GOLD), "..", "scripts", "gen_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


PA_GOLD = os.path.join(GOLD, "paged_attn_ref.npz")


@pytest.mark.parametrize("name", ["gqa_sinks", "mha_alibi", "softcap"])
def test_oracle_attention_reproduces_reference_kernel_golden(oracle, name):
    gg = _pa_cases()
    case = gg.PA_CASES[name]
    q, kc, vc, bt, ctxs, alibi, sinks, scale = gg.paged_attn_case(**case)
    ref = np.load(PA_GOLD)
    want = oracle.paged_attention_ref(q, kc, vc, bt.astype(np.int32), ctxs, float(scale), float(case.get("softcap", 1.0)), alibi, sinks)
    pabs = oracle.paged_attention_ref(q, kc, np.abs(vc), bt.astype(np.int32), ctxs, float(scale), float(case.get("softcap", 1.0)), alibi, sinks)
    for v in ("v1", "v2"):
        err = np.abs(ref[f"{name}_{v}"] - want)
        assert (err <= 3e-5 * pabs + 1e-6).all(), (name, v, float(err.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gqa_sinks", "mha_alibi", "softcap"])
def test_hip_attention_reproduces_reference_kernel_golden(oracle, dev, name):
    """paged_attention_v1_f32 / v2_f32 (f32 cache) against the outputs of the reference's own kernels, to f32 summation order."""
    import torch
    from mistralrs_amd import paged_attn
    gg = _pa_cases()
    case = gg.PA_CASES[name]
    q, kc, vc, bt, ctxs, alibi, sinks, scale = gg.paged_attn_case(**case)
    ref = np.load(PA_GOLD)
    pabs = oracle.paged_attention_ref(q, kc, np.abs(vc), bt.astype(np.int32), ctxs, float(scale), float(case.get("softcap", 1.0)), alibi, sinks)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if a is not None else None
    for v in ("v1", "v2"):
        got = paged_attn.paged_attention(t(q), t(kc), t(vc), t(bt.astype(np.int32)), t(np.array(ctxs, dtype=np.int32)), max(ctxs), float(scale),
                                         softcapping=float(case.get("softcap", 1.0)), alibi_slopes=t(alibi), sinks=t(sinks), force=v).cpu().numpy()
        err = np.abs(got - ref[f"{name}_{v}"])
        assert np.isfinite(got).all() and (err <= 6e-5 * pabs + 2e-6).all(), (name, v, float(err.max()))
