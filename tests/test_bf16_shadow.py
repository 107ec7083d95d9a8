"""The selectable bf16 prompt path on a bf16 shadow copy of the weights (csrc/ext_gemm_lt.hip, round 6): plain bf16 x bf16 -> f32 library GEMMs (hipBLASLt) instead of
the fused block-dequant kernels.  Role in the reference: the prompt GEMMs of GgufMatMul::forward / fast_mmq (mistralrs-quant/src/gguf/fast_mmq.rs:528-635); north_star:
"MFMA only on the bf16 prefill GEMM".  Same arithmetic as the fused kernels (weights rounded to bf16 once, activations to bf16, f32 accumulation), library summation order:
tolerances, not bit identity -- the bit-exact prompt path is the default one (tests/test_prefill_exact.py)."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.gpu
def test_library_gemm_on_bf16_rows_equals_f32_matmul_of_the_bf16_values(dev, request):
    if request.config.getoption("--host-emulation"):
        pytest.skip("hipBLASLt needs the device")
    import torch
    from mistralrs_amd import _lib
    L = _lib.load("ext")
    L.mrs_lt_gemm_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mrs_rows_f32_to_bf16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(3)
    for N, K, T, ldo in ((4096, 4096, 512, 4096), (1024, 4096, 300, 1024), (4096, 14336, 129, 4096), (512, 512, 40, 640)):
        w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
        x = torch.randn(T, K, generator=g).to(dev)
        xb = torch.empty(T, K, dtype=torch.bfloat16, device=dev)
        assert L.mrs_rows_f32_to_bf16(x.data_ptr(), K, T, K, xb.data_ptr(), st) == 0
        assert torch.equal(xb, x.to(torch.bfloat16)), "row conversion is round-to-nearest-even"
        base = torch.randn(T, ldo, generator=g).to(dev)
        for acc in (0, 1):
            out = base.clone()
            assert L.mrs_lt_gemm_bf16(w.data_ptr(), xb.data_ptr(), out.data_ptr(), ldo, N, K, T, acc, st) == 0
            torch.cuda.synchronize()
            want = xb.float() @ w.float().t() + (base[:, :N] if acc else 0)
            err = float((out[:, :N] - want).abs().max() / want.abs().max())
            assert err <= 2e-5, (N, K, T, acc, err)
            if ldo > N:
                assert torch.equal(out[:, N:], base[:, N:]), "columns past N must stay untouched"


@pytest.mark.gpu
def test_bf16_shadow_prefill_matches_the_fused_dequant_prefill_and_the_exact_one(oracle, dev, request):
    if request.config.getoption("--host-emulation"):
        pytest.skip("hipBLASLt needs the device")
    from tests.test_dec_model import Q4KM, _mk
    cfg, w, m, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16")
    assert m.bf16_shadow, "a small model takes the shadow copy by default (MRS_PREFILL_BF16_SHADOW=auto)"
    prompt = [(1000 + 37 * i) % cfg.vocab_size for i in range(70)]
    m.set_prefill_mode(1)
    exact = m.prefill(prompt, 0).float().cpu().numpy()
    m.set_prefill_mode(2)
    fused = m.prefill(prompt, 0).float().cpu().numpy()
    m.set_prefill_mode(0)
    shadow = m.prefill(prompt, 0).float().cpu().numpy()
    scale = float(np.abs(exact).max())
    assert float(np.abs(shadow - fused).max()) <= 5e-3 * scale, float(np.abs(shadow - fused).max()) / scale  # same bf16 arithmetic, different f32 summation order
    assert float(np.abs(shadow - exact).max()) <= 6e-2 * scale, float(np.abs(shadow - exact).max()) / scale  # bf16 operands vs the int8 reference arithmetic
    assert int(shadow.argmax()) == int(fused.argmax())


@pytest.mark.gpu
def test_shadow_built_after_loading_and_dropped(oracle, dev, request, monkeypatch):
    """Llama.build_bf16_shadow() on a model loaded without the copy (what bench.py's 70B leg does once everything else is recorded) gives the logits of a model loaded with
    it; drop_bf16_shadow() returns the bf16 path to the fused block-dequant kernels."""
    if request.config.getoption("--host-emulation"):
        pytest.skip("hipBLASLt needs the device")
    from tests.test_dec_model import Q4KM, _mk
    cfg, w, m_auto, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16")
    monkeypatch.setenv("MRS_PREFILL_BF16_SHADOW", "0")
    _, _, m, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16")
    assert m_auto.bf16_shadow and not m.bf16_shadow
    prompt = [(1000 + 37 * i) % cfg.vocab_size for i in range(40)]
    m.set_prefill_mode(0)
    m_auto.set_prefill_mode(0)
    fused = m.prefill(prompt, 0).float().cpu().numpy()  # no copy: mode 0 = the fused kernels
    assert m.build_bf16_shadow()
    got = m.prefill(prompt, 0).float().cpu().numpy()
    want = m_auto.prefill(prompt, 0).float().cpu().numpy()
    assert np.array_equal(got, want), "shadow built after loading != shadow built while loading"
    m.drop_bf16_shadow()
    assert not m.bf16_shadow
    assert np.array_equal(m.prefill(prompt, 0).float().cpu().numpy(), fused)
