"""GPU parity: the prefill GEMM (fused block-dequant -> bf16 MFMA, f32 accumulate) vs oracle A.

Expected value = sum_k bf16(x) * bf16(dequant(W)) evaluated in f64 (the kernel's defined arithmetic: both operands are
rounded to bf16 once, products and sums are f32 on the matrix cores).  Tolerance: f32 accumulation of K terms in the MFMA's
order, 2^-19 * sum|terms| (>= 8 eps sqrt(K) for K <= 16384); the distance to the exact-dequant f32 matmul is checked against
the bf16 input-rounding bound 2^-8 * sum|terms| (what "MFMA on the bf16 prefill GEMM" costs in accuracy).
"""
import numpy as np
import pytest

from tests.util import round_through

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tname", ["Q4_K", "Q5_K", "Q6_K", "Q8_0"])
@pytest.mark.parametrize("m,n,k", [(512, 256, 1024), (130, 384, 512), (1, 128, 256), (37, 200, 2048)])
def test_prefill_gemm_vs_oracle(oracle, dev, tname, m, n, k):
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_gemm
    t = getattr(oracle, tname)
    rng = np.random.default_rng(m * 7 + n + k)
    w = oracle.random_blocks(t, n, k, seed=n + k, d_scale=0.02)
    x = (rng.standard_normal((m, k)) * rng.uniform(0.2, 3.0, (m, 1))).astype(np.float32)
    wt = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), w, dev)
    got = fast_gemm.plain(wt, torch.from_numpy(x).to(dev)).cpu().numpy().astype(np.float64)
    wd = oracle.dequantize(t, w, k)
    xb, wb = round_through(x, "bf16").astype(np.float64), round_through(wd, "bf16").astype(np.float64)
    want = xb @ wb.T
    mag = np.abs(xb) @ np.abs(wb).T
    err = np.abs(got - want)
    assert (err <= 2.0 ** -19 * mag + 1e-30).all(), float((err / (2.0 ** -19 * mag + 1e-30)).max())
    exact = x.astype(np.float64) @ wd.astype(np.float64).T
    assert (np.abs(got - exact) <= 2.0 ** -8 * mag + 1e-30).all()


def test_prefill_gemm_accumulate_and_errors(oracle, dev):
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_gemm
    t = oracle.Q4_K
    w = oracle.random_blocks(t, 128, 512, seed=3, d_scale=0.02)
    wt = QTensor.from_numpy(GgmlDType.from_id(t), (128, 512), w, dev)
    x = torch.randn(40, 512, device=dev)
    base = torch.randn(40, 128, device=dev)
    o1 = fast_gemm.plain(wt, x)
    o2 = fast_gemm.plain(wt, x, out=base.clone(), accumulate=True)
    assert torch.equal(o2, base + o1)
    with pytest.raises(ValueError, match="shape mismatch"):
        fast_gemm.plain(wt, torch.randn(4, 256, device=dev))
    with pytest.raises(ValueError, match="unsupported quant dtype"):
        fast_gemm.plain(QTensor.from_numpy(GgmlDType.Q4_0, (32, 64), oracle.random_blocks(oracle.Q4_0, 32, 64), dev), torch.randn(4, 64, device=dev))


@pytest.mark.parametrize("tname", ["Q4_K", "Q5_K", "Q6_K", "Q8_0"])
@pytest.mark.parametrize("m,n,k,split", [(512, 256, 1024, True), (300, 384, 2048, True), (256, 128, 512, False), (130, 200, 4096, True), (1, 128, 256, True)])
def test_prefill_gemm_large_m_vs_oracle(oracle, dev, tname, m, n, k, split):
    """256-row-tile kernel (bf16 activations, ping-pong waves, split-K partials) against the same oracle and bounds."""
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_gemm
    t = getattr(oracle, tname)
    rng = np.random.default_rng(m * 7 + n + k)
    w = oracle.random_blocks(t, n, k, seed=n + k, d_scale=0.02)
    x = (rng.standard_normal((m, k)) * rng.uniform(0.2, 3.0, (m, 1))).astype(np.float32)
    wt = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), w, dev)
    xt = torch.from_numpy(x).to(dev)
    got = fast_gemm.plain_bf16(wt, xt, split_k=split).cpu().numpy().astype(np.float64)
    wd = oracle.dequantize(t, w, k)
    xb, wb = round_through(x, "bf16").astype(np.float64), round_through(wd, "bf16").astype(np.float64)
    want = xb @ wb.T
    mag = np.abs(xb) @ np.abs(wb).T
    err = np.abs(got - want)
    assert (err <= 2.0 ** -19 * mag + 1e-30).all(), float((err / (2.0 ** -19 * mag + 1e-30)).max())
    base = torch.randn(m, n, device=dev)
    acc = fast_gemm.plain_bf16(wt, fast_gemm.to_slabs(xt), out=base.clone(), accumulate=True, split_k=split)
    assert torch.equal(acc, base + torch.from_numpy(got.astype(np.float32)).to(dev))
    assert torch.equal(fast_gemm.plain_bf16(wt, xt, split_k=split), torch.from_numpy(got.astype(np.float32)).to(dev))  # deterministic


@pytest.mark.gpu
@pytest.mark.parametrize("tname", ["Q4_K", "Q5_K", "Q6_K", "Q8_0"])
@pytest.mark.parametrize("m,n,k,split", [(512, 384, 1024, False), (300, 200, 2048, True), (257, 128, 512, False)])
def test_wave_specialised_gemm_equals_ping_pong_kernel(oracle, dev, tname, m, n, k, split):
    """gemm_qc_kernel (producer / consumer waves, the default) and gemm_qb_kernel feed the same MFMAs the same operands in the same k
    order: identical bits (ragged M / N tiles, split-K)."""
    import ctypes as C
    import torch
    from mistralrs_amd import _lib
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_gemm
    t = getattr(oracle, tname)
    rng = np.random.default_rng(m + n + k)
    w = oracle.random_blocks(t, n, k, seed=3 + n + k, d_scale=0.02)
    wt = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), w, dev)
    xt = torch.from_numpy((rng.standard_normal((m, k)) * rng.uniform(0.2, 3.0, (m, 1))).astype(np.float32)).to(dev)
    setv = _lib.sym("ext", "mrs_gemm_set_variant", [C.c_int], None)
    try:
        setv(0)
        old = fast_gemm.plain_bf16(wt, xt, split_k=split).clone()
        setv(1)
        new = fast_gemm.plain_bf16(wt, xt, split_k=split)
    finally:
        setv(-1)
    assert torch.equal(old, new)


@pytest.mark.parametrize("tname", ["Q4_K", "Q6_K", "Q8_0"])
@pytest.mark.parametrize("E,tokens,topk,n,k", [(4, 70, 2, 200, 512), (3, 300, 2, 128, 1024), (8, 9, 3, 130, 256)])
def test_moe_grouped_gemm_on_matrix_cores(oracle, dev, tname, E, tokens, topk, n, k):
    """mrs_moe_gemm_q_bf16 (grouped MoE GEMM of a prompt, dispatch tables of launch_moe_dispatch): per expert bit-identical to the dense bf16 MFMA
    GEMM on that expert's gathered rows; the down form (rows in sorted order, routing weights, f32 atomics into a zeroed buffer) == the sum of the
    weighted per-route results (top-k 2: order-free; top-k 3: f32 order tolerance).  Includes an expert without routes and ragged row tiles."""
    import ctypes as C
    import torch
    from mistralrs_amd import _lib
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_gemm
    t = getattr(oracle, tname)
    rng = np.random.default_rng(E + tokens + n)
    w = np.concatenate([oracle.random_blocks(t, n, k, seed=11 + e + n, d_scale=0.02) for e in range(E)], axis=0)
    wt = QTensor.from_numpy(GgmlDType.from_id(t), (E * n, k), w, dev)
    ids = np.stack([rng.permutation(E - 1)[:topk] if E > topk else rng.permutation(E)[:topk] for _ in range(tokens)]).astype(np.int32)  # expert E - 1 stays empty when E > topk
    routes = tokens * topk
    idt = torch.from_numpy(ids.reshape(-1)).to(dev)
    bounds, sorted_ids = torch.zeros(E + 1, dtype=torch.int32, device=dev), torch.zeros(routes, dtype=torch.int32, device=dev)
    counts, cursors = torch.zeros(E, dtype=torch.int32, device=dev), torch.zeros(E, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.sym("quant", "launch_moe_dispatch", [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 3)(idt.data_ptr(), bounds.data_ptr(), sorted_ids.data_ptr(), None, routes, E, topk,
                                                                                                   counts.data_ptr(), cursors.data_ptr(), st)
    x = torch.from_numpy((rng.standard_normal((tokens, k)) * rng.uniform(0.2, 3.0, (tokens, 1))).astype(np.float32)).to(dev)
    xs = fast_gemm.to_slabs(x)
    out = torch.full((routes, n), float("nan"), device=dev)
    fast_gemm.moe_grouped_bf16(wt, E, xs, bounds, sorted_ids, topk, True, out)
    b, so = bounds.cpu().numpy(), sorted_ids.cpu().numpy()
    assert b[-1] == routes
    per_route = torch.empty(routes, n, device=dev)  # by flat route index
    for e in range(E):
        pos = np.arange(b[e], b[e + 1])
        if len(pos) == 0:
            continue
        we = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), w[e * n:(e + 1) * n], dev)
        rows = torch.from_numpy(so[pos] // topk).to(dev)
        want = fast_gemm.plain_bf16(we, x[rows].contiguous(), split_k=False)
        assert torch.equal(out[torch.from_numpy(pos).to(dev)], want), e
        per_route[torch.from_numpy(so[pos]).to(dev).long()] = want
    # down form: the routes' activations in sorted order, scaled by the routing weights, summed per token
    rw = torch.from_numpy(rng.uniform(0.1, 0.9, routes).astype(np.float32)).to(dev)
    xr = x[torch.from_numpy(so // topk).to(dev).long()].contiguous()  # row pos = the token of sorted position pos
    acc = torch.zeros(tokens, n, device=dev)
    fast_gemm.moe_grouped_bf16(wt, E, fast_gemm.to_slabs(xr), bounds, sorted_ids, topk, False, acc, route_w=rw)
    ref = (per_route * rw[:, None]).reshape(tokens, topk, n)
    if topk == 2:
        assert torch.equal(acc, ref[:, 0] + ref[:, 1])
    else:
        assert float((acc - ref.sum(1)).abs().max()) <= 1e-5 * float(ref.abs().sum(1).max())


@pytest.mark.parametrize("tname", ["Q4_K", "Q5_K", "Q6_K", "Q8_0"])
@pytest.mark.parametrize("m,n,k,act", [(300, 192, 512, 0), (70, 64, 1024, 1), (512, 448, 256, 0)])
def test_fused_gate_up_glu_gemm(oracle, dev, tname, m, n, k, act):
    """mrs_gemm_q_bf16_glu (gate and up rows interleaved in one B tile, act(gate) * up -> bf16 slabs in the epilogue) == the two GEMMs followed by the
    GLU-to-slabs kernel, bit for bit, through both GEMM kernels."""
    import ctypes as C
    import torch
    from mistralrs_amd import _lib
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_gemm
    t = getattr(oracle, tname)
    rng = np.random.default_rng(m + n + k)
    wg = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), oracle.random_blocks(t, n, k, seed=1 + n, d_scale=0.02), dev)
    wu = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), oracle.random_blocks(t, n, k, seed=2 + n, d_scale=0.02), dev)
    xs = fast_gemm.to_slabs(torch.from_numpy((rng.standard_normal((m, k)) * rng.uniform(0.2, 3.0, (m, 1))).astype(np.float32)).to(dev))
    want = fast_gemm.glu_slabs(fast_gemm.plain_bf16(wg, xs, split_k=False), fast_gemm.plain_bf16(wu, xs, split_k=False), act)
    setv = _lib.sym("ext", "mrs_gemm_set_variant", [C.c_int], None)
    try:
        for v in (0, 1):
            setv(v)
            got = fast_gemm.fused_glu_bf16(wg, wu, xs, act)
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), v
    finally:
        setv(-1)


def test_slab_producers(dev):
    """to_slabs is the layout [K/64][M][64] of the bf16-rounded matrix; the fused producers equal the unfused op + to_slabs bit for bit."""
    import torch
    from mistralrs_amd import ops
    from mistralrs_amd.gguf import fast_gemm
    torch.manual_seed(5)
    x = torch.randn(77, 320, device=dev) * 3.0
    want = x.to(torch.bfloat16).reshape(77, 5, 64).permute(1, 0, 2).contiguous()
    assert torch.equal(fast_gemm.to_slabs(x), want)
    wide = torch.randn(77, 400, device=dev)
    assert torch.equal(fast_gemm.to_slabs(wide[:, 16:336]), wide[:, 16:336].to(torch.bfloat16).reshape(77, 5, 64).permute(1, 0, 2).contiguous())
    w = torch.rand(320, device=dev) + 0.5
    assert torch.equal(fast_gemm.rms_norm_slabs(x, w, 1e-5), fast_gemm.to_slabs(ops.rms_norm(x, w, 1e-5)))
    g, u = torch.randn(77, 320, device=dev) * 4.0, torch.randn(77, 320, device=dev)
    for act in (0, 1, 2):
        assert torch.equal(fast_gemm.glu_slabs(g, u, act), fast_gemm.to_slabs(ops.fused_glu(g, u, act)))
    with pytest.raises(ValueError, match="multiple of 64"):
        fast_gemm.to_slabs(torch.randn(4, 96, device=dev))
    with pytest.raises(ValueError, match="multiple of 64"):
        fast_gemm.glu_slabs(torch.randn(4, 96, device=dev), torch.randn(4, 96, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("tname", ["Q4_K", "Q6_K"])
@pytest.mark.parametrize("variant", [0, 1])  # 0 = gemm_qb_kernel (ping-pong waves), 1 = gemm_qc_kernel (producer / consumer waves, the runner's default)
@pytest.mark.parametrize("m,n,k", [(512, 28672, 4096), (2048, 28672, 4096), (512, 4096, 14336), (2048, 4096, 14336), (512, 6144, 4096), (2048, 6144, 4096)])
def test_prefill_gemm_at_bench_shapes(oracle, dev, tname, variant, m, n, k):
    """The shapes bench.py's prompt runs (Llama-3-8B: gate+up 28672 x 4096, down 4096 x 14336, qkv 6144 x 4096 at M = 512 / 2048), both kernels, with
    and without split-K (the runner passes a workspace, so the launcher splits wherever there are fewer tiles than CUs): vs oracle A on 384 sampled
    weight rows (the f64 reference over all 28672 rows would take minutes), same 2^-19 sum|terms| bound; the unsampled rows are covered by the
    bit-equality of the two kernels and of split / unsplit accumulation order being deterministic (second call identical)."""
    import ctypes as C
    import torch
    from mistralrs_amd import _lib
    from mistralrs_amd.gguf import GgmlDType, QTensor, fast_gemm
    t = getattr(oracle, tname)
    rng = np.random.default_rng(m + n + k)
    w = oracle.random_blocks(t, n, k, seed=n + k, d_scale=0.02)
    x = (rng.standard_normal((m, k)) * rng.uniform(0.2, 3.0, (m, 1))).astype(np.float32)
    wt = QTensor.from_numpy(GgmlDType.from_id(t), (n, k), w, dev)
    xt = torch.from_numpy(x).to(dev)
    rows = np.sort(rng.choice(n, size=384, replace=False))
    wd = oracle.dequantize(t, w[rows], k)
    xb, wb = round_through(x, "bf16").astype(np.float64), round_through(wd, "bf16").astype(np.float64)
    want = xb @ wb.T
    mag = np.abs(xb) @ np.abs(wb).T
    setv = _lib.sym("ext", "mrs_gemm_set_variant", [C.c_int], None)
    try:
        setv(variant)
        for split in (True, False):
            got_t = fast_gemm.plain_bf16(wt, xt, split_k=split)
            got = got_t[:, torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.float64)
            err = np.abs(got - want)
            assert (err <= 2.0 ** -19 * mag + 1e-30).all(), (split, float((err / (2.0 ** -19 * mag + 1e-30)).max()))
            assert torch.equal(fast_gemm.plain_bf16(wt, xt, split_k=split), got_t)  # deterministic
    finally:
        setv(-1)

