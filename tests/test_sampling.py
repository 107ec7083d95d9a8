"""Top-k over a large vocabulary (csrc/sampling.hip, reference ABI topk_large_f32_packed[_batched]) against the restatement (oracle.topk_large_packed), and the
restatement against the reference's own kernels compiled for the host (oracle/_ref/libref_topk.so: sort.cu:1502-1823 on fibers).  Values / indices exactly;
the softmax normaliser to 2e-6 relative (expf differs in the last ulp between libm, CUDA and ROCm; the summation association is the reference's)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VP, I, F, LL = C.c_void_p, C.c_int, C.c_float, C.c_int64


def _logits(n, seed, kind):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * 3).astype(np.float32)
    if kind == "ties":  # few distinct values, +-0.0 mixed: order decided by the index rule
        x = rng.integers(-3, 4, n).astype(np.float32)
        x[x == 0] = np.where(rng.random((x == 0).sum()) < 0.5, np.float32(-0.0), np.float32(0.0))
    elif kind == "nan":
        x[rng.integers(0, n, max(1, n // 50))] = np.nan
        x[rng.integers(0, n, max(1, n // 50))] = -np.inf
        x[n // 2] = np.inf
    elif kind == "sparse":  # fewer finite values than k in most chunks
        m = rng.random(n) < 0.995
        x[m] = -np.inf
    return x


CASES = [(5000, 5, 1.3, 2048, "normal", 0), (128256, 40, 0.7, 2048, "normal", 1), (4100, 128, 1.0, 2048, "ties", 2), (3000, 7, 2.5, 2048, "nan", 3),
         (20000, 16, 1.0, 256, "normal", 4), (9000, 64, 0.9, 2048, "sparse", 5), (1, 1, 1.0, 2048, "normal", 6), (2048, 3, 1.0, 2048, "ties", 7),
         (2049, 2, 0.5, 2048, "normal", 8), (6000, 9, 1.0, 4096, "normal", 9), (3000, 30, 1.0, 128, "ties", 11)]
# many chunks x large k: the staged keys do not cover the lists (stage 2 reads the deeper candidates from memory).  Device only: the fiber runs of this shape take minutes
# (checked once on the host emulation and against the reference kernels: 313 chunks, k = 100)
GPU_ONLY_CASES = [(20000, 100, 1.0, 64, "normal", 10)]
IDS = [f"n{c[0]}k{c[1]}c{c[3]}{c[4]}" for c in CASES]


def _ref():
    p = os.path.join(ROOT, "oracle", "_ref", "libref_topk.so")
    if not os.path.exists(p):
        pytest.skip("libref_topk.so not built (needs /root/reference: make -C oracle ref)")
    return C.CDLL(p)


def _same(got, want, k, what):
    np.testing.assert_array_equal(got[:k].view(np.uint32), want[:k].view(np.uint32), err_msg=f"{what}: values")
    np.testing.assert_array_equal(got[k:2 * k], want[k:2 * k], err_msg=f"{what}: indices")
    if np.isnan(want[2 * k]):
        assert np.isnan(got[2 * k])
    else:
        np.testing.assert_allclose(got[2 * k], want[2 * k], rtol=2e-6, err_msg=f"{what}: denom")
    np.testing.assert_array_equal(got[2 * k + 1:], want[2 * k + 1:], err_msg=f"{what}: global max")


@pytest.mark.parametrize("n,k,temp,chunk,kind,seed", CASES, ids=IDS)
def test_restatement_matches_reference_kernels(oracle, n, k, temp, chunk, kind, seed):
    lib = _ref()
    x = _logits(n, seed, kind)
    k = min(k, n)
    nb = (n + chunk - 1) // chunk
    bv, bi = np.zeros(nb * k, np.float32), np.zeros(nb * k, np.uint32)
    bm, bs, packed = np.zeros(nb, np.float32), np.zeros(nb, np.float32), np.zeros(2 * k + 2, np.float32)
    vp = lambda a: a.ctypes.data_as(VP)
    assert lib.ref_topk_large_f32_packed(vp(x), vp(bv), vp(bi), vp(bm), vp(bs), vp(packed), n, k, chunk, nb, F(1.0 / temp)) == 0
    r = oracle.topk_large_packed(x, k, 1.0 / temp, chunk)
    _same(r["packed"], packed, k, "restatement vs reference")
    np.testing.assert_array_equal(r["block_values"].reshape(-1).view(np.uint32), bv.view(np.uint32))
    np.testing.assert_array_equal(r["block_indices"].reshape(-1), bi)
    np.testing.assert_array_equal(r["block_maxes"], bm)


def check_product(be, oracle, n, k, temp, chunk, kind, seed):
    x = _logits(n, seed, kind)
    k = min(k, n)
    nb = (n + chunk - 1) // chunk
    xb = be.buf(x)
    bv, bi, bm, bs = be.buf(np.zeros(nb * k, np.float32)), be.buf(np.zeros(nb * k, np.uint32)), be.buf(np.zeros(nb, np.float32)), be.buf(np.zeros(nb, np.float32))
    pk = be.buf(np.zeros(2 * k + 2, np.float32))
    be.sym("topk_large_f32_packed", [VP, VP, VP, VP, VP, VP, I, I, I, I, F, LL])(xb.ptr, bv.ptr, bi.ptr, bm.ptr, bs.ptr, pk.ptr, n, k, chunk, nb, 1.0 / temp, be.stream or 0)
    r = oracle.topk_large_packed(x, k, 1.0 / temp, chunk)
    _same(pk.numpy(), r["packed"], k, "product vs restatement")
    np.testing.assert_array_equal(bv.numpy().view(np.uint32), r["block_values"].reshape(-1).view(np.uint32))
    np.testing.assert_array_equal(bi.numpy(), r["block_indices"].reshape(-1))
    np.testing.assert_array_equal(bm.numpy(), r["block_maxes"])
    fin = ~np.isnan(r["block_sums"])
    np.testing.assert_allclose(bs.numpy()[fin], r["block_sums"][fin], rtol=2e-6)
    # the unpacked entry point writes the same numbers
    vo, io, so = be.buf(np.zeros(k, np.float32)), be.buf(np.zeros(k, np.uint32)), be.buf(np.zeros(2, np.float32))
    be.sym("topk_large_f32", [VP, VP, VP, VP, VP, VP, VP, VP, I, I, I, I, F, LL])(xb.ptr, bv.ptr, bi.ptr, bm.ptr, bs.ptr, vo.ptr, io.ptr, so.ptr, n, k, chunk, nb, 1.0 / temp,
                                                                                   be.stream or 0)
    np.testing.assert_array_equal(vo.numpy().view(np.uint32), pk.numpy()[:k].view(np.uint32))
    np.testing.assert_array_equal(io.numpy().astype(np.float32), pk.numpy()[k:2 * k])
    np.testing.assert_array_equal(so.numpy().view(np.uint32), pk.numpy()[2 * k:].view(np.uint32))


def check_batched(be, oracle, rows=3, n=7000, k=11):
    xs = np.stack([_logits(n, 20 + r, "normal" if r else "ties") for r in range(rows)])
    temps = np.array([0.6, 1.0, 1.9][:rows], dtype=np.float32)
    nb = (n + 2047) // 2048
    xb, tb = be.buf(xs), be.buf((1.0 / temps).astype(np.float32))
    bv, bi = be.buf(np.zeros(rows * nb * k, np.float32)), be.buf(np.zeros(rows * nb * k, np.uint32))
    bm, bs, pk = be.buf(np.zeros(rows * nb, np.float32)), be.buf(np.zeros(rows * nb, np.float32)), be.buf(np.zeros(rows * (2 * k + 2), np.float32))
    be.sym("topk_large_f32_packed_batched", [VP, VP, VP, VP, VP, VP, VP, I, I, I, I, I, LL])(xb.ptr, tb.ptr, bv.ptr, bi.ptr, bm.ptr, bs.ptr, pk.ptr, rows, n, k, 2048, nb,
                                                                                              be.stream or 0)
    got = pk.numpy().reshape(rows, 2 * k + 2)
    for r in range(rows):
        _same(got[r], oracle.topk_large_packed(xs[r], k, float(np.float32(1.0) / temps[r]), 2048)["packed"], k, f"batched row {r}")


@pytest.mark.parametrize("n,k,temp,chunk,kind,seed", [c for c in CASES if c[0] <= 20000], ids=[i for c, i in zip(CASES, IDS) if c[0] <= 20000])
def test_topk_host_emulation(oracle, n, k, temp, chunk, kind, seed):
    from tests.abi_backends import HostBackend
    check_product(HostBackend(), oracle, n, k, temp, chunk, kind, seed)


def test_topk_batched_host_emulation(oracle):
    from tests.abi_backends import HostBackend
    check_batched(HostBackend(), oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,temp,chunk,kind,seed", CASES + GPU_ONLY_CASES, ids=IDS + ["n20000k100c64normal"])
def test_topk_gpu(oracle, dev, n, k, temp, chunk, kind, seed):
    from tests.abi_backends import GpuBackend
    check_product(GpuBackend(dev), oracle, n, k, temp, chunk, kind, seed)


@pytest.mark.gpu
def test_topk_batched_gpu(oracle, dev):
    from tests.abi_backends import GpuBackend
    check_batched(GpuBackend(dev), oracle)


def test_host_half_top_p_min_p(oracle):
    """sampler.rs:1189-1236 restated twice (oracle.sample_topk_host, mistralrs_amd.sampler.filtered_probs): same weights; the cuts behave as the reference's tests expect
    (top-p keeps the shortest prefix whose mass reaches top_p * kept mass, min-p drops candidates below min_p * the best probability)."""
    from mistralrs_amd import sampler
    x = _logits(5000, 3, "normal")
    k, temp = 12, 0.8
    packed = oracle.topk_large_packed(x, k, 1.0 / temp)["packed"]
    for top_p, min_p in [(1.0, 0.0), (0.9, 0.0), (0.5, 0.05), (1.0, 0.3), (0.01, 0.0)]:
        ids, rep, probs = sampler.filtered_probs(packed, k, temp, top_p, min_p)
        ids2, rep2, probs2 = oracle.sample_topk_host(packed, k, temp, top_p, min_p)
        np.testing.assert_array_equal(ids, ids2)
        np.testing.assert_array_equal(rep, rep2)
        np.testing.assert_array_equal(probs, probs2)
        full = np.exp(x.astype(np.float64) / temp - (x.astype(np.float64) / temp).max())
        np.testing.assert_allclose(rep, (full / full.sum())[ids], rtol=1e-5)  # probabilities under the FULL softmax
        assert probs[0] > 0 and np.all((probs == 0) | (probs == rep))
        if 0 < top_p < 1:
            kept = np.nonzero(probs)[0]
            assert kept.size == kept.max() + 1 or min_p > 0  # a prefix
    rng = np.random.default_rng(0)
    tok, p = sampler.sample(packed, k, temp, 0.9, 0.0, rng)
    assert tok in packed[k:2 * k].astype(np.int64) and 0 < p <= 1


@pytest.mark.gpu
def test_sampler_class_gpu(oracle, dev):
    import torch
    from mistralrs_amd import sampler
    x = _logits(128256, 11, "normal")
    tk = sampler.TopK(128256, 50, dev, max_rows=2)
    got = tk(torch.from_numpy(x).to(dev), 0.7).cpu().numpy()[0]
    _same(got, oracle.topk_large_packed(x, 50, float(np.float32(1.0 / 0.7)))["packed"], 50, "TopK class")
    two = torch.from_numpy(np.stack([x, x[::-1].copy()])).to(dev)
    got2 = tk(two, [0.7, 1.5]).cpu().numpy()
    _same(got2[1], oracle.topk_large_packed(x[::-1].copy(), 50, float(np.float32(1.0) / np.float32(1.5)))["packed"], 50, "TopK class, batched row")
    with pytest.raises(ValueError):
        tk(torch.from_numpy(x).to(dev), 0.0)
    with pytest.raises(ValueError):
        sampler.TopK(128256, 129, dev)


@pytest.mark.gpu
def test_sampled_generation_on_the_runner(oracle, dev):
    """top_k = 1 reproduces the greedy ids of the runner; with top_k > 1 every drawn token is one of that step's k candidates, the reporting probabilities are
    probabilities, and a fixed seed reproduces the run."""
    import torch
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dec_model import _mk, Q4KM
    from mistralrs_amd import sampler
    cfg, w, m, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16")
    prompt = [(1000 + 7 * i) % cfg.vocab_size for i in range(12)]
    greedy = []
    lg = m.prefill(prompt, 0)
    for i in range(6):
        greedy.append(int(lg.argmax()))
        m.set_state([greedy[-1]], [len(prompt) + i])
        lg = m.forward_logits(1)[0]
    toks, probs = sampler.generate(m, prompt, 6, top_k=1)
    assert toks == greedy and all(p == 1.0 for p in probs)  # sample_cuda_top1_row: logprob 0
    a = sampler.generate(m, prompt, 8, top_k=20, temperature=1.5, top_p=0.9, min_p=0.02, seed=5)
    b = sampler.generate(m, prompt, 8, top_k=20, temperature=1.5, top_p=0.9, min_p=0.02, seed=5)
    assert a == b and all(0 < p <= 1 for p in a[1]) and all(0 <= t < cfg.vocab_size for t in a[0])


# ---------------------------------------------------------------- greedy: top1_large_f32_packed[_batched]
TOP1 = [(5000, 2048, "normal", 0), (128256, 2048, "normal", 1), (4100, 2048, "ties", 2), (3000, 2048, "nan", 3), (3000, 512, "sparse", 5), (1, 2048, "normal", 6),
        (2049, 2048, "ties", 7), (700, 100, "allinf", 8)]


def _top1_rows(n, kind, seed, rows=3):
    if kind == "allinf":
        return np.full((rows, n), -np.inf, dtype=np.float32)
    xs = np.stack([_logits(n, seed + 31 * r, kind if r != 1 else "normal") for r in range(rows)])
    if kind == "ties":
        xs[2, :] = np.float32(-0.0)
        xs[2, n // 3] = np.float32(0.0)  # equal to the -0.0 before it: the first index wins
    return xs


@pytest.mark.parametrize("n,chunk,kind,seed", TOP1, ids=[f"n{c[0]}c{c[1]}{c[2]}" for c in TOP1])
def test_top1_restatement_matches_reference_kernels(oracle, n, chunk, kind, seed):
    lib = _ref()
    xs = _top1_rows(n, kind, seed)
    rows, nb = xs.shape[0], (n + chunk - 1) // chunk
    bv, bi = np.zeros(rows * nb, np.float32), np.zeros(rows * nb, np.uint32)
    packed, toks = np.zeros(rows * 2, np.float32), np.zeros(rows, np.uint32)
    vp = lambda a: a.ctypes.data_as(VP)
    assert lib.ref_top1_large_f32_packed_batched(vp(xs), vp(bv), vp(bi), vp(packed), vp(toks), rows, n, chunk, nb) == 0
    for r in range(rows):
        p, t, v, i = oracle.top1_large_packed(xs[r], chunk)
        np.testing.assert_array_equal(p.view(np.uint32) if not np.isnan(p).any() else np.isnan(p), packed[2 * r:2 * r + 2].view(np.uint32) if not np.isnan(p).any() else np.isnan(packed[2 * r:2 * r + 2]))
        assert int(t) == int(toks[r])
        np.testing.assert_array_equal(np.isnan(v), np.isnan(bv[r * nb:(r + 1) * nb]))
        np.testing.assert_array_equal(v[~np.isnan(v)].view(np.uint32), bv[r * nb:(r + 1) * nb][~np.isnan(v)].view(np.uint32))
        np.testing.assert_array_equal(i, bi[r * nb:(r + 1) * nb])


def check_top1_product(be, oracle, n, chunk, kind, seed):
    xs = _top1_rows(n, kind, seed)
    rows, nb = xs.shape[0], (n + chunk - 1) // chunk
    xb = be.buf(xs)
    bv, bi = be.buf(np.zeros(rows * nb, np.float32)), be.buf(np.zeros(rows * nb, np.uint32))
    pk, tk = be.buf(np.zeros(rows * 2, np.float32)), be.buf(np.zeros(rows, np.uint32))
    be.sym("top1_large_f32_packed_batched", [VP, VP, VP, VP, VP, I, I, I, I, LL])(xb.ptr, bv.ptr, bi.ptr, pk.ptr, tk.ptr, rows, n, chunk, nb, be.stream or 0)
    got_p, got_t = pk.numpy().reshape(rows, 2), tk.numpy()
    for r in range(rows):
        p, t, v, i = oracle.top1_large_packed(xs[r], chunk)
        assert int(got_t[r]) & 0xFFFFFFFF == int(t), (r, got_t[r], t)  # the device buffer comes back as int32
        if np.isnan(p).any():
            assert np.isnan(got_p[r]).all()
        else:
            np.testing.assert_array_equal(got_p[r].view(np.uint32), p.view(np.uint32))
    # single-row entry point, token ids only
    t1 = be.buf(np.zeros(1, np.uint32))
    x0 = be.buf(xs[0].copy())
    be.sym("top1_large_f32_packed", [VP, VP, VP, VP, VP, I, I, I, LL])(x0.ptr, bv.ptr, bi.ptr, None, t1.ptr, n, chunk, nb, be.stream or 0)
    assert int(t1.numpy()[0]) & 0xFFFFFFFF == int(oracle.top1_large_packed(xs[0], chunk)[1])


@pytest.mark.parametrize("n,chunk,kind,seed", [c for c in TOP1 if c[0] <= 5000], ids=[f"n{c[0]}c{c[1]}{c[2]}" for c in TOP1 if c[0] <= 5000])
def test_top1_host_emulation(oracle, n, chunk, kind, seed):
    from tests.abi_backends import HostBackend
    check_top1_product(HostBackend(), oracle, n, chunk, kind, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("n,chunk,kind,seed", TOP1, ids=[f"n{c[0]}c{c[1]}{c[2]}" for c in TOP1])
def test_top1_gpu(oracle, dev, n, chunk, kind, seed):
    from tests.abi_backends import GpuBackend
    check_top1_product(GpuBackend(dev), oracle, n, chunk, kind, seed)


# ---------------------------------------------------------------- sampler pre-processing: apply_sparse_penalties_f32 / apply_sparse_logits_bias_f32 (sort.cu:8-110)
def check_penalties_and_bias(be, n=5003, n_tokens=300, seed=0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * 4).astype(np.float32)
    ids = rng.permutation(n + 50)[:n_tokens].astype(np.uint32)  # unique; some >= n (ignored)
    counts = rng.integers(0, 5, n_tokens).astype(np.float32)    # zeros: skipped
    biases = rng.standard_normal(n_tokens).astype(np.float32)
    xb, ib, cb, bb = be.buf(x), be.buf(ids), be.buf(counts), be.buf(biases)
    for fp, pp, rp in [(0.3, 0.1, 1.0), (0.0, 0.0, 1.3), (0.7, -0.2, 0.8)]:
        dst = be.buf(np.zeros(n, np.float32))
        be.sym("apply_sparse_penalties_f32", [VP, VP, VP, VP, I, I, F, F, F, LL])(xb.ptr, dst.ptr, ib.ptr, cb.ptr, n, n_tokens, fp, pp, rp, be.stream or 0)
        want = x.copy()
        for t, c in zip(ids, counts):
            if t < n and c > 0:
                v = np.float32(want[t] - np.float32(np.float64(c) * np.float64(np.float32(fp)) + np.float64(np.float32(pp))))  # count * f + p in one rounding (fma)
                if np.float32(rp) != np.float32(1.0):
                    v = np.float32(v / np.float32(rp)) if v > 0 else np.float32(v * np.float32(rp))
                want[t] = v
        np.testing.assert_array_equal(dst.numpy().view(np.uint32), want.view(np.uint32))
    dst = be.buf(np.zeros(n, np.float32))
    be.sym("apply_sparse_logits_bias_f32", [VP, VP, VP, VP, I, I, LL])(xb.ptr, dst.ptr, ib.ptr, bb.ptr, n, n_tokens, be.stream or 0)
    want = x.copy()
    for t, b in zip(ids, biases):
        if t < n:
            want[t] = np.float32(want[t] + b)
    np.testing.assert_array_equal(dst.numpy().view(np.uint32), want.view(np.uint32))
    # no listed tokens: a plain copy
    dst = be.buf(np.zeros(n, np.float32))
    be.sym("apply_sparse_logits_bias_f32", [VP, VP, VP, VP, I, I, LL])(xb.ptr, dst.ptr, ib.ptr, bb.ptr, n, 0, be.stream or 0)
    np.testing.assert_array_equal(dst.numpy().view(np.uint32), x.view(np.uint32))


def test_penalties_and_bias_host_emulation():
    from tests.abi_backends import HostBackend
    check_penalties_and_bias(HostBackend())


@pytest.mark.gpu
def test_penalties_and_bias_gpu(dev):
    from tests.abi_backends import GpuBackend
    check_penalties_and_bias(GpuBackend(dev), n=128256, n_tokens=4000, seed=1)
