"""CPU: pin the oracle against the REFERENCE'S OWN arithmetic (oracle/_ref, built by oracle/build_ref.sh from the
device functions of /root/reference/mistralrs-quant/kernels -- compiled for the host, never copied):

  * block decode of all 10 MMVQ formats == kernels/gguf_affine_packed/marlin_gguf_affine_repack.cu
    get_quant / get_affine_params (w = scale*q - offset), bit for bit;
  * the Q8_1 matvec oracle == SUM of the reference's vec_dot_<type>_q8_1 terms (mmvq_gguf.cu:388-685), to f32
    rounding of the individual terms (the reference adds f32 terms; both sides combine them in f64);
  * GLU activations == apply_glu_activation (mmvq_gguf.cu:44-85).
These libraries travel to the GPU box as prebuilt files; when absent (fresh clone without /root/reference) the tests skip.
"""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def _ref(name):
    p = os.path.join(REF_DIR, name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not built (needs /root/reference: make -C oracle ref)")
    return C.CDLL(p)


TYPES = [2, 3, 6, 7, 8, 10, 11, 12, 13, 14]


@pytest.mark.parametrize("t", TYPES)
def test_block_decode_matches_reference_format_spec(oracle, t):
    lib = _ref("libref_affine.so")
    n, k = 5, 768
    blocks = oracle.random_blocks(t, n, k, seed=100 + t, d_scale=0.02)
    want = np.empty((n, k), dtype=np.float32)
    assert lib.ref_dequantize(t, blocks.ctypes.data_as(C.c_void_p), n, k, want.ctypes.data_as(C.c_void_p)) == 0
    got = oracle.dequantize(t, blocks, k)
    # reference computes scale*q - offset in f32; the oracle's decode must agree to the last bit (only the sign of an
    # exact zero may differ: the affine form yields +0 where d*0 yields -0 for a negative scale)
    np.testing.assert_array_equal(got, want)
    nz = want != 0
    np.testing.assert_array_equal(got[nz].view(np.uint32), want[nz].view(np.uint32))


@pytest.mark.parametrize("t", TYPES)
def test_q8_1_matvec_matches_reference_vec_dot(oracle, t):
    lib = _ref("libref_mmvq.so")
    rng = np.random.default_rng(t)
    n, k, b = 7, 1024, 3
    w = oracle.random_blocks(t, n, k, seed=200 + t, d_scale=0.02)
    x = (rng.standard_normal((b, k)) * np.array([[0.5], [2.0], [30.0]])).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    want = np.empty((b, n), dtype=np.float64)
    assert lib.ref_mmvq(t, w.ctypes.data_as(C.c_void_p), n, k, y.ctypes.data_as(C.c_void_p), y.shape[1] // 36, b,
                        want.ctypes.data_as(C.c_void_p)) == 0
    got, mag = oracle.matmul_q8_1_mag(t, w, n, k, y)
    # each reference term is an f32 expression of a few products: relative 2^-22 of the term magnitudes (+ f32 output rounding)
    tol = 2.0 ** -21 * mag.astype(np.float64) + 2.0 ** -23 * np.abs(want) + 1e-30
    assert (np.abs(got - want) <= tol).all(), float(np.max(np.abs(got - want) / tol))


def test_glu_activations_match_reference(oracle):
    lib = _ref("libref_mmvq.so")
    lib.ref_glu_activation.restype = C.c_float
    lib.ref_glu_activation.argtypes = [C.c_float, C.c_int]
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-12, 12, 481), [0.0, -0.0, 1e-6, 30.0, -30.0]]).astype(np.float32)
    for act in range(5):
        for v in xs:
            a, r = L.orc_glu_act(float(v), act), lib.ref_glu_activation(float(v), act)
            assert abs(a - r) <= 2e-6 * max(1.0, abs(r)), (act, float(v), a, r)


# ---------------------------------------------------------------------------------------------------------------------
# RoPE and paged-cache data movement: the reference's own __global__ kernels (kernels/rotary/rotary.cu,
# mistralrs-paged-attn/src/cuda/{reshape_and_cache,gather_kv_cache,copy_blocks}_kernel.cu, f32 instantiations) run on the host,
# one thread at a time (oracle/ref_shim/cache_driver.inc -> libref_cache.so)
def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("rot_pairs,hd", [(32, 64), (16, 64), (64, 128)])
def test_rope_matches_reference_kernel(oracle, neox, rot_pairs, hd):
    lib = _ref("libref_cache.so")
    rng = np.random.default_rng(rot_pairs + hd + int(neox))
    T, H, KVH, max_pos = 5, 3, 2, 40
    q = rng.standard_normal((T, H, hd)).astype(np.float32)
    k = rng.standard_normal((T, KVH, hd)).astype(np.float32)
    ang = rng.uniform(0, 6.28, (max_pos, rot_pairs))
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    pos = rng.integers(0, max_pos, T).astype(np.uint32)
    want_q, want_k = q.copy(), k.copy()
    lib.ref_rotary_f32(_vp(want_q), _vp(want_k), _vp(cos), _vp(sin), _vp(pos), int(neox), hd, T, rot_pairs, H, KVH, C.c_long(H * hd), C.c_long(KVH * hd))
    np.testing.assert_array_equal(oracle.rope(q, cos, sin, pos.astype(np.int32), neox).view(np.uint32), want_q.view(np.uint32))
    np.testing.assert_array_equal(oracle.rope(k, cos, sin, pos.astype(np.int32), neox).view(np.uint32), want_k.view(np.uint32))
    # the positions-free entry takes cos/sin rows per token
    want2 = q.copy()
    k2 = k.copy()
    lib.ref_rotary_f32(_vp(want2), _vp(k2), _vp(np.ascontiguousarray(cos[pos])), _vp(np.ascontiguousarray(sin[pos])), None, int(neox), hd, T, rot_pairs, H, KVH,
                       C.c_long(H * hd), C.c_long(KVH * hd))
    np.testing.assert_array_equal(want2.view(np.uint32), want_q.view(np.uint32))


@pytest.mark.parametrize("bs,x", [(16, 4), (32, 4), (8, 8)])
def test_cache_ops_match_reference_kernels(oracle, bs, x):
    lib = _ref("libref_cache.so")
    rng = np.random.default_rng(bs + x)
    kvh, hd, nb, T = 3, 32, 7, 19
    key = rng.standard_normal((T, kvh, hd)).astype(np.float32)
    val = rng.standard_normal((T, kvh, hd)).astype(np.float32)
    slots = rng.permutation(nb * bs)[:T].astype(np.int64)
    slots[4] = -1
    kc_ref = rng.standard_normal((nb, kvh, hd // x, bs, x)).astype(np.float32)
    vc_ref = rng.standard_normal((nb, kvh, hd, bs)).astype(np.float32)
    kc, vc = kc_ref.copy(), vc_ref.copy()
    lib.ref_reshape_and_cache_f32(_vp(key), _vp(val), _vp(kc_ref), _vp(vc_ref), _vp(slots), T, kvh, hd, bs, x, kvh * hd, kvh * hd)
    oracle.kv_cache_write(kc, vc, key, val, slots)
    np.testing.assert_array_equal(kc, kc_ref)
    np.testing.assert_array_equal(vc, vc_ref)
    # gather two sequences of different lengths through their block tables
    lens = [bs + 3, 2 * bs]
    tables = np.stack([rng.permutation(nb)[:3], rng.permutation(nb)[:3]]).astype(np.int32)
    cu = np.array([0, lens[0], lens[0] + lens[1]], dtype=np.int32)
    n = int(cu[-1])
    k_out, v_out = np.zeros((n, kvh, hd), np.float32), np.zeros((n, kvh, hd), np.float32)
    lib.ref_gather_kv_cache_f32(_vp(kc), _vp(vc), _vp(k_out), _vp(v_out), _vp(tables), _vp(cu), n, 2, bs, 3, kvh, hd, x)
    for s in range(2):
        kg, vg = oracle.kv_cache_gather(kc, vc, tables[s], lens[s])
        np.testing.assert_array_equal(kg, k_out[cu[s]:cu[s + 1]])
        np.testing.assert_array_equal(vg, v_out[cu[s]:cu[s + 1]])
    # copy_blocks: per-layer block copies src -> dst
    layers_k = [rng.standard_normal((nb, 24)).astype(np.float32) for _ in range(2)]
    layers_v = [rng.standard_normal((nb, 40)).astype(np.float32) for _ in range(2)]
    want_k, want_v = [a.copy() for a in layers_k], [a.copy() for a in layers_v]
    mapping = np.array([0, 5, 2, 6, 0, 3], dtype=np.int64)
    for a in want_k + want_v:
        for s_, d_ in mapping.reshape(-1, 2):
            a[d_] = a[s_]
    kp = np.array([a.ctypes.data for a in layers_k], dtype=np.int64)
    vp = np.array([a.ctypes.data for a in layers_v], dtype=np.int64)
    lib.ref_copy_blocks_f32(_vp(kp), _vp(vp), _vp(mapping), 2, 3, 24, 40)
    for got, want in zip(layers_k + layers_v, want_k + want_v):
        np.testing.assert_array_equal(got, want)


# ---------------------------------------------------------------------------------------------------------------------
# Paged attention: the reference's own paged_attention_v1 / v2 / v2_reduce kernels (pagedattention.cuh:56-667, f32 path), executed on
# host fibers with real block barriers and warp shuffles (oracle/ref_shim/fiber_shim.h -> libref_pa.so).  The oracle evaluates the same
# semantics in f64; the kernels in f32 (exp evaluated exactly on the host), so the comparison is to f32 rounding of the sums.
@pytest.mark.parametrize("hd,bs,heads,kvh", [(128, 32, 8, 2), (64, 16, 4, 4), (96, 8, 6, 3)])
@pytest.mark.parametrize("variant", ["plain", "softcap", "alibi", "sinks"])
@pytest.mark.parametrize("v2", [0, 1])
def test_paged_attention_oracle_matches_reference_kernels(oracle, hd, bs, heads, kvh, variant, v2):
    lib = _ref("libref_pa.so")
    rng = np.random.default_rng(hd + bs + 7 * v2 + len(variant))
    ctxs = [1, bs - 1, bs, 2 * bs + 3, 530] if v2 else [1, bs - 1, bs + 1, 77]
    seqs = len(ctxs)
    x = 4  # 16 bytes of f32
    max_blocks = (max(ctxs) + bs - 1) // bs + 1
    nb = seqs * max_blocks + 1
    kc = rng.standard_normal((nb, kvh, hd // x, bs, x)).astype(np.float32)
    vc = rng.standard_normal((nb, kvh, hd, bs)).astype(np.float32)
    bt = rng.permutation(nb)[: seqs * max_blocks].reshape(seqs, max_blocks).astype(np.uint32)
    for s, c in enumerate(ctxs):  # stale slots past the context may hold NaN: the kernels zero them (pagedattention.cuh:407-423)
        if c % bs:
            vc[bt[s, c // bs], :, :, c % bs:] = np.nan
    q = (rng.standard_normal((seqs, heads, hd)) * 1.5).astype(np.float32)
    scale = np.float32(1.0 / np.sqrt(hd))
    softcap = np.float32(8.0 if variant == "softcap" else 1.0)
    alibi = (-rng.uniform(0.01, 0.3, heads)).astype(np.float32) if variant == "alibi" else None
    sinks = rng.standard_normal(heads).astype(np.float32) if variant == "sinks" else None
    max_ctx = max(ctxs)
    parts = (max_ctx + 511) // 512
    out = np.zeros((seqs, heads, hd), np.float32)
    es, ml = np.zeros((seqs, heads, parts), np.float32), np.zeros((seqs, heads, parts), np.float32)
    tmp = np.zeros((seqs, heads, parts, hd), np.float32)
    cl = np.array(ctxs, dtype=np.uint32)
    rc = lib.ref_paged_attention_f32(v2, _vp(out), _vp(es), _vp(ml), _vp(tmp), _vp(q), _vp(kc), _vp(vc), kvh, C.c_float(scale), C.c_float(softcap),
                                     _vp(bt), _vp(cl), bs, max_ctx, seqs, heads, hd, max_blocks, _vp(alibi) if alibi is not None else None,
                                     heads * hd, kvh * hd * bs, hd * bs, _vp(sinks) if sinks is not None else None)
    assert rc == 0
    vz = np.nan_to_num(vc, nan=0.0)
    want = oracle.paged_attention_ref(q, kc, vz, bt.astype(np.int32), ctxs, float(scale), float(softcap), alibi, sinks)
    pabs = oracle.paged_attention_ref(q, kc, np.abs(vz), bt.astype(np.int32), ctxs, float(scale), float(softcap), alibi, sinks)
    assert np.isfinite(out).all()
    err = np.abs(out - want.reshape(out.shape))
    tol = 3e-5 * pabs.reshape(out.shape) + 1e-6
    assert (err <= tol).all(), float((err / tol).max())


@pytest.mark.parametrize("k", [32, 96, 512, 4096, 520])
def test_q8_1_quantizer_matches_reference_kernel(oracle, k):
    """The reference's mmvq_gguf_quantize_q8_1_f32 kernel with its real warp reductions (butterfly order), run on host fibers: block
    scale d = half(amax / 127), quants roundf(x / d), and the half block sum, bit for bit; rows padded to 512 with zero blocks."""
    lib = _ref("libref_q8_1.so")
    rng = np.random.default_rng(k)
    rows = 3
    x = (rng.standard_normal((rows, k)) * rng.uniform(0.01, 30.0, (rows, 1))).astype(np.float32)
    x[0, :32] = 0.0            # an all-zero block (amax == 0 branch)
    if k >= 96:
        x[1, 64:96] = np.float32(1e-30)  # denormal-scale block
    kp = oracle.pad512(k)
    want = np.zeros((rows, kp // 32 * 36), dtype=np.uint8)
    lib.ref_quantize_q8_1_f32(_vp(x), _vp(want), k, kp, rows)
    got = oracle.quantize_q8_1(x)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("rows,cols", [(3, 4096), (2, 1000), (1, 14336), (2, 24)])
def test_rms_norm_family_matches_reference_kernels(oracle, dt, rows, cols):
    """add_rms_norm_* / rms_norm_residual_* of mistralrs-core/src/cuda/sort.cu run on host fibers (block reduction with real barriers and
    __shfl_down): the expected-value expressions that tests/test_glue_ops.py holds the HIP kernels to are the reference's results --
    residual sum bit-exact, normed outputs to one ulp of the dtype (the f32 sum of squares is order-dependent, rsqrtf approximate)."""
    from tests.util import ULP, round_through
    import torch
    lib = _ref("libref_rms.so")
    code = {"f16": 0, "bf16": 1, "f32": 2}[dt]
    td = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dt]
    rng = np.random.default_rng(cols + code)
    x = round_through(rng.standard_normal((rows, cols)).astype(np.float32), dt)
    r = round_through(rng.standard_normal((rows, cols)).astype(np.float32), dt)
    w = round_through(1 + 0.1 * rng.standard_normal(cols).astype(np.float32), dt)
    sc = round_through(np.array([0.5], np.float32), dt)
    raw = lambda a: torch.from_numpy(a).to(td).contiguous()
    xt, rt, wt, st = raw(x), raw(r), raw(w), raw(sc)
    rd, nd, dst = torch.zeros_like(xt), torch.zeros_like(xt), torch.zeros_like(xt)
    p = lambda t: C.c_void_p(t.data_ptr())
    eps = 1e-5
    assert lib.ref_add_rms_norm(code, p(xt), p(rt), p(wt), p(rd), p(nd), rows, cols, C.c_float(eps)) == 0
    assert lib.ref_rms_norm_residual(code, p(xt), p(rt), p(wt), p(st), p(dst), rows, cols, C.c_float(eps)) == 0
    tol = lambda want: 1.01 * ULP[dt] * np.abs(want) + 4e-6 * np.abs(want) + 1e-7
    s = round_through(x + r, dt)
    np.testing.assert_array_equal(rd.float().numpy(), s)                      # residual_out = T(x + r)
    want = oracle.rms_norm(s, w, eps)                                          # norm over the ROUNDED sum
    assert (np.abs(nd.float().numpy() - (round_through(want, dt) if dt != "f32" else want)) <= tol(want)).all()
    want = (r + oracle.rms_norm(x, w, eps)) * sc[0]                            # (residual + rms(x) * w) * scale
    assert (np.abs(dst.float().numpy() - (round_through(want, dt) if dt != "f32" else want)) <= tol(want) + 4e-6 * np.abs(r)).all()


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("b", [1, 3, 8])
def test_matvec_oracle_matches_reference_mmvq_kernels(oracle, t, b):
    """The reference's complete MMVQ kernels (mmvq_core_impl and the fused-GLU variant, mmvq_gguf.cu:724-996, f32 destination) executed
    on host fibers with the launch geometry of their launchers: the Q8_1 matvec oracle (f64 combination of the same terms) agrees to the
    f32 accumulation bound the GPU tests hold the HIP kernels to, for every format and for the single-column, 2..4 and 5..8 batch shapes."""
    lib = _ref("libref_mmvq_kernel.so")
    rng = np.random.default_rng(31 * t + b)
    n, k = 22, 1024
    w = oracle.random_blocks(t, n, k, seed=300 + t, d_scale=0.02)
    x = (rng.standard_normal((b, k)) * rng.uniform(0.2, 5.0, (b, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    stride = y.shape[1] // 36
    out = np.zeros((b, n), dtype=np.float32)
    assert lib.ref_mmvq_kernel_plain_f32(t, _vp(w), _vp(y), _vp(out), k, n, stride, n, b) == 0
    want, mag = oracle.matmul_q8_1_mag(t, w, n, k, y)
    tol = 8 * 2.0 ** -23 * np.sqrt(k / 16) * mag.astype(np.float64) + 2.0 ** -23 * np.abs(want) + 1e-30
    err = np.abs(out.astype(np.float64) - want)
    assert (err <= tol).all(), float((err / tol).max())
    # fused gate/up + activation: dst = act(gate . y) * (up . y)
    wu = oracle.random_blocks(t, n, k, seed=900 + t, d_scale=0.02)
    wantu, magu = oracle.matmul_q8_1_mag(t, wu, n, k, y)
    for act in (0, 1, 2):
        glu = np.zeros((b, n), dtype=np.float32)
        assert lib.ref_mmvq_kernel_fused_glu_f32(t, _vp(w), _vp(wu), _vp(y), _vp(glu), k, n, stride, n, b, act) == 0
        exp = oracle.fused_glu(want.astype(np.float32), wantu.astype(np.float32), act).astype(np.float64)
        # error propagation: d(act(g) u) <= |u| dg (act' <= ~1.13) + |act(g)| du, plus f32 rounding of the product
        ag = np.abs(oracle.fused_glu(want.astype(np.float32), np.ones_like(wantu, dtype=np.float32), act).astype(np.float64))
        tg = 8 * 2.0 ** -23 * np.sqrt(k / 16) * mag
        tu = 8 * 2.0 ** -23 * np.sqrt(k / 16) * magu
        tolg = 1.2 * np.abs(wantu) * tg + ag * tu + 4 * 2.0 ** -23 * np.abs(exp) + 1e-30
        errg = np.abs(glu.astype(np.float64) - exp)
        assert (errg <= tolg).all(), (act, float((errg / tolg).max()))


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
def test_fused_glu_matches_reference_kernels(oracle, dt, act):
    """fused_glu_kernel (scalar, strided rows) and fused_glu_kernel_vec4 of mistralrs-quant/kernels/ops/ops.cu on the host: out =
    T(act(float(a))) * b with the product rounded to T -- the expression tests/test_glue_ops.py holds the HIP kernel to."""
    from tests.util import ULP, round_through
    import torch
    lib = _ref("libref_glu.so")
    code = {"f16": 0, "bf16": 1, "f32": 2}[dt]
    td = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dt]
    rng = np.random.default_rng(act + code)
    for rows, cols, width in ((5, 1001, 2 * 1001), (4, 1024, 2 * 1024)):  # odd width -> scalar kernel; aligned -> vec4 kernel
        wide = round_through((rng.standard_normal((rows, width)) * 3).astype(np.float32), dt)
        wt = torch.from_numpy(wide).to(td).contiguous()
        out = torch.zeros(rows, cols, dtype=td)
        es = wt.element_size()
        a_ptr, b_ptr = wt.data_ptr(), wt.data_ptr() + cols * es
        assert lib.ref_fused_glu(code, C.c_void_p(a_ptr), C.c_void_p(b_ptr), C.c_void_p(out.data_ptr()), rows, cols, width, width, act) == 0
        a, b = wide[:, :cols], wide[:, cols:]
        want = round_through(round_through(oracle.fused_glu(a, np.ones_like(a), act), dt) * b, dt)
        tol = 2 * ULP[dt] * np.abs(want) + 2e-6 * np.abs(b) * (np.abs(a) + 1)
        assert (np.abs(out.float().numpy() - want) <= tol + 1e-30).all()


ROUTER_CASES = [
    dict(E=8, k=2, score=1, weight=0, renorm=True),                       # Mixtral: softmax -> top-2 -> renormalise
    dict(E=8, k=2, score=1, weight=0, renorm=False),
    dict(E=64, k=6, score=2, weight=0, renorm=True, bias=True, norm_min=1e-20, oscale=2.5),   # DeepSeek-style sigmoid + bias
    dict(E=128, k=8, score=0, weight=1, renorm=False),                    # top-k of raw logits, softmax over the picks
    dict(E=256, k=8, score=2, weight=2, renorm=True, escale=True, clamp=(-3.0, 3.0)),
    dict(E=576, k=40, score=1, weight=0, renorm=True),                    # more picks than lanes
    dict(E=4, k=4, score=1, weight=1, renorm=True),
    dict(E=1, k=1, score=1, weight=0, renorm=True),
]


def _router_inputs(case, rows=7, seed=0):
    rng = np.random.default_rng(seed + case["E"])
    x = (rng.standard_normal((rows, case["E"])) * 2).astype(np.float32)
    x[0, :] = 0.25                       # a full tie: ids must come out as 0, 1, 2, ...
    if case["E"] >= 8:
        x[1, 3] = np.nan                 # NaN logits are never selected
    bias = (rng.standard_normal(case["E"]) * 0.1).astype(np.float32) if case.get("bias") else None
    esc = rng.uniform(0.5, 2.0, case["E"]).astype(np.float32) if case.get("escale") else None
    return x, bias, esc


@pytest.mark.parametrize("case", ROUTER_CASES, ids=[f"E{c['E']}k{c['k']}s{c['score']}w{c['weight']}" for c in ROUTER_CASES])
def test_moe_router_oracle_matches_reference_kernel(oracle, case):
    """moe_router_topk_kernel (sort.cu:1186-1357) on host fibers: ids identical (incl. tie and NaN rules), weights to f32 rounding."""
    lib = _ref("libref_router.so")
    x, bias, esc = _router_inputs(case)
    rows, E, k = x.shape[0], case["E"], case["k"]
    ids = np.zeros((rows, k), dtype=np.uint32)
    w = np.zeros((rows, k), dtype=np.float32)
    clamp = case.get("clamp")
    rc = lib.ref_moe_router_topk_f32(_vp(x), _vp(w), _vp(ids), _vp(bias) if bias is not None else None, _vp(esc) if esc is not None else None,
                                     rows, E, k, case["score"], case["weight"], int(case["renorm"]), int(clamp is not None),
                                     C.c_float(clamp[0] if clamp else 0.0), C.c_float(clamp[1] if clamp else 0.0),
                                     C.c_float(case.get("norm_min", 0.0)), C.c_float(case.get("oscale", 1.0)))
    assert rc == 0
    gi, gw = oracle.moe_router_topk(x, k, case["score"], case["weight"], case["renorm"], bias, esc, clamp, case.get("norm_min", 0.0), case.get("oscale", 1.0))
    np.testing.assert_array_equal(gi, ids)
    np.testing.assert_allclose(gw, w, rtol=3e-6, atol=1e-9)
    if bias is None:
        assert list(ids[0, : min(k, 4)]) == list(range(min(k, 4)))  # full tie: lowest expert ids first


def _imoe_case(oracle, t, seed=0):
    rng = np.random.default_rng(seed + t)
    E, n, k, batch, topk = 5, 12, 512, 3, 2
    w = np.concatenate([oracle.random_blocks(t, n, k, seed=50 + e + t, d_scale=0.02) for e in range(E)], axis=0)  # [E*n, row_bytes]
    idx = rng.integers(0, E, size=batch * topk).astype(np.uint32)
    return E, n, k, batch, topk, w, idx, rng


@pytest.mark.parametrize("t", TYPES + [9])  # 9: Q8_1 weights (indexed_moe.cu:483-502)
@pytest.mark.parametrize("input_dim1", [1, 2])
def test_indexed_moe_forward_matches_reference_kernel(oracle, t, input_dim1):
    """indexed_moe_forward_<t>_q8_1 (kernels/indexed_moe/indexed_moe.cu:806-1013) on host fibers == per task the Q8_1 matvec oracle with the
    weights of expert indices[task]; input row = token (input_dim1 == 1: shared by the token's top-k slots) or task."""
    lib = _ref("libref_imoe.so")
    E, n, k, batch, topk, w, idx, rng = _imoe_case(oracle, t)
    rows_in = batch if input_dim1 == 1 else batch * topk
    x = (rng.standard_normal((rows_in, k)) * rng.uniform(0.3, 4.0, (rows_in, 1))).astype(np.float32)
    y = oracle.quantize_q8_1(x)
    kp = oracle.pad512(k)
    out = np.zeros((batch * topk, n), dtype=np.float32)
    assert lib.ref_indexed_moe_forward(t, _vp(w), _vp(y), _vp(idx), _vp(out), n, k, batch, topk, kp, input_dim1) == 0
    for task in range(batch * topk):
        e = int(idx[task])
        row = task // topk if input_dim1 == 1 else task
        want, mag = oracle.matmul_q8_1_mag(t, w[e * n:(e + 1) * n], n, k, y[row:row + 1])
        tol = 8 * 2.0 ** -23 * np.sqrt(k / 16) * mag[0].astype(np.float64) + 2.0 ** -23 * np.abs(want[0]) + 1e-30
        assert (np.abs(out[task].astype(np.float64) - want[0]) <= tol).all(), task


def _rope16(x, cos, sin, pos, neox, dt):
    """RoPE with the arithmetic in the tensor dtype, as the reference's f16 / bf16 kernel instantiations run it (rotary.cu:9-33):
    every product and the final sum / difference round to dt."""
    from tests.util import round_through
    r = lambda a: round_through(np.asarray(a, dtype=np.float32), dt)
    out = x.copy()
    pairs = cos.shape[1]
    c, s = cos[pos][:, None, :], sin[pos][:, None, :]
    if neox:
        a, b = x[..., :pairs], x[..., pairs:2 * pairs]
    else:
        a, b = x[..., 0:2 * pairs:2], x[..., 1:2 * pairs:2]
    xo, yo = r(r(a * c) - r(b * s)), r(r(b * c) + r(a * s))
    if neox:
        out[..., :pairs], out[..., pairs:2 * pairs] = xo, yo
    else:
        out[..., 0:2 * pairs:2], out[..., 1:2 * pairs:2] = xo, yo
    return out


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("neox", [False, True])
def test_rope_16bit_matches_reference_kernel(dt, neox):
    from tests.util import round_through
    import torch
    lib = _ref("libref_half.so")
    td = torch.float16 if dt == "f16" else torch.bfloat16
    rng = np.random.default_rng(11 + int(neox))
    T, H, KVH, hd, pairs, max_pos = 5, 3, 2, 64, 24, 40
    q = round_through(rng.standard_normal((T, H, hd)).astype(np.float32), dt)
    k = round_through(rng.standard_normal((T, KVH, hd)).astype(np.float32), dt)
    ang = rng.uniform(0, 6.28, (max_pos, pairs))
    cos, sin = round_through(np.cos(ang).astype(np.float32), dt), round_through(np.sin(ang).astype(np.float32), dt)
    pos = rng.integers(0, max_pos, T).astype(np.uint32)
    tq, tk, tc, ts = (torch.from_numpy(a).to(td).contiguous() for a in (q, k, cos, sin))
    p = lambda t: C.c_void_p(t.data_ptr())
    lib.ref_rotary_16(0 if dt == "f16" else 1, p(tq), p(tk), p(tc), p(ts), _vp(pos), int(neox), hd, T, pairs, H, KVH, C.c_long(H * hd), C.c_long(KVH * hd))
    np.testing.assert_array_equal(tq.float().numpy(), _rope16(q, cos, sin, pos, neox, dt))
    np.testing.assert_array_equal(tk.float().numpy(), _rope16(k, cos, sin, pos, neox, dt))


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("bits", [8, 4, 3, 2, 1])
def test_hqq_16bit_dequantize_matches_reference_kernel(dt, bits):
    """f16 / bf16 instantiations of the HQQ dequantize kernels (`(T(q) - zero) * scale` evaluated in T) == the oracle, bit for bit."""
    from oracle import hqq_oracle as H
    from tests.util import round_through
    import torch
    lib = _ref("libref_half.so")
    td = torch.float16 if dt == "f16" else torch.bfloat16
    rng = np.random.default_rng(bits)
    h, w = 6, 40
    q = rng.integers(0, 2 ** bits, size=(H.PACK[bits] * h, w)).astype(np.uint32 if bits == 3 else np.uint8)
    packed = H.pack(bits, q)
    scale = round_through((rng.uniform(0.001, 0.05, w) * rng.choice([1.0, -1.0], w)).astype(np.float32), dt)
    zero = round_through(rng.uniform(0.0, 2 ** bits - 1.0, w).astype(np.float32), dt)
    ts, tz = torch.from_numpy(scale).to(td), torch.from_numpy(zero).to(td)
    out = torch.zeros(H.PACK[bits] * h, w, dtype=td)
    pk = np.ascontiguousarray(packed)
    assert lib.ref_hqq_dequantize_16(0 if dt == "f16" else 1, bits, _vp(pk), C.c_void_p(ts.data_ptr()), C.c_void_p(tz.data_ptr()), C.c_void_p(out.data_ptr()), h, w) == 0
    np.testing.assert_array_equal(out.float().numpy().view(np.uint32), H.dequantize(bits, packed, scale, zero, dt).view(np.uint32))
