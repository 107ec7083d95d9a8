"""Decode engine (csrc/dec_core2.cuh, ext_dec.hip): GEMV phases in the arithmetic of the reference CPU path
(GgufMatMul::forward_raw -> candle QMatMul on f32 activations, mistralrs-quant/src/gguf/mod.rs:465-478): Q8_K / Q8_0 activation quantization,
integer block dots, f32 combination.  Checked against oracle B (oracle/ggml_oracle.c orc_matmul_cpu: dot_kquant_q8K / dot_legacy_q8).
Same test bodies on the wave64 host emulation (CPU suite) and on the MI355X (`-m gpu`).

Two bars per case (round 3):
  * BIT EQUALITY with the engine-order restatement (oracle/cpu_path_oracle.c: orc_gemv_engine, orc_rms_norm_engine, orc_fused_glu_engine,
    orc_attention_engine): the same integers, the same f32 products, the kernel's documented summation order -- written from the format
    definitions and the order's description, not from the kernel source;
  * the generic ggml order (orc_matmul_cpu) within the f32-order tolerance: with identical activations the integer parts are exact, so only the
    f32 summation order over the blocks of a row differs: 2e-5 * max|want| is generous."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend


class Mat(C.Structure):
    _fields_ = [("planes", C.c_void_p), ("type", C.c_int), ("n", C.c_longlong), ("k", C.c_longlong)]


def repack(be, O, t, packed, n, k):
    nbytes = be.sym("mrs_dec_repack_bytes", [C.c_int, C.c_longlong, C.c_longlong], C.c_size_t)(t, n, k)
    assert nbytes > 0
    src = be.buf(np.ascontiguousarray(packed).reshape(-1))
    dst = be.buf(np.zeros(nbytes, dtype=np.uint8))
    assert be.sym("mrs_dec_repack", [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p], C.c_int)(src.ptr, t, n, k, dst.ptr, be.stream) == 0
    return dst, Mat(dst.ptr, t, n, k)


def _weights(O, t, n, k, seed):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((n, k)) * 0.05).astype(np.float32)
    return O.quantize(t, w).reshape(n, -1)


PROJ = [C.POINTER(Mat), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]


def check_proj(O, be, tname, n, k, b, mode, seed=0):
    t = getattr(O, tname)
    packed = _weights(O, t, n, k, seed)
    keep, m = repack(be, O, t, packed, n, k)
    rng = np.random.default_rng(seed + 1)
    x = rng.standard_normal((b, k)).astype(np.float32)
    x[0, : min(256, k)] *= 0.0 if seed % 2 else 1.0  # an all-zero activation block now and then
    want = O.matmul_cpu(t, packed, n, k, x)
    base = rng.standard_normal((b, n)).astype(np.float32)
    xb, ob = be.buf(x), be.buf(base.copy())
    fn = be.sym("mrs_dec_proj", PROJ, C.c_int)
    rs = 0.5
    assert fn(C.byref(m), n, None, xb.ptr, k, None, 0.0, ob.ptr, n, mode, rs, None, b, be.stream) == 0
    got = ob.numpy()
    eng = np.concatenate([O.gemv_engine(t, packed, n, k, r) for r in x], axis=0)
    if mode:
        want = base * np.float32(rs) + want
        eng = base * np.float32(rs) + eng * np.float32(1.0)
    assert np.array_equal(got, eng), (tname, n, k, b, "engine-order oracle", float(np.abs(got - eng).max()))
    tol = 2e-5 * np.abs(want).max()
    assert np.abs(got - want).max() <= tol, (tname, n, k, b, float(np.abs(got - want).max()), tol)


CASES = [("Q4_K", 70, 512, 1, 0), ("Q4_K", 33, 1024, 2, 1), ("Q4_K", 16, 4096, 1, 1), ("Q4_K", 9, 768, 3, 0), ("Q4_K", 24, 3584, 1, 0),
         ("Q5_K", 40, 512, 1, 0), ("Q5_K", 12, 2048, 2, 1),
         ("Q6_K", 50, 512, 1, 0), ("Q6_K", 20, 4096, 1, 1), ("Q6_K", 7, 768, 8, 0), ("Q6_K", 10, 3584, 2, 0),
         ("Q8_0", 40, 512, 1, 0), ("Q8_0", 18, 1280, 2, 1), ("Q8_0", 8, 4096, 1, 0),
         # activation rows of a 70B down_proj (K = 28672): 5 and 8 columns exceed the LDS budget of one launch and run as column groups (advisor, round 2)
         ("Q4_K", 5, 28672, 5, 1), ("Q6_K", 3, 28672, 8, 0)]


@pytest.mark.parametrize("tname,n,k,b,mode", CASES)
def test_proj_host_emulation(oracle, tname, n, k, b, mode):
    check_proj(oracle, HostBackend(), tname, n, k, b, mode, seed=n + k)


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,b,mode", CASES + [("Q4_K", 4096, 4096, 1, 1), ("Q4_K", 4096, 14336, 1, 1), ("Q6_K", 4096, 14336, 1, 1), ("Q6_K", 1024, 4096, 4, 0),
                                                     ("Q8_0", 4096, 4096, 2, 1), ("Q5_K", 2048, 4096, 1, 0), ("Q6_K", 32064, 4096, 1, 0)])
def test_proj_gpu(oracle, dev, tname, n, k, b, mode):
    check_proj(oracle, GpuBackend(dev), tname, n, k, b, mode, seed=n + k)


def check_norm_proj(O, be, tname, n, k, b):
    """RMSNorm fused into the prologue (candle's x / sqrt(mean + eps) * w): bit-equal to the engine-order restatement (same expression, the
    kernel's summation tree); against candle's in-order sum a 1-ulp difference of the norm can move single quants by one step, so that bar
    is the quantization step, not the f32 order."""
    t = getattr(O, tname)
    packed = _weights(O, t, n, k, 3)
    keep, m = repack(be, O, t, packed, n, k)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((b, k)).astype(np.float32)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32)
    want = O.matmul_cpu(t, packed, n, k, O.rms_norm_candle(x, nw, 1e-5))
    xb, nb, ob = be.buf(x), be.buf(nw), be.buf(np.zeros((b, n), dtype=np.float32))
    assert be.sym("mrs_dec_proj", PROJ, C.c_int)(C.byref(m), n, None, xb.ptr, k, nb.ptr, 1e-5, ob.ptr, n, 0, 1.0, None, b, be.stream) == 0
    got = ob.numpy()
    eng = np.concatenate([O.gemv_engine(t, packed, n, k, r) for r in O.rms_norm_engine(x, nw, 1e-5)], axis=0)
    assert np.array_equal(got, eng), (tname, n, k, b, "engine-order oracle", float(np.abs(got - eng).max()))
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max()
    assert np.median(np.abs(got - want)) <= 2e-5 * np.abs(want).max()  # most outputs see identical quants


@pytest.mark.parametrize("tname,n,k,b", [("Q4_K", 24, 1024, 2), ("Q6_K", 16, 512, 1), ("Q8_0", 16, 512, 1)])
def test_norm_proj_host_emulation(oracle, tname, n, k, b):
    check_norm_proj(oracle, HostBackend(), tname, n, k, b)


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,b", [("Q4_K", 1024, 4096, 2), ("Q6_K", 2048, 4096, 1), ("Q8_0", 512, 4096, 3)])
def test_norm_proj_gpu(oracle, dev, tname, n, k, b):
    check_norm_proj(oracle, GpuBackend(dev), tname, n, k, b)


GLU = [C.POINTER(Mat), C.POINTER(Mat), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]


def check_gate_up(O, be, tname, n, k, b, experts=0, sel=0):
    t = getattr(O, tname)
    E = max(experts, 1)
    pg, pu = _weights(O, t, E * n, k, 11), _weights(O, t, E * n, k, 12)
    kg, mg = repack(be, O, t, pg, E * n, k)
    ku, mu = repack(be, O, t, pu, E * n, k)
    rng = np.random.default_rng(13)
    x = rng.standard_normal((b, k)).astype(np.float32)
    g = O.matmul_cpu(t, pg[sel * n:(sel + 1) * n], n, k, x)
    u = O.matmul_cpu(t, pu[sel * n:(sel + 1) * n], n, k, x)
    want = O.fused_glu(g, u, 0)
    xb, ob = be.buf(x), be.buf(np.zeros((b, n), dtype=np.float32))
    selb = be.buf(np.array([sel], dtype=np.int32)) if experts else None
    fn = be.sym("mrs_dec_gate_up", GLU, C.c_int)
    assert fn(C.byref(mg), C.byref(mu), n, selb.ptr if selb else None, xb.ptr, k, None, 0.0, 0, ob.ptr, n, b, be.stream) == 0
    got = ob.numpy()
    ge = np.concatenate([O.gemv_engine(t, pg[sel * n:(sel + 1) * n], n, k, r) for r in x], axis=0)
    ue = np.concatenate([O.gemv_engine(t, pu[sel * n:(sel + 1) * n], n, k, r) for r in x], axis=0)
    eng = O.fused_glu_engine(ge, ue)
    assert np.array_equal(got, eng), (tname, n, k, b, "engine-order oracle", float(np.abs(got - eng).max()))
    assert np.abs(got - want).max() <= 3e-5 * np.abs(want).max() + 1e-7


@pytest.mark.parametrize("tname,n,k,b,experts,sel", [("Q4_K", 64, 512, 1, 0, 0), ("Q6_K", 20, 768, 2, 0, 0), ("Q8_0", 36, 512, 1, 0, 0), ("Q4_K", 48, 512, 1, 3, 2)])
def test_gate_up_host_emulation(oracle, tname, n, k, b, experts, sel):
    check_gate_up(oracle, HostBackend(), tname, n, k, b, experts, sel)


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,b,experts,sel", [("Q4_K", 14336, 4096, 1, 0, 0), ("Q6_K", 1000, 4096, 2, 0, 0), ("Q8_0", 2048, 4096, 1, 0, 0), ("Q4_K", 1024, 4096, 1, 4, 3),
                                                     ("Q5_K", 512, 2048, 8, 0, 0)])
def test_gate_up_gpu(oracle, dev, tname, n, k, b, experts, sel):
    check_gate_up(oracle, GpuBackend(dev), tname, n, k, b, experts, sel)


QKV = [C.POINTER(Mat)] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]


def check_qkv(O, be, tq, tv, heads, kvh, hd, k, b, kv_dtype=1):
    """q/k/v + interleaved RoPE + paged cache write vs the oracle: rope(matmul_cpu(...)), cache pages through kv_cache_gather."""
    nq, nkv, bs = heads * hd, kvh * hd, 32
    pq, pk, pv = _weights(O, getattr(O, tq), nq, k, 21), _weights(O, getattr(O, tq), nkv, k, 22), _weights(O, getattr(O, tv), nkv, k, 23)
    kq, mq = repack(be, O, getattr(O, tq), pq, nq, k)
    kk, mk = repack(be, O, getattr(O, tq), pk, nkv, k)
    kv, mv = repack(be, O, getattr(O, tv), pv, nkv, k)
    rng = np.random.default_rng(24)
    x = rng.standard_normal((b, k)).astype(np.float32)
    nw = (1.0 + 0.05 * rng.standard_normal(k)).astype(np.float32)
    maxpos = 64
    inv = 1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))
    fr = np.arange(maxpos, dtype=np.float32)[:, None] * inv[None, :]
    cos, sin = np.cos(fr).astype(np.float32), np.sin(fr).astype(np.float32)
    pos = np.array([5 + 7 * i for i in range(b)], dtype=np.int32)
    nblocks = 2 * b
    slots = np.array([(2 * i) * bs + int(pos[i]) % bs for i in range(b)], dtype=np.int64)
    xn = O.rms_norm(x, nw, 1e-5)
    q = O.rope(O.matmul_cpu(getattr(O, tq), pq, nq, k, xn).reshape(b, heads, hd), cos, sin, pos, False).reshape(b, nq)
    kk_ = O.rope(O.matmul_cpu(getattr(O, tq), pk, nkv, k, xn).reshape(b, kvh, hd), cos, sin, pos, False).reshape(b, nkv)
    vv = O.matmul_cpu(getattr(O, tv), pv, nkv, k, xn)
    cache_np = np.float16 if kv_dtype == 0 else None
    kc = be.buf(np.zeros((nblocks, kvh, hd // 8, bs, 8), dtype=np.uint16))
    vc = be.buf(np.zeros((nblocks, kvh, hd, bs), dtype=np.uint16))
    xb, nb, qb = be.buf(x), be.buf(nw), be.buf(np.zeros((b, nq), dtype=np.float32))
    sb, pb, cb, snb = be.buf(slots), be.buf(pos), be.buf(cos), be.buf(sin)
    fn = be.sym("mrs_dec_qkv", QKV, C.c_int)
    assert fn(C.byref(mq), C.byref(mk), C.byref(mv), xb.ptr, k, nb.ptr, 1e-5, qb.ptr, kc.ptr, vc.ptr, sb.ptr, pb.ptr, cb.ptr, snb.ptr, hd, hd // 2, kvh, bs, kv_dtype, b,
              be.stream) == 0
    got_q = qb.numpy()
    tol = 2e-3 * max(np.abs(q).max(), np.abs(kk_).max(), np.abs(vv).max())
    assert np.abs(got_q - q).max() <= tol

    def dec(a16):
        a16 = np.asarray(a16).astype(np.uint16)
        return a16.view(np.float16).astype(np.float32) if kv_dtype == 0 else O.from_bf16_bits(a16)
    kcn, vcn = dec(kc.numpy().reshape(nblocks, kvh, hd // 8, bs, 8)), dec(vc.numpy().reshape(nblocks, kvh, hd, bs))
    for i in range(b):
        blk, off = int(slots[i]) // bs, int(slots[i]) % bs
        gk = kcn[blk, :, :, off, :].reshape(kvh * hd)
        gv = vcn[blk, :, :, off].reshape(kvh * hd)
        rel = 2.0 ** -8 if kv_dtype == 1 else 2.0 ** -11
        assert np.abs(gk - kk_[i]).max() <= tol + rel * np.abs(kk_[i]).max()
        assert np.abs(gv - vv[i]).max() <= tol + rel * np.abs(vv[i]).max()
    # untouched slots stay zero
    assert np.count_nonzero(kcn) <= b * kvh * hd and np.count_nonzero(vcn) <= b * kvh * hd


@pytest.mark.parametrize("tq,tv,heads,kvh,hd,k,b,kvd", [("Q4_K", "Q6_K", 4, 2, 64, 512, 1, 1), ("Q4_K", "Q4_K", 2, 1, 128, 512, 2, 0), ("Q8_0", "Q8_0", 2, 2, 64, 256, 1, 1)])
def test_qkv_host_emulation(oracle, tq, tv, heads, kvh, hd, k, b, kvd):
    check_qkv(oracle, HostBackend(), tq, tv, heads, kvh, hd, k, b, kvd)


@pytest.mark.gpu
@pytest.mark.parametrize("tq,tv,heads,kvh,hd,k,b,kvd", [("Q4_K", "Q6_K", 32, 8, 128, 4096, 1, 1), ("Q4_K", "Q4_K", 32, 8, 128, 4096, 3, 0), ("Q8_0", "Q8_0", 8, 2, 128, 2048, 2, 1),
                                                        ("Q6_K", "Q6_K", 8, 1, 128, 1024, 1, 1)])
def test_qkv_gpu(oracle, dev, tq, tv, heads, kvh, hd, k, b, kvd):
    check_qkv(oracle, GpuBackend(dev), tq, tv, heads, kvh, hd, k, b, kvd)


ACT_IMG = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
PROJ_IMG_ = [C.POINTER(Mat), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
GLU_IMG = [C.POINTER(Mat), C.POINTER(Mat), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
QKV_IMG = [C.POINTER(Mat)] * 3 + [C.c_void_p] * 8 + [C.c_int] * 7 + [C.c_void_p]


def _act_image(be, t, x, nw, k, b):
    """mrs_dec_act_image: the batch's activation image built once (one workgroup per column, column groups that fit LDS)"""
    nbytes = be.sym("mrs_dec_act_image_bytes", [C.c_int, C.c_int], C.c_size_t)(k, b)
    img = be.buf(np.full(nbytes, 0xA5, np.uint8))  # dead slots of the image keep this fill: whatever they hold must not reach a result
    assert be.sym("mrs_dec_act_image", ACT_IMG, C.c_int)(x.ptr, k, nw.ptr if nw is not None else None, 1e-5, k, t, b, img.ptr, be.stream) == 0
    return img


def check_image_proj(O, be, tname, n, k, b, norm):
    """mrs_dec_proj (every GEMV workgroup quantizes the b columns itself) against mrs_dec_act_image + mrs_dec_proj_img: the same bits (batched decode of the runner)"""
    t = getattr(O, tname)
    packed = _weights(O, t, n, k, 31)
    keep, m = repack(be, O, t, packed, n, k)
    rng = np.random.default_rng(32)
    x = rng.standard_normal((b, k)).astype(np.float32)
    x[b - 1, :256] = 0.0
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32)
    base = rng.standard_normal((b, n)).astype(np.float32)
    xb, nb = be.buf(x), (be.buf(nw) if norm else None)
    o1, o2 = be.buf(base.copy()), be.buf(base.copy())
    assert be.sym("mrs_dec_proj", PROJ, C.c_int)(C.byref(m), n, None, xb.ptr, k, nb.ptr if norm else None, 1e-5, o1.ptr, n, 1, 0.5, None, b, be.stream) == 0
    img = _act_image(be, t, xb, nb, k, b)
    assert be.sym("mrs_dec_proj_img", PROJ_IMG_, C.c_int)(C.byref(m), n, img.ptr, o2.ptr, n, 1, 0.5, b, be.stream) == 0
    a1, a2 = o1.numpy(), o2.numpy()
    assert np.array_equal(a1, a2), (tname, n, k, b, norm, float(np.abs(a1 - a2).max()))
    xe = O.rms_norm_engine(x, nw, 1e-5) if norm else x
    eng = base * np.float32(0.5) + np.concatenate([O.gemv_engine(t, packed, n, k, r) for r in xe], axis=0) * np.float32(1.0)
    assert np.array_equal(a2, eng), (tname, n, k, b, norm, "engine-order oracle")


def check_image_gate_up(O, be, tname, n, k, b):
    t = getattr(O, tname)
    pg, pu = _weights(O, t, n, k, 41), _weights(O, t, n, k, 42)
    kg, mg = repack(be, O, t, pg, n, k)
    ku, mu = repack(be, O, t, pu, n, k)
    rng = np.random.default_rng(43)
    x = rng.standard_normal((b, k)).astype(np.float32)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32)
    xb, nb = be.buf(x), be.buf(nw)
    o1, o2 = be.buf(np.zeros((b, n), np.float32)), be.buf(np.zeros((b, n), np.float32))
    assert be.sym("mrs_dec_gate_up", GLU, C.c_int)(C.byref(mg), C.byref(mu), n, None, xb.ptr, k, nb.ptr, 1e-5, 0, o1.ptr, n, b, be.stream) == 0
    img = _act_image(be, t, xb, nb, k, b)
    assert be.sym("mrs_dec_gate_up_img", GLU_IMG, C.c_int)(C.byref(mg), C.byref(mu), n, img.ptr, 0, o2.ptr, n, b, be.stream) == 0
    a1, a2 = o1.numpy(), o2.numpy()
    assert np.array_equal(a1, a2) and np.isfinite(a2).all(), (tname, n, k, b, float(np.abs(a1 - a2).max()))


def check_image_qkv(O, be, tq, tv, heads, kvh, k, b, neox):
    hd, bs = 128, 32
    nq, nkv = heads * hd, kvh * hd
    pq, pk, pv = _weights(O, getattr(O, tq), nq, k, 51), _weights(O, getattr(O, tq), nkv, k, 52), _weights(O, getattr(O, tv), nkv, k, 53)
    kq, mq = repack(be, O, getattr(O, tq), pq, nq, k)
    kk, mk = repack(be, O, getattr(O, tq), pk, nkv, k)
    kv, mv = repack(be, O, getattr(O, tv), pv, nkv, k)
    rng = np.random.default_rng(54)
    x = rng.standard_normal((b, k)).astype(np.float32)
    nw = (1.0 + 0.05 * rng.standard_normal(k)).astype(np.float32)
    inv = 1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))
    fr = np.arange(64, dtype=np.float32)[:, None] * inv[None, :]
    pos = np.array([5 + 7 * i for i in range(b)], dtype=np.int32)
    slots = np.array([(2 * i) * bs + int(pos[i]) % bs for i in range(b)], dtype=np.int64)
    xb, nb, sb, pb, cb, snb = be.buf(x), be.buf(nw), be.buf(slots), be.buf(pos), be.buf(np.cos(fr).astype(np.float32)), be.buf(np.sin(fr).astype(np.float32))
    res = []
    for use_img in (False, True):
        kc, vc = be.buf(np.zeros((2 * b, kvh, hd // 8, bs, 8), dtype=np.uint16)), be.buf(np.zeros((2 * b, kvh, hd, bs), dtype=np.uint16))
        qb = be.buf(np.zeros((b, nq), dtype=np.float32))
        if use_img:
            img = _act_image(be, getattr(O, tq), xb, nb, k, b)
            assert be.sym("mrs_dec_qkv_img", QKV_IMG, C.c_int)(C.byref(mq), C.byref(mk), C.byref(mv), img.ptr, qb.ptr, kc.ptr, vc.ptr, sb.ptr, pb.ptr, cb.ptr, snb.ptr, hd,
                                                                hd // 2, kvh, bs, 1, b, neox, be.stream) == 0
        else:
            assert be.sym("mrs_dec_qkv_neox" if neox else "mrs_dec_qkv", QKV, C.c_int)(C.byref(mq), C.byref(mk), C.byref(mv), xb.ptr, k, nb.ptr, 1e-5, qb.ptr, kc.ptr, vc.ptr,
                                                                                      sb.ptr, pb.ptr, cb.ptr, snb.ptr, hd, hd // 2, kvh, bs, 1, b, be.stream) == 0
        res.append((qb.numpy(), kc.numpy(), vc.numpy()))
    for a1, a2 in zip(*res):
        assert np.array_equal(a1, a2)
    assert np.count_nonzero(res[1][0]) > 0 and np.count_nonzero(res[1][1]) > 0


@pytest.mark.parametrize("tname,n,k,b,norm", [("Q4_K", 24, 1024, 3, True), ("Q6_K", 16, 512, 8, False), ("Q4_K", 8, 14336, 8, False), ("Q5_K", 12, 768, 5, True), ("Q8_0", 16, 512, 4, True)])
def test_image_proj_host_emulation(oracle, tname, n, k, b, norm):
    check_image_proj(oracle, HostBackend(), tname, n, k, b, norm)  # (k = 14336, b = 8: two column groups of four)


def test_image_gate_up_qkv_host_emulation(oracle):
    check_image_gate_up(oracle, HostBackend(), "Q4_K", 32, 512, 4)
    check_image_gate_up(oracle, HostBackend(), "Q8_0", 16, 512, 2)
    check_image_qkv(oracle, HostBackend(), "Q4_K", "Q6_K", 2, 1, 512, 3, 0)
    check_image_qkv(oracle, HostBackend(), "Q4_K", "Q4_K", 2, 2, 256, 2, 1)


def test_image_entry_points_refuse_what_they_cannot_do_host_emulation(oracle):
    """argument checks of the batched-image entry points (return -1, nothing launched): column counts outside 1..8, K not a multiple of 256, a misaligned image, a weight
    type without a decode layout, a missing image; mrs_dec_qkv refuses head sizes that are not powers of two (its epilogue indexes with shifts and masks)"""
    O, be = oracle, HostBackend()
    k, b = 512, 2
    x = be.buf(np.ones((8, k), np.float32))
    nbytes = be.sym("mrs_dec_act_image_bytes", [C.c_int, C.c_int], C.c_size_t)(k, 8)
    img = be.buf(np.zeros(nbytes + 64, np.uint8))
    base = (img.ptr + 15) & ~15
    fn = be.sym("mrs_dec_act_image", ACT_IMG, C.c_int)
    assert fn(x.ptr, k, None, 0.0, k, O.Q4_K, b, base, be.stream) == 0
    for bad in ((x.ptr, k, None, 0.0, k, O.Q4_K, 0, base), (x.ptr, k, None, 0.0, k, O.Q4_K, 9, base), (x.ptr, k, None, 0.0, 300, O.Q4_K, b, base),
                (x.ptr, k, None, 0.0, k, O.Q4_K, b, base + 4), (x.ptr, k, None, 0.0, k, 0, b, base), (None, k, None, 0.0, k, O.Q4_K, b, base), (x.ptr, k, None, 0.0, k, O.Q4_K, b, None)):
        assert fn(*bad, be.stream) == -1, bad
    t = O.Q4_K
    keep, m = repack(be, O, t, _weights(O, t, 16, k, 1), 16, k)
    out = be.buf(np.zeros((b, 16), np.float32))
    assert be.sym("mrs_dec_gate_up_img", GLU_IMG, C.c_int)(C.byref(m), C.byref(m), 16, None, 0, out.ptr, 16, b, be.stream) == -1
    assert be.sym("mrs_dec_proj_img", PROJ_IMG_, C.c_int)(C.byref(m), 16, None, out.ptr, 16, 0, 1.0, b, be.stream) == -1
    # head size 96 (not a power of two): refused by the q / k / v phase
    hd, heads = 96, 2
    kq, mq = repack(be, O, t, _weights(O, t, heads * hd, k, 2), heads * hd, k)
    nw, qb = be.buf(np.ones(k, np.float32)), be.buf(np.zeros((1, heads * hd), np.float32))
    kc, vc = be.buf(np.zeros((2, heads, hd // 8, 32, 8), np.uint16)), be.buf(np.zeros((2, heads, hd, 32), np.uint16))
    sb, pb = be.buf(np.zeros(1, np.int64)), be.buf(np.zeros(1, np.int32))
    cs = be.buf(np.ones((4, hd // 2), np.float32))
    assert be.sym("mrs_dec_qkv", QKV, C.c_int)(C.byref(mq), C.byref(mq), C.byref(mq), x.ptr, k, nw.ptr, 1e-5, qb.ptr, kc.ptr, vc.ptr, sb.ptr, pb.ptr, cs.ptr, cs.ptr, hd, hd // 2,
                                               heads, 32, 1, 1, be.stream) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,b,norm", [("Q4_K", 4096, 14336, 8, False), ("Q6_K", 4096, 14336, 5, False), ("Q4_K", 2048, 4096, 8, True), ("Q6_K", 512, 28672, 8, False),
                                              ("Q4_K", 1000, 4096, 2, True), ("Q8_0", 1024, 4096, 8, True), ("Q8_0", 512, 14336, 6, False)])
def test_image_proj_gpu(oracle, dev, tname, n, k, b, norm):
    check_image_proj(oracle, GpuBackend(dev), tname, n, k, b, norm)


@pytest.mark.gpu
def test_image_gate_up_qkv_gpu(oracle, dev):
    be = GpuBackend(dev)
    check_image_gate_up(oracle, be, "Q4_K", 14336, 4096, 8)
    check_image_gate_up(oracle, be, "Q8_0", 2048, 4096, 3)
    check_image_qkv(oracle, be, "Q4_K", "Q6_K", 32, 8, 4096, 8, 0)
    check_image_qkv(oracle, be, "Q4_K", "Q4_K", 32, 8, 4096, 4, 1)


ATTN_F32 = [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]
ATTN_Q8K = [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]
PROJ_IMG = [C.POINTER(Mat), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]


ATTN2 = [C.c_void_p] * 9 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 11 + [C.c_void_p]


def _gather_kv(kc, vc, bt_row, ctx, from16):
    """paged cache (K [blocks, kvh, hd/8, 32, 8], V [blocks, kvh, hd, 32]) -> k, v [ctx, kvh, hd] f32"""
    pos = np.arange(ctx)
    blk, off = bt_row[pos // 32], pos % 32
    k = from16(kc[blk, :, :, off, :]).reshape(ctx, kc.shape[1], -1)                 # [ctx, kvh, hd/8, 8]
    v = from16(vc[blk, :, :, off])                                                   # [ctx, kvh, hd]
    return k, v


def check_attention(O, be, heads, kvh, ctxs, max_ctx, kv_dtype=1, n_out=24, window=0):
    """mrs_dec_attention (split waves + last-arriver merge + Q8_K image, ONE launch): the f32 result equals the engine-order restatement
    (orc_attention_engine) BIT FOR BIT, agrees with the restated in-tree CPU attention (single_q.rs order) to f32 rounding, and o_proj on the
    image equals o_proj on the f32 vector bit for bit.  """
    hd, bs, b = 128, 32, len(ctxs)
    nq = heads * hd
    rng = np.random.default_rng(heads + kvh + sum(ctxs))
    mbs = (max_ctx + bs - 1) // bs
    nblocks = b * mbs + 1
    to16 = (lambda a: a.astype(np.float16).view(np.uint16)) if kv_dtype == 0 else O.to_bf16_bits
    from16 = (lambda a: a.view(np.float16).astype(np.float32)) if kv_dtype == 0 else O.from_bf16_bits
    kc_np = to16((rng.standard_normal((nblocks, kvh, hd // 8, bs, 8)) * 0.7).astype(np.float32))
    vc_np = to16(rng.standard_normal((nblocks, kvh, hd, bs)).astype(np.float32))
    bt_np = rng.permutation(nblocks - 1)[: b * mbs].reshape(b, mbs).astype(np.uint32) + 1
    kc, vc, bt = be.buf(kc_np), be.buf(vc_np), be.buf(bt_np)
    cl = be.buf(np.asarray(ctxs, dtype=np.uint32))
    q_np = (rng.standard_normal((b, nq)) * 0.5).astype(np.float32)
    q = be.buf(q_np)
    splits = be.sym("mrs_decode_attention_max_splits", [C.c_int], C.c_int)(max_ctx)
    po, pm, pl = be.buf(np.zeros((b, heads, splits, hd), np.float32)), be.buf(np.zeros((b, heads, splits), np.float32)), be.buf(np.zeros((b, heads, splits), np.float32))
    ticket = be.buf(np.zeros(b * kvh, np.uint32))
    scale = np.float32(1.0 / np.sqrt(np.float32(hd)))
    nimg = be.sym("mrs_dec_act_image_bytes", [C.c_int, C.c_int], C.c_size_t)(nq, b)
    S_ = nq // 256
    assert nimg == b * (4 * (((S_ + 3) // 4) * 256 + 16) + 4 * ((S_ + 3) // 4) * (48 + 80))  # per column: 4 chunks of quants (16-byte skew each); scales (48 covers the Q8_0 mode) and 16 sums at a stride of 80 for the 4 * ceil(S / 4) superblock slots of a row group's tiles
    img, got = be.buf(np.zeros(nimg, np.uint8)), be.buf(np.full((b, nq), np.nan, np.float32))
    fn = be.sym("mrs_dec_attention", ATTN2, C.c_int)
    even = (heads // kvh) % 2 == 0
    for rep in range(2):  # twice: the arrival counters must be back at zero
        rc = fn(got.ptr, img.ptr, ticket.ptr, po.ptr, pm.ptr, pl.ptr, q.ptr, kc.ptr, vc.ptr, kvh, scale, bt.ptr, cl.ptr, bs, max_ctx, b, heads, hd, mbs, nq, kvh * hd * bs,
                hd * bs, kv_dtype, window, be.stream)
        assert rc == (1 if even else 0)
        assert not ticket.numpy().any()
    res = got.numpy()
    nblk_max = (max_ctx + 31) // 32
    bpw = 1 if nblk_max <= 64 else (nblk_max + 63) // 64
    for i, ctx in enumerate(ctxs):
        k, v = _gather_kv(kc_np, vc_np, bt_np[i], ctx, from16)
        eng = O.attention_engine(q_np[i].reshape(heads, hd), k, v, scale, bpw, window)
        assert np.array_equal(res[i].reshape(heads, hd), eng), ("engine-order oracle", i, ctx, float(np.abs(res[i].reshape(heads, hd) - eng).max()))
        lo = max(0, ctx - window) if window else 0   # the query sits at position ctx - 1: keys <= ctx - 1 - window are masked
        cpu = O.attention_single_q_cpu(q_np[i].reshape(heads, hd), k[lo:], v[lo:], scale, 1)
        assert np.abs(res[i].reshape(heads, hd) - cpu).max() <= 2e-6 * max(1.0, np.abs(cpu).max()), ("single_q.rs order", i, ctx)
    if not even:
        return
    # o_proj on the image == o_proj on the f32 vector (the same quantizer ran in a different kernel)
    t = O.Q4_K
    packed = _weights(O, t, n_out, nq, 7)
    keep, m = repack(be, O, t, packed, n_out, nq)
    base = rng.standard_normal((b, n_out)).astype(np.float32)
    o_ref, o_img = be.buf(base.copy()), be.buf(base.copy())
    assert be.sym("mrs_dec_proj", PROJ, C.c_int)(C.byref(m), n_out, None, got.ptr, nq, None, 0.0, o_ref.ptr, n_out, 1, 0.5, None, b, be.stream) == 0
    assert be.sym("mrs_dec_proj_img", PROJ_IMG, C.c_int)(C.byref(m), n_out, img.ptr, o_img.ptr, n_out, 1, 0.5, b, be.stream) == 0
    np.testing.assert_array_equal(o_img.numpy(), o_ref.numpy())
    eng_o = base * np.float32(0.5) + np.concatenate([O.gemv_engine(t, packed, n_out, nq, r) for r in res], axis=0) * np.float32(1.0)
    np.testing.assert_array_equal(o_img.numpy(), eng_o)


@pytest.mark.parametrize("heads,kvh,ctxs,max_ctx,kvd", [(4, 2, [70], 128, 1), (4, 1, [33, 200], 224, 1), (2, 1, [1], 64, 0), (8, 2, [500, 17, 96], 512, 1)])
def test_fused_attention_host_emulation(oracle, heads, kvh, ctxs, max_ctx, kvd):
    check_attention(oracle, HostBackend(), heads, kvh, ctxs, max_ctx, kvd)


@pytest.mark.parametrize("heads,kvh,ctxs,max_ctx,kvd", [(2, 2, [45], 64, 1), (8, 1, [100, 31], 128, 0), (4, 2, [2300], 4096, 1)])
def test_attention_more_shapes_host_emulation(oracle, heads, kvh, ctxs, max_ctx, kvd):
    check_attention(oracle, HostBackend(), heads, kvh, ctxs, max_ctx, kvd)


@pytest.mark.gpu
@pytest.mark.parametrize("heads,kvh,ctxs,max_ctx,kvd", [(32, 8, [700], 832, 1), (32, 8, [1024, 3, 515], 1024, 1), (8, 4, [129], 160, 0), (64, 8, [333, 1000], 1024, 1)])
def test_fused_attention_gpu(oracle, dev, heads, kvh, ctxs, max_ctx, kvd):
    check_attention(oracle, GpuBackend(dev), heads, kvh, ctxs, max_ctx, kvd, n_out=256)


def check_attention_handoff_reuse(O, be, iters, heads=32, kvh=8, ctx=700, max_ctx=832):
    """The split -> last-arriver hand-off inside mrs_dec_attention under the conditions that expose a stale read: the SAME partial buffers, ticket, image and output for
    `iters` consecutive calls, a NEW query (hence new partials at the same addresses) every call, the merging workgroup's caches warm from the call before.  Every call
    must equal the engine-order restatement bit for bit (the ticket is a relaxed agent-scope atomic since round 5: sc1 payload -> drained -> flag, sc1 loads in the merge)."""
    hd, bs = 128, 32
    nq = heads * hd
    rng = np.random.default_rng(99)
    mbs = (max_ctx + bs - 1) // bs
    nblocks = mbs + 1
    kc_np = O.to_bf16_bits((rng.standard_normal((nblocks, kvh, hd // 8, bs, 8)) * 0.7).astype(np.float32))
    vc_np = O.to_bf16_bits(rng.standard_normal((nblocks, kvh, hd, bs)).astype(np.float32))
    bt_np = rng.permutation(nblocks - 1)[:mbs].reshape(1, mbs).astype(np.uint32) + 1
    kc, vc, bt = be.buf(kc_np), be.buf(vc_np), be.buf(bt_np)
    cl = be.buf(np.asarray([ctx], dtype=np.uint32))
    splits = be.sym("mrs_decode_attention_max_splits", [C.c_int], C.c_int)(max_ctx)
    po, pm, pl = be.buf(np.zeros((1, heads, splits, hd), np.float32)), be.buf(np.zeros((1, heads, splits), np.float32)), be.buf(np.zeros((1, heads, splits), np.float32))
    ticket = be.buf(np.zeros(kvh, np.uint32))
    nimg = be.sym("mrs_dec_act_image_bytes", [C.c_int, C.c_int], C.c_size_t)(nq, 1)
    img, got = be.buf(np.zeros(nimg, np.uint8)), be.buf(np.full((1, nq), np.nan, np.float32))
    fn = be.sym("mrs_dec_attention", ATTN2, C.c_int)
    scale = np.float32(1.0 / np.sqrt(np.float32(hd)))
    k, v = _gather_kv(kc_np, vc_np, bt_np[0], ctx, O.from_bf16_bits)
    q = be.buf(np.zeros((1, nq), np.float32))
    for it in range(iters):
        q_np = (rng.standard_normal((1, nq)) * (0.5 + 0.1 * (it % 5))).astype(np.float32)
        q.write(q_np)
        rc = fn(got.ptr, img.ptr, ticket.ptr, po.ptr, pm.ptr, pl.ptr, q.ptr, kc.ptr, vc.ptr, kvh, scale, bt.ptr, cl.ptr, bs, max_ctx, 1, heads, hd, mbs, nq, kvh * hd * bs,
                hd * bs, 1, 0, be.stream)
        assert rc == 1
        res = got.numpy()
        eng = O.attention_engine(q_np[0].reshape(heads, hd), k, v, scale, 1 if mbs <= 64 else (mbs + 63) // 64, 0)  # blocks per split: mrs_dec_attention's rule
        assert np.array_equal(res[0].reshape(heads, hd), eng), ("stale hand-off?", it, float(np.abs(res[0].reshape(heads, hd) - eng).max()))
        assert not ticket.numpy().any()


def test_attention_handoff_reuse_host_emulation(oracle):
    check_attention_handoff_reuse(oracle, HostBackend(), 3, heads=8, kvh=2, ctx=200, max_ctx=224)


@pytest.mark.gpu
def test_attention_handoff_reuse_gpu(oracle, dev):
    check_attention_handoff_reuse(oracle, GpuBackend(dev), 80)


@pytest.mark.gpu
def test_attention_handoff_under_uneven_load_long_context_gpu(oracle, dev):
    """ADVICE round 5 (the relaxed ticket): the same hand-off with 63 splits per kv head (context 4000 of 4096: every XCD takes part, the merging workgroup reads 63 partials
    per head) while a side stream keeps part of the chip busy with large copies -- uneven load and a warm merging CU are the conditions under which a stale partial would
    show (MI355X_MICROARCH.md, "test every hand-off under UNEVEN load").  60 calls on the same buffers, each equal to the engine-order restatement bit for bit."""
    import torch
    side = torch.cuda.Stream()
    a, b = torch.empty(48 << 20, dtype=torch.uint8, device=dev), torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    stop = {"n": 0}

    class LoadedBackend(GpuBackend):
        def sym(self, name, argtypes, restype=None):
            f = super().sym(name, argtypes, restype)
            if name != "mrs_dec_attention":
                return f

            def g(*args):
                with torch.cuda.stream(side):  # ~30 us of copies on a third of the CUs' memory pipes around every attention launch
                    for _ in range(2 + stop["n"] % 3):
                        b.copy_(a, non_blocking=True)
                stop["n"] += 1
                return f(*args)
            return g
    check_attention_handoff_reuse(oracle, LoadedBackend(dev), 60, heads=32, kvh=8, ctx=4000, max_ctx=4096)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("heads,kvh,ctxs,max_ctx,kvd", [(32, 8, [4000, 70], 4096, 1), (32, 32, [300], 512, 1), (8, 1, [2047], 2048, 0), (32, 8, [768] * 8, 1024, 1)])
def test_attention_more_shapes_gpu(oracle, dev, heads, kvh, ctxs, max_ctx, kvd):
    check_attention(oracle, GpuBackend(dev), heads, kvh, ctxs, max_ctx, kvd, n_out=256)


@pytest.mark.parametrize("heads,kvh,ctxs,max_ctx,window", [(32, 8, [700, 40, 257], 832, 256), (8, 2, [2047, 100], 2048, 100), (32, 8, [4096], 4096, 4096)])
def test_attention_sliding_window_host_emulation(oracle, heads, kvh, ctxs, max_ctx, window):
    check_attention(oracle, HostBackend(), heads, kvh, ctxs, max_ctx, 1, window=window)


@pytest.mark.gpu
@pytest.mark.parametrize("heads,kvh,ctxs,max_ctx,window", [(32, 8, [700, 40, 257], 832, 256), (8, 2, [2047, 100], 2048, 100), (32, 8, [4096, 33], 4096, 1024)])
def test_attention_sliding_window_gpu(oracle, dev, heads, kvh, ctxs, max_ctx, window):
    check_attention(oracle, GpuBackend(dev), heads, kvh, ctxs, max_ctx, 1, n_out=256, window=window)


GLU_TOPK = [C.POINTER(Mat), C.POINTER(Mat), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p]


def check_gate_up_topk(O, be, tname, n, k, experts, sels):
    """mrs_dec_gate_up_topk (all top-k experts of a token in one launch) == one mrs_dec_gate_up launch per expert, bit for bit."""
    t = getattr(O, tname)
    pg, pu = _weights(O, t, experts * n, k, 31), _weights(O, t, experts * n, k, 32)
    kg, mg = repack(be, O, t, pg, experts * n, k)
    ku, mu = repack(be, O, t, pu, experts * n, k)
    rng = np.random.default_rng(33)
    x = rng.standard_normal((1, k)).astype(np.float32)
    nw = (1.0 + 0.05 * rng.standard_normal(k)).astype(np.float32)
    xb, nb = be.buf(x), be.buf(nw)
    selb = be.buf(np.array(sels, dtype=np.int32))
    one = be.buf(np.zeros((len(sels), n), dtype=np.float32))
    fn = be.sym("mrs_dec_gate_up", GLU, C.c_int)
    for i in range(len(sels)):
        assert fn(C.byref(mg), C.byref(mu), n, selb.ptr + 4 * i, xb.ptr, k, nb.ptr, 1e-5, 0, one.ptr + 4 * n * i, n, 1, be.stream) == 0
    allk = be.buf(np.full((len(sels), n), np.nan, dtype=np.float32))
    rc = be.sym("mrs_dec_gate_up_topk", GLU_TOPK, C.c_int)(C.byref(mg), C.byref(mu), n, selb.ptr, len(sels), xb.ptr, nb.ptr, 1e-5, 0, allk.ptr, n, be.stream)
    assert rc == 0
    np.testing.assert_array_equal(allk.numpy(), one.numpy())


@pytest.mark.parametrize("tname,n,k,experts,sels", [("Q4_K", 64, 512, 4, [3, 1]), ("Q6_K", 48, 768, 3, [0, 2, 1]), ("Q4_K", 32, 512, 2, [1, 0])])
def test_gate_up_topk_host_emulation(oracle, tname, n, k, experts, sels):
    check_gate_up_topk(oracle, HostBackend(), tname, n, k, experts, sels)


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,experts,sels", [("Q4_K", 14336, 4096, 4, [3, 2]), ("Q6_K", 1000, 4096, 4, [3, 0]), ("Q4_K", 2048, 4096, 8, [1, 6, 3])])
def test_gate_up_topk_gpu(oracle, dev, tname, n, k, experts, sels):
    check_gate_up_topk(oracle, GpuBackend(dev), tname, n, k, experts, sels)


def check_proj_top2(O, be, tname, n, k, experts, sels):
    """mrs_dec_proj_top2 (both experts' down projections of a token in one launch) == two accumulating mrs_dec_proj launches, bit for bit."""
    t = getattr(O, tname)
    pw = _weights(O, t, experts * n, k, 41)
    keep, m = repack(be, O, t, pw, experts * n, k)
    rng = np.random.default_rng(42)
    x = rng.standard_normal((2, k)).astype(np.float32)
    base = rng.standard_normal((1, n)).astype(np.float32)
    rw = rng.uniform(0.2, 0.8, 2).astype(np.float32)
    xb, selb, wb = be.buf(x), be.buf(np.array(sels, dtype=np.int32)), be.buf(rw)
    two, one = be.buf(base.copy()), be.buf(base.copy())
    fn = be.sym("mrs_dec_proj", PROJ, C.c_int)
    for i in range(2):
        assert fn(C.byref(m), n, selb.ptr + 4 * i, xb.ptr + 4 * k * i, k, None, 0.0, two.ptr, n, 1, 0.5 if i == 0 else 1.0, wb.ptr + 4 * i, 1, be.stream) == 0
    T2 = [C.POINTER(Mat), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    assert be.sym("mrs_dec_proj_top2", T2, C.c_int)(C.byref(m), n, selb.ptr, xb.ptr, k, one.ptr, 0.5, wb.ptr, be.stream) == 0
    np.testing.assert_array_equal(one.numpy(), two.numpy())


@pytest.mark.parametrize("tname,n,k,experts,sels", [("Q4_K", 80, 512, 4, [3, 1]), ("Q6_K", 32, 768, 3, [0, 2]), ("Q4_K", 16, 1024, 2, [1, 1])])
def test_proj_top2_host_emulation(oracle, tname, n, k, experts, sels):
    check_proj_top2(oracle, HostBackend(), tname, n, k, experts, sels)


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,experts,sels", [("Q4_K", 4096, 14336, 4, [3, 2]), ("Q6_K", 4096, 14336, 4, [3, 0]), ("Q4_K", 1000, 4096, 8, [1, 6])])
def test_proj_top2_gpu(oracle, dev, tname, n, k, experts, sels):
    check_proj_top2(oracle, GpuBackend(dev), tname, n, k, experts, sels)


def _check_router_norm(be, tokens, E, K, topk):
    """mrs_moe_router_topk_norm (RmsNorm inside the router's workgroup) == mrs_rms_norm_f32 followed by mrs_moe_router_topk: same ids, same weights bit for bit."""
    rng = np.random.default_rng(E + K + tokens)
    h = (rng.standard_normal((tokens, K)) * 1.7).astype(np.float32)
    nw = (1.0 + 0.05 * rng.standard_normal(K)).astype(np.float32)
    gw = (rng.standard_normal((E, K)) * 0.05).astype(np.float32)
    hb, nb, gb = be.buf(h), be.buf(nw), be.buf(gw)
    xn = be.buf(np.zeros((tokens, K), np.float32))
    ids_a, w_a = be.buf(np.full((tokens, topk), -1, np.int32)), be.buf(np.zeros((tokens, topk), np.float32))
    ids_b, w_b = be.buf(np.full((tokens, topk), -1, np.int32)), be.buf(np.zeros((tokens, topk), np.float32))
    be.sym("mrs_rms_norm_f32", [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_int64])(hb.ptr, nb.ptr, xn.ptr, tokens, K, 1e-5, 0)
    R = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert be.sym("mrs_moe_router_topk", R, C.c_int)(xn.ptr, gb.ptr, tokens, E, K, topk, 1, ids_a.ptr, w_a.ptr, None, be.stream) == 0
    RN = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert be.sym("mrs_moe_router_topk_norm", RN, C.c_int)(hb.ptr, nb.ptr, 1e-5, gb.ptr, tokens, E, K, topk, 1, ids_b.ptr, w_b.ptr, be.stream) == 0
    np.testing.assert_array_equal(ids_b.numpy(), ids_a.numpy())
    np.testing.assert_array_equal(w_b.numpy(), w_a.numpy())
    # round 6: the split router (one workgroup per expert and token + the last arriver's top-k), twice on the same scratch (the tickets must return to zero)
    nbytes = be.sym("mrs_moe_router_split_scratch_bytes", [C.c_int, C.c_int], C.c_size_t)(tokens, E)
    scratch = be.buf(np.zeros(nbytes, np.uint8))
    RS = RN[:-1] + [C.c_void_p, C.c_void_p]
    for _ in range(2):
        ids_c, w_c = be.buf(np.full((tokens, topk), -1, np.int32)), be.buf(np.zeros((tokens, topk), np.float32))
        assert be.sym("mrs_moe_router_topk_norm_split", RS, C.c_int)(hb.ptr, nb.ptr, 1e-5, gb.ptr, tokens, E, K, topk, 1, ids_c.ptr, w_c.ptr, scratch.ptr, be.stream) == 0
        np.testing.assert_array_equal(ids_c.numpy(), ids_a.numpy())
        np.testing.assert_array_equal(w_c.numpy(), w_a.numpy())
    assert not scratch.numpy().view(np.uint32).reshape(tokens, E + 1)[:, E].any(), "router tickets must be zero at rest"


@pytest.mark.parametrize("tokens,E,K,topk", [(1, 8, 4096, 2), (3, 4, 512, 2), (2, 16, 1024, 3), (1, 3, 256, 1)])
def test_router_with_fused_norm_host_emulation(tokens, E, K, topk):
    _check_router_norm(HostBackend(), tokens, E, K, topk)


@pytest.mark.gpu
@pytest.mark.parametrize("tokens,E,K,topk", [(1, 8, 4096, 2), (8, 8, 4096, 2), (2, 64, 2048, 6)])
def test_router_with_fused_norm_gpu(dev, tokens, E, K, topk):
    _check_router_norm(GpuBackend(dev), tokens, E, K, topk)
