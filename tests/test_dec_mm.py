"""Batched decode on the matrix cores (csrc/ext_dec_mm.hip): mrs_dec_mm_proj / mrs_dec_mm_gate_up / mrs_dec_mm_qkv on the MFMA-order copy of the weights
(mrs_gemm_qi_repack) against the vector-ALU launches of the decode engine on the decode-layout copy (mrs_dec_proj_img / mrs_dec_gate_up_img / mrs_dec_qkv_img): the
SAME BITS for every column count 1..8 -- a token's result must not depend on which kernel produced it -- and, for the plain projection, bit equality with the
engine-order restatement of the oracle (oracle/cpu_path_oracle.c orc_gemv_engine).  Reference role: MMVQ's batch 1..8 from one pass over the weights
(mistralrs-quant/kernels/mmvq_gguf/mmvq_gguf.cu:724-792, gguf/fast_mmvq.rs:52).  Same bodies on the wave64 host emulation (CPU suite) and on the MI355X."""
import ctypes as C

import numpy as np
import pytest

from tests.abi_backends import GpuBackend, HostBackend
from tests.test_dec_engine import ACT_IMG, GLU_IMG, PROJ_IMG_, QKV_IMG, Mat, _act_image, _weights, repack

MM_PROJ = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
MM_GLU = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
MM_QKV = [C.c_void_p, C.c_int, C.c_int] * 3 + [C.c_int, C.c_void_p] + [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]


def qi_repack(be, t, packed, n, k):
    nbytes = be.sym("mrs_gemm_qi_repack_bytes", [C.c_int, C.c_longlong, C.c_longlong], C.c_size_t)(t, n, k)
    assert nbytes > 0
    src = be.buf(np.ascontiguousarray(packed).reshape(-1))
    dst = be.buf(np.zeros(nbytes, dtype=np.uint8))
    assert be.sym("mrs_gemm_qi_repack", [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p], C.c_int)(src.ptr, t, n, k, dst.ptr, be.stream) == 0
    return dst


def check_mm_proj(O, be, tname, n, k, b, norm, mode=1):
    t = getattr(O, tname)
    packed = _weights(O, t, n, k, 61)
    keep, m = repack(be, O, t, packed, n, k)
    qi = qi_repack(be, t, packed, n, k)
    rng = np.random.default_rng(62)
    x = rng.standard_normal((b, k)).astype(np.float32)
    x[b - 1, :256] = 0.0  # an all-zero activation block
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32)
    base = rng.standard_normal((b, n)).astype(np.float32)
    xb, nb = be.buf(x), (be.buf(nw) if norm else None)
    o1, o2 = be.buf(base.copy()), be.buf(base.copy())
    img = _act_image(be, t, xb, nb, k, b)
    assert be.sym("mrs_dec_mm_supported", [C.c_int, C.c_int, C.c_int], C.c_int)(t, k, b) == 1
    assert be.sym("mrs_dec_proj_img", PROJ_IMG_, C.c_int)(C.byref(m), n, img.ptr, o1.ptr, n, mode, 0.5, b, be.stream) == 0
    assert be.sym("mrs_dec_mm_proj", MM_PROJ, C.c_int)(qi.ptr, t, n, k, img.ptr, o2.ptr, n, mode, 0.5, b, be.stream) == 0
    a1, a2 = o1.numpy(), o2.numpy()
    assert np.array_equal(a1, a2), (tname, n, k, b, norm, float(np.abs(a1 - a2).max()))
    xe = O.rms_norm_engine(x, nw, 1e-5) if norm else x
    eng = np.concatenate([O.gemv_engine(t, packed, n, k, r) for r in xe], axis=0)
    if mode:
        eng = base * np.float32(0.5) + eng * np.float32(1.0)
    assert np.array_equal(a2, eng), (tname, n, k, b, norm, "engine-order oracle")


def check_mm_gate_up(O, be, tname, n, k, b):
    t = getattr(O, tname)
    pg, pu = _weights(O, t, n, k, 71), _weights(O, t, n, k, 72)
    kg, mg = repack(be, O, t, pg, n, k)
    ku, mu = repack(be, O, t, pu, n, k)
    qg, qu = qi_repack(be, t, pg, n, k), qi_repack(be, t, pu, n, k)
    rng = np.random.default_rng(73)
    x = rng.standard_normal((b, k)).astype(np.float32)
    nw = (1.0 + 0.1 * rng.standard_normal(k)).astype(np.float32)
    xb, nb = be.buf(x), be.buf(nw)
    o1, o2 = be.buf(np.zeros((b, n), np.float32)), be.buf(np.zeros((b, n), np.float32))
    img = _act_image(be, t, xb, nb, k, b)
    assert be.sym("mrs_dec_gate_up_img", GLU_IMG, C.c_int)(C.byref(mg), C.byref(mu), n, img.ptr, 0, o1.ptr, n, b, be.stream) == 0
    assert be.sym("mrs_dec_mm_gate_up", MM_GLU, C.c_int)(qg.ptr, qu.ptr, t, n, k, img.ptr, 0, o2.ptr, n, b, be.stream) == 0
    a1, a2 = o1.numpy(), o2.numpy()
    assert np.array_equal(a1, a2) and np.isfinite(a2).all() and np.count_nonzero(a2) > 0, (tname, n, k, b, float(np.abs(a1 - a2).max()))


def check_mm_qkv(O, be, tq, tv, heads, kvh, k, b, kvd=1):
    hd, bs = 128, 32
    nq, nkv = heads * hd, kvh * hd
    Tq, Tv = getattr(O, tq), getattr(O, tv)
    pq, pk, pv = _weights(O, Tq, nq, k, 81), _weights(O, Tq, nkv, k, 82), _weights(O, Tv, nkv, k, 83)
    kq, mq = repack(be, O, Tq, pq, nq, k)
    kk, mk = repack(be, O, Tq, pk, nkv, k)
    kv, mv = repack(be, O, Tv, pv, nkv, k)
    qq, qk, qv = qi_repack(be, Tq, pq, nq, k), qi_repack(be, Tq, pk, nkv, k), qi_repack(be, Tv, pv, nkv, k)
    rng = np.random.default_rng(84)
    x = rng.standard_normal((b, k)).astype(np.float32)
    nw = (1.0 + 0.05 * rng.standard_normal(k)).astype(np.float32)
    inv = 1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))
    fr = np.arange(64, dtype=np.float32)[:, None] * inv[None, :]
    pos = np.array([5 + 7 * i for i in range(b)], dtype=np.int32)
    slots = np.array([(2 * i) * bs + int(pos[i]) % bs for i in range(b)], dtype=np.int64)
    if b > 1:
        slots[b - 1] = -1  # a padded sequence: nothing is written for it
    xb, nb, sb, pb, cb, snb = be.buf(x), be.buf(nw), be.buf(slots), be.buf(pos), be.buf(np.cos(fr).astype(np.float32)), be.buf(np.sin(fr).astype(np.float32))
    img = _act_image(be, Tq, xb, nb, k, b)
    res = []
    for mm in (False, True):
        kc, vc = be.buf(np.zeros((2 * b, kvh, hd // 8, bs, 8), dtype=np.uint16)), be.buf(np.zeros((2 * b, kvh, hd, bs), dtype=np.uint16))
        qb = be.buf(np.zeros((b, nq), dtype=np.float32))
        if mm:
            assert be.sym("mrs_dec_mm_qkv", MM_QKV, C.c_int)(qq.ptr, Tq, nq, qk.ptr, Tq, nkv, qv.ptr, Tv, nkv, k, img.ptr, qb.ptr, kc.ptr, vc.ptr, sb.ptr, pb.ptr, cb.ptr,
                                                              snb.ptr, hd, hd // 2, kvh, bs, kvd, b, be.stream) == 0
        else:
            assert be.sym("mrs_dec_qkv_img", QKV_IMG, C.c_int)(C.byref(mq), C.byref(mk), C.byref(mv), img.ptr, qb.ptr, kc.ptr, vc.ptr, sb.ptr, pb.ptr, cb.ptr, snb.ptr, hd,
                                                                hd // 2, kvh, bs, kvd, b, 0, be.stream) == 0
        res.append((qb.numpy(), kc.numpy(), vc.numpy()))
    for a1, a2 in zip(*res):
        assert np.array_equal(a1, a2)
    assert np.count_nonzero(res[1][0]) > 0 and np.count_nonzero(res[1][1]) > 0 and np.count_nonzero(res[1][2]) > 0


@pytest.mark.parametrize("tname,n,k,b,norm", [("Q4_K", 40, 1024, 3, True), ("Q6_K", 32, 512, 8, False), ("Q5_K", 33, 768, 5, True), ("Q8_0", 64, 512, 4, True),
                                              ("Q4_K", 32, 3584, 8, False), ("Q4_K", 16, 256, 1, False), ("Q6_K", 64, 1792, 2, False), ("Q4_K", 96, 2560, 8, True)])
def test_mm_proj_host_emulation(oracle, tname, n, k, b, norm):
    check_mm_proj(oracle, HostBackend(), tname, n, k, b, norm)  # (n = 33 / 40 / 16: a ragged last panel; k = 768: runs of one superblock, the fourth empty; k = 256: three empty runs)


def test_mm_store_gate_up_qkv_host_emulation(oracle):
    be = HostBackend()
    check_mm_proj(oracle, be, "Q6_K", 96, 512, 2, True, mode=0)
    check_mm_gate_up(oracle, be, "Q4_K", 64, 512, 4)
    check_mm_gate_up(oracle, be, "Q8_0", 32, 512, 2)
    check_mm_qkv(oracle, be, "Q4_K", "Q6_K", 2, 1, 512, 3)
    check_mm_qkv(oracle, be, "Q5_K", "Q5_K", 2, 2, 256, 2, kvd=0)


def test_mm_entry_points_refuse_what_they_cannot_do_host_emulation(oracle):
    O, be = oracle, HostBackend()
    k, b, n = 512, 2, 32
    t = O.Q4_K
    packed = _weights(O, t, n, k, 3)
    qi = qi_repack(be, t, packed, n, k)
    x = be.buf(np.ones((b, k), np.float32))
    img = _act_image(be, t, x, None, k, b)
    out = be.buf(np.zeros((b, n), np.float32))
    sup = be.sym("mrs_dec_mm_supported", [C.c_int, C.c_int, C.c_int], C.c_int)
    assert sup(t, k, 8) == 1 and sup(t, k, 9) == 0 and sup(t, 300, 2) == 0 and sup(0, k, 2) == 0 and sup(t, 28672, 8) == 0 and sup(t, 28672, 4) == 1 and sup(t, 14336, 8) == 1
    fn = be.sym("mrs_dec_mm_proj", MM_PROJ, C.c_int)
    assert fn(qi.ptr, t, n, k, img.ptr, out.ptr, n, 0, 1.0, b, be.stream) == 0
    for bad in ((None, t, n, k, img.ptr, out.ptr, n, 0, 1.0, b), (qi.ptr, 0, n, k, img.ptr, out.ptr, n, 0, 1.0, b), (qi.ptr, t, n, 300, img.ptr, out.ptr, n, 0, 1.0, b),
                (qi.ptr, t, n, k, None, out.ptr, n, 0, 1.0, b), (qi.ptr, t, n, k, img.ptr, None, n, 0, 1.0, b), (qi.ptr, t, n, k, img.ptr, out.ptr, n, 0, 1.0, 9),
                (qi.ptr, t, n, k, img.ptr + 4, out.ptr, n, 0, 1.0, b)):
        assert fn(*bad, be.stream) == -1, bad
    # gate / up of different reduction formats share no activation image; q / k / v must agree on it too
    assert be.sym("mrs_dec_mm_qkv", MM_QKV, C.c_int)(qi.ptr, t, n, qi.ptr, t, n, qi.ptr, O.Q8_0, n, k, img.ptr, out.ptr, out.ptr, out.ptr, out.ptr, out.ptr, out.ptr, out.ptr,
                                                      128, 64, 1, 32, 1, b, be.stream) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("tname,n,k,b,norm", [("Q4_K", 4096, 4096, 8, True), ("Q4_K", 4096, 14336, 8, False), ("Q6_K", 4096, 14336, 5, False), ("Q6_K", 32064, 4096, 8, True),
                                              ("Q5_K", 1000, 4096, 3, True), ("Q8_0", 1024, 4096, 8, True), ("Q4_K", 512, 28672, 4, False), ("Q4_K", 4096, 4096, 1, True)])
def test_mm_proj_gpu(oracle, dev, tname, n, k, b, norm):
    check_mm_proj(oracle, GpuBackend(dev), tname, n, k, b, norm)


@pytest.mark.gpu
def test_mm_store_gate_up_qkv_gpu(oracle, dev):
    be = GpuBackend(dev)
    check_mm_proj(oracle, be, "Q6_K", 8192, 4096, 8, True, mode=0)
    check_mm_gate_up(oracle, be, "Q4_K", 14336, 4096, 8)
    check_mm_gate_up(oracle, be, "Q8_0", 2048, 4096, 3)
    check_mm_gate_up(oracle, be, "Q6_K", 1024, 4096, 2)
    check_mm_qkv(oracle, be, "Q4_K", "Q6_K", 32, 8, 4096, 8)
    check_mm_qkv(oracle, be, "Q4_K", "Q4_K", 32, 8, 4096, 4, kvd=0)
    check_mm_qkv(oracle, be, "Q8_0", "Q8_0", 8, 2, 2048, 2)
