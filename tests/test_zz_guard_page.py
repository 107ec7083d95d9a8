"""No kernel may read past a weight tensor.  The weights are placed so that they END at a PROT_NONE guard page (host emulation of the product kernels,
in a subprocess: a stray read is a segfault).  Regression: waves of the last workgroup that own no rows used to issue their first prefetch at
first_row >= N -- megabytes past an lm_head-sized tensor; on the MI355X that was a page fault for a [128256, 8192] Q6_K head (70B shapes)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, mmap, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import oracle as O
O.build()
from tests.abi_backends import HostBackend
be = HostBackend()
libc = C.CDLL(None, use_errno=True)
libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
n, K, sym = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
t = O.Q6_K if sym in ("norm_proj", "gemm_q6", "gemm_qi_q6", "mmq_mfma_q6") else O.Q4_K
w = O.random_blocks(t, n, K if sym != "hqq" else 256, seed=1, d_scale=0.02).reshape(-1)
page = mmap.PAGESIZE
keep = []
def guarded(a):
    """Copy of the bytes of `a` that ENDS at a PROT_NONE page (16-byte aligned start when the size allows it)."""
    a = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    size = (a.size + page - 1) // page * page + page
    mm = mmap.mmap(-1, size)
    keep.append(mm)
    base = C.addressof(C.c_char.from_buffer(mm))
    assert libc.mprotect(base + size - page, page, 0) == 0
    p = base + size - page - a.size
    C.memmove(p, a.ctypes.data, a.size)
    return p
dst = guarded(w)
x = np.random.default_rng(0).standard_normal((1, K)).astype(np.float32)
nw = np.ones(K, np.float32)
xb, nb, ob = be.buf(x), be.buf(nw), be.buf(np.full((1, n), np.nan, np.float32))
if sym == "norm_proj":
    fn = be.sym("mrs_decode_norm_proj", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int)
    assert fn(dst, t, n, K, xb.ptr, nb.ptr, 1e-5, ob.ptr, n, 1, be.stream) == 0
    want = O.matmul_q8_1(t, w.reshape(n, -1), n, K, O.quantize_q8_1(O.rms_norm(x, nw, 1e-5)))
elif sym in ("gemm", "gemm_q6"):
    # prompt GEMM (both kernels behind mrs_gemm_q_bf16_multi): weights AND the bf16 activation slabs end at guard pages; ragged N and M tiles
    import torch
    M = 70
    xm = np.random.default_rng(1).standard_normal((M, K)).astype(np.float32)
    slabs = be.buf(np.zeros((K // 64, M, 64), np.uint16))
    xmb = be.buf(xm)
    assert be.sym("mrs_convert_f32_bf16_slabs", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int)(xmb.ptr, K, M, K, slabs.ptr, be.stream) == 0
    sp = guarded(slabs.numpy())
    out = be.buf(np.full((M, n), np.nan, np.float32))
    wp, np_, op, ld = (C.c_void_p * 1)(dst), (C.c_int * 1)(n), (C.c_void_p * 1)(out.ptr), (C.c_int * 1)(n)
    fn = be.sym("mrs_gemm_q_bf16_multi", [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p], C.c_int)
    assert fn(1, wp, np_, op, ld, t, K, sp, M, 0, None, 0, be.stream) == 0
    from tests.util import round_through
    want = round_through(xm, "bf16").astype(np.float64) @ round_through(O.dequantize(t, w.reshape(n, -1), K), "bf16").astype(np.float64).T
    got = out.numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-3 * np.abs(want).max()
    print("guard page intact"); sys.exit(0)
elif sym in ("gemm_qi", "gemm_qi_q6"):
    # prompt GEMM in the decode engine's arithmetic (ext_gemm_qi.hip): the repack reads GGUF blocks that end at a guard page (the padding rows of the last
    # 32-row panel must not be fetched), the GEMM reads an MFMA-order copy and operand buffers that end at guard pages; ragged N and token tiles
    M = 150
    nb2 = be.sym("mrs_gemm_qi_repack_bytes", [C.c_int, C.c_longlong, C.c_longlong], C.c_size_t)(t, n, K)
    assert nb2 > 0
    lay = be.buf(np.zeros(nb2, np.uint8))
    assert be.sym("mrs_gemm_qi_repack", [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p], C.c_int)(dst, t, n, K, lay.ptr, be.stream) == 0
    lp = guarded(lay.numpy())
    xm = np.random.default_rng(1).standard_normal((M, K)).astype(np.float32)
    xp = guarded(xm)
    ab = be.sym("mrs_qi_act_bytes", [C.c_int, C.c_int], C.c_size_t)(M, K)
    actb = be.buf(np.zeros(ab, np.uint8))
    assert be.sym("mrs_qi_quantize", [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int)(
        xp, None, K, None, 0.0, M, K, actb.ptr, None, be.stream) == 0
    ap = guarded(actb.numpy())
    out = be.buf(np.full((M, n), np.nan, np.float32))
    fn = be.sym("mrs_gemm_qi", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int)
    assert fn(lp, t, n, K, ap, M, out.ptr, n, 0, be.stream) == 0
    want = np.concatenate([O.gemv_engine(t, w.reshape(n, -1), n, K, r) for r in xm], axis=0)
    assert np.array_equal(out.numpy(), want)
    print("guard page intact"); sys.exit(0)
elif sym in ("mmq", "mmq_mfma", "mmq_mfma_q6"):
    # prompt-sized drop-in launcher (launch_mmq_gguf_q4_k / q6_k): weights and the block_q8_1_mmq activations end at guard pages; >= 48 columns take the
    # matrix-core kernels (ragged 128 x 128 tiles: surplus rows / columns re-read the last one)
    cols = 37 if sym == "mmq" else 50
    xm = np.random.default_rng(2).standard_normal((cols, K)).astype(np.float32)
    y = O.quantize_q8_1_mmq(xm, O.mmq_layout(t))
    yp = guarded(y)
    D = be.buf(np.full((cols, n), 7.0, np.float32))
    P, L, I = C.c_void_p, C.c_int64, C.c_int
    fn = be.sym("launch_mmq_gguf_q6_k" if sym == "mmq_mfma_q6" else "launch_mmq_gguf_q4_k", [P, P, P, P] + [L] * 5 + [I, I, L, I, I, P])
    fn(None, dst, yp, D.ptr, K, n, cols, K // 256, n, 0, 256, 160 << 10, 64, 0, be.stream)
    want, mag = O.matmul_q8_1_mmq(t, w.reshape(n, -1), n, K, y)
    got = D.numpy().astype(np.float64)
    assert np.isfinite(got).all() and (np.abs(got - want) <= 1e-4 * (mag + 1e-30) + 1e-6).all()
    print("guard page intact"); sys.exit(0)
elif sym == "imoe":
    # expert-indexed GEMV (launch_indexed_moe_forward_q4k_q8_1): the stacked experts end at the guard page, the last task picks the last expert
    E, batch, topk = 3, 2, 2
    we = np.concatenate([O.random_blocks(t, n, K, seed=5 + e, d_scale=0.02) for e in range(E)], axis=0)
    wp = guarded(we)
    idx = np.array([0, 2, 1, 2], dtype=np.uint32)
    xm = np.random.default_rng(3).standard_normal((batch, K)).astype(np.float32)
    y = O.quantize_q8_1(xm)
    yb, ib = be.buf(y), be.buf(idx)
    out = be.buf(np.full((batch * topk, n), np.nan, np.float32))
    fn = be.sym("launch_indexed_moe_forward_q4k_q8_1", [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p])
    fn(wp, yb.ptr, ib.ptr, out.ptr, n, K, batch, topk, O.pad512(K), 1, be.stream)
    got = out.numpy()
    for task in range(batch * topk):
        e = int(idx[task])
        want = O.matmul_q8_1(t, we[e * n:(e + 1) * n], n, K, y[task // topk: task // topk + 1])
        assert np.abs(got[task] - want[0]).max() <= 1e-4 * np.abs(want).max()
    print("guard page intact"); sys.exit(0)
elif sym == "hqq":
    # fused HQQ GEMV: packed weights, scales and zeros end at guard pages
    from oracle import hqq_oracle as H
    wf = (np.random.default_rng(4).standard_normal((n, K)) * 0.05).astype(np.float32)
    wq, scale, zero = H.quantize(wf, 4, 64)
    inv, zr = scale.reshape(-1).astype(np.float32), zero.reshape(-1).astype(np.float32)
    wd = H.dequantize(4, wq, inv, zr).reshape(-1)[: n * K].reshape(n, K)
    fn = be.sym("mrs_hqq_gemv", [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int)
    assert fn(4, 0, guarded(wq), guarded(inv), guarded(zr), None, xb.ptr, K, ob.ptr, n, n, K, 1, be.stream) == 0
    want = x @ wd.T
    got = ob.numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    print("guard page intact"); sys.exit(0)
elif sym == "attn":
    # decode attention (split + merge kernels and the one-launch variant): K / V pages end at guard pages and the sequence's last block IS the last page
    heads, kvh, hd, bs, ctx = 4, 2, 128, 32, n
    mbs = (ctx + bs - 1) // bs
    nblocks = mbs + 1
    rng = np.random.default_rng(5)
    kc = O.to_bf16_bits((rng.standard_normal((nblocks, kvh, hd // 8, bs, 8)) * 0.7).astype(np.float32))
    vc = O.to_bf16_bits(rng.standard_normal((nblocks, kvh, hd, bs)).astype(np.float32))
    kp, vp = guarded(kc), guarded(vc)
    bt = be.buf(np.arange(nblocks - mbs, nblocks, dtype=np.uint32)[::-1].copy().reshape(1, mbs))  # includes block nblocks - 1
    cl = be.buf(np.array([ctx], np.uint32))
    q = be.buf((rng.standard_normal((1, heads * hd)) * 0.5).astype(np.float32))
    splits = be.sym("mrs_decode_attention_max_splits", [C.c_int], C.c_int)(mbs * bs)
    po, pm, pl = be.buf(np.zeros((1, heads, splits, hd), np.float32)), be.buf(np.zeros((1, heads, splits), np.float32)), be.buf(np.zeros((1, heads, splits), np.float32))
    ref, got = be.buf(np.zeros((1, heads * hd), np.float32)), be.buf(np.zeros((1, heads * hd), np.float32))
    A = [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]
    B = [C.c_void_p] * 5 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 10 + [C.c_void_p]
    sc = 1.0 / np.sqrt(hd)
    assert be.sym("mrs_decode_attention_f32_f32_bf16", A, C.c_int)(ref.ptr, pl.ptr, pm.ptr, po.ptr, q.ptr, kp, vp, kvh, sc, bt.ptr, cl.ptr, bs, mbs * bs, 1, heads, hd, mbs, heads * hd,
                                                                   kvh * hd * bs, hd * bs, 1, be.stream) == 0
    # the engine's one-launch split + last-arriver merge (reference fast_exp, run_barrier merge order) vs the round-1 kernels (v_exp_f32, their own merge order): f32 rounding apart
    A2 = [C.c_void_p] * 9 + [C.c_int, C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 11 + [C.c_void_p]
    nimg = be.sym("mrs_dec_act_image_bytes", [C.c_int, C.c_int], C.c_size_t)(heads * hd, 1)
    img2, got2, ticket = be.buf(np.zeros(nimg, np.uint8)), be.buf(np.zeros((1, heads * hd), np.float32)), be.buf(np.zeros(kvh, np.uint32))
    assert be.sym("mrs_dec_attention", A2, C.c_int)(got2.ptr, img2.ptr, ticket.ptr, po.ptr, pm.ptr, pl.ptr, q.ptr, kp, vp, kvh, sc, bt.ptr, cl.ptr, bs, mbs * bs, 1, heads, hd, mbs,
                                                    heads * hd, kvh * hd * bs, hd * bs, 1, 0, be.stream) == 1
    assert np.isfinite(ref.numpy()).all() and np.abs(ref.numpy() - got2.numpy()).max() <= 4e-6 * max(1.0, np.abs(ref.numpy()).max())
    print("guard page intact"); sys.exit(0)
elif sym == "prefill_attn":
    # prompt attention (MFMA flash kernel reading the paged cache with 16-byte loads): pages end at guard pages, the prompt's last block is the last page,
    # the block table itself ends at a guard page too; T = n tokens (ragged last block)
    heads, kvh, hd, bs, T = 4, 2, 128, 32, n
    mbs = (T + bs - 1) // bs
    nblocks = mbs + 1
    rng = np.random.default_rng(6)
    kc = O.to_bf16_bits((rng.standard_normal((nblocks, kvh, hd // 8, bs, 8)) * 0.7).astype(np.float32))
    vc = O.to_bf16_bits(rng.standard_normal((nblocks, kvh, hd, bs)).astype(np.float32))
    btab = np.arange(nblocks - mbs, nblocks, dtype=np.uint32)[::-1].copy()
    q = (rng.standard_normal((T, heads * hd)) * 0.5).astype(np.float32)
    out = be.buf(np.full((T, heads * hd), np.nan, np.float32))
    fn = be.sym("mrs_prefill_attention_f32_bf16", [C.c_void_p] * 5 + [C.c_int] * 10 + [C.c_float, C.c_void_p], C.c_int)
    assert fn(guarded(q), guarded(kc), guarded(vc), guarded(btab), out.ptr, T, 0, heads, kvh, hd, bs, heads * hd, heads * hd, kvh * hd * bs, hd * bs, 1.0 / np.sqrt(hd),
              be.stream) == 0
    assert np.isfinite(out.numpy()).all()
    print("guard page intact"); sys.exit(0)
elif sym == "dec_proj":
    # decode engine: the repacked planes end at the guard page (buffer loads: out-of-range lanes must stay out of range)
    class Mat(C.Structure):
        _fields_ = [("planes", C.c_void_p), ("type", C.c_int), ("n", C.c_longlong), ("k", C.c_longlong)]
    nbytes = be.sym("mrs_dec_repack_bytes", [C.c_int, C.c_longlong, C.c_longlong], C.c_size_t)(t, n, K)
    planes = be.buf(np.zeros(nbytes, np.uint8))
    src = be.buf(w)
    assert be.sym("mrs_dec_repack", [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p], C.c_int)(src.ptr, t, n, K, planes.ptr, be.stream) == 0
    m = Mat(guarded(planes.numpy()), t, n, K)
    PROJ = [C.POINTER(Mat), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    assert be.sym("mrs_dec_proj", PROJ, C.c_int)(C.byref(m), n, None, xb.ptr, K, None, 0.0, ob.ptr, n, 0, 1.0, None, 1, be.stream) == 0
    want = O.matmul_cpu(t, w.reshape(n, -1), n, K, x)
    got = ob.numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    print("guard page intact"); sys.exit(0)
else:
    kp = O.pad512(K)
    y = be.buf(O.quantize_q8_1(x))
    fn = be.sym("launch_mmvq_gguf_q4_k_f32_plain", [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p])
    fn(dst, y.ptr, ob.ptr, K, n, kp // 32, n, 1, be.stream)
    want = O.matmul_q8_1(t, w.reshape(n, -1), n, K, O.quantize_q8_1(x))
got = ob.numpy()
assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
print("guard page intact")
'''


@pytest.mark.parametrize("n,k,sym,env", [(600, 512, "norm_proj", {"MRS_PROJ_WGS": "2"}), (129, 1024, "mmvq", {}), (2049, 256, "mmvq", {}),
                                         (200, 512, "gemm", {"MRS_GEMM_VARIANT": "1"}), (200, 512, "gemm", {"MRS_GEMM_VARIANT": "0"}), (130, 256, "gemm_q6", {}), (200, 512, "gemm_qi", {}), (130, 256, "gemm_qi_q6", {}),
                                         (70, 512, "dec_proj", {}), (2049, 256, "dec_proj", {}),
                                         (70, 512, "mmq", {}), (129, 256, "mmq", {}), (70, 512, "mmq_mfma", {}), (129, 256, "mmq_mfma", {}), (70, 512, "mmq_mfma_q6", {}), (50, 512, "imoe", {}), (128, 256, "hqq", {}), (64, 2064, "hqq", {}),
                                         (96, 256, "attn", {}), (33, 256, "attn", {}), (70, 256, "prefill_attn", {}), (33, 256, "prefill_attn", {})])
def test_row_less_waves_do_not_read_past_the_tensor(n, k, sym, env):
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, str(n), str(k), sym], capture_output=True, text=True, timeout=600,
                       env={**os.environ, **env})
    assert r.returncode == 0 and "guard page intact" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
