#!/usr/bin/env python
"""Generate the frozen golden vectors under tests/golden/ (run in the build container, where /root/reference exists).

Expected values come from the REFERENCE'S OWN device functions compiled for the host (oracle/_ref: vec_dot_*_q8_1 and the
affine format spec), NOT from the oracle, so the fixtures pin both the oracle (tests -m "not gpu") and the HIP kernels
(tests -m gpu) to the reference arithmetic even on the GPU box, where /root/reference does not exist.

    python scripts/gen_golden.py        # rewrites tests/golden/*.npz (small: a few KB each)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TYPES = [2, 3, 6, 7, 8, 10, 11, 12, 13, 14]


PA_CASES = {
    "gqa_sinks": dict(hd=128, bs=32, heads=8, kvh=2, ctxs=(1, 31, 33, 530), sinks=True),
    "mha_alibi": dict(hd=64, bs=16, heads=4, kvh=4, ctxs=(1, 15, 17, 200), alibi=True),
    "softcap": dict(hd=96, bs=8, heads=6, kvh=3, ctxs=(3, 8, 9, 70), softcap=6.0),
}


def paged_attn_case(hd, bs, heads, kvh, ctxs, sinks=False, alibi=False, softcap=1.0):
    """Deterministic inputs from index formulas (shared with tests/test_golden.py): f32 K / V caches in the reference layouts (x = 4)."""
    seqs = len(ctxs)
    max_blocks = (max(ctxs) + bs - 1) // bs + 1
    nb = seqs * max_blocks + 1
    while np.gcd(7, nb) != 1:
        nb += 1
    bt = ((np.arange(seqs * max_blocks, dtype=np.int64) * 7 + 3) % nb).reshape(seqs, max_blocks).astype(np.uint32)
    kc = O.patterned(nb * kvh * hd * bs, 5, 1.0).reshape(nb, kvh, hd // 4, bs, 4).astype(np.float32)
    vc = O.patterned(nb * kvh * hd * bs, 11, 1.0).reshape(nb, kvh, hd, bs).astype(np.float32)
    q = O.patterned(seqs * heads * hd, 2, 1.5).reshape(seqs, heads, hd).astype(np.float32)
    al = (0.05 * (1 + np.arange(heads))).astype(np.float32) * np.float32(-1.0) if alibi else None
    sk = (O.patterned(heads, 9, 1.0)).astype(np.float32) if sinks else None
    return q, kc, vc, bt, list(ctxs), al, sk, np.float32(1.0 / np.sqrt(hd))


def main():
    O.build()
    mm = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_mmvq.so"))
    af = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_affine.so"))
    os.makedirs(GOLD, exist_ok=True)
    for t in TYPES:
        rng = np.random.default_rng(1000 + t)
        n, k, b = 6, 1024, 3
        w = O.random_blocks(t, n, k, seed=7000 + t, d_scale=0.02)
        # reference test pattern (fast_mmq.rs:1545-1554) for one row + gaussian rows with very different magnitudes
        x = np.stack([O.patterned(k, 3, 1.0), rng.standard_normal(k) * 0.05, rng.standard_normal(k) * 40.0]).astype(np.float32)
        y = O.quantize_q8_1(x)
        out = np.empty((b, n), dtype=np.float64)
        assert mm.ref_mmvq(t, w.ctypes.data_as(C.c_void_p), n, k, y.ctypes.data_as(C.c_void_p), y.shape[1] // 36, b, out.ctypes.data_as(C.c_void_p)) == 0
        deq = np.empty((n, k), dtype=np.float32)
        assert af.ref_dequantize(t, w.ctypes.data_as(C.c_void_p), n, k, deq.ctypes.data_as(C.c_void_p)) == 0
        _, mag = O.matmul_q8_1_mag(t, w, n, k, y)
        np.savez_compressed(os.path.join(GOLD, f"mmvq_{O.TYPE_NAMES[t]}.npz"), type=np.int32(t), n=np.int32(n), k=np.int32(k), w=w, x=x,
                            y_q8_1=y, out_ref=out, mag=mag, dequant_ref=deq)
        print(f"golden: mmvq_{O.TYPE_NAMES[t]}.npz  out[0,:3]={out[0,:3]}")
    # HQQ: expected values from the reference's own dequantize_* / pack_* kernels run on the host (oracle/_ref/libref_hqq.so)
    hq = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_hqq.so"))
    from oracle import hqq_oracle as H
    for bits in (8, 4, 3, 2, 1):
        rng = np.random.default_rng(4000 + bits)
        p, w = H.PACK[bits], 52
        h = 7
        q = rng.integers(0, 2 ** bits, size=(p * h, w)).astype(np.uint32 if bits == 3 else np.uint8)
        packed = np.zeros((h, w), dtype=np.int32 if bits == 3 else np.uint8)
        assert hq.ref_hqq_pack(bits, q.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p), C.c_size_t(p * h), C.c_size_t(w)) == 0
        scale = (rng.uniform(0.001, 0.05, w) * rng.choice([1.0, -1.0], w)).astype(np.float32)
        zero = rng.uniform(0.0, 2 ** bits - 1.0, w).astype(np.float32)
        out = np.empty((p * h, w), dtype=np.float32)
        assert hq.ref_hqq_dequantize_f32(bits, packed.ctypes.data_as(C.c_void_p), scale.ctypes.data_as(C.c_void_p), zero.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), h, w) == 0
        np.savez_compressed(os.path.join(GOLD, f"hqq_{bits}bit.npz"), bits=np.int32(bits), q=q, packed_ref=packed, scale=scale, zero=zero, out_ref=out)
        print(f"golden: hqq_{bits}bit.npz  out[0,:3]={out[0,:3]}")
    # RoPE + paged-cache data movement: outputs of the reference's own kernels run on the host (oracle/_ref/libref_cache.so)
    ca = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_cache.so"))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(77)
    T, H, KVH, hd, pairs, max_pos = 6, 4, 2, 64, 24, 50  # partial rotary: 48 of 64 dims
    q = rng.standard_normal((T, H, hd)).astype(np.float32)
    k = rng.standard_normal((T, KVH, hd)).astype(np.float32)
    ang = rng.uniform(0, 6.28, (max_pos, pairs))
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    pos = rng.integers(0, max_pos, T).astype(np.uint32)
    rope = {}
    for neox in (0, 1):
        qo, ko = q.copy(), k.copy()
        ca.ref_rotary_f32(vp(qo), vp(ko), vp(cos), vp(sin), vp(pos), neox, hd, T, pairs, H, KVH, C.c_long(H * hd), C.c_long(KVH * hd))
        rope[f"q_out_{neox}"], rope[f"k_out_{neox}"] = qo, ko
    kvh, hd2, nb, bs, x, T2 = 3, 32, 6, 16, 4, 23
    key = rng.standard_normal((T2, kvh, hd2)).astype(np.float32)
    val = rng.standard_normal((T2, kvh, hd2)).astype(np.float32)
    slots = rng.permutation(nb * bs)[:T2].astype(np.int64)
    slots[7] = -1
    kc = np.zeros((nb, kvh, hd2 // x, bs, x), np.float32)
    vc = np.zeros((nb, kvh, hd2, bs), np.float32)
    ca.ref_reshape_and_cache_f32(vp(key), vp(val), vp(kc), vp(vc), vp(slots), T2, kvh, hd2, bs, x, kvh * hd2, kvh * hd2)
    tables = np.stack([rng.permutation(nb)[:3], rng.permutation(nb)[:3]]).astype(np.int32)
    cu = np.array([0, bs + 5, 3 * bs + 5], dtype=np.int32)
    n = int(cu[-1])
    k_out, v_out = np.zeros((n, kvh, hd2), np.float32), np.zeros((n, kvh, hd2), np.float32)
    ca.ref_gather_kv_cache_f32(vp(kc), vp(vc), vp(k_out), vp(v_out), vp(tables), vp(cu), n, 2, bs, 3, kvh, hd2, x)
    np.savez_compressed(os.path.join(GOLD, "cache_ops.npz"), q=q, k=k, cos=cos, sin=sin, pos=pos, **rope, key=key, val=val, slots=slots,
                        kc_ref=kc, vc_ref=vc, tables=tables, cu=cu, k_gather_ref=k_out, v_gather_ref=v_out)
    print("golden: cache_ops.npz")
    # paged attention: outputs of the reference's own v1 / v2 (+ reduce) kernels run on host fibers (oracle/_ref/libref_pa.so).  The
    # inputs are index formulas (paged_attn_case below), so the fixture only stores the reference outputs.
    pa = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_pa.so"))
    outs = {}
    for name, case in PA_CASES.items():
        q, kc, vc, bt, ctxs, alibi, sinks, scale = paged_attn_case(**case)
        seqs, heads, hd = q.shape
        kvh, bs = kc.shape[1], kc.shape[3]
        parts = (max(ctxs) + 511) // 512
        for v2 in (0, 1):
            out = np.zeros((seqs, heads, hd), np.float32)
            es, ml = np.zeros((seqs, heads, parts), np.float32), np.zeros((seqs, heads, parts), np.float32)
            tmp = np.zeros((seqs, heads, parts, hd), np.float32)
            cl = np.array(ctxs, dtype=np.uint32)
            rc = pa.ref_paged_attention_f32(v2, vp(out), vp(es), vp(ml), vp(tmp), vp(q), vp(kc), vp(vc), kvh, C.c_float(scale), C.c_float(case.get("softcap", 1.0)),
                                            vp(bt), vp(cl), bs, max(ctxs), seqs, heads, hd, bt.shape[1], vp(alibi) if alibi is not None else None,
                                            heads * hd, kvh * hd * bs, hd * bs, vp(sinks) if sinks is not None else None)
            assert rc == 0
            outs[f"{name}_v{1 + v2}"] = out
    np.savez_compressed(os.path.join(GOLD, "paged_attn_ref.npz"), **outs)
    print("golden: paged_attn_ref.npz", {k: v.shape for k, v in outs.items()})


if __name__ == "__main__":
    main()
