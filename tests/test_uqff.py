"""UQFF read / write of GGUF-quantized layers (mistralrs_amd/uqff.py; reference mistralrs-quant/src/uqff/mod.rs, gguf/mod.rs:260-280,755-793):
on-disk entry names / dtypes / shapes read back from the raw safetensors header, round trip of the packed bytes, tensor-parallel shards
(rows, block-aligned columns, stacked experts) equal to slices of the dequantized weight, bias handling, and the reference's error cases."""
import json
import struct

import numpy as np
import pytest


def _mods():
    from mistralrs_amd import uqff
    from mistralrs_amd.distributed import Shard
    from mistralrs_amd.gguf import GgmlDType
    return uqff, Shard, GgmlDType


def _header(path):
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        return json.loads(f.read(n))


def test_on_disk_layout_and_round_trip(oracle, tmp_path):
    uqff, Shard, G = _mods()
    n, k = 6, 512
    packed = oracle.random_blocks(oracle.Q4_K, n, k, seed=3)
    bias = np.arange(n, dtype=np.float32)
    p = str(tmp_path / "m.uqff")
    layers = {}
    layers.update(uqff.serialize_gguf_layer("model.layers.0.self_attn.q_proj", G.Q4K, (n, k), packed, bias))
    layers.update(uqff.serialize_gguf_layer("lm_head", G.Q6K, (4, 256), oracle.random_blocks(oracle.Q6_K, 4, 256, seed=4)))
    uqff.write(p, layers)
    h = _header(p)
    assert h["uqff.version.major"] == {"dtype": "U32", "shape": [], "data_offsets": h["uqff.version.major"]["data_offsets"]}
    pre = "model.layers.0.self_attn.q_proj"
    assert h[f"{pre}.weight.format"]["dtype"] == "U8" and h[f"{pre}.weight.format"]["shape"] == []
    assert h[f"{pre}.weight"]["dtype"] == "U8" and h[f"{pre}.weight"]["shape"] == [n * k // 256 * 144]
    assert h[f"{pre}.weight.dtype"]["dtype"] == "U32" and h[f"{pre}.weight.shape"] == {**h[f"{pre}.weight.shape"], "dtype": "U32", "shape": [2]}
    assert h[f"{pre}.bias"]["dtype"] == "F32" and "lm_head.bias" not in h
    r = uqff.UqffReader(p)
    assert r.version == (1, 2, 0) and r.serde_type(pre) == uqff.SERDE_GGUF
    assert r.load_u32_scalar(f"{pre}.weight.dtype") == 12 and r.load_u32_vec(f"{pre}.weight.shape") == [n, k]
    lay = r.load_gguf_layer(pre)
    assert lay.dtype == G.Q4K and lay.shape == (n, k) and np.array_equal(lay.packed, packed.reshape(-1))
    assert np.array_equal(lay.bias, bias) and lay.bias_mode == "full"
    assert r.load_gguf_layer("lm_head").dtype == G.Q6K and r.load_gguf_layer("lm_head").bias is None


@pytest.mark.parametrize("tname", ["q4_k", "q6_k", "q8_0", "q5_1"])
def test_shards_equal_slices_of_the_dequantized_weight(oracle, tmp_path, tname):
    uqff, Shard, G = _mods()
    t = {v: k for k, v in oracle.TYPE_NAMES.items()}[tname]
    dt = G.from_id(t)
    n, k = 8, 1024
    packed = oracle.random_blocks(t, n, k, seed=t)
    full = oracle.dequantize(t, packed, k).reshape(n, k)
    bias = np.linspace(-1, 1, n).astype(np.float32)
    p = str(tmp_path / "s.uqff")
    uqff.write(p, uqff.serialize_gguf_layer("w", dt, (n, k), packed, bias))
    r = uqff.UqffReader(p)
    for rank in range(4):  # column-parallel: rows, bias narrowed with them
        lay = r.load_gguf_layer("w", Shard(dim=0, rank=rank, world_size=4))
        assert lay.shape == (2, k) and lay.bias_mode == "narrow" and np.array_equal(lay.bias, bias[2 * rank:2 * rank + 2])
        assert np.array_equal(oracle.dequantize(t, lay.packed, k).reshape(2, k), full[2 * rank:2 * rank + 2])
    world = 4 if dt.block_size == 256 else 8
    for rank in range(world):  # row-parallel: block-aligned columns, bias left to the caller (added after the all-reduce)
        lay = r.load_gguf_layer("w", Shard(dim=1, rank=rank, world_size=world))
        kk = k // world
        assert lay.shape == (n, kk) and lay.bias is None and lay.bias_mode == "skip"
        assert np.array_equal(oracle.dequantize(t, lay.packed, kk).reshape(n, kk), full[:, rank * kk:(rank + 1) * kk])
    lay = r.load_gguf_layer("w", Shard(dim=0, offset=3, length=2))  # Shard::Offset (KV-head replication)
    assert np.array_equal(oracle.dequantize(t, lay.packed, k).reshape(2, k), full[3:5])
    assert r.load_gguf_layer("w", Shard(dim=0, offset=0, length=n)).shape == (n, k)  # the whole range = a full load


def test_matches_distributed_shard_qtensor(oracle, tmp_path):
    """The UQFF loader and the in-memory TP sharding cut the same bytes."""
    import torch
    uqff, Shard, G = _mods()
    from mistralrs_amd.distributed import shard_qtensor
    from mistralrs_amd.gguf import QTensor
    n, k = 12, 768
    packed = oracle.random_blocks(oracle.Q5_K, n, k, seed=9)
    p = str(tmp_path / "d.uqff")
    uqff.write(p, uqff.serialize_gguf_layer("w", G.Q5K, (n, k), packed))
    r = uqff.UqffReader(p)
    w = QTensor(G.Q5K, (n, k), torch.from_numpy(packed.reshape(-1).copy()))
    for sh in (Shard(0, 1, 3), Shard(1, 2, 3), Shard(dim=0, offset=4, length=4)):
        assert np.array_equal(r.load_gguf_layer("w", sh).packed, shard_qtensor(w, sh).data.numpy())


def test_stacked_experts_shard_on_the_ffn_dim(oracle, tmp_path):
    uqff, Shard, G = _mods()
    e, n, k = 3, 4, 256
    packed = oracle.random_blocks(oracle.Q4_K, e * n, k, seed=2)
    full = oracle.dequantize(oracle.Q4_K, packed, k).reshape(e, n, k)
    p = str(tmp_path / "e.uqff")
    uqff.write(p, uqff.serialize_gguf_layer("experts.gate", G.Q4K, (e, n, k), packed))
    lay = uqff.UqffReader(p).load_gguf_layer("experts.gate", Shard(dim=1, rank=1, world_size=2))
    assert lay.shape == (e, 2, k)
    assert np.array_equal(oracle.dequantize(oracle.Q4_K, lay.packed, k).reshape(e, 2, k), full[:, 2:4])


def test_error_cases(oracle, tmp_path):
    uqff, Shard, G = _mods()
    packed = oracle.random_blocks(oracle.Q4_K, 6, 512, seed=1)
    p = str(tmp_path / "x.uqff")
    uqff.write(p, uqff.serialize_gguf_layer("w", G.Q4K, (6, 512), packed))
    r = uqff.UqffReader(p)
    with pytest.raises(ValueError, match="not divisible by world size 4"):
        r.load_gguf_layer("w", Shard(0, 0, 4))
    with pytest.raises(ValueError, match="block alignment"):
        r.load_gguf_layer("w", Shard(dim=1, offset=128, length=256))
    with pytest.raises(ValueError, match="exceeds dimension 0"):
        r.load_gguf_layer("w", Shard(dim=0, offset=4, length=4))
    with pytest.raises(ValueError, match="outside world size"):
        uqff.shard_range(Shard(0, 2, 2), [6, 512])
    with pytest.raises(ValueError, match="Cannot shard dimension 2"):
        uqff.shard_range(Shard(2, 0, 2), [6, 512])
    with pytest.raises(ValueError, match="Missing `v.weight.format`"):
        r.load_gguf_layer("v")
    with pytest.raises(ValueError, match="only 10 are available"):
        uqff.slice_blocked_data(np.zeros(10, np.uint8), [6, 512], 256, 144, 0, 0, 3)
    assert uqff.bias_shard(None, 2) == "full" and uqff.bias_shard((1, 0, 4), 2) == "skip" and uqff.bias_shard((0, 2, 2), 2) == ("narrow", 0, 2, 2)
    assert uqff.bias_shard((0, 0, 1), 1) == "skip" and uqff.bias_shard(None, 1) == "full"
    from safetensors.numpy import save_file
    q = str(tmp_path / "old.uqff")
    save_file({"uqff.version.major": np.array(2, np.uint32), "uqff.version.minor": np.array(0, np.uint32), "uqff.version.patch": np.array(0, np.uint32)}, q)
    with pytest.raises(ValueError, match="major version 2"):
        uqff.UqffReader(q)
