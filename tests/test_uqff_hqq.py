"""UQFF entries of HQQ layers (mistral.rs_amd/uqff.py `serialize_hqq_layer` / `UqffReader.load_hqq_layer`, hqq.py `HqqLayer.serialize_uqff` / `from_uqff`;
reference: mistralrs-quant/src/hqq/mod.rs:797-826 `from_uqff`, :1268-1336 `uqff_type` / `serialize_uqff` / `isq_type_from_uqff`, test :1516-1550
`hqq4_uqff_embedding_matches_dequantized_gather`).  Byte / dtype logic on CPU tensors; the device round trip is the `-m gpu` test."""
import numpy as np
import pytest


def _parts(bits, dtype):
    import torch
    rng = np.random.default_rng(bits)
    n, k, g = 96, 32, 64
    packed_rows = {8: g, 4: g // 2}[bits]
    w_q = torch.from_numpy(rng.integers(0, 256, (packed_rows, n * k // g), dtype=np.uint8))
    scales = torch.from_numpy(rng.uniform(0.01, 0.1, (1, n * k // g)).astype(np.float32)).to(dtype)
    zeros = torch.from_numpy(rng.uniform(0, 15, (1, n * k // g)).astype(np.float32)).to(dtype)
    return w_q, scales, zeros, (n, k)


@pytest.mark.parametrize("bits,dt", [(4, "bfloat16"), (8, "float16"), (4, "float32")])
def test_hqq_uqff_entries_round_trip(tmp_path, bits, dt):
    """Key names, scalar encodings and tensor dtypes as `serialize_uqff` writes them; 0 optimization steps <-> None; bias optional; bf16 scales survive."""
    import torch
    from safetensors import safe_open
    from mistralrs_amd import uqff
    dtype = getattr(torch, dt)
    w_q, scales, zeros, shape = _parts(bits, dtype)
    bias = torch.arange(96, dtype=torch.float32).to(dtype) if bits == 4 else None
    ent = uqff.serialize_hqq_layer("model.layers.0.q", w_q, scales, zeros, shape, bits, 64, 0, None if bits == 8 else 2, False, True, bias)
    assert list(ent)[:11] == [f"model.layers.0.q.weight{s}" for s in (".format", "", ".scales", ".zeros", ".shape", ".bits", ".group_size", ".axis",
                                                                      ".optimization_steps", ".round_zeros", ".channel_wise")]
    p = str(tmp_path / "h.uqff")
    uqff.write(p, ent)
    with safe_open(p, framework="pt") as f:   # what the reference's reader sees: dtypes and scalar shapes
        assert f.get_tensor("model.layers.0.q.weight.format").dtype == torch.uint8 and int(f.get_tensor("model.layers.0.q.weight.format")) == 2
        assert f.get_tensor("model.layers.0.q.weight.bits").dtype == torch.uint8 and f.get_tensor("model.layers.0.q.weight.bits").dim() == 0
        assert f.get_tensor("model.layers.0.q.weight.group_size").dtype == torch.uint32
        assert f.get_tensor("model.layers.0.q.weight.scales").dtype == dtype
        assert f.get_tensor("uqff.version.major").dtype == torch.uint32
    r = uqff.UqffReader(p)
    assert r.serde_type("model.layers.0.q") == uqff.SERDE_HQQ and r.isq_type("model.layers.0.q") == f"HQQ{bits}"
    d = r.load_hqq_layer("model.layers.0.q")
    assert torch.equal(d.w_q, w_q) and torch.equal(d.scales, scales) and torch.equal(d.zeros, zeros) and d.scales.dtype == dtype
    assert d.w_shape == shape and (d.bits, d.group_size, d.axis) == (bits, 64, 0)
    assert d.optimization_steps == (None if bits == 8 else 2) and d.round_zeros is False and d.channel_wise is True
    assert (d.bias is None) == (bias is None) and (bias is None or torch.equal(d.bias, bias))
    from mistralrs_amd.distributed import Shard
    with pytest.raises(ValueError, match="do not support sharded loading"):
        r.load_hqq_layer("model.layers.0.q", Shard(0, 0, 2))
    assert r.load_hqq_layer("model.layers.0.q", Shard(0, 0, 1)).bits == bits   # world size 1 is the full load
    with pytest.raises(ValueError, match="not a GGUF-quantized layer"):
        r.load_gguf_layer("model.layers.0.q")


def test_hqq_uqff_refuses_widths_without_a_uqff_type():
    """hqq/mod.rs:1268-1279: 3 / 2 / 1 bit have no UQFF type."""
    import torch
    from mistralrs_amd import uqff
    w_q, scales, zeros, shape = _parts(4, torch.float32)
    for bits in (3, 2, 1):
        with pytest.raises(ValueError, match="unsupported HQQ bit width"):
            uqff.serialize_hqq_layer("x", w_q, scales, zeros, shape, bits, 64, 0, None, False, True)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [4, 8])
def test_hqq_layer_uqff_round_trip_on_device(dev, tmp_path, bits):
    """hqq/mod.rs:1516-1550 restated: quantize the reference's test weight (sin / cos pattern, 96 x 32, group 64, 2 optimisation steps), write the layer as
    UQFF, read it back onto the device: same packed tensors, and the reloaded layer dequantizes to the same values."""
    import torch
    from mistralrs_amd import uqff
    from mistralrs_amd.hqq import HqqConfig, HqqLayer
    idx = torch.arange(96 * 32, dtype=torch.float32)
    w = (torch.sin(idx * 0.017) + torch.cos(idx * 0.013) * 0.25).reshape(96, 32).to(dev)
    layer = HqqLayer.quantize(w, HqqConfig(bits=bits, group_size=64, axis=0, optimization_steps=2, round_zeros=False, channel_wise=True))
    assert layer.uqff_type() == f"HQQ{bits}"
    p = str(tmp_path / "hqq.uqff")
    uqff.write(p, layer.serialize_uqff("test.embedding"))
    back = HqqLayer.from_uqff(uqff.UqffReader(p), "test.embedding", dev)
    assert torch.equal(back.w_q, layer.w_q) and torch.equal(back.scales, layer.scales) and torch.equal(back.zeros, layer.zeros)
    assert back.cfg == layer.cfg and back.w_shape == layer.w_shape
    assert float((back.dequantize() - layer.dequantize()).abs().max()) <= 1e-6
