"""GGUF container (mirror of the reference's archive.rs tests, :1443-2114: synthetic files, dtype catalog, alignment, truncation,
endianness rejection) + the loader: a GGUF file written from a synthetic checkpoint loads into a runner that is bit-identical to
one built directly from the same tensors (GPU)."""
import os
import struct

import numpy as np
import pytest


def _tiny(oracle, tmp_path, alignment=32, drop_output=False):
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.gguf import GgmlDType, archive
    from mistralrs_amd.llama import LlamaConfig
    from oracle import llama_ref
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, vocab_size=512, head_dim=64,
                      rope_theta=10000.0, max_position_embeddings=256, max_batch=1, max_context_len=128)
    types = dict(embd=oracle.Q4_K, q=oracle.Q4_K, k=oracle.Q4_K, v=oracle.Q6_K, o=oracle.Q4_K, gate=oracle.Q4_K, up=oracle.Q4_K, down=oracle.Q6_K, output=oracle.Q6_K)
    w = llama_ref.synth_weights(cfg, types, seed=4)
    md = {"general.architecture": "llama", "llama.embedding_length": 256, "llama.feed_forward_length": 512, "llama.block_count": 2,
          "llama.attention.head_count": 4, "llama.attention.head_count_kv": 2, "llama.attention.layer_norm_rms_epsilon": 1e-5,
          "llama.rope.freq_base": 10000.0, "llama.rope.dimension_count": 64, "llama.context_length": 256, "llama.vocab_size": 512}
    tensors = {}
    for name, val in w.items():
        if drop_output and name == "output.weight":
            continue
        if isinstance(val, tuple):
            t, packed = val
            dt = GgmlDType.from_id(t)
            tensors[name] = (dt, (packed.shape[0], packed.shape[1] // dt.type_size * dt.block_size), packed)
        else:
            tensors[name] = (GgmlDType.F32, (val.size,), val.astype(np.float32).view(np.uint8))
    path = os.path.join(tmp_path, "tiny.gguf")
    archive.write_gguf(path, md, tensors, alignment=alignment)
    return cfg, w, path, archive


@pytest.mark.parametrize("alignment", [32, 64, 256])
def test_roundtrip_and_config(oracle, tmp_path, alignment):
    cfg, w, path, archive = _tiny(oracle, str(tmp_path), alignment)
    with archive.GgufArchive(path) as ar:
        assert ar.version == 3 and ar.alignment == alignment and ar.data_start % alignment == 0
        assert set(ar.tensors) == set(w)
        for name, val in w.items():
            got = ar.tensor_bytes(name)
            want = val[1].reshape(-1) if isinstance(val, tuple) else val.astype(np.float32).view(np.uint8)
            np.testing.assert_array_equal(got, want)
            assert ar.tensors[name].offset % alignment == 0
        c = ar.llama_config(max_context_len=128)
        assert (c.hidden_size, c.intermediate_size, c.num_layers, c.num_heads, c.num_kv_heads, c.vocab_size, c.head_dim) == (256, 512, 2, 4, 2, 512, 64)
        assert c.rope_interleaved and abs(c.rms_eps - 1e-5) < 1e-12 and c.rope_theta == 10000.0
        assert ar.tensors["blk.0.attn_v.weight"].dtype.name == "Q6K" and ar.tensors["blk.0.attn_v.weight"].shape == (128, 256)


def test_rejects_bad_files(oracle, tmp_path):
    cfg, w, path, archive = _tiny(oracle, str(tmp_path))
    raw = open(path, "rb").read()
    def write(b, name):
        p = os.path.join(str(tmp_path), name)
        open(p, "wb").write(b)
        return p
    with pytest.raises(archive.GgufError, match="bad magic"):
        archive.GgufArchive(write(b"GGML" + raw[4:], "magic.gguf"))
    with pytest.raises(archive.GgufError, match="big-endian"):
        archive.GgufArchive(write(raw[:4] + struct.pack(">I", 3) + raw[8:], "be.gguf"))
    with pytest.raises(archive.GgufError, match="unsupported GGUF version"):
        archive.GgufArchive(write(raw[:4] + struct.pack("<I", 7) + raw[8:], "v7.gguf"))
    with pytest.raises(archive.GgufError, match="truncated"):
        archive.GgufArchive(write(raw[: len(raw) - 1000], "trunc.gguf"))
    with pytest.raises(archive.GgufError, match="truncated header"):
        archive.GgufArchive(write(raw[:40], "hdr.gguf"))
    with pytest.raises(archive.GgufError, match="architecture"):
        p2 = os.path.join(str(tmp_path), "arch.gguf")
        archive.write_gguf(p2, {"general.architecture": "gpt2"}, {})
        archive.GgufArchive(p2).llama_config()


@pytest.mark.gpu
@pytest.mark.parametrize("tied", [False, True])
def test_loaded_model_matches_direct_build(oracle, dev, tmp_path, tied):
    import torch
    from mistralrs_amd.gguf import GgmlDType, QTensor
    from mistralrs_amd.llama import Llama
    cfg, w, path, archive = _tiny(oracle, str(tmp_path), drop_output=tied)
    m = archive.load_llama(path, dev, max_context_len=128, max_batch=1)
    direct = Llama(cfg, dev, max_new_tokens=16)
    for name, val in w.items():
        if tied and name == "output.weight":
            val = w["token_embd.weight"]
        if isinstance(val, tuple):
            dt = GgmlDType.from_id(val[0])
            direct.set_tensor(name, QTensor.from_numpy(dt, (val[1].shape[0], val[1].shape[1] // dt.type_size * dt.block_size), val[1], dev))
        else:
            direct.set_tensor(name, torch.from_numpy(val))
    for pos, tok in enumerate([3, 77, 200, 511]):
        m.set_state([tok], [pos]); direct.set_state([tok], [pos])
        assert torch.equal(m.forward_logits(1), direct.forward_logits(1))


@pytest.mark.gpu
def test_mixtral_gguf_loads_and_matches_direct_build(oracle, dev, tmp_path):
    """Mixtral-style file: architecture "llama" + expert_count / expert_used_count, F32 router, rank-3 stacked expert tensors."""
    import torch
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.gguf import GgmlDType, QTensor, archive
    from mistralrs_amd.llama import Llama, LlamaConfig
    from oracle import llama_ref
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, vocab_size=512, head_dim=64,
                      rope_theta=10000.0, max_position_embeddings=256, max_batch=1, max_context_len=128, num_experts=4, num_experts_per_tok=2)
    types = dict(embd=oracle.Q4_K, q=oracle.Q4_K, k=oracle.Q4_K, v=oracle.Q6_K, o=oracle.Q4_K, gate=oracle.Q4_K, up=oracle.Q4_K, down=oracle.Q6_K, output=oracle.Q6_K)
    w = llama_ref.synth_weights(cfg, types, seed=9)
    md = {"general.architecture": "llama", "llama.embedding_length": 256, "llama.feed_forward_length": 512, "llama.block_count": 2,
          "llama.attention.head_count": 4, "llama.attention.head_count_kv": 2, "llama.attention.layer_norm_rms_epsilon": 1e-5,
          "llama.rope.freq_base": 10000.0, "llama.rope.dimension_count": 64, "llama.context_length": 256, "llama.vocab_size": 512,
          "llama.expert_count": 4, "llama.expert_used_count": 2}
    tensors = {}
    for name, val in w.items():
        if isinstance(val, tuple):
            dt = GgmlDType.from_id(val[0])
            rows, cols = val[1].shape[0], val[1].shape[1] // dt.type_size * dt.block_size
            shape = (4, rows // 4, cols) if name.endswith("_exps.weight") else (rows, cols)
            tensors[name] = (dt, shape, val[1])
        else:
            tensors[name] = (GgmlDType.F32, val.shape, val.astype(np.float32).view(np.uint8))
    path = os.path.join(str(tmp_path), "tiny_moe.gguf")
    archive.write_gguf(path, md, tensors)
    ar = archive.GgufArchive(path)
    got_cfg = ar.llama_config(max_context_len=128, max_batch=1)
    assert got_cfg.num_experts == 4 and got_cfg.num_experts_per_tok == 2 and ar.tensors["blk.0.ffn_gate_exps.weight"].shape == (4, 512, 256)
    ar.close()
    m = archive.load_llama(path, dev, max_context_len=128, max_batch=1)
    direct = Llama(cfg, dev, max_new_tokens=16)
    for name, val in w.items():
        if isinstance(val, tuple):
            dt = GgmlDType.from_id(val[0])
            direct.set_tensor(name, QTensor.from_numpy(dt, (val[1].shape[0], val[1].shape[1] // dt.type_size * dt.block_size), val[1], dev))
        else:
            direct.set_tensor(name, torch.from_numpy(val))
    for pos, tok in enumerate([3, 77, 200, 511]):
        m.set_state([tok], [pos]); direct.set_state([tok], [pos])
        assert torch.equal(m.forward_logits(1), direct.forward_logits(1))


def _meta_only(archive, tmp_path, md, name="m.gguf"):
    """a GGUF with metadata `md` and one small tensor (config synthesis does not read tensor data)"""
    from mistralrs_amd.gguf import GgmlDType
    path = os.path.join(str(tmp_path), name)
    archive.write_gguf(path, md, {"token_embd.weight": (GgmlDType.F32, (4, 8), np.zeros(32, np.float32).view(np.uint8))})
    return path


def test_config_synthesis_follows_normal_config_rs(oracle, tmp_path):
    """gguf/normal_config.rs:748-796,897-925: head_dim = attention.key_length when present (else embedding_length / head_count, NOT
    rope.dimension_count), attention.sliding_window (0 = absent), and head_count_kv / context_length / the norm epsilon are REQUIRED -- a file
    without them is rejected with the reference's message instead of getting a silent default (advisor, rounds 1-2)."""
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.gguf import archive
    base = {"general.architecture": "mistral", "mistral.embedding_length": 4096, "mistral.feed_forward_length": 14336, "mistral.block_count": 32,
            "mistral.attention.head_count": 32, "mistral.attention.head_count_kv": 8, "mistral.attention.layer_norm_rms_epsilon": 1e-5,
            "mistral.rope.freq_base": 1000000.0, "mistral.context_length": 32768, "mistral.vocab_size": 32000, "mistral.attention.sliding_window": 4096}
    with archive.GgufArchive(_meta_only(archive, tmp_path, base)) as ar:
        c = ar.llama_config()
        assert (c.head_dim, c.sliding_window, c.num_kv_heads, c.max_position_embeddings, c.rope_theta) == (128, 4096, 8, 32768, 1000000.0)
    with archive.GgufArchive(_meta_only(archive, tmp_path, {**base, "mistral.attention.sliding_window": 0, "mistral.attention.key_length": 128,
                                                            "mistral.rope.dimension_count": 128}, "b.gguf")) as ar:
        c = ar.llama_config()
        assert c.sliding_window is None and c.head_dim == 128
    # key_length wins over embedding_length / head_count (e.g. Mistral-Nemo: 5120 / 32 = 160, key_length 128)
    nemo = {**base, "mistral.embedding_length": 5120, "mistral.attention.key_length": 128, "mistral.rope.dimension_count": 128}
    with archive.GgufArchive(_meta_only(archive, tmp_path, nemo, "c.gguf")) as ar:
        assert ar.llama_config().head_dim == 128
    for missing, msg in (("mistral.attention.head_count_kv", "missing required key `mistral.attention.head_count_kv`"),
                         ("mistral.context_length", "missing required key `mistral.context_length`"),
                         ("mistral.attention.layer_norm_rms_epsilon", "requires `mistral.attention.layer_norm_rms_epsilon` or `mistral.attention.layer_norm_epsilon`")):
        md = {k: v for k, v in base.items() if k != missing}
        with archive.GgufArchive(_meta_only(archive, tmp_path, md, "d.gguf")) as ar:
            with pytest.raises(archive.GgufError, match=msg.replace(".", r"\.").replace("`", "`")):
                ar.llama_config()
    with archive.GgufArchive(_meta_only(archive, tmp_path, {**base, "mistral.attention.head_count_kv": 0}, "e.gguf")) as ar:
        with pytest.raises(archive.GgufError, match="has no non-zero value"):
            ar.llama_config()
    with archive.GgufArchive(_meta_only(archive, tmp_path, {**base, "mistral.embedding_length": 4100}, "f.gguf")) as ar:
        with pytest.raises(archive.GgufError, match="is not an integer"):
            ar.llama_config()
