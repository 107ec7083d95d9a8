"""Run one 512-token prefill of the bench model under rocprofv3."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import mistralrs_amd
from mistralrs_amd.llama import LlamaConfig
dev = torch.device("cuda:0")
cfg = LlamaConfig.llama3_8b(max_batch=1, max_context_len=832, max_position_embeddings=8192)
m = bench.build_model(cfg, dev, seed=0, max_new_tokens=16)
prompt = [(1000 + i % 2048) % cfg.vocab_size for i in range(512)]
m.prefill(prompt, 0); torch.cuda.synchronize()
m.prefill(prompt, 0); torch.cuda.synchronize()
