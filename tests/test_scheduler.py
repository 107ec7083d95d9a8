"""Paged-attention scheduler (mistralrs_amd/scheduler.py) on the C++ KV cache manager: the reference's own scheduler unit tests for text sequences, restated
(mistralrs-core/src/paged_attention/scheduler.rs:1883-2072, fixtures :1183-1197,1225-1275: 8-token blocks, 128 blocks, max 8 sequences, 4096 batched
tokens, 8 decode steps before prefill; sequences of `len` ones, timestamp = id, state RunningCompletion unless said otherwise), preemption under an exhausted
pool, and -- on the device / host emulation -- an end-to-end run: sequences admitted, chunk-prefilled, decoded in batches, preempted and recomputed produce the
same tokens and logits, bit for bit, as each sequence run alone."""
import numpy as np
import pytest


def _sched(blocks=128, block_size=8, **kw):
    import mistralrs_amd  # noqa: F401
    from mistralrs_amd.kv_cache_manager import KVCacheManager
    from mistralrs_amd.scheduler import PagedAttentionScheduler, SchedulerConfig
    cfg = SchedulerConfig(max_num_seqs=kw.pop("max_num_seqs", 8), max_num_batched_tokens=kw.pop("max_num_batched_tokens", 4096),
                          max_decode_steps_before_prefill=kw.pop("max_decode_steps_before_prefill", 8))
    return PagedAttentionScheduler(cfg, KVCacheManager(blocks, block_size, True, [0]))


def _seq(i, n, state="running_completion", computed=None, tok=1):
    from mistralrs_amd.scheduler import Sequence
    s = Sequence(id=i, tokens=[tok] * n, timestamp=i + 1)
    s.state = state
    s.num_computed_tokens = n if computed is None and state == "running_completion" else (computed or 0)
    return s


def test_prompt_chunk_size_stays_within_the_batch_token_budget():  # scheduler.rs:2014-2025
    s = _sched()
    assert s.prompt_chunk_size(0) is None and s.prompt_chunk_size(1) == 4096 and s.prompt_chunk_size(8) == 512 and s.prompt_chunk_size(16) == 256
    assert s.prompt_chunk_size(7) == 585 and s.prompt_chunk_size(7) * 7 <= 4096


def test_token_budget_caps_prompt_admission():  # :2027-2043
    s = _sched(max_num_batched_tokens=3)
    for i in range(5):
        s.waiting.append(_seq(i, 4, "waiting"))
    out = s.schedule()
    assert out.kind == "prompt" and len(out.scheduled) == 3 and out.prompt_chunk_size == 1 and len(s.waiting) == 2


def test_token_budget_fairly_rotates_completion_batches():  # :2045-2071
    s = _sched(max_num_batched_tokens=2)
    for i in range(3):
        s.running.append(_seq(i, 4))
    assert [q.id for q in s.schedule().scheduled] == [0, 1]
    assert [q.id for q in s.schedule().scheduled] == [2, 0]


def test_active_decodes_are_prioritized_over_new_prompts():  # :1962-1977
    s = _sched()
    s.running.append(_seq(0, 4))
    s.waiting.append(_seq(1, 7, "waiting"))
    out = s.schedule()
    assert out.kind == "completion" and len(out.scheduled) == 1 and not out.scheduled[0].is_prompt and len(s.waiting) == 1


def test_underfilled_decode_gets_one_completion_turn_before_refill():  # :1883-1905
    s = _sched()
    s.running.append(_seq(0, 4, "running_prompt", computed=4))
    s.waiting.append(_seq(1, 7, "waiting"))
    first = s.schedule()
    assert first.kind == "completion" and len(first.scheduled) == 1 and not first.scheduled[0].is_prompt and len(s.waiting) == 1
    second = s.schedule()
    assert second.kind == "prompt" and len(second.scheduled) == 1 and second.scheduled[0].is_prompt and not s.waiting


def test_running_prompt_tail_within_budget_finishes_before_decode():  # :1908-1930
    s = _sched()
    s.running.append(_seq(0, 8))
    p = _seq(1, 12, "running_prompt", computed=8)
    p.prefix_cache_len = 8
    s.running.append(p)
    out = s.schedule()
    assert out.kind == "prompt" and [q.id for q in out.scheduled] == [1] and out.scheduled[0].is_prompt


def test_uniform_prompt_batch_preempts_other_lengths_to_the_front_of_waiting():  # bucket_and_preempt_sequences :457-545 (requires_uniform_prompt_batch)
    s = _sched()
    for i, n in enumerate((5, 9, 5)):
        s.waiting.append(_seq(i, n, "waiting", tok=i + 2))
    out = s.schedule()
    assert out.kind == "prompt" and [q.id for q in out.scheduled] == [0, 2] and out.preempted == [1]
    assert [q.id for q in s.waiting] == [1] and s.waiting[0].state == "waiting" and [q.id for q in s.running] == [0, 2]
    assert not s.kv.get_block_ids(1)  # its blocks went back to the pool (unknown request: None)


def test_sequence_larger_than_the_cache_is_rejected_and_pool_pressure_preempts_from_the_back():
    from mistralrs_amd.scheduler import FINISHED_IGNORED
    s = _sched(blocks=6, block_size=8)  # 6 blocks, one of them the null block: 5 usable = 40 tokens
    cap = s.kv.num_gpu_blocks() * 8
    big = _seq(9, cap + 1, "waiting")
    s.waiting.append(big)
    s.waiting.append(_seq(0, 16, "waiting", tok=3))
    s.waiting.append(_seq(1, 16, "waiting", tok=4))
    out = s.schedule()
    assert big.state == FINISHED_IGNORED and "exceeds the total KV cache capacity" in big.error
    assert out.kind == "prompt" and [q.id for q in out.scheduled] == [0, 1]
    for q in out.scheduled:  # both prompts computed; both now need a third block for their next token
        q.num_computed_tokens = len(q)
        q.tokens.append(7)
        q.state = "running_completion"
    free0 = s.kv.num_free_blocks()
    out = s.schedule()
    if free0 >= 2:
        assert out.kind == "completion" and len(out.scheduled) == 2 and out.preempted == []
    else:  # the younger sequence gives its blocks back and waits at the FRONT of the queue with nothing computed (_preempt :1026-1073)
        assert out.kind == "completion" and [q.id for q in out.scheduled] == [0] and out.preempted == [1]
        assert s.waiting[0].id == 1 and s.waiting[0].num_computed_tokens == 0 and s.waiting[0].state == "waiting"


class _FakeRunner:
    """The slice of mistralrs_amd.llama.Llama that PagedEngine drives, with a "model" whose logits are a hash of the token ids found IN THE PAGES of the
    positions 0 .. pos of the row's block table: any mistake in block tables, slot mappings, chunk boundaries, preemption or recomputation changes them."""

    def __init__(self, num_blocks, block_size=8, max_batch=8, max_ctx=96, vocab=97):
        import torch
        from types import SimpleNamespace
        self.cfg = SimpleNamespace(block_size=block_size, max_batch=max_batch, max_context_len=max_ctx, vocab_size=vocab,
                                   max_blocks_per_seq=(max_ctx + block_size - 1) // block_size + 1)
        self.device = torch.device("cpu")
        self.block_tables = torch.zeros(max_batch, self.cfg.max_blocks_per_seq, dtype=torch.int32)
        self.pages = np.full(num_blocks * block_size, -1, dtype=np.int64)
        self.num_blocks = num_blocks

    def _slot(self, row, pos):
        bs = self.cfg.block_size
        return int(self.block_tables[row, pos // bs]) * bs + pos % bs

    def set_state(self, ids, positions):
        self._ids, self._pos = list(ids), list(positions)

    def forward_logits(self, b):
        import torch
        assert b == len(self._ids)
        for i in range(b):  # reshape_and_cache of every row first, then attention
            self.pages[self._slot(i, self._pos[i])] = self._ids[i]
        out = torch.empty(b, self.cfg.vocab_size)
        for i in range(b):
            ctx = [int(self.pages[self._slot(i, p)]) for p in range(self._pos[i] + 1)]
            assert -1 not in ctx, "a row read a page nobody wrote"
            g = np.random.default_rng(abs(hash(tuple(ctx))) % (2 ** 32))
            out[i] = torch.from_numpy(g.standard_normal(self.cfg.vocab_size).astype(np.float32))
        return out


def test_engine_loop_bookkeeping_on_a_fake_runner():
    """PagedEngine + scheduler + C++ KV manager end to end on the CPU: 7 sequences on a pool that holds about three of them; every sequence must generate
    exactly what it generates alone (prompt chunks, batched decodes, preemption + recomputation, prefix-cache hits of the shared prompt head)."""
    import torch
    from mistralrs_amd.kv_cache_manager import KVCacheManager
    from mistralrs_amd.scheduler import PagedAttentionScheduler, PagedEngine, SchedulerConfig, Sequence
    lens = [(5, 9), (20, 12), (41, 6), (12, 20), (20, 7), (33, 5), (3, 30)]
    head = [11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26]  # two full 8-token blocks shared by some prompts
    prompts = [(head if i % 2 else []) + [(31 * i + 7 * j * j) % 90 for j in range(n)] for i, (n, _) in enumerate(lens)]
    nb = 14
    m = _FakeRunner(nb)
    mgr = KVCacheManager(nb, 8, True, [0])
    sched = PagedAttentionScheduler(SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=16, max_decode_steps_before_prefill=3), mgr)
    seqs = [Sequence(id=i + 1, tokens=list(p), max_new_tokens=nn) for i, (p, (_, nn)) in enumerate(zip(prompts, lens))]
    for s in seqs:
        sched.add_seq(s)
    eng = PagedEngine(m, sched)
    eng.run(max_steps=5000)
    assert all(s.state == "done" and len(s.generated) == nn for s, (_, nn) in zip(seqs, lens))
    assert eng.steps["prompt"] > len(seqs) and eng.steps["completion"] > 0 and eng.steps["preemptions"] > 0, eng.steps
    assert mgr.num_free_blocks() == mgr.num_usable_blocks()
    for s, p, (_, nn) in zip(seqs, prompts, lens):
        solo = _FakeRunner(16)
        solo.block_tables[0] = torch.arange(1, 1 + solo.cfg.max_blocks_per_seq, dtype=torch.int32) % 16
        toks, lg = list(p), None
        for pos in range(len(p) + nn - 1):
            solo.set_state([toks[pos]], [pos])
            lg = solo.forward_logits(1)[0]
            if pos >= len(p) - 1:
                toks.append(int(lg.argmax()))
        assert toks[len(p):] == s.generated, (s.id, toks[len(p):], s.generated)
        assert torch.equal(lg, s.last_logits), s.id


@pytest.mark.gpu
def test_engine_loop_admits_chunks_batches_and_preempts_bit_exactly(oracle, dev, request):
    """Six sequences (prompts of 5 .. 70 tokens, 6 .. 14 new tokens) through PagedEngine on a pool too small to hold them all: the scheduler admits,
    chunk-prefills (token budget 16), batches the decodes, preempts and later recomputes; every sequence ends with the SAME tokens and the same final
    logits, bit for bit, as when it is run alone on a fresh runner (the prompt chunks and the recomputation go through the batch <= 8 decode kernels, whose
    rows are bit-identical to single-sequence decoding)."""
    import torch
    from mistralrs_amd.kv_cache_manager import KVCacheManager
    from mistralrs_amd.scheduler import PagedAttentionScheduler, PagedEngine, SchedulerConfig, Sequence
    from tests.test_dec_model import Q4KM, _mk
    emu = request.config.getoption("--host-emulation")
    if emu:
        pytest.skip("~20 minutes on the host emulation; the bookkeeping runs on the fake runner in the CPU suite, the kernels' batch == single property in test_dec_model.py")
    lens = [(5, 6), (33, 8), (70, 6), (12, 14), (33, 7), (20, 6)]
    cfg, w, m, cos, sin = _mk(oracle, dev, Q4KM(oracle), "bf16", max_batch=8, max_ctx=128, max_new=8)
    prompts = [[(1000 + 13 * i + 7 * j * j) % cfg.vocab_size for j in range(n)] for i, (n, _) in enumerate(lens)]
    pool = 7 if not emu else 5  # 32-token blocks: not enough for everyone at once
    mgr = KVCacheManager(pool, cfg.block_size, True, [0])
    assert pool <= m.num_blocks
    sched = PagedAttentionScheduler(SchedulerConfig(max_num_seqs=8, max_num_batched_tokens=16, max_decode_steps_before_prefill=3), mgr)
    seqs = [Sequence(id=i + 1, tokens=list(p), max_new_tokens=nn) for i, (p, (_, nn)) in enumerate(zip(prompts, lens))]
    for s in seqs:
        sched.add_seq(s)
    eng = PagedEngine(m, sched)
    eng.run(max_steps=4000)
    assert all(s.state == "done" and len(s.generated) == nn for s, (_, nn) in zip(seqs, lens))
    assert eng.steps["prompt"] > len(seqs) and eng.steps["completion"] > 0  # prompts were chunked; decodes were batched
    if not emu:
        assert eng.steps["preemptions"] > 0, eng.steps
    assert mgr.num_free_blocks() == mgr.num_usable_blocks()  # everything returned to the pool
    for s, p, (_, nn) in zip(seqs, prompts, lens):
        cfg2, _, solo, _, _ = _mk(oracle, dev, Q4KM(oracle), "bf16", max_batch=8, max_ctx=128, max_new=8)
        toks, lg = list(p), None
        for pos in range(len(p) + nn - 1):
            solo.set_state([toks[pos]], [pos])
            lg = solo.forward_logits(1)[0]
            if pos >= len(p) - 1:
                toks.append(int(lg.argmax()))
        # the last forward of the scheduled run consumed token [-2] and produced token [-1]
        assert toks[len(p):] == s.generated, (s.id, toks[len(p):], s.generated)
        assert torch.equal(lg, s.last_logits), s.id
