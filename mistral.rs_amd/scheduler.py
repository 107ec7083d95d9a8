"""Paged-attention scheduler on the C++ KV cache manager: host-side mirror of `PagedAttentionScheduler`
(mistralrs-core/src/paged_attention/scheduler.rs) for text sequences, and a small engine loop that drives the C++ model runner with it.

What is restated (line references into scheduler.rs):
  * two queues, `waiting` and `running`; a scheduling pass returns EITHER a prompt batch OR a completion batch (:563-956);
  * `completion_is_due` (:246-282): decode steps and prompt work alternate under `max_decode_steps_before_prefill`; a running prompt that fits the
    token budget goes first; a waiting prompt that fits the free blocks interrupts decoding after at least one decode step;
  * admission of waiting sequences (:588-769): reject what can never fit, chain-hash the prompt, prefix-cache lookup (`get_computed_blocks`),
    `allocate_slots`; when the pool is short the sequence waits, and after `WAITING_TIMEOUT` passes running sequences are preempted from the back;
  * prompt batches are uniform (`requires_uniform_prompt_batch`: same uncached length and cached prefix, `bucket_and_preempt_sequences` :457-545 -- the
    others go back to the FRONT of the waiting queue) and every sequence gets `max_num_batched_tokens / batch` prompt tokens per pass (:136-138);
  * completion passes (:791-868) reserve the slot of the next token for every running sequence in FCFS order, preempting from the back
    (`_preempt` :1026-1073: state Waiting, computed tokens 0, full blocks published to the prefix cache, blocks freed, FRONT of the waiting queue), then
    take a round-robin window of the running rows under the token budget (`completion_batch_indices` :303-330);
  * `free_finished_sequence_groups` (:957-1024): finished sequences publish their full blocks and release them.
Not restated: multimodal features, adapters, speculative staging, recurrent state, packed prefill, scheduler-visible chunk plans, metrics.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import List, Optional

from .kv_cache_manager import KVCacheManager, compute_block_hashes

WAITING_TIMEOUT = 64  # scheduler.rs:48
WAITING, RUNNING_PROMPT, RUNNING_COMPLETION, DONE, FINISHED_IGNORED = "waiting", "running_prompt", "running_completion", "done", "finished_ignored"


@dataclass
class Sequence:
    """The slice of `sequence::Sequence` the scheduler reads."""
    id: int
    tokens: List[int]                 # prompt + generated so far
    max_new_tokens: int = 16
    timestamp: int = 0
    state: str = WAITING
    num_computed_tokens: int = 0      # tokens whose K / V are in the pages
    prefix_cache_len: int = 0
    prompt_len: int = 0
    error: Optional[str] = None
    generated: List[int] = field(default_factory=list)
    last_logits: object = None

    def __post_init__(self):
        if not self.prompt_len:
            self.prompt_len = len(self.tokens)

    def __len__(self):
        return len(self.tokens)

    @property
    def num_uncomputed_tokens(self) -> int:
        return len(self.tokens) - self.num_computed_tokens

    @property
    def is_prompt(self) -> bool:
        return self.state == RUNNING_PROMPT

    @property
    def is_completion(self) -> bool:
        return self.state == RUNNING_COMPLETION

    @property
    def is_finished(self) -> bool:
        return self.state in (DONE, FINISHED_IGNORED)


@dataclass
class SchedulerConfig:
    max_num_seqs: int
    max_num_batched_tokens: int
    max_decode_steps_before_prefill: int = 4


@dataclass
class SchedulerOutput:
    """Either ALL prompt or ALL completion (PagedAttentionSchedulerOutput, scheduler.rs:61-72)."""
    kind: str                         # "prompt" | "completion" | "idle"
    scheduled: List[Sequence]
    num_cached_tokens: List[int]
    prompt_chunk_size: Optional[int]
    preempted: List[int]


class PagedAttentionScheduler:
    def __init__(self, config: SchedulerConfig, kv_cache_manager: KVCacheManager, prefix_caching: bool = True):
        assert config.max_num_seqs > 0 and config.max_num_batched_tokens > 0 and config.max_decode_steps_before_prefill > 0
        self.config, self.kv = config, kv_cache_manager
        self.block_size = kv_cache_manager.block_size()
        self.waiting: deque = deque()
        self.running: deque = deque()
        self.prefix_caching_enabled = prefix_caching
        self.decode_steps_since_prefill = 0
        self.completion_cursor = 0
        self.waiting_counts: dict = {}
        self.preempted_sequence_ids: List[int] = []
        self._clock = 0

    # ---- queue interface (Scheduler trait, scheduler.rs:1082-1110)
    def add_seq(self, seq: Sequence) -> None:
        self._clock += 1
        if not seq.timestamp:
            seq.timestamp = self._clock
        seq.state = WAITING
        self.waiting.append(seq)

    def waiting_len(self) -> int:
        return len(self.waiting)

    def running_len(self) -> int:
        return len(self.running)

    # ---- helpers
    def _hashes(self, seq: Sequence):
        return compute_block_hashes(seq.tokens, self.block_size)

    def _preempt(self, seq: Sequence) -> None:  # scheduler.rs:1026-1073
        if seq.is_finished:
            return
        seq.state, seq.prefix_cache_len = WAITING, 0
        self.preempted_sequence_ids.append(seq.id)
        computed = seq.num_computed_tokens
        seq.num_computed_tokens = 0
        if self.prefix_caching_enabled:
            self.kv.cache_blocks(seq.id, self._hashes(seq), computed)
        self.kv.free(seq.id)
        self.waiting.appendleft(seq)

    def _reject_front_of_waiting(self, reason: str) -> None:
        seq = self.waiting.popleft()
        seq.state, seq.error = FINISHED_IGNORED, reason
        self.waiting_counts.pop(seq.id, None)
        self.kv.free(seq.id)

    def prompt_chunk_size(self, batch: int) -> Optional[int]:  # :136-138
        return max(self.config.max_num_batched_tokens // batch, 1) if batch > 0 else None

    def _waiting_prompt_fits_free_blocks(self) -> bool:  # :284-300
        if not self.waiting:
            return False
        prompt_blocks = -(-len(self.waiting[0]) // self.block_size)
        decode_reserve = sum(1 for s in self.running if s.is_completion)
        return prompt_blocks + decode_reserve <= self.kv.num_free_blocks()

    def completion_is_due(self) -> bool:  # :246-282
        if not any(s.is_completion for s in self.running):
            return False
        running_prompt_tokens = sum(s.num_uncomputed_tokens for s in self.running if s.is_prompt)
        if 0 < running_prompt_tokens <= self.config.max_num_batched_tokens:
            return False
        has_running_prompt, has_waiting_prompt = running_prompt_tokens > 0, bool(self.waiting)
        if (not has_running_prompt and has_waiting_prompt and len(self.running) < self.config.max_num_seqs
                and self._waiting_prompt_fits_free_blocks() and self.decode_steps_since_prefill > 0):
            return False
        has_prompt = has_running_prompt or has_waiting_prompt
        return (not has_prompt) or self.decode_steps_since_prefill < self.config.max_decode_steps_before_prefill

    def _bucket_and_preempt(self, seqs: deque) -> deque:  # :457-545, BatchKind::Prompt with require_uniform_length
        if len(seqs) <= 1:
            return seqs
        key = lambda s: (len(s) - s.prefix_cache_len, s.prefix_cache_len)
        first = key(seqs[0])
        selected, rejected = deque(s for s in seqs if key(s) == first), [s for s in seqs if key(s) != first]
        if not rejected:
            return selected
        ids = {s.id for s in rejected}
        for s in reversed(rejected):
            self._preempt(s)
        self.running = deque(s for s in self.running if s.id not in ids)
        return selected

    @staticmethod
    def completion_batch_indices(rows: List[Sequence], cursor: int, token_budget: int):  # :303-330
        if not rows:
            return [], cursor
        n, start, remaining, selected, last = len(rows), cursor % len(rows), token_budget, [], cursor % len(rows)
        for off in range(n):
            i = (start + off) % n
            cost = max(rows[i].num_uncomputed_tokens, 1)
            if cost <= remaining or not selected:
                remaining = max(remaining - cost, 0)
                selected.append(i)
                last = i
        return selected, (last + 1) % n

    # ---- the scheduling pass (scheduler.rs:563-956)
    def schedule(self) -> SchedulerOutput:
        self.preempted_sequence_ids = []
        for s in self.running:
            if s.is_prompt and s.num_computed_tokens == len(s):
                s.state = RUNNING_COMPLETION
        scheduled: deque = deque()
        completion_due = self.completion_is_due()
        if not completion_due:
            scheduled.extend(s for s in self.running if s.is_prompt)
        while not completion_due and self.waiting:
            seq = self.waiting[0]
            if len(self.running) >= self.config.max_num_seqs or len(scheduled) >= self.config.max_num_batched_tokens:
                break
            n_tok = len(seq)
            cap = self.kv.num_gpu_blocks() * self.block_size
            if n_tok > cap:
                self._reject_front_of_waiting(f"Sequence {seq.id} with {n_tok} tokens exceeds the total KV cache capacity of {cap} tokens.")
                continue
            hashes = self._hashes(seq)
            hit = self.kv.get_computed_blocks(hashes, n_tok) if self.prefix_caching_enabled else None
            cached_ids = list(hit.block_ids) if hit else []
            n_cached = hit.num_computed_tokens if hit else 0
            cached_ids = cached_ids[: n_cached // self.block_size]
            ok = self.kv.allocate_slots(seq.id, n_tok, cached_ids) is not None
            if ok:
                self.waiting_counts.pop(seq.id, None)
            else:
                cnt = self.waiting_counts.get(seq.id, 0) + 1
                self.waiting_counts[seq.id] = cnt
                if cnt <= WAITING_TIMEOUT:
                    break
                allocated = False
                while self.running:  # starved: preempt running sequences from the back until it fits (:687-716)
                    victim = self.running.pop()
                    waiting_seq = self.waiting.popleft()
                    self._preempt(victim)
                    self.waiting.appendleft(waiting_seq)
                    scheduled = deque(s for s in scheduled if s.id != victim.id)
                    if self.kv.allocate_slots(seq.id, n_tok, cached_ids) is not None:
                        allocated = True
                        break
                if not allocated:
                    self._reject_front_of_waiting(f"Sequence {seq.id} with {n_tok} tokens cannot be scheduled: KV cache exhausted even after preempting all running sequences.")
                    continue
                self.waiting_counts.pop(seq.id, None)
            seq.state, seq.prefix_cache_len, seq.num_computed_tokens = RUNNING_PROMPT, n_cached, n_cached
            self.waiting.popleft()
            self.running.append(seq)
            scheduled.append(seq)
        if scheduled:
            batch = self._bucket_and_preempt(scheduled)
            self.decode_steps_since_prefill = 0
            return SchedulerOutput("prompt", list(batch), [s.prefix_cache_len for s in batch], self.prompt_chunk_size(len(batch)), list(self.preempted_sequence_ids))
        # ---- completion: reserve the next token's slot for every running sequence, FCFS, preempting from the back (:791-850)
        prompt_running = deque(s for s in self.running if s.is_prompt)
        rows = deque(sorted((s for s in self.running if not s.is_prompt), key=lambda s: s.timestamp))
        kept: deque = deque()
        while rows:
            seq = rows.popleft()
            n_tok = len(seq) if seq.num_uncomputed_tokens > 0 else len(seq) + 1
            gave_up = False
            while self.kv.allocate_slots(seq.id, n_tok, []) is None:
                if rows:
                    self._preempt(rows.pop())
                else:
                    self._preempt(seq)
                    gave_up = True
                    break
            if not gave_up:
                kept.append(seq)
        self.running = kept
        for s in self.running:
            s.state = RUNNING_COMPLETION
        if self.prefix_caching_enabled:  # publish newly full blocks eagerly (:887-930); idempotent
            for s in self.running:
                self.kv.cache_blocks(s.id, self._hashes(s), s.num_computed_tokens)
        live = list(self.running)
        sel, self.completion_cursor = self.completion_batch_indices(live, self.completion_cursor, self.config.max_num_batched_tokens)
        self.running.extend(prompt_running)
        if not sel:
            return SchedulerOutput("idle", [], [], None, list(self.preempted_sequence_ids))
        self.decode_steps_since_prefill = min(self.decode_steps_since_prefill + 1, self.config.max_decode_steps_before_prefill)
        return SchedulerOutput("completion", [live[i] for i in sel], [], None, list(self.preempted_sequence_ids))

    def free_finished_sequence_groups(self) -> None:  # :957-1024
        keep = deque()
        for s in self.running:
            if s.is_finished:
                if self.prefix_caching_enabled:
                    self.kv.cache_blocks(s.id, self._hashes(s), s.num_computed_tokens)
                self.kv.free(s.id)
                self.waiting_counts.pop(s.id, None)
            else:
                keep.append(s)
        self.running = keep


class PagedEngine:
    """Drives `mistralrs_amd.llama.Llama` with the scheduler: greedy decoding of many sequences through the batch <= 8 decode engine, prompts in chunks of
    `prompt_chunk_size` tokens through the same kernels (so a preempted and recomputed sequence reproduces its logits bit for bit), block tables and slot
    mappings from the KV cache manager (pipeline/inputs_processor.rs:900-922)."""

    def __init__(self, model, scheduler: PagedAttentionScheduler):
        self.m, self.s = model, scheduler
        self.cfg = model.cfg
        self.steps = {"prompt": 0, "completion": 0, "preemptions": 0}

    def _table(self, seq: Sequence):
        import torch
        return torch.tensor(self.s.kv.get_block_table(seq.id, self.cfg.max_blocks_per_seq), dtype=self.m.block_tables.dtype, device=self.m.device)

    def _finish_token(self, seq: Sequence, logits) -> None:
        tok = int(logits.argmax())
        seq.last_logits = logits.clone()
        seq.tokens.append(tok)
        seq.generated.append(tok)
        if len(seq.generated) >= seq.max_new_tokens or len(seq.tokens) >= self.cfg.max_context_len:
            seq.state = DONE

    def _p2p_guard(self) -> None:
        """Tensor parallel: before tokens are handed out, the per-rank error word of the peer-mailbox all-reduce is MAX-reduced over the ranks (Llama.p2p_sync_error);
        a timed-out sum is NaN, so the step is not usable -- the route has been dropped on every rank, the caller re-runs the request on RCCL."""
        f = getattr(self.m, "p2p_sync_error", None)
        if f is not None and f():
            raise RuntimeError("p2p all-reduce timed out during this step: the route has been dropped on every rank (RCCL from now on); re-schedule the step")

    def step(self) -> str:
        out = self.s.schedule()
        self.steps["preemptions"] += len(out.preempted)
        if out.kind == "prompt":
            self.steps["prompt"] += 1
            for seq in out.scheduled:
                a = seq.num_computed_tokens
                b = min(len(seq), a + (out.prompt_chunk_size or len(seq)))
                table = self._table(seq)
                last = None
                for i in range(a, b, min(8, self.cfg.max_batch)):  # rows of one launch = consecutive positions of ONE sequence
                    ids = seq.tokens[i:min(b, i + min(8, self.cfg.max_batch))]
                    self.m.block_tables[: len(ids)] = table
                    self.m.set_state(ids, list(range(i, i + len(ids))))
                    last = self.m.forward_logits(len(ids))[len(ids) - 1]
                self._p2p_guard()
                seq.num_computed_tokens = b
                if b == len(seq):  # the prompt's last token produced the first new token (llama.rs:514-517: logits of the last position only)
                    seq.state = RUNNING_COMPLETION
                    self._finish_token(seq, last)
        elif out.kind == "completion":
            self.steps["completion"] += 1
            # every scheduled row decodes in THIS step (the scheduler has reserved its slot and counted the step): launches of up to max_batch rows
            for r0 in range(0, len(out.scheduled), self.cfg.max_batch):
                rows = out.scheduled[r0: r0 + self.cfg.max_batch]
                for i, seq in enumerate(rows):
                    self.m.block_tables[i] = self._table(seq)
                self.m.set_state([s.tokens[-1] for s in rows], [len(s) - 1 for s in rows])
                logits = self.m.forward_logits(len(rows)).clone()
                self._p2p_guard()
                for i, seq in enumerate(rows):
                    seq.num_computed_tokens = len(seq)
                    self._finish_token(seq, logits[i])
        self.s.free_finished_sequence_groups()
        return out.kind

    def run(self, max_steps: int = 10000) -> None:
        for _ in range(max_steps):
            if not self.s.waiting and not self.s.running:
                return
            self.step()
        raise RuntimeError("PagedEngine.run: sequences did not finish")
