"""Tensor-parallel sharding + collectives: host-side mirror of `mistralrs-quant/src/distributed/`.

  * Shard / shard_qtensor ........ `Shard::{Simple,Offset}` applied to packed GGUF blocks: dim 0 = whole rows (a contiguous
                                    byte range), dim 1 = columns cut at quant-block multiples (uqff/mod.rs:224-300,
                                    gguf/weight_source.rs:524-563 `slice_blocked_data`)
  * compute_kv_shard / compute_n_kv_groups / validate_tp_* ... distributed/layers.rs:2657-2733 (KV-head replication when
                                    world_size > n_kv_heads), same error text
  * ColumnParallel / RowParallel placement of the Llama tensors ... models/llama.rs:320-470 via distributed/layers.rs:695-975,1160-1616
  * SumAllReduce .................. distributed/mod.rs:390-453: sum all-reduce of the row-parallel partial outputs.  CPU tensors go
                                    through torch.distributed (gloo); on the GPU the C++ runner issues RCCL ncclAllReduce on its own
                                    stream (csrc/ext_comm.hip), one process per GPU, id hand-off as core/distributed.rs:569-795.
Everything here is byte/shape logic on torch tensors of any device; nothing imports the oracle.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from .gguf.qtensor import GgmlDType, QTensor


@dataclass(frozen=True)
class Shard:
    """`Shard::Simple{dim, rank, world_size}` (offset is None) or `Shard::Offset{dim, offset, len}`."""
    dim: int = 0
    rank: int = 0
    world_size: int = 1
    offset: int | None = None
    length: int | None = None
    # > 0: the tensor is `stacked` experts along the row axis ([E * n, k]); a dim-0 shard then means rows [lo, hi) of EACH expert and must go through
    # shard_stacked_experts -- shard_qtensor refuses it (advisor, round 2: first rows / world of the stack would be whole leading experts, silently wrong)
    stacked: int = 0

    def bounds(self, size: int) -> tuple[int, int]:
        if self.offset is not None:
            return self.offset, self.offset + self.length
        if size % self.world_size:
            raise ValueError(f"The size of dimension {self.dim} ({size}) must be divisible by the world size ({self.world_size}).")
        step = size // self.world_size
        return self.rank * step, (self.rank + 1) * step


def validate_tp_kv_heads(total_num_kv_heads: int, tensor_parallel_size: int) -> None:
    if total_num_kv_heads == 0:
        raise ValueError("Total number of KV heads must be greater than 0.")
    if tensor_parallel_size <= total_num_kv_heads:
        if total_num_kv_heads % tensor_parallel_size:
            raise ValueError(f"Total number of KV heads ({total_num_kv_heads}) must be divisible by tensor parallel size "
                             f"({tensor_parallel_size}) when KV heads are partitioned.")
    elif tensor_parallel_size % total_num_kv_heads:
        raise ValueError(f"Tensor parallel size ({tensor_parallel_size}) must be divisible by total number of KV heads "
                         f"({total_num_kv_heads}) when KV heads are replicated.")


def validate_tp_head_layout(total_num_attention_heads: int, total_num_kv_heads: int, tensor_parallel_size: int) -> None:
    if total_num_attention_heads == 0:
        raise ValueError("Total number of attention heads must be greater than 0.")
    if total_num_attention_heads % tensor_parallel_size:
        raise ValueError(f"Total number of attention heads ({total_num_attention_heads}) must be divisible by tensor parallel size "
                         f"({tensor_parallel_size}).")
    validate_tp_kv_heads(total_num_kv_heads, tensor_parallel_size)


def compute_kv_shard(total_num_kv_heads: int, head_dim: int, rank: int, world_size: int) -> Shard:
    """distributed/layers.rs:2692-2716 (rows of k_proj / v_proj owned by `rank`)."""
    if world_size == 1:
        return Shard()
    validate_tp_kv_heads(total_num_kv_heads, world_size)
    if world_size <= total_num_kv_heads:
        return Shard(0, rank, world_size)
    kv_replicate = world_size // total_num_kv_heads
    num_kv_heads = max(total_num_kv_heads // world_size, 1)
    kv_shard_id = (rank // kv_replicate) * num_kv_heads
    return Shard(0, rank, world_size, offset=kv_shard_id * head_dim, length=head_dim)


def compute_n_kv_groups(total_num_kv_heads: int, num_attention_heads: int, world_size: int) -> int:
    """distributed/layers.rs:2718-2733."""
    validate_tp_head_layout(num_attention_heads, total_num_kv_heads, world_size)
    kv_replicate = world_size // total_num_kv_heads if world_size > total_num_kv_heads else 1
    return (num_attention_heads // total_num_kv_heads) // kv_replicate


def shard_qtensor(w: QTensor, shard: Shard) -> QTensor:
    """Apply a shard to packed blocks [N][K/blk].  dim 0: rows; dim 1: columns, which must fall on block boundaries
    (256 for the K-quants, 32 for Q8_0 & co: uqff/mod.rs:277-284)."""
    n, k = w.shape
    dt = w.dtype
    rb = dt.row_bytes(k)
    data = w.data.view(n, rb)
    if shard.world_size == 1 and shard.offset is None:
        return w
    if shard.stacked:
        raise ValueError(f"stacked-expert shard ({shard.stacked} experts): use shard_stacked_experts / shard_llama_tensor, not shard_qtensor")
    if shard.dim == 0:
        lo, hi = shard.bounds(n)
        return QTensor(dt, (hi - lo, k), data[lo:hi].contiguous().view(-1))
    if shard.dim == 1:
        lo, hi = shard.bounds(k)
        if lo % dt.block_size or hi % dt.block_size:
            raise ValueError(f"Cannot slice {dt.name} columns [{lo}, {hi}): not a multiple of the block size {dt.block_size}.")
        b0, b1 = lo // dt.block_size * dt.type_size, hi // dt.block_size * dt.type_size
        return QTensor(dt, (n, hi - lo), data[:, b0:b1].contiguous().view(-1))
    raise ValueError("quantized weights shard along dim 0 or 1")


def shard_stacked_experts(w: QTensor, num_experts: int, shard: Shard) -> QTensor:
    """Shard of experts stacked along the row axis [E * n, k] (Mixtral `ffn_{gate,up,down}_exps`): every expert is cut like a dense FFN matrix
    (moe/experts/mod.rs:332-339: experts sharded on the ffn dimension, one all-reduce per MoE block) -- dim 0: rows [lo, hi) of EACH expert;
    dim 1: the same column blocks of every row."""
    rows, k = w.shape
    if rows % num_experts:
        raise ValueError(f"stacked experts: {rows} rows are not a multiple of {num_experts} experts")
    if shard.dim == 1:
        return shard_qtensor(w, shard)
    n = rows // num_experts
    rb = w.dtype.row_bytes(k)
    lo, hi = shard.bounds(n)
    data = w.data.view(num_experts, n, rb)[:, lo:hi].contiguous()
    return QTensor(w.dtype, (num_experts * (hi - lo), k), data.view(-1))


def llama_tensor_shard(name: str, cfg_total: dict, rank: int, world_size: int) -> Shard | None:
    """Placement of a GGUF tensor under TP (models/llama.rs:320-470): q/k/v/gate/up column-parallel (dim 0), attn_output /
    ffn_down row-parallel (dim 1), embeddings / norms / lm_head replicated (None)."""
    if world_size == 1:
        return None
    leaf = name.split(".")[-2] if name.startswith("blk.") else name
    E = int(cfg_total.get("num_experts", 0) or 0) if leaf.endswith("_exps") else 0
    if leaf.endswith("_exps") and E <= 0:
        raise ValueError(f"{name}: stacked experts need cfg_total['num_experts']")
    if leaf in ("attn_q", "ffn_gate", "ffn_up", "ffn_gate_exps", "ffn_up_exps"):  # stacked experts: per expert (shard_stacked_experts)
        return Shard(0, rank, world_size, stacked=E)
    if leaf in ("attn_k", "attn_v"):
        return compute_kv_shard(cfg_total["num_kv_heads"], cfg_total["head_dim"], rank, world_size)
    if leaf in ("attn_output", "ffn_down", "ffn_down_exps"):
        return Shard(1, rank, world_size, stacked=E)
    return None


def shard_llama_tensor(name: str, w: QTensor, cfg_total: dict, rank: int, world_size: int) -> QTensor:
    """One entry point for a loader: the rank's shard of GGUF tensor `name` (dense linears through shard_qtensor, stacked experts per expert)."""
    sh = llama_tensor_shard(name, cfg_total, rank, world_size)
    if sh is None:
        return w
    if sh.stacked:
        return shard_stacked_experts(w, sh.stacked, Shard(sh.dim, sh.rank, sh.world_size))
    return shard_qtensor(w, sh)


def local_dims(num_heads: int, num_kv_heads: int, intermediate_size: int, world_size: int) -> tuple[int, int, int]:
    """(heads, kv_heads, ffn) owned by one rank."""
    validate_tp_head_layout(num_heads, num_kv_heads, world_size)
    if intermediate_size % world_size:
        raise ValueError(f"The size of dimension 0 ({intermediate_size}) must be divisible by the world size ({world_size}).")
    return num_heads // world_size, max(num_kv_heads // world_size, 1), intermediate_size // world_size


class SumAllReduce:
    """distributed/mod.rs:390-453: out-of-place in the reference, in place here (the runner's buffer is the residual stream)."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class RcclComm:
    """RCCL communicator owned by the C++ runner's library (csrc/ext_comm.hip): rank 0 creates the unique id, it is handed to
    the other ranks through torch.distributed (the role of the daemon hand-off in core/distributed.rs:665-704), every rank calls
    ncclCommInitRank.  One process per GPU."""

    def __init__(self, rank: int, world_size: int, device: torch.device):
        import torch.distributed as dist
        from . import _lib
        L = _lib.load("ext")
        L.mrs_comm_unique_id.argtypes = [C.c_void_p]
        L.mrs_comm_init.restype = C.c_void_p
        L.mrs_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mrs_comm_all_reduce_sum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.mrs_comm_destroy.argtypes = [C.c_void_p]
        L.mrs_last_error.restype = C.c_char_p
        self._L, self.rank, self.world_size = L, rank, world_size
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            if L.mrs_comm_unique_id(uid.data_ptr()) != 0:
                raise RuntimeError((L.mrs_last_error() or b"").decode())
        uid_dev = uid.to(device)
        dist.broadcast(uid_dev, src=0)
        uid = uid_dev.cpu()
        torch.cuda.set_device(device)
        self.handle = L.mrs_comm_init(uid.data_ptr(), rank, world_size)
        if not self.handle:
            raise RuntimeError((L.mrs_last_error() or b"").decode())

    def nranks(self) -> int:
        """Ranks of the communicator as RCCL counts them (ncclCommCount)."""
        self._L.mrs_comm_nranks.argtypes = [C.c_void_p]
        return int(self._L.mrs_comm_nranks(self.handle))

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.dtype == torch.float32 and t.is_contiguous()
        if self._L.mrs_comm_all_reduce_sum_f32(self.handle, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream) != 0:
            raise RuntimeError((self._L.mrs_last_error() or b"").decode())
        return t


class P2PAllReduce:
    """One-shot all-reduce over peer-mapped mailboxes (csrc/ext_p2p.hip) for the decode-sized messages of tensor parallelism: every rank
    allocates a mailbox, the 64-byte IPC handles travel through torch.distributed (all_gather_object), every rank opens its peers' mailboxes
    (hipIpcOpenMemHandle; the processes need HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack).  Replaces ncclAllReduce
    (mistralrs-quant/src/distributed/mod.rs:584-587) for messages of <= max_elems f32; larger ones stay on RCCL (Llama.set_comm)."""

    def __init__(self, rank: int, world_size: int, device: torch.device, max_elems: int = 8 * 8192):
        import torch.distributed as dist
        from . import _lib
        L = _lib.load("ext")
        L.mrs_p2p_mailbox_bytes.restype = C.c_size_t
        L.mrs_p2p_mailbox_bytes.argtypes = [C.c_int, C.c_size_t]
        L.mrs_ipc_get_handle.argtypes = [C.c_void_p, C.c_void_p]
        L.mrs_ipc_open_handle.restype = C.c_void_p
        L.mrs_ipc_open_handle.argtypes = [C.c_void_p]
        L.mrs_p2p_create.restype = C.c_void_p
        L.mrs_p2p_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_size_t]
        L.mrs_p2p_all_reduce_sum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.mrs_p2p_error.argtypes = [C.c_void_p]
        L.mrs_last_error.restype = C.c_char_p
        self._L, self.rank, self.world_size, self.max_elems = L, rank, world_size, max_elems
        torch.cuda.set_device(device)
        # fine-grained / uncached device memory (hipExtMallocWithFlags inside the extension): peers write into it while our kernel polls it
        L.mrs_p2p_alloc_mailbox.restype = C.c_void_p
        L.mrs_p2p_alloc_mailbox.argtypes = [C.c_size_t]
        L.mrs_p2p_free_mailbox.argtypes = [C.c_void_p]
        # Every step that can fail on one rank only is followed by an exchange of the outcome, so that all ranks leave through the same door
        # (a rank that raised while its peers sit in a collective would hang the job instead of falling back to RCCL).
        err, mine = None, (C.c_char * 64)()
        self.handle, self._peer_ptrs = None, []
        self.mailbox_ptr = L.mrs_p2p_alloc_mailbox(L.mrs_p2p_mailbox_bytes(world_size, max_elems))
        if not self.mailbox_ptr:
            err = "mailbox allocation: " + (L.mrs_last_error() or b"").decode()
        elif L.mrs_ipc_get_handle(self.mailbox_ptr, mine) != 0:
            err = "hipIpcGetMemHandle: " + (L.mrs_last_error() or b"").decode()
        got = [None] * world_size
        dist.all_gather_object(got, (err, bytes(mine)))
        bad = [f"rank {r}: {e}" for r, (e, _) in enumerate(got) if e]
        if bad:
            self._release()
            raise RuntimeError("p2p all-reduce unavailable (" + "; ".join(bad) + ")")
        ptrs = (C.c_void_p * world_size)()
        for r, (_, hb) in enumerate(got):
            if r == rank:
                ptrs[r] = self.mailbox_ptr
                continue
            p = L.mrs_ipc_open_handle(C.create_string_buffer(hb, 64))
            if not p:
                err = f"hipIpcOpenMemHandle(rank {r}): " + (L.mrs_last_error() or b"").decode()
                break
            ptrs[r] = p
            self._peer_ptrs.append(p)
        got2 = [None] * world_size
        dist.all_gather_object(got2, err)  # also the barrier: every mailbox is zeroed and mapped before the first granule is written
        bad = [f"rank {r}: {e}" for r, e in enumerate(got2) if e]
        if bad:
            self._release()
            raise RuntimeError("p2p all-reduce unavailable (" + "; ".join(bad) + ")")
        self.handle = L.mrs_p2p_create(rank, world_size, ptrs, max_elems)
        got3 = [None] * world_size
        dist.all_gather_object(got3, None if self.handle else "mrs_p2p_create: " + (L.mrs_last_error() or b"").decode())
        bad = [f"rank {r}: {e}" for r, e in enumerate(got3) if e]
        if bad:  # the last step that can fail on one rank only: exchanged like the others, so no rank starts polling a peer that gave up
            self._release()
            raise RuntimeError("p2p all-reduce unavailable (" + "; ".join(bad) + ")")

    def _release(self) -> None:
        """Free everything this object opened, in reverse order (idempotent): the Comm, the peers' mapped mailboxes, the own mailbox."""
        L = self._L
        if getattr(self, "handle", None):
            L.mrs_p2p_destroy.argtypes = [C.c_void_p]
            L.mrs_p2p_destroy(self.handle)
            self.handle = None
        L.mrs_ipc_close_handle.argtypes = [C.c_void_p]
        for p in getattr(self, "_peer_ptrs", []):
            L.mrs_ipc_close_handle(p)
        self._peer_ptrs = []
        if getattr(self, "mailbox_ptr", None):
            L.mrs_p2p_free_mailbox(self.mailbox_ptr)
            self.mailbox_ptr = None

    def close(self) -> None:
        """Release the mailbox, the peer mappings and the Comm.  The caller must have synchronised the device (no all-reduce in flight) and detached the
        object from any runner (Llama.set_p2p(None)) first."""
        self._release()

    # no __del__: freeing the mailbox needs a device synchronisation and a rank barrier first (peers post into it through their IPC mapping, a captured graph holds
    # its address by value) -- only close() releases, after the caller has done both

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.dtype == torch.float32 and t.is_contiguous()
        rc = self._L.mrs_p2p_all_reduce_sum_f32(self.handle, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("message larger than the mailboxes: use RCCL" if rc == -2 else (self._L.mrs_last_error() or b"").decode())
        return t

    def error(self) -> int:
        return int(self._L.mrs_p2p_error(self.handle))
