"""Multi-GPU for the sampling path: prompt-sharded replicas.

Every sample's 200-step trajectory, decode and vocode are independent, so the global batch is cut
into contiguous per-rank slices; there is NO collective inside a sample.  The only communication is
  * one RCCL broadcast of the hot-path weights from rank 0 at start-up (bucketed flat buffers; over
    xGMI a broadcast is per-link bound, ~150 GB/s => ~12-25 ms for 1.7-3.2 GB, amortised to zero),
  * an optional gather of the finished waveforms on rank 0.
The reference has no inference-time distribution at all (its collectives live in dead training code,
SURVEY.md §2.4); one process per GPU with torch.distributed (backend "nccl" == RCCL on ROCm).

RNG contract under sharding: the reference draws noise for the GLOBAL batch from one CPU generator
(ddim.py:191,351).  Each rank therefore draws the same global-shaped tensors from the same seed and
keeps its slice (`DDIMSampler.noise_shard`), so an N-rank run reproduces the single-process result.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # ALDM_DIST_BACKEND=gloo lets two ranks share ONE GPU (tests); RCCL needs one GPU per rank
            backend = os.environ.get("ALDM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [start, stop) of the global batch owned by `rank` (as even as possible;
    a prompt's n_gen candidates stay on one rank when the caller shards prompts, ddpm.py:1562)."""
    base, extra = divmod(n_global, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def candidate_rows(n_prompts: int, n_gen: int, rank: int, world: int) -> torch.Tensor:
    """Rows of the GLOBAL candidate-major batch (row = candidate * n_prompts + prompt, the layout `torch.cat([z] * n_gen)` of
    ddpm.py:1515-1530 gives) that belong to `rank` when PROMPTS are sharded contiguously: all n_gen candidates of the rank's
    prompts, again candidate-major — so the re-ranking of ddpm.py:1559-1564 (`best = i + argmax * B`) runs locally with B =
    the rank's prompt count."""
    lo, hi = shard_range(n_prompts, rank, world)
    p = torch.arange(lo, hi)
    return (torch.arange(n_gen)[:, None] * n_prompts + p[None, :]).reshape(-1)


def broadcast_tensors(tensors: Iterable[torch.Tensor], src: int = 0, bucket_bytes: int = 512 << 20,
                      even_alone: bool = False) -> int:
    """One-time weight broadcast: tensors are packed into flat buckets (few large collectives instead
    of ~2000 tiny ones), broadcast from `src`, and copied back.  Returns bytes sent.  even_alone: run the
    collectives in a 1-rank group too (tests: loads RCCL and launches its broadcast on a single GPU)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not even_alone):
        return 0
    total = 0
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size, total
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.broadcast(flat, src=src)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        total += flat.numel() * flat.element_size()
        bucket, size = [], 0

    last = None
    for t in tensors:
        if not t.is_floating_point():
            continue
        nbytes = t.numel() * t.element_size()
        kind = (t.device, t.dtype)   # a flat bucket holds one device and one dtype (no silent promotion)
        if bucket and (size + nbytes > bucket_bytes or kind != last):
            flush()
        last = kind
        bucket.append(t.data)
        size += nbytes
    flush()
    return total


def broadcast_module(module: torch.nn.Module, src: int = 0, even_alone: bool = False) -> int:
    """Broadcast every parameter and buffer of the hot-path modules from rank `src`, then drop the
    packed-weight caches so the kernels re-pack from the received values."""
    sent = broadcast_tensors(list(module.parameters()) + list(module.buffers()), src=src, even_alone=even_alone)
    for m in module.modules():
        # everything derived from the old values goes: packed weights, cached context K/V projections, captured graphs
        if hasattr(m, "invalidate_packed"):
            m.invalidate_packed()
        if hasattr(m, "_pk"):
            m._pk = None
        if hasattr(m, "_kv"):
            m._kv = None
        if hasattr(m, "_graph_cache"):
            from .ddim import drop_graph_entries
            drop_graph_entries(m._graph_cache)
    return sent


def gather_waveforms(local, dst: int = 0):
    """Collect per-rank np.float32 [b, 1, T] waveforms on `dst` in rank order (host-side result gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    import numpy as np
    return np.concatenate(out, axis=0)
