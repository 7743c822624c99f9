"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The encoder has no cross-frame state (LayerNorm only, reference configs/infer/*.yaml:49), so
frames shard embarrassingly: rank r owns a contiguous block of frames (the reference's own
multi-thread mode batches EXTRACTOR_BATCHSIZE consecutive frames the same way,
system/core.py:141-143).  The only exchange step is the gather of per-frame results to rank 0,
where the sequential pose-graph logic of system/core.py runs: descriptors (131 x 256 fp32 =
134 144 B per frame) and the speculative consecutive-frame edges (R, T, rmse, confidence,
6x6 information = 56 floats per edge, pipeline.EDGE_FLOATS).  That is ~1 MB per 8-frame round: latency-bound on xGMI,
so whole shards are gathered with ONE collective per step rather than per frame.

The reference has no inference-time collective; this module is new.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist



def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of rank `rank`; the first n_frames % world ranks get one extra."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _lone(force: bool) -> bool:
    """no process group, or one rank (and the caller did not ask for the collective anyway)"""
    return not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not force)


def gather_to_root(t: torch.Tensor, root: int = 0, force: bool = False) -> Optional[torch.Tensor]:
    """Equal-shaped per-rank tensor (F, ...) -> on `root`: (world*F, ...) in rank order; None elsewhere.
    A no-op (returns t) when torch.distributed is not initialised or world == 1.
    force=True issues the collective on a one-rank group too: the only way to execute the RCCL call path (communicator
    creation, the collective on the caller's stream, owned receive buffers) on a single-GPU box -- tests/test_gpu_rccl.py."""
    if _lone(force):
        return t
    world, rank = dist.get_world_size(), dist.get_rank()
    t = t.contiguous()
    if t.is_cuda and dist.get_backend() == "gloo":  # dry-run backend: gloo gathers host tensors only
        out = gather_to_root(t.cpu(), root, force)
        return out.to(t.device) if out is not None else None
    if rank == root:
        # one tensor of its own per rank (not views of one buffer: a collective is handed plain allocations), joined after
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, gather_list=parts, dst=root)
        return torch.cat(parts, dim=0)
    dist.gather(t, gather_list=None, dst=root)
    return None


def gather_step_results(desc: torch.Tensor, edges_packed: torch.Tensor, root: int = 0, force: bool = False):
    """One step's exchange: descriptors (F,131,S) and packed edges (E,EDGE_FLOATS) -> rank 0, as ONE collective:
    the two tensors travel as one flat fp32 row block per rank (a collective costs a launch and a stream hand-over
    on every rank whatever its size; the edge rows are 14 KB next to 8.6 MB of descriptors)."""
    if _lone(force):
        return desc, edges_packed
    nd, ne = desc.numel(), edges_packed.numel()
    flat = torch.cat([desc.reshape(-1), edges_packed.reshape(-1).to(desc.dtype)]).unsqueeze(0)   # (1, nd + ne)
    out = gather_to_root(flat, root, force)
    if out is None:
        return None, None
    world = out.shape[0]
    d = out[:, :nd].reshape((world * desc.shape[0],) + tuple(desc.shape[1:]))
    e = out[:, nd:].reshape((world * edges_packed.shape[0],) + tuple(edges_packed.shape[1:]))
    return d, e


def exchange_halo(last_desc: torch.Tensor, last_pcd: Optional[torch.Tensor], force: bool = False):
    """Ring hand-over of a block's LAST frame to the rank that owns the following block: the first frame of rank r's
    block is registered against the last frame of rank r-1's (reference odometry.py:103-127 registers every new scan
    against its predecessor).  Sends (descriptor (131,S), scan (3,N) or None) to rank+1, returns what rank-1 sent --
    one point-to-point message per rank and step (134 KB + 786 KB at 65 536 points).  Rank 0 receives the last frame of
    the whole window: the predecessor of ITS first frame in the NEXT step (the caller keeps it until then)."""
    if _lone(force):
        return last_desc, last_pcd  # one rank: its own last frame precedes its next block
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [last_desc.reshape(-1)] + ([last_pcd.reshape(-1)] if last_pcd is not None else [])
    flat = torch.cat(parts).contiguous()
    host = flat.is_cuda and dist.get_backend() == "gloo"  # dry-run backend: gloo moves host tensors only
    send = flat.cpu() if host else flat
    recv = torch.empty_like(send)
    ops_ = [dist.P2POp(dist.isend, send, (rank + 1) % world), dist.P2POp(dist.irecv, recv, (rank - 1) % world)]
    for req in dist.batch_isend_irecv(ops_):
        req.wait()
    recv = recv.to(flat.device) if host else recv
    nd = last_desc.numel()
    return recv[:nd].view_as(last_desc), (recv[nd:].view_as(last_pcd) if last_pcd is not None else None)
