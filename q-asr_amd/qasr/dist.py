"""Utterance-sharded multi-GPU inference helpers: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).  The path has exactly two exchange steps
(SURVEY §8e): a one-time broadcast of the packed int-weight blob from the rank that calibrated it, and a
per-step gather of greedy tokens (int32 [B_local, T']) to rank 0.  No collective sits between layers:
utterances are independent in static-quantisation mode."""
import torch
import torch.distributed as dist


def shard_range(n_utts: int, world: int, rank: int):
    """Contiguous utterance shard [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, extra = divmod(n_utts, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_bytes(data, src: int, device):
    """Broadcast a bytes object from `src` to every rank (length first, then the payload as uint8)."""
    rank = dist.get_rank()
    n = torch.tensor([len(data) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return data if rank == src else buf.cpu().numpy().tobytes()


def broadcast_blob(blob, src: int, device):
    """The packed model from the calibrating rank to every rank; a receiving rank validates it (qasr_blob_check: every
    offset, index and shape, host-only) before it may reach an engine."""
    data = broadcast_bytes(blob, src, device)
    if dist.get_rank() != src:
        from . import engine
        engine.blob_check(data)
    return data


def broadcast_tensors(tensors, src: int, device):
    """Broadcast a list of float tensors whose shapes are known on every rank."""
    out = []
    for t in tensors:
        t = t.to(device).contiguous()
        dist.broadcast(t, src)
        out.append(t)
    return out


def gather_tokens(tokens: torch.Tensor, dst: int = 0, bufs=None):
    """Gather equally-shaped int32 token tensors to `dst`; returns the list on dst, None elsewhere.
    `bufs` lets the caller reuse pre-allocated receive buffers across steps."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst and bufs is None:
        bufs = [torch.empty_like(tokens) for _ in range(world)]
    dist.gather(tokens, bufs if rank == dst else None, dst=dst)
    return bufs if rank == dst else None


def gather_ragged_tokens(tokens: torch.Tensor, dst: int = 0):
    """Gather token tensors whose first dimension differs across ranks (uneven shards)."""
    world = dist.get_world_size()
    n = torch.tensor([tokens.shape[0]], dtype=torch.int64, device=tokens.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(s.item() for s in sizes))
    pad = torch.zeros((m,) + tuple(tokens.shape[1:]), dtype=tokens.dtype, device=tokens.device)
    pad[:tokens.shape[0]] = tokens
    got = gather_tokens(pad, dst)
    if got is None:
        return None
    return torch.cat([g[:int(s.item())] for g, s in zip(got, sizes)], dim=0)
