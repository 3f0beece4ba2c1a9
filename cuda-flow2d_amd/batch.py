"""Sharding of independent image pairs over the GPUs of one node: one process per GPU,
pair k -> rank k mod world, no data-path collective.  torch.distributed is used only for the start/stop
barrier, the broadcast of rank 0's parameter block, the max-over-ranks of the elapsed time and, after the work, ONE
all_gather of the flow fields (or of per-pair digests).
Backend "nccl" is RCCL on ROCm; "gloo" is what the CPU tests use."""
import os


def world_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend="nccl", device=None):
    """Joins the process group when launched by torch.distributed.run; a no-op for a single process."""
    rank, local_rank, world = world_info()
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run even a single rank joins a group
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            kwargs = {"device_id": device} if (device is not None and backend == "nccl") else {}
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def pairs_of_rank(total_pairs, rank, world):
    """Global pair indices owned by `rank`: k mod world == rank (every pair exactly once, balanced +-1)."""
    return list(range(rank, total_pairs, world))


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def _collective_device(device):
    """Tensors of a collective must live where the backend works: the current GPU for nccl (RCCL), host for gloo."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return device
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"


def _staged_through_host(local):
    """gloo gathers host tensors only: a device block handed to it (bench.py --rehearse-on-one-gpu, N ranks sharing one
    GPU) goes through host memory.  RCCL takes device blocks as they are."""
    import torch.distributed as dist
    return local.is_cuda and dist.get_backend() == "gloo"


def broadcast_params(values, device=None):
    """Rank 0's parameter block (a flat sequence of numbers, e.g. levels, scale, outer, inner, alpha, ...) on every
    rank: all ranks of a batch solve with the same parameters whatever their own command line said."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(x) for x in values]
    t = torch.tensor([float(x) for x in values], dtype=torch.float64, device=_collective_device(device))
    dist.broadcast(t, src=0)
    return t.tolist()


def all_gather_fields(local, out=None):
    """ONE collective for the flow fields of a batch: every rank contributes its block `local` (a tensor of any one
    shape on all ranks -- bench.py: [2, pairs_per_rank, H, W], the u planes then the v planes of the rank's lock-step
    group; on the GPU for RCCL) and receives [world, *local.shape]; with the shard rule of pairs_of_rank, pair k is
    rank k % world's entry k // world.  all_gather moves each byte once per receiving rank
    (a ring all-gather over xGMI), unlike a sum-reduction of a dense zero-padded array."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0) if out is None else out.copy_(local.unsqueeze(0))
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if _staged_through_host(local):
        return out.copy_(all_gather_fields(local.cpu()))
    # the collective's own layout is the concatenation along dim 0: [world * pairs_per_rank, ...]
    dist.all_gather_into_tensor(out.view((-1,) + tuple(local.shape[1:])), local.contiguous())
    return out


def gather_fields_to_root(local, out=None, root=0):
    """The same blocks delivered to ONE rank (BASELINE.json: "broadcast/gather"): rank `root` receives
    [world, *local.shape], every other rank sends its block once and gets None.  Each byte crosses xGMI once in total
    (the root's seven links in parallel) instead of once per receiving rank as in all_gather_fields: 1/world of the
    traffic when only the root consumes the fields.  RCCL runs it as grouped send/recv."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0) if out is None else out.copy_(local.unsqueeze(0))
    world, rank = dist.get_world_size(), dist.get_rank()
    if _staged_through_host(local):
        at_root = gather_fields_to_root(local.cpu(), root=root)
        if rank != root:
            return None
        return at_root.to(local.device) if out is None else out.copy_(at_root)
    if rank != root:
        dist.gather(local.contiguous(), dst=root)
        return None
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.gather(local.contiguous(), gather_list=list(out.unbind(0)), dst=root)
    return out


def gather_fields(local, total_pairs, height, width, device=None):
    """local: {global_pair_index: (u, v)} float32 arrays of this rank's pairs.  Returns on every rank a float32
    array [total_pairs, 2, height, width] with all flow fields (host-array front end of all_gather_fields; ranks
    with fewer pairs than the others pad their block)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if initialised else 1
    per_rank = -(-total_pairs // world) if total_pairs else 0
    block = torch.zeros((per_rank, 2, height, width), dtype=torch.float32,
                        device=_collective_device(device) if initialised else "cpu")
    for k, (u, v) in local.items():
        block[k // world, 0] = torch.from_numpy(np.ascontiguousarray(u)).to(block.device)
        block[k // world, 1] = torch.from_numpy(np.ascontiguousarray(v)).to(block.device)
    everything = all_gather_fields(block).cpu().numpy()
    out = np.zeros((total_pairs, 2, height, width), np.float32)
    for k in range(total_pairs):
        out[k] = everything[k % world, k // world]
    return out


def max_over_ranks(value, device=None):
    """The slowest rank's value (the bench contract times the whole job by its slowest rank)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests(local, total_pairs, device=None):
    """local: {global_pair_index: float digest}.  Returns the list of all pairs' digests on every rank."""
    import torch
    import torch.distributed as dist
    initialised = dist.is_available() and dist.is_initialized()
    mine = torch.zeros(total_pairs, dtype=torch.float64, device=_collective_device(device) if initialised else "cpu")
    for k, d in local.items():
        mine[k] = d
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)  # owners are disjoint, so the sum is a gather
    return mine.tolist()


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
