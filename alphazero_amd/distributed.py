"""The two exchange steps of multi-GPU self-play, over torch.distributed (RCCL on GPUs,
gloo in CPU tests).  Games shard across ranks with no data-path collective; these replace
the reference's Redis model server and SQLite sink between processes:

  broadcast_model / broadcast_packed     <-  ModelServer.update / latest   (alphazero/model_server.py:19-40)
  gather_records / gather_records_device <-  ReplayBufferDataset.add_game   (alphazero/replay_buffer.py:165-167)

The *_packed / *_device forms are the ones the self-play loop uses: weights travel in kernel-ready form
and records never leave device memory; the others move torch modules and host-side record lists.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import GameRecord

RECORD_BYTES = C.sizeof(GameRecord)


def shard_uids(rank, world_size, games_per_rank):
    """(first_game_uid, uid_stride) for az_engine_config: rank r owns uids r*G+g + k*world*G."""
    return rank * games_per_rank, world_size * games_per_rank


def _unique_state_tensors(model):
    """state_dict tensors with aliases removed (the reference's tower repeats one block)."""
    seen, out = set(), []
    for t in model.state_dict().values():
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            out.append(t)
    return out


def broadcast_model(model, version, src=0, group=None):
    """One broadcast of [version | all weights and BN statistics] from `src`; every rank returns
    the version and has the weights loaded in place.  ~137 KB for the 6x32 net: latency-bound."""
    tensors = _unique_state_tensors(model)
    dev = tensors[0].device
    flat = torch.cat([torch.tensor([float(version)], dtype=torch.float64, device=dev).view(torch.uint8)] +
                     [t.detach().contiguous().view(-1).view(torch.uint8) for t in tensors])
    dist.broadcast(flat, src, group=group)
    version = int(flat[:8].view(torch.float64).item())
    off = 8
    with torch.no_grad():
        for t in tensors:
            n = t.numel() * t.element_size()
            t.copy_(flat[off:off + n].view(t.dtype).view(t.shape))
            off += n
    return version


def broadcast_packed(net, version, src=0, group=None):
    """New weights for the <= 32-channel HIP network in kernel-ready form: ONE in-place broadcast of
    `net.wire` (FusedC4Net: 256-byte header {version, num_blocks} + packed blob, ~80 KB) from `src`.
    The publishing rank packs once (FusedC4Net.load_module); receivers evaluate straight from the
    buffer the collective wrote -- no state_dict, no re-packing, no copies.  Returns the version on
    every rank (one 16-byte host read).  Replaces ModelServer.update / latest
    (alphazero/model_server.py:19-40) between self-play ranks."""
    hdr = net.wire[:16].view(torch.int64)
    if dist.get_rank(group) == src:
        hdr.copy_(torch.tensor([int(version), int(net.num_blocks)], dtype=torch.int64))
    if net.wire.is_cuda and dist.get_backend(group) == 'gloo':
        # CPU-only rehearsals of the multi-rank path: gloo moves host memory, so stage through it
        staged = net.wire.cpu()
        dist.broadcast(staged, src, group=group)
        net.wire.copy_(staged)
    else:
        dist.broadcast(net.wire, src, group=group)
    version, blocks = hdr.tolist()
    net.num_blocks = int(blocks)
    return int(version)


def gather_records_device(records, head, dst=0, group=None):
    """Finished games of every rank collected on `dst` without leaving device memory:
    (records, head) as returned by SelfPlayEngine.take_finished_device on each rank ->
    (uint8 tensor [sum n, RECORD_BYTES] on `dst` in rank order -- an empty tensor elsewhere --, OR of
    the ranks' error bits, this rank's own count).  One all_gather of the 8-byte heads, whose host read is the only
    synchronisation, then one gather of the records padded to the largest count.  Replaces
    ReplayBufferDataset.add_game across processes (alphazero/replay_buffer.py:165-167, 28-47)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    home = records.device
    if records.is_cuda and dist.get_backend(group) == 'gloo':
        # gloo has no device gather: CPU-only rehearsals of the multi-rank path stage through the host
        head = head.cpu()
    heads = [torch.empty_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    hs = torch.stack(heads).tolist()                     # one host read for all counts
    counts = [h[0] for h in hs]
    flags = 0
    for h in hs:
        flags |= h[1]
    cap = max(counts)
    empty = records[:0]
    if cap == 0:
        return empty, flags, counts[rank]
    if records.shape[0] >= cap:
        send = records[:cap]
    else:
        # a rank whose engine (and so its record buffer) is smaller than another rank's largest hand-over: every
        # rank must send `cap` rows, so the short buffer is padded (its own count is below its own size)
        send = torch.zeros(cap, records.shape[1], dtype=records.dtype, device=records.device)
        send[:records.shape[0]] = records
    if send.device != head.device:
        send = send.to(head.device)
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return torch.cat([bufs[r][:counts[r]] for r in range(world)]).to(home), flags, counts[rank]
    dist.gather(send, None, dst=dst, group=group)
    return empty, flags, counts[rank]


def records_to_bytes(records):
    """list of az_game_record (ctypes) -> uint8 array [n, RECORD_BYTES]."""
    out = np.zeros((len(records), RECORD_BYTES), np.uint8)
    for i, r in enumerate(records):
        out[i] = np.frombuffer(bytes(r), np.uint8)
    return out


def bytes_to_records(arr):
    arr = np.ascontiguousarray(arr, np.uint8).reshape(-1, RECORD_BYTES)
    return [GameRecord.from_buffer_copy(arr[i].tobytes()) for i in range(arr.shape[0])]


def allreduce_gradients(model, group=None):
    """Data-parallel training step, the exchange part: every parameter gradient -- and every
    floating-point buffer, i.e. the batch-norm running statistics each rank has just updated from its
    own share of the batch -- averaged over the ranks with ONE all-reduce of a flat fp32 bucket
    (137 KB for the 6x32 net, 1.25 MB for 10x128: latency-bound on xGMI, so one large message instead
    of one per tensor).  Call between backward() and optimizer.step(); ranks that start from the same
    weights then stay bit-identical, so no weight broadcast is needed afterwards.  The reference
    trains in one process (training.py:42-75)."""
    seen, grads, bufs = set(), [], []
    for p in model.parameters():                       # the tower repeats one block: each tensor once
        if p.grad is not None and p.data_ptr() not in seen:
            seen.add(p.data_ptr())
            grads.append(p.grad)
    for b in model.buffers():
        if b.dtype.is_floating_point and b.data_ptr() not in seen:
            seen.add(b.data_ptr())
            bufs.append(b)
    tensors = grads + bufs
    if not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def gather_record_bytes(raw, dst=0, device=None, group=None):
    """Finished games of every rank collected on `dst`, in the raw form of
    SelfPlayEngine.drain_finished_bytes: uint8 [n, RECORD_BYTES] in, uint8 [sum n, RECORD_BYTES] out on
    `dst` (rank order), an empty array elsewhere; dst=None returns them on every rank.  Fixed-size
    records, one count exchange + one padded all_gather; nothing is parsed on the way."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = device or ('cuda' if dist.get_backend(group) == 'nccl' else 'cpu')
    mine = torch.from_numpy(np.ascontiguousarray(raw, np.uint8).reshape(-1, RECORD_BYTES)).to(device)
    parts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, torch.tensor([mine.shape[0]], dtype=torch.int64, device=device), group=group)
    counts = torch.cat(parts).tolist()                  # one host read for all counts
    cap = max(max(counts), 1)
    padded = torch.zeros(cap, RECORD_BYTES, dtype=torch.uint8, device=device)
    padded[:mine.shape[0]] = mine
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    if dst is not None and rank != dst:
        return np.zeros((0, RECORD_BYTES), np.uint8)
    return np.concatenate([bufs[r][:counts[r]].cpu().numpy() for r in range(world)], axis=0)


def gather_records(records, dst=0, device=None, group=None):
    """gather_record_bytes for lists of az_game_record: list of all ranks' records on `dst`, []
    elsewhere."""
    return bytes_to_records(gather_record_bytes(records_to_bytes(records), dst, device, group))
