"""Multi-GPU execution: one process per GPU, samples sharded, one result gather.

Frame samples are independent (weights read-only, every normalisation statistic
is per sample - SURVEY 8e), so the path shards with NO data-path collective:
rank r runs the contiguous slice ``shard_range(B, r, world)`` of the batch on its
own GPU.  The only exchange is the gather of the (small) score tensors towards
the host-side LP solver: one flat padded ``all_gather_into_tensor`` over
RCCL/xGMI (backend "nccl" on ROCm), a few MB per step - latency-bound, so a
single flat buffer beats per-tensor collectives on the point-to-point xGMI mesh.
The reference has no distributed code at all; this is new.
"""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous, balanced split of B samples: first (B % world) ranks get one extra."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def flatten_results(results):
    """results: list of per-sample (det, links, new, end) device tensors -> (flat fp32, layout).
    layout lets ``unflatten_results`` rebuild the tensors on any rank."""
    parts, layout = [], []
    for det, links, new, end in results:
        item = [tuple(det.shape), [tuple(l.shape) for l in links], tuple(new.shape), tuple(end.shape)]
        layout.append(item)
        parts += [det.reshape(-1)] + [l.reshape(-1) for l in links] + [new.reshape(-1), end.reshape(-1)]
    flat = torch.cat(parts) if parts else torch.zeros(0)
    return flat, layout


def unflatten_results(flat, layout):
    out, o = [], 0

    def take(shape):
        nonlocal o
        n = 1
        for s in shape:
            n *= s
        t = flat[o:o + n].view(*shape)
        o += n
        return t

    for dshape, lshapes, nshape, eshape in layout:
        det = take(dshape)
        links = [take(s) for s in lshapes]
        out.append((det, links, take(nshape), take(eshape)))
    return out


def gather_flat(flat, group=None, equal=False, force=False):
    """All-gather variable-length fp32 vectors: returns the list of every rank's vector.
    One length exchange + one padded all_gather_into_tensor (no per-tensor collectives);
    ``equal=True`` (every rank holds the same number of elements) skips the length exchange and
    its device-to-host read-back.  A one-rank group returns its input without any collective unless
    ``force`` (tests / ``bench.py --force-dist``: the N-rank call sequence on one GPU)."""
    if not dist.is_available() or not dist.is_initialized():
        return [flat]
    if dist.get_world_size(group) == 1 and not force:
        return [flat]
    world = dist.get_world_size(group)
    if equal:
        out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(out, flat.contiguous(), group=group)
        return list(out.view(world, -1).unbind(0))
    n = torch.tensor([flat.numel()], dtype=torch.int64, device=flat.device)
    sizes = torch.empty(world, dtype=torch.int64, device=flat.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes = sizes.tolist()
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=flat.dtype, device=flat.device)
    pad[:flat.numel()] = flat
    out = torch.empty(world * mx, dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * mx:r * mx + sizes[r]] for r in range(world)]


def gather_results(results, group=None, same_layout=False, force=False):
    """Per-rank list of per-sample results -> list over ALL samples in global (rank-major) order.
    ``same_layout=True`` (every rank holds identically shaped samples, e.g. the synthetic
    benchmark) skips the python-object exchange of shapes: one length + one data collective.
    ``force``: run the pack / collective / unpack sequence even in a one-rank group."""
    if not dist.is_available() or not dist.is_initialized():
        return list(results)
    if dist.get_world_size(group) == 1 and not force:
        return list(results)  # single process: the results already are the global list (no packing, no copy)
    flat, layout = flatten_results(results)
    world = dist.get_world_size(group)
    if same_layout:
        layouts = [layout] * world
    else:
        layouts = [None] * world
        dist.all_gather_object(layouts, layout, group=group)
    flats = gather_flat(flat, group, equal=same_layout, force=force)
    out = []
    for f, l in zip(flats, layouts):
        out += unflatten_results(f, l)
    return out
