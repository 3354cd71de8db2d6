"""Multi-GPU exchange for the sharded batch-expand step.

One process per GPU.  Samples and obstacles are replicated; rank g owns a contiguous range of the
library's cell-sorted sample order (`mpfmt_set_shard`), builds the CSC columns of its own samples and
sweeps its own edges.  The only data-path exchange is ONE all-gather of the per-shard free-edge masks
(RCCL over xGMI when the process group backend is "nccl"; gloo on CPU in the tests), preceded by a
world-sized all-gather of the mask lengths so the shards can be padded to a common size.
"""
import torch


class DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, nelem, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def all_gather_mask(local_words, dist, world, device=None):
    """local_words: 1-D int64 tensor (this rank's free-edge mask words, may be empty).
    Returns (gathered, counts): `gathered` is a [world, max_words] int64 tensor holding every rank's
    words (zero padded), `counts` the true word count of each rank.  Two collectives: lengths, payload."""
    device = local_words.device if device is None else device
    cnt = torch.tensor([local_words.numel()], dtype=torch.int64, device=device)
    counts = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, cnt)
    mx = max(int(counts.max().item()), 1)
    send = torch.zeros(mx, dtype=torch.int64, device=device)
    if local_words.numel():
        send[:local_words.numel()] = local_words
    out = torch.empty(world * mx, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, send)
    return out.view(world, mx), counts


def split_gathered(gathered, counts):
    """Per-rank word tensors with the padding removed."""
    return [gathered[g, :int(counts[g].item())] for g in range(gathered.shape[0])]


def sharded_step(ctx, r, dist, world, device):
    """One batch-expand step on this rank's shard of `ctx` (a `_lib.Context` with set_shard done):
    r-disc graph + edge sweep on the GPU, then the all-gather of the free-edge mask.
    Returns (local nnz, gathered mask [world, max_words], counts)."""
    nnz = ctx.graph_step_device(r)
    _, _, _, fptr = ctx.graph_device_ptrs()
    words = (nnz + 63) // 64
    if words:
        local = torch.as_tensor(DevArray(fptr, words), device=device)
    else:
        local = torch.zeros(0, dtype=torch.int64, device=device)
    gathered, counts = all_gather_mask(local, dist, world, device)
    return nnz, gathered, counts
