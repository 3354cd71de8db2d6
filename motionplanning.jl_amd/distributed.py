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


class MaskGather:
    """The per-step exchange with ONE collective in the steady state.

    `all_gather_mask` needs two (lengths, then payload) because a rank does not know the others' mask lengths.  From the
    second step on every rank does: it has seen all lengths of the previous step, so all ranks derive the same capacity
    from them, put their own length in front of their words and gather capacity + 1 words each.  The lengths come back
    with the payload; if one exceeds the capacity (a rank's shard grew by more than the 5 % margin) every rank sees that
    and all of them repeat the payload exchange at the exact size -- the same decision everywhere, so the collectives
    stay matched.  Returns the same (gathered [world, >= max words], counts) as `all_gather_mask`: rows are zero padded
    beyond `counts[g]` words.  The returned tensor is a VIEW of a buffer the next call overwrites (one-step lifetime):
    clone it to keep a step's mask while the next gather is in flight."""

    def __init__(self, dist, world, device=None):
        self.dist, self.world, self.device = dist, world, device
        self.cap = None
        self._send = self._out = None
        self._last_m = 0

    def __call__(self, local_words):
        dist, world = self.dist, self.world
        device = local_words.device if self.device is None else self.device
        n = int(local_words.numel())
        if self.cap is None:
            gathered, counts = all_gather_mask(local_words, dist, world, device)
            self.cap = int(int(counts.max().item()) * 1.05) + 8
            return gathered, counts
        cap = self.cap
        if self._send is None or self._send.numel() != cap + 1:
            self._send = torch.zeros(cap + 1, dtype=torch.int64, device=device)
            self._out = torch.empty(world * (cap + 1), dtype=torch.int64, device=device)
            self._last_m = 0
        send = self._send
        send[0] = n
        m = min(n, cap)
        if m:
            send[1:1 + m] = local_words[:m]
        if self._last_m > m:
            send[1 + m:1 + self._last_m] = 0          # a shorter mask than last step: no stale words in the padding
        self._last_m = m
        dist.all_gather_into_tensor(self._out, send)
        view = self._out.view(world, cap + 1)
        counts = view[:, 0].clone()
        mx = int(counts.max().item())
        if mx <= cap:
            if mx < cap // 2:                          # shards shrank a lot: tighten for the next step
                self.cap = int(mx * 1.05) + 8
            return view[:, 1:], counts
        # a shard outgrew the capacity: every rank sees the same lengths and repeats the payload at the exact size
        self.cap = int(mx * 1.05) + 8
        send2 = torch.zeros(mx, dtype=torch.int64, device=device)
        if n:
            send2[:n] = local_words
        out2 = torch.empty(world * mx, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(out2, send2)
        return out2.view(world, mx), counts


def split_gathered(gathered, counts):
    """Per-rank word tensors with the padding removed."""
    return [gathered[g, :int(counts[g].item())] for g in range(gathered.shape[0])]


def sharded_step(ctx, r, dist, world, device, gather=None):
    """One batch-expand step on this rank's shard of `ctx` (a `_lib.Context` with set_shard done):
    r-disc graph + edge sweep on the GPU, then the all-gather of the free-edge mask (through `gather`, a MaskGather kept by
    the caller across steps, when given: one collective per step instead of two).
    Returns (local nnz, gathered mask [world, >= max_words], counts)."""
    nnz = ctx.graph_step_device(r)
    _, _, _, fptr = ctx.graph_device_ptrs()
    words = (nnz + 63) // 64
    if words:
        local = torch.as_tensor(DevArray(fptr, words), device=device)
    else:
        local = torch.zeros(0, dtype=torch.int64, device=device)
    gathered, counts = gather(local) if gather is not None else all_gather_mask(local, dist, world, device)
    return nnz, gathered, counts
