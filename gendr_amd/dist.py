"""Multi-GPU helpers: one process per GPU, the batch (view) axis is sharded.

The rasterizer itself needs no collective: every batch item is rendered and differentiated
independently (``kernel.cu:714,903``: ``bn = i / (is*is)``, no cross-batch term).  A collective appears
only in callers whose loss couples views (BASELINE config 4: all-gather of rendered views; shared-geometry
optimisation such as ``experiments/opt_shape.py:86``: sum of the vertex gradient over ranks).
``torch.distributed`` with backend "nccl" is RCCL on ROCm (xGMI within a node); the same code runs on
"gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank=None, world=None):
    """Contiguous [start, stop) slice of ``n`` batch items owned by ``rank`` (remainder to the low ranks)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(t, rank=None, world=None):
    """The slice of the leading (batch / view) axis this rank renders."""
    a, b = shard_range(t.shape[0], rank, world)
    return t[a:b]


class _GatherViews(torch.autograd.Function):
    """All-gather of equally sized per-rank view blocks; backward = reduce-scatter of the gradient
    (each rank receives the sum over ranks of the gradient w.r.t. its own block)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, grad):
        group = ctx.group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        grad = grad.contiguous()
        n = grad.shape[0] // world
        if dist.get_backend(group) == 'gloo':          # gloo has no reduce_scatter: all-reduce and slice
            dist.all_reduce(grad, group=group)
            return grad[rank * n:(rank + 1) * n].clone(), None
        out = torch.empty((n,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
        dist.reduce_scatter_tensor(out, grad, group=group)
        return out, None


def gather_views(images, group=None):
    """[B_local, ...] on every rank -> [world * B_local, ...] on every rank, differentiable.
    A no-op without an initialised process group (single GPU)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return images
    return _GatherViews.apply(images, group)


def sum_over_ranks(t, group=None):
    """In-place sum of a (small) tensor over ranks, e.g. the gradient of geometry shared by all views."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t
