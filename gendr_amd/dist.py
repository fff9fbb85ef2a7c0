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


# bench.py sets this to a list; every collective of gather_views then appends a (start, end) pair of
# torch.cuda.Event recorded on the current stream around it (CUDA tensors only).  None = no timing.
PROFILE_EVENTS = None


def shard_range(n, rank=None, world=None, group=None):
    """Contiguous [start, stop) slice of ``n`` batch items owned by ``rank`` (remainder to the low ranks)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(t, rank=None, world=None, group=None):
    """The slice of the leading (batch / view) axis this rank renders."""
    a, b = shard_range(t.shape[0], rank, world, group)
    return t[a:b]


class _Timed:
    """Brackets a collective with two events on the current stream when bench.py asked for it."""

    def __init__(self, t):
        self.on = PROFILE_EVENTS is not None and t.is_cuda

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            PROFILE_EVENTS.append((self.a, b))
        return False


def _check_equal_blocks(n_local, group):
    """all_gather_into_tensor / reduce_scatter_tensor need the same block size on every rank: unequal blocks hang
    or corrupt on NCCL / RCCL instead of failing.  One tiny all-reduce (min and max of the local size) and a host read on
    EVERY call: whether to run it must not depend on anything a rank knows by itself (a per-rank cache keyed on the local
    block size let a ragged step send one rank into the all-gather while another issued this all-reduce -- mismatched
    collectives, i.e. the hang the check exists to prevent).  Callers that shard evenly pass
    ``assume_equal_blocks=True`` and skip it."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError('gather_views: the equal-block check reads a value back on the host and cannot run inside a HIP graph capture; '
                           'callers that shard evenly pass assume_equal_blocks=True (bench.py does)')
    t = torch.tensor([n_local, -n_local], dtype=torch.int64)
    if dist.get_backend(group) != 'gloo':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    lo, hi = -int(t[1]), int(t[0])
    if lo != hi:
        raise ValueError('gather_views needs equally sized per-rank view blocks (got %d..%d views per rank); '
                         'pad the batch or shard it evenly' % (lo, hi))


def _direct_all_gather(out, x, group):
    """The all-gather as ONE round of point-to-point transfers: every rank sends its block to every peer and receives every
    peer's block, all 2 (N - 1) transfers posted together (``batch_isend_irecv``: one RCCL group call).  On a node whose GPUs
    are fully connected by point-to-point xGMI links (MI355X: 7 links per GPU) every transfer has its own link, so the
    exchange takes one block time instead of the N - 1 serialised hops of a ring (0.9 ms for config 4's RGBA
    views).  Same result as ``all_gather_into_tensor``, bit for bit (pure copies)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = x.shape[0]
    out[rank * n:(rank + 1) * n].copy_(x)
    ops = []
    for step in range(1, world):                     # peers in rotating order: no two ranks start on the same receiver
        to, frm = (rank + step) % world, (rank - step) % world
        ops.append(dist.P2POp(dist.isend, x, dist.get_global_rank(group, to) if group is not None else to, group))
        ops.append(dist.P2POp(dist.irecv, out[frm * n:(frm + 1) * n], dist.get_global_rank(group, frm) if group is not None else frm, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def _direct_reduce_scatter(out, grad, group):
    """Backward of the direct all-gather: rank r receives block r of every peer's gradient in one round and sums the N blocks in
    rank order (a fixed order: deterministic, unlike a ring's)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = out.shape[0]
    recv = torch.empty((world,) + tuple(out.shape), dtype=grad.dtype, device=grad.device)
    recv[rank].copy_(grad[rank * n:(rank + 1) * n])
    ops = []
    for step in range(1, world):
        to, frm = (rank + step) % world, (rank - step) % world
        ops.append(dist.P2POp(dist.isend, grad[to * n:(to + 1) * n], dist.get_global_rank(group, to) if group is not None else to, group))
        ops.append(dist.P2POp(dist.irecv, recv[frm], dist.get_global_rank(group, frm) if group is not None else frm, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.sum(recv, dim=0, out=out)


def _algorithm(direct):
    """'direct' | 'collective': the argument, else GENDR_ALLGATHER, else the library's collective."""
    import os
    if direct is None:
        direct = os.environ.get('GENDR_ALLGATHER', 'collective') == 'direct'
    return bool(direct)


class _GatherViews(torch.autograd.Function):
    """All-gather of equally sized per-rank view blocks; backward = reduce-scatter of the gradient
    (each rank receives the sum over ranks of the gradient w.r.t. its own block)."""

    @staticmethod
    def forward(ctx, x, group, checked, direct=False):
        ctx.group = group
        ctx.direct = direct
        world = dist.get_world_size(group)
        if not checked:
            _check_equal_blocks(x.shape[0], group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        with _Timed(x):
            if direct:
                _direct_all_gather(out, x, group)
            else:
                dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, grad):
        group = ctx.group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n = grad.shape[0] // world
        if ctx.direct:
            grad = grad.contiguous()
            out = torch.empty((n,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
            with _Timed(grad):
                _direct_reduce_scatter(out, grad, group)
            return out, None, None, None
        if dist.get_backend(group) == 'gloo':          # gloo has no reduce_scatter: all-reduce a private copy and slice
            total = grad.clone(memory_format=torch.contiguous_format)   # never reduce in place into autograd's own buffer
            with _Timed(total):
                dist.all_reduce(total, group=group)
            return total[rank * n:(rank + 1) * n].clone(), None, None, None
        grad = grad.contiguous()
        out = torch.empty((n,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
        with _Timed(grad):
            dist.reduce_scatter_tensor(out, grad, group=group)
        return out, None, None, None


def gather_views(images, group=None, assume_equal_blocks=False, direct=None):
    """[B_local, ...] on every rank -> [world * B_local, ...] on every rank, differentiable.
    A no-op without an initialised process group (single GPU).  Every rank must hand over the same number of
    views; that is verified with one small all-reduce per call unless ``assume_equal_blocks`` (callers that
    sharded with an even ``shard_range`` already know).  The check reads the result back on the host: a step that is to be
    captured in a HIP graph has to pass ``assume_equal_blocks=True`` (the check raises inside a capture instead of breaking it).
    ``direct=True`` (or GENDR_ALLGATHER=direct): the one-round point-to-point form (``_direct_all_gather``) instead of the
    library's collective, whose algorithm RCCL chooses -- on point-to-point xGMI a ring serialises N - 1 hops; not measured on a
    multi-GPU node by any session, so the library's collective stays the default."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return images
    return _GatherViews.apply(images, group, assume_equal_blocks, _algorithm(direct))


def sum_over_ranks(t, group=None):
    """In-place sum of a (small) tensor over ranks, e.g. the gradient of geometry shared by all views."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return t
