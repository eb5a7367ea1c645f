"""Multi-GPU decomposition of a render: split the samples of every pixel over ranks.

The reference has one parallel construct, an OpenMP loop over image columns
(CW/raytracer/Source/main.cpp:121). Here the unit of distribution is the sample index:
a path's random stream is a function of (seed, pixel, global sample index) only, so
rank r rendering samples [b_r, e_r) of ALL pixels produces exactly the paths a single
GPU would produce for that range, and the per-rank images just add up. One process per
GPU; the only exchange is one sum-reduce of the fp32 image (3*W*H floats) to rank 0 per
render call -- RCCL over xGMI with the "nccl" backend, gloo in the CPU tests.
"""


def rank_sample_range(sample_begin, sample_count, rank, world):
    """Contiguous sub-range (begin, count) of rank `rank` out of `world`; the first
    `sample_count % world` ranks get one sample more."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(sample_count, world)
    begin = sample_begin + rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def reduce_image(rank_image, total, rank, world, group=None, via_host=False, always=False):
    """Sum the per-rank images into rank 0's running image `total` (None on other ranks).
    via_host: the backend cannot reduce device tensors (gloo with images in HBM): the image makes
    the round trip through host memory. always: call the collective even in a world of one rank
    (a rehearsal of the backend on a one-GPU machine)."""
    if world > 1 or always:
        import torch.distributed as dist
        if via_host and rank_image.is_cuda:
            staged = rank_image.cpu()
            dist.reduce(staged, dst=0, op=dist.ReduceOp.SUM, group=group)
            if rank == 0:
                rank_image.copy_(staged)
        else:
            dist.reduce(rank_image, dst=0, op=dist.ReduceOp.SUM, group=group)
    if rank == 0:
        total += rank_image


def render_distributed(render_fn, sample_begin, sample_count, rank, world, rank_image, total, group=None,
                       via_host=False, always=False):
    """render_fn(begin, count, image) must ADD samples [begin, begin+count) of every pixel into
    `image` (Scene.render_accumulate_dev, or any renderer with the same contract). After the
    call rank 0's `total` has gained the whole range [sample_begin, sample_begin+sample_count)."""
    begin, count = rank_sample_range(sample_begin, sample_count, rank, world)
    rank_image.zero_()
    if count:
        render_fn(begin, count, rank_image)
    reduce_image(rank_image, total, rank, world, group, via_host, always)


ITEM_SAMPLES = 16  # samples per work item of the path kernel (one lane adds them in sample order)


def rank_item_range(sample_begin, sample_count, rank, world, item=ITEM_SAMPLES):
    """rank_sample_range in whole work items: every share is a multiple of `item` samples except
    the last non-empty one, which takes the ragged rest. These are the shares render_chained needs."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    n_items = -(-sample_count // item)
    first, count = rank_sample_range(0, n_items, rank, world)
    lo = min(first * item, sample_count)
    hi = min((first + count) * item, sample_count)
    return sample_begin + lo, hi - lo


def render_chained(render_items_fn, fold_fn, sample_begin, sample_count, rank, world, image, group=None,
                   via_host=False):
    """The image ONE rank would produce for [sample_begin, sample_begin + sample_count), bit for bit,
    from `world` ranks. render_items_fn(begin, count) traces a share and keeps its item sums;
    fold_fn(image) joins them to the image's running values in sample order
    (Scene.render_items_dev / Scene.fold_items_dev). All ranks trace at once; then the image
    travels rank 0 -> 1 -> ... -> world-1 -> 0, each rank folding its share in turn (a 12.6 MB hop
    per rank at 1024 x 1024 against tens of milliseconds of tracing). On entry rank 0's `image`
    holds the running image (the other ranks' contents are overwritten); on return rank 0's does
    again. via_host: the backend cannot move device tensors (gloo)."""
    begin, count = rank_item_range(sample_begin, sample_count, rank, world)
    if count:
        render_items_fn(begin, count)
    if world == 1:
        if count:
            fold_fn(image)
        return
    import torch.distributed as dist

    def recv(src):
        if via_host and image.is_cuda:
            staged = image.cpu()
            dist.recv(staged, src=src, group=group)
            image.copy_(staged)
        else:
            dist.recv(image, src=src, group=group)

    def send(dst):
        dist.send(image.cpu() if via_host and image.is_cuda else image, dst=dst, group=group)

    if rank > 0:
        recv(rank - 1)
    if count:
        fold_fn(image)
    send((rank + 1) % world)
    if rank == 0:
        recv(world - 1)
