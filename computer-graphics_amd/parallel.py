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


def reduce_image(rank_image, total, rank, world, group=None, via_host=False):
    """Sum the per-rank images into rank 0's running image `total` (None on other ranks).
    via_host: the backend cannot reduce device tensors (gloo with images in HBM): the image makes
    the round trip through host memory."""
    if world > 1:
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
                       via_host=False):
    """render_fn(begin, count, image) must ADD samples [begin, begin+count) of every pixel into
    `image` (Scene.render_accumulate_dev, or any renderer with the same contract). After the
    call rank 0's `total` has gained the whole range [sample_begin, sample_begin+sample_count)."""
    begin, count = rank_sample_range(sample_begin, sample_count, rank, world)
    rank_image.zero_()
    if count:
        render_fn(begin, count, rank_image)
    reduce_image(rank_image, total, rank, world, group, via_host)
