"""Data-parallel recognition of a shared list of lines across ranks (one process per GPU).

No collective on the data path: every rank computes the same
deterministic partition locally, recognises its own buckets on its own GPU and the decoded
strings are gathered on the host (`torch.distributed.all_gather_object`, any backend; gloo is
enough since only Python strings move).
"""
from .batching import shard_buckets, width_buckets


def my_buckets(widths, rank, world_size, max_batch=256, width_spread=1.25):
    buckets = width_buckets(widths, max_batch, width_spread)
    return shard_buckets(buckets, widths, world_size)[rank]


def recognize_sharded(predict_lines, images, widths, rank=0, world_size=1, gather=None,
                      max_batch=256, width_spread=1.25):
    """predict_lines: callable(list of images) -> list of str (e.g. CRNN.predict_lines).
    widths: resized width of every image (the sort key).  gather: callable(obj) -> list of the
    objects of all ranks (defaults to torch.distributed.all_gather_object when world_size > 1).
    Returns the texts of ALL images, in input order, on every rank."""
    mine = my_buckets(widths, rank, world_size, max_batch, width_spread)
    part = {}
    for b in mine:
        for i, t in zip(b, predict_lines([images[i] for i in b])):
            part[i] = t
    if world_size == 1:
        parts = [part]
    elif gather is not None:
        parts = gather(part)
    else:
        import torch.distributed as dist
        parts = [None] * world_size
        dist.all_gather_object(parts, part)
    out = [None] * len(images)
    for p in parts:
        for i, t in p.items():
            out[i] = t
    assert all(t is not None for t in out)
    return out
