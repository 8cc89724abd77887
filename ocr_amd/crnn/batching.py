"""Width bucketing and multi-GPU sharding of independent text lines.

Every line is independent, so data parallelism is a pure
host-side partition: sort by resized width, cut into buckets whose padding
waste is bounded, deal buckets to ranks by cost.  No collective is involved;
the only gather is of decoded strings on the host.
"""


def width_buckets(widths, max_batch=256, width_spread=1.25, min_cols=0):
    """Indices of `widths` grouped into buckets of <= max_batch lines whose widest line is at most
    width_spread x the narrowest.  Lines narrower than 8 (decoded as '' by predict) ride along in
    the first bucket: the library skips them.
    min_cols: a bucket is not cut for its width spread while it holds fewer than min_cols padded
    columns (lines x widest line).  A small batch is latency-bound on the device (the recurrence takes
    T sequential steps whatever the batch), so for a page's worth of crops one padded batch beats
    several exact ones; the per-line width masks keep the results identical either way."""
    order = sorted(range(len(widths)), key=lambda i: (widths[i], i))
    buckets, cur, lo = [], [], None
    for i in order:
        w = max(widths[i], 8)
        if cur and (len(cur) >= max_batch or (w > lo * width_spread and len(cur) * w >= min_cols)):
            buckets.append(cur)
            cur, lo = [], None
        if lo is None:
            lo = w
        cur.append(i)
    if cur:
        buckets.append(cur)
    return buckets


def padded_waste(widths, buckets):
    """Fraction of padded columns that are padding."""
    used = padded = 0
    for b in buckets:
        wm = max(widths[i] for i in b)
        padded += wm * len(b)
        used += sum(widths[i] for i in b)
    return 0.0 if padded == 0 else 1.0 - used / padded


def shard_buckets(buckets, widths, world_size):
    """Greedy longest-processing-time assignment of buckets to ranks, cost = padded columns.
    Returns a list (per rank) of lists of buckets.  Deterministic, so every rank can compute the
    same partition locally without communicating."""
    cost = [max(widths[i] for i in b) * len(b) for b in buckets]
    order = sorted(range(len(buckets)), key=lambda j: (-cost[j], j))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for j in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(buckets[j])
        load[r] += cost[j]
    return out
