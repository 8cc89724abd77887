"""Host logic of the multi-GPU path: width bucketing, cost-balanced sharding and the
world-size-2 gather (gloo, CPU).  The recogniser is replaced by a deterministic stand-in so the
test needs no GPU; what is under test is the partition/gather, not the arithmetic."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from ocr_amd.crnn.batching import padded_waste, shard_buckets, width_buckets
from ocr_amd.crnn.parallel import my_buckets, recognize_sharded


def _widths(n, seed=0):
    rs = np.random.RandomState(seed)
    return [int(w) for w in rs.choice(np.arange(64, 1025, 32), size=n)]


def test_buckets_cover_everything_once():
    w = _widths(1000) + [3, 5, 7]
    b = width_buckets(w, max_batch=64, width_spread=1.25)
    flat = sorted(i for bb in b for i in bb)
    assert flat == list(range(len(w)))
    for bb in b:
        assert len(bb) <= 64
        ws = [max(w[i], 8) for i in bb]
        assert max(ws) <= min(ws) * 1.25
    assert padded_waste(w, b) < 0.12
    assert width_buckets([], 8) == []
    assert width_buckets([280] * 600, 256) == [list(range(256)), list(range(256, 512)), list(range(512, 600))]


def test_shards_balanced_and_deterministic():
    w = _widths(5000, seed=3)
    b = width_buckets(w, 256)
    for world in (1, 2, 4, 8):
        sh = shard_buckets(b, w, world)
        assert sorted(i for r in sh for bb in r for i in bb) == list(range(len(w)))
        cost = [sum(max(w[i] for i in bb) * len(bb) for bb in r) for r in sh]
        assert max(cost) <= 1.15 * (sum(cost) / world) + 256 * 1024
        assert sh == shard_buckets(b, w, world)
    assert my_buckets(w, 1, 2) == shard_buckets(b, w, 2)[1]


def _fake_predict(images):
    return ["%d:%d" % (im.shape[1], int(im.sum()) % 9973) for im in images]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(7)
    imgs = [rs.randint(0, 256, size=(32, int(w))).astype(np.uint8) for w in _widths(300, seed=5)]
    widths = [im.shape[1] for im in imgs]
    calls = []

    def predict(batch):
        calls.append(len(batch))
        return _fake_predict(batch)

    out = recognize_sharded(predict, imgs, widths, rank, world, max_batch=32)
    dist.barrier()
    q.put((rank, out == _fake_predict(imgs), sum(calls)))
    dist.destroy_process_group()


def test_world_size_2_gather_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(n for _, _, n in res) == 300 and all(n > 90 for _, _, n in res)


def test_single_rank_path():
    rs = np.random.RandomState(1)
    imgs = [rs.randint(0, 256, size=(32, w)).astype(np.uint8) for w in (64, 1024, 96, 96, 500)]
    out = recognize_sharded(_fake_predict, imgs, [im.shape[1] for im in imgs])
    assert out == _fake_predict(imgs)
