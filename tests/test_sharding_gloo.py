"""The N>1 path on CPU: world_size 2 and 3 over gloo.  Each rank takes its hypothesis range,
gets that range's counts (from the oracle here; from the GPU kernel in production), packs its
local first maximum and merges with all-reduce(max).  The merged winner must equal the
single-rank first maximum, including ties."""
import os
import socket

import numpy as np
import pytest

from sfm_hip.sharding import merge_best, pack_key, shard_range, unpack_key


def test_shard_ranges_cover_everything_once():
    for total in (0, 1, 7, 100, 10**6 + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_key_order_is_count_then_lowest_index():
    assert pack_key(5, 100) > pack_key(4, 0)
    assert pack_key(5, 3) > pack_key(5, 4)
    assert unpack_key(pack_key(123, 456789)) == (123, 456789)
    with pytest.raises(ValueError):
        pack_key(1, 1 << 33)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, counts_all, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys = []
    for counts in counts_all:  # several problems merged in one call
        lo, hi = shard_range(len(counts), rank, world)
        local = counts[lo:hi]
        if hi > lo:
            i = int(np.argmax(local))  # first maximum of the shard
            keys.append(pack_key(int(local[i]), lo + i))
        else:
            keys.append(0)
    t = torch.tensor(keys, dtype=torch.int64)
    merge_best(t)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_merge_matches_single_rank_first_maximum(tmp_path, world):
    import torch.multiprocessing as mp

    rng = np.random.default_rng(world)
    counts_all = [rng.integers(0, 50, 1001), rng.integers(0, 3, 17), np.zeros(5, np.int64), rng.integers(0, 1000, 4096)]
    counts_all[0][[10, 700, 900]] = 77  # a three-way tie across shards: the lowest id must win
    mp.spawn(_worker, args=(world, _free_port(), counts_all, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        for k, counts in enumerate(counts_all):
            c, i = unpack_key(got[k])
            assert (c, i) == (int(counts.max()), int(np.argmax(counts)))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_in_process_shard_and_merge_names_the_single_rank_winner(world):
    """The index partition + packed-max merge without any communication library: for every world size the merged key
    is the first maximum of the whole range, also when the maximum occurs in several shards."""
    import numpy as np

    from sfm_hip import sharding

    rng = np.random.default_rng(world)
    for total in (1000, 12345, 8):
        counts = rng.integers(0, 50, size=total)
        counts[rng.integers(0, total, size=5)] = 77  # tied maxima scattered over the range
        want = (int(counts.max()), int(np.argmax(counts)))
        keys = []
        for r in range(world):
            lo, hi = sharding.shard_range(total, r, world)
            if hi > lo:
                local = counts[lo:hi]
                keys.append(sharding.pack_key(int(local.max()), lo + int(np.argmax(local))))
        assert sharding.unpack_key(max(keys)) == want
