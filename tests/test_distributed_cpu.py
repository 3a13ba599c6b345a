"""N>1 path on CPU: world_size-2 gloo rehearsal of what bench.py does across
GPUs -- disjoint shards, distinct seeds, barrier, MAX of elapsed, SUM of units."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from svo_pro_universal_amd import dist_utils as du, synth
    dist = du.init("gloo", rank, world)
    lo, hi = du.shard_range(11, rank, world)
    seeds = [du.problem_seed(rank, i) for i in range(3)]
    # each rank builds its own scenes (no image rendering needed for the plumbing test)
    n_feat = sum(synth.make_align_scene(s, n_features=20 + rank, render_images=False).n_features for s in seeds)
    dist.barrier()
    elapsed, units = du.combine(dist, world, 0.5 + rank, n_feat)
    q.put((rank, lo, hi, seeds, elapsed, units))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_sharding_and_combine():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, s0, e0, u0), (r1, lo1, hi1, s1, e1, u1) = out
    assert (lo0, hi0, lo1, hi1) == (0, 6, 6, 11)          # disjoint, complete, balanced
    assert not set(s0) & set(s1)                            # distinct problems per rank
    assert e0 == e1 == 1.5                                  # MAX over ranks
    assert u0 == u1 == 3 * 20 + 3 * 21                      # SUM over ranks


def test_shard_range_properties():
    from svo_pro_universal_amd.dist_utils import shard_range
    for n in (0, 1, 7, 8, 1024, 1025):
        for w in (1, 2, 4, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
