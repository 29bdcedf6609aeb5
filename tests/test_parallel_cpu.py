"""The N > 1 path (sharding, ragged gather, bucketed gradient all-reduce, MAX-over-ranks timing) under
world_size = 2 with the gloo backend on CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from setok_amd import parallel as P
    try:
        # 1. sharding: contiguous, disjoint, covering
        n = 257
        s, e = P.shard_range(n, rank, world)
        spans = [None] * world
        dist.all_gather_object(spans, (s, e))
        assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
        images = torch.arange(n * 2, dtype=torch.float32).reshape(n, 2)
        assert torch.equal(P.shard_batch(images, rank, world), images[s:e])
        # 2. ragged gather: per-image token counts differ per rank (dynamic-k)
        g = torch.Generator().manual_seed(100 + rank)
        counts = torch.randint(1, 9, (e - s,), generator=g).tolist()
        packed = torch.cat([torch.full((c, 3), float(s + i)) for i, c in enumerate(counts)])
        rows, all_counts = P.gather_ragged(packed, counts)
        assert len(all_counts) == n and rows.shape[0] == sum(all_counts)
        off = 0
        for i, c in enumerate(all_counts):                      # image order == global order, rows labelled by image id
            assert bool((rows[off:off + c] == float(i)).all()); off += c
        # 3. bucketed gradient all-reduce == mean of per-rank gradients, several buckets
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(sz)) for sz in (1000, 37, 4096, 5)]
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        ncoll = P.allreduce_gradients(params, bucket_bytes=8192)
        assert ncoll >= 2
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, torch.full_like(p, (i + 1) * (1 + world) / 2.0))
        # 3b. the training step's per-module reduction: plain fp32 gradient tensors (the hand-written backward's buffers), summed
        #     over ranks (HeadTrainer divides by the world size inside AdamW), through the trainer's own hook
        from setok_amd.training import HeadTrainer
        tr = HeadTrainer.__new__(HeadTrainer)
        tr.group, tr.bucket_bytes, tr._comm_stream, tr._pending, tr._comm_bytes, tr._wait_events = None, 4096, None, [], 0, []
        gmod = {"out.weight": torch.full((300, 7), float(rank + 1)), "out.bias": torch.full((300,), 10.0 * (rank + 1))}
        tr._allreduce_module("out", gmod)
        assert torch.allclose(gmod["out.weight"], torch.full((300, 7), 3.0)) and torch.allclose(gmod["out.bias"], torch.full((300,), 30.0))
        assert tr.world == world
        st = tr.comm_stats()
        assert st["allreduce_bytes_per_step"] == (300 * 7 + 300) * 4 and st["exposed_allreduce_ms"] == 0.0 and st["world"] == world
        # 4. timing rule
        assert P.max_over_ranks(1.0 + rank) == float(world)
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_range_edge_cases():
    from setok_amd import parallel as P
    assert [P.shard_range(5, r, 8) for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    assert P.shard_range(256, 3, 8) == (96, 128)
    assert P.max_over_ranks(0.5) == 0.5          # not initialised -> identity


def test_bench_launches_n_ranks_by_itself():
    """`python bench.py --gpus N` without a launcher must become N ranks (torch.distributed.run on 127.0.0.1) and report `n_gpus: N` with one
    record per rank — the driver's SCALE command shape.  `--launch-check` runs the launcher and the rank bookkeeping on CPU over gloo."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and sorted(x["rank"] for x in d["ranks"]) == [0, 1]
    assert abs(d["slowest_seconds"] - 0.02) < 1e-9                      # MAX over the ranks
    # the launcher and the flag must agree: WORLD_SIZE=1 with --gpus 2 is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "disagree" in r.stderr
