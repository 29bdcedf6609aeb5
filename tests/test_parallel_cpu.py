"""The N > 1 path (sharding, ragged gather, bucketed gradient all-reduce — concatenating and zero-copy flat-bucket forms — MAX-over-ranks
timing) under world_size = 2 and 8 with the gloo backend on CPU, and the benchmark's self-launcher at 2 and 8 ranks."""
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
        tot1 = world * (world + 1) / 2.0
        assert torch.allclose(gmod["out.weight"], torch.full((300, 7), tot1)) and torch.allclose(gmod["out.bias"], torch.full((300,), 10.0 * tot1))
        assert tr.world == world
        st = tr.comm_stats()
        assert st["allreduce_bytes_per_step"] == (300 * 7 + 300) * 4 and st["exposed_allreduce_ms"] == 0.0 and st["world"] == world
        # 3c. the zero-copy form: gradients written into the views of pre-allocated flat buckets, the collective runs on the buckets themselves
        #     (bucket order = module order; a module larger than bucket_bytes splits at parameter boundaries)
        mods = {"out": [("out.weight", (300, 7)), ("out.bias", (300,))],
                "inter_encoder": [("inter_encoder.a", (5000,)), ("inter_encoder.b", (5000,)), ("inter_encoder.c", (10,))]}
        bk = P.GradBuckets(mods, bucket_bytes=24000)
        assert bk.order == [("out", 0), ("inter_encoder", 0), ("inter_encoder", 1)]
        for i, n in enumerate(bk.names()):
            bk.view(n).fill_(float((rank + 1) * (i + 1)))
        ptrs = {n: bk.view(n).data_ptr() for n in bk.names()}
        assert bk.allreduce("out") == 1 and bk.allreduce("inter_encoder") == 2
        tot = world * (world + 1) / 2.0
        for i, n in enumerate(bk.names()):
            assert bk.view(n).data_ptr() == ptrs[n] and torch.allclose(bk.view(n), torch.full_like(bk.view(n), tot * (i + 1))), n
        tr2 = HeadTrainer.__new__(HeadTrainer)
        tr2.group, tr2.bucket_bytes, tr2._comm_stream, tr2._pending, tr2._comm_bytes, tr2._wait_events, tr2.buckets = None, 4096, None, [], 0, [], bk
        for i, n in enumerate(bk.names()):
            bk.view(n).fill_(float(rank + 1))
        tr2._allreduce_module("out", {n: bk.view(n) for n in ("out.weight", "out.bias")})            # in the buckets: reduced in place, no copies
        assert torch.allclose(bk.view("out.weight"), torch.full((300, 7), tot)) and bk.view("out.weight").data_ptr() == ptrs["out.weight"]
        assert torch.allclose(bk.view("inter_encoder.a"), torch.full((5000,), float(rank + 1)))         # the other module's buckets untouched
        # 4. timing rule
        assert P.max_over_ranks(1.0 + rank) == float(world)
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_world_size_n_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_shard_range_edge_cases():
    from setok_amd import parallel as P
    assert [P.shard_range(5, r, 8) for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    assert P.shard_range(256, 3, 8) == (96, 128)
    assert P.max_over_ranks(0.5) == 0.5          # not initialised -> identity


@pytest.mark.parametrize("n", [2, 8])
def test_bench_launches_n_ranks_by_itself(n):
    """`python bench.py --gpus N` without a launcher must become N ranks (torch.distributed.run on 127.0.0.1) and report `n_gpus: N` with one
    record per rank — the driver's SCALE command shape.  `--launch-check` runs the launcher and the rank bookkeeping on CPU over gloo."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch-check"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and sorted(x["rank"] for x in d["ranks"]) == list(range(n))
    assert abs(d["slowest_seconds"] - 0.01 * n) < 1e-9                  # MAX over the ranks
    if n != 2:
        return
    # the launcher and the flag must agree: WORLD_SIZE=1 with --gpus 2 is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "disagree" in r.stderr
