"""The N>1 path on CPU: world_size 2, gloo.  Checks the batch sharding, the barrier / max / gather plumbing that
bench.py uses, and that the union of the shards equals the global batch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from portfft_amd.sharding import shard_range
    for total in (1, 7, 8, 65536, 524288, 1000003):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(524288, 8, 3) == (196608, 262144)  # config 4: 65536 transforms per GPU
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_two_rank_gloo_run(oracle):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and len(r["table"]) == 2
    (lo0, hi0, s0, t0), (lo1, hi1, s1, t1) = r["table"]
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37)
    for got, exp in zip((s0, s1), r["expected"]):
        assert abs(got - exp) <= 1e-5 * exp
    assert r["max_elapsed"] >= max(t0, t1) - 1e-9


def test_forced_distributed_world_size_one():
    """PFFT_BENCH_FORCE_DIST=1: the whole multi-rank code path (rendezvous, groups, barrier, max, gather) at world size 1
    -- what `torchrun --nproc-per-node 1 bench.py --gpus 1` runs on a one-GPU box to prove the RCCL branch; here over
    gloo (a request for 'nccl' without a GPU must fall back to gloo collectively and say why)."""
    code = (
        "import os, sys, json\n"
        "sys.path.insert(0, %r)\n"
        "from portfft_amd.sharding import process_group\n"
        "out = {}\n"
        "for backend in ('gloo', 'nccl'):\n"
        "    pg = process_group(backend)\n"
        "    pg.barrier()\n"
        "    out[backend] = [pg.active, pg.backend, pg.max(2.5), pg.gather([1.0, 2.0]), pg.fallback_reason]\n"
        "pg.close()\n"
        "print(json.dumps(out))\n" % ROOT)
    env = dict(os.environ, PFFT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["gloo"][:4] == [True, "gloo", 2.5, [[1.0, 2.0]]] and r["gloo"][4] is None
    # no GPU here: the RCCL sub-group cannot come up, every rank (the one there is) agrees on gloo and reports the reason
    assert r["nccl"][0] is True and r["nccl"][2:4] == [2.5, [[1.0, 2.0]]]
    assert r["nccl"][1] == "gloo" and r["nccl"][4]
