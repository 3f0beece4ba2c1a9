"""world_size-2 CPU test (gloo) of the multi-GPU plumbing bench.py uses: pair sharding without a data-path
collective, parameter broadcast, barrier, max-over-ranks timing and the digest / flow-field gathers.  The per-pair work is stood in for by
the CPU oracle on a tiny pair (this test checks the sharding logic, not the HIP path)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import importlib, json, os, sys, time
    sys.path.insert(0, %r)
    import importlib.util
    spec = importlib.util.spec_from_file_location("flow2d_batch", os.path.join(%r, "cuda-flow2d_amd", "batch.py"))
    batch = importlib.util.module_from_spec(spec); spec.loader.exec_module(batch)
    from oracle import oracle as O
    import numpy as np
    rank, local_rank, world = batch.init(backend="gloo")
    total = 5
    mine = batch.pairs_of_rank(total, rank, world)
    digests, fields = {}, {}
    # rank 0's parameter block wins: rank 1 starts from a different (wrong) one
    levels, scale, outer, inner, alpha = batch.broadcast_params([3, 0.5, 2, 2, 35.0] if rank == 0 else [9, 0.9, 7, 7, 1.0])
    batch.barrier()
    t0 = time.perf_counter()
    for k in mine:
        f0, f1 = O.synthetic_pair(48, 32, 2.0 * np.cos(k), 2.0 * np.sin(k))
        u, v, _ = O.compute_flow(f0, f1, int(levels), scale, int(outer), int(inner), alpha, 0.001, 0.001, 5, 1.5)
        digests[k] = float(u.sum(dtype=np.float64) + 2.0 * v.sum(dtype=np.float64))
        fields[k] = (u, v)
    time.sleep(0.05 * (rank + 1))
    batch.barrier()
    elapsed = time.perf_counter() - t0
    slowest = batch.max_over_ranks(elapsed)
    allp = batch.gather_digests(digests, total)
    allf = batch.gather_fields(fields, total, 32, 48)
    field_digests = [float(allf[k, 0].sum(dtype=np.float64) + 2.0 * allf[k, 1].sum(dtype=np.float64)) for k in range(total)]
    print(json.dumps({"rank": rank, "world": world, "mine": mine, "elapsed": elapsed, "slowest": slowest,
                      "digests": allp, "field_digests": field_digests, "params": [levels, scale, outer, inner, alpha]}))
    batch.shutdown()
""") % (ROOT, ROOT)


def run_workers(tmp_path, world, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = []
    for rank in range(world):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(__import__("json").loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    return outs


def test_four_ranks_uneven_pair_counts(tmp_path):
    """5 pairs on 4 ranks: rank 0 owns two, the others one; the all_gather blocks are padded and every pair's
    field arrives on every rank exactly once."""
    outs = run_workers(tmp_path, 4, 29541)
    assert [o["mine"] for o in outs] == [[0, 4], [1], [2], [3]]
    assert all(o["digests"] == outs[0]["digests"] for o in outs) and all(d != 0.0 for d in outs[0]["digests"])
    assert all(o["field_digests"] == outs[0]["digests"] for o in outs)
    assert all(o["slowest"] == outs[0]["slowest"] for o in outs)
    assert all(o["params"] == [3.0, 0.5, 2.0, 2.0, 35.0] for o in outs)


def test_two_rank_sharding_with_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(__import__("json").loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["mine"] == [0, 2, 4] and outs[1]["mine"] == [1, 3]       # pair k -> rank k mod world
    assert sorted(outs[0]["mine"] + outs[1]["mine"]) == list(range(5))      # every pair exactly once
    assert outs[0]["slowest"] == outs[1]["slowest"] >= max(o["elapsed"] for o in outs) - 1e-9
    assert outs[0]["digests"] == outs[1]["digests"] and all(d != 0.0 for d in outs[0]["digests"])
    assert outs[0]["params"] == outs[1]["params"] == [3.0, 0.5, 2.0, 2.0, 35.0]     # rank 0's block everywhere
    assert outs[0]["field_digests"] == outs[1]["field_digests"] == outs[0]["digests"]  # gathered fields are the fields
    # single-process reference: the digests do not depend on how the pairs were sharded
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import oracle as O
    for k, d in enumerate(outs[0]["digests"]):
        f0, f1 = O.synthetic_pair(48, 32, 2.0 * np.cos(k), 2.0 * np.sin(k))
        u, v, _ = O.compute_flow(f0, f1, 3, 0.5, 2, 2, 35.0, 0.001, 0.001, 5, 1.5)
        assert d == float(u.sum(dtype=np.float64) + 2.0 * v.sum(dtype=np.float64))


def test_shard_rule():
    import importlib.util
    spec = importlib.util.spec_from_file_location("flow2d_batch", os.path.join(ROOT, "cuda-flow2d_amd", "batch.py"))
    batch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(batch)
    for total in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            shards = [batch.pairs_of_rank(total, r, world) for r in range(world)]
            assert sorted(sum(shards, [])) == list(range(total))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert batch.pairs_of_rank(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]  # config 4: 8 pairs per GPU
