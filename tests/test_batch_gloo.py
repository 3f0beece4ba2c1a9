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


def test_bench_launcher_path_from_a_plain_shell():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its own ranks (fresh child processes under
    torch.distributed.run) and relays rank 0's line; --plumbing-check runs the rank plumbing on gloo without the flow
    computation, so the launcher path is covered on a box without GPUs."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-check"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1                                   # ONE line, from rank 0
    line = lines[0]
    assert line["plumbing_check"] is True and line["n_gpus"] == 2 and line["collectives_ok"] is True
    assert line["params"][:4] == [8.0, 0.5, 10.0, 5.0]       # rank 0's block (the default workload) reached rank 1
    assert "value" not in line and "metric" not in line      # not a measurement


def test_bench_launcher_refuses_without_devices_before_spawning():
    """On a box without GPUs `bench.py --gpus 2` stops at the device check (no CPU fallback), with a message and a
    non-zero status, instead of demanding a launcher."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs are visible: the launcher would run the real bench")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode != 0 and "HIP device(s) visible" in p.stderr and p.stdout.strip() == ""


def test_gather_to_root_matches_all_gather(tmp_path):
    """gather_fields_to_root: rank 0 gets what all_gather_fields gives everyone, the other ranks get None."""
    worker = textwrap.dedent("""
        import importlib.util, json, os, sys
        spec = importlib.util.spec_from_file_location("flow2d_batch", os.path.join(%r, "cuda-flow2d_amd", "batch.py"))
        batch = importlib.util.module_from_spec(spec); spec.loader.exec_module(batch)
        import torch
        rank, _, world = batch.init(backend="gloo")
        local = (torch.arange(2 * 3 * 4 * 8, dtype=torch.float32).view(2, 3, 4, 8) + 1000.0 * rank)
        everywhere = batch.all_gather_fields(local)
        at_root = batch.gather_fields_to_root(local)
        again = batch.gather_fields_to_root(local, out=at_root)   # into a buffer that already exists
        print(json.dumps({"rank": rank, "root_none": at_root is None,
                          "equal": bool(at_root is not None and torch.equal(at_root, everywhere) and again is at_root),
                          "own": bool(torch.equal(everywhere[rank], local))}))
        batch.shutdown()
    """) % ROOT
    script = tmp_path / "worker.py"
    script.write_text(worker)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        outs.append(__import__("json").loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["equal"] and not outs[0]["root_none"]
    assert all(o["root_none"] for o in outs[1:]) and all(o["own"] for o in outs)
