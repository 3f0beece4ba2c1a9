"""flow2d_batch, the C++ multi-GPU batch driver (host layer + librccl, no Python on its path), with one rank on the
GPU box: pairs from raw files through OpticalFlowBatch2D (lanes, lock-step groups formed by the object), RCCL broadcast
of the parameter block, grouped send/recv gather to rank 0, flow fields written as the reference's float32 raws.  Every
field must equal the oracle's, and the two ways of starting ranks (threads of one process, separate processes with an
id file) must deliver the same bits."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "cuda-flow2d_amd", "host", "flow2d_batch")


def run_tool(args, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([TOOL] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    assert p.returncode == 0, (p.returncode, p.stderr[-1500:])
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    return lines[0]


@pytest.mark.parametrize("constancy,tool_constancy", [(0, 0), (1, 1)])
def test_batch_tool_matches_the_oracle(flow2d, oracle, tmp_path, constancy, tool_constancy):
    w, h, n = 208, 144, 7
    p = (3, 0.5, 2, 5, 35.0, 0.001, 0.001, 5, 1.5)
    pairs_dir, out_dir = tmp_path / "pairs", tmp_path / "out"
    pairs_dir.mkdir()
    out_dir.mkdir()
    pairs = [oracle.synthetic_pair(w, h, 2.0 * np.cos(k), 2.0 * np.sin(k), seed=200 + k, noise=True) for k in range(n)]
    for k, (f0, f1) in enumerate(pairs):
        f0.astype("<f4").tofile(pairs_dir / ("pair_%04d_0.raw" % k))
        f1.astype("<f4").tofile(pairs_dir / ("pair_%04d_1.raw" % k))
    common = ["--pairs", n, "--width", w, "--height", h, "--lanes", 2, "--group", 3, "--levels", p[0], "--scale", p[1],
              "--outer", p[2], "--inner", p[3], "--alpha", p[4], "--median", p[7], "--sigma", p[8], "--constancy",
              tool_constancy, "--pairs-dir", pairs_dir]
    line = run_tool(["--gpus", 1, "--out-dir", out_dir] + common)
    assert line["world"] == 1 and line["pairs"] == n and line["pairs_per_s"] > 0
    for k, (f0, f1) in enumerate(pairs):
        ou, ov, _ = oracle.compute_flow(f0, f1, *p, constancy)
        u = np.fromfile(out_dir / ("flow_%04d_u.raw" % k), "<f4").reshape(h, w)
        v = np.fromfile(out_dir / ("flow_%04d_v.raw" % k), "<f4").reshape(h, w)
        assert np.array_equal(u, ou) and np.array_equal(v, ov), k
    # the same job as one rank of a one-process "world" started with an id file: same digest of all fields
    other = run_tool(["--rank", 0, "--world", 1, "--id-file", tmp_path / "nccl.id"] + common)
    assert other["flows_fnv1a"] == line["flows_fnv1a"]


def test_batch_tool_synthetic_and_exit_codes(tmp_path):
    line = run_tool(["--gpus", 1, "--pairs", 5, "--width", 160, "--height", 96, "--levels", 3, "--outer", 2, "--lanes", 2,
                     "--group", 2, "--repeat", 2])
    assert line["pairs"] == 5 and line["repeat"] == 2 and line["group"] == 2
    env = dict(os.environ)
    # a pair file that does not exist: exit code 2, like the CLI for a frame that cannot be loaded
    p = subprocess.run([TOOL, "--gpus", "1", "--pairs", "1", "--width", "64", "--height", "48", "--pairs-dir", str(tmp_path)],
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
    assert p.returncode == 2
    p = subprocess.run([TOOL, "--no-such-flag"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60)
    assert p.returncode == 3
    p = subprocess.run([TOOL, "--gpus", "99"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60)
    assert p.returncode == 1


def test_run_batch8_script_with_one_rank(tmp_path):
    """tools/run_batch8.sh, the turnkey launcher of the one-process-per-GPU form (fresh child processes, an id file and a run
    id in a directory of the job's own), at --world 1 on the one GPU of the box: same digest as the threaded form, the
    job's directory gone afterwards; a stale id file of another run at a fixed path does not confuse the process form."""
    job = ["--pairs", 3, "--width", 160, "--height", 96, "--levels", 3, "--outer", 2, "--lanes", 2, "--group", 2, "--repeat", 2,
           "--constancy", 3]
    line = run_tool(["--gpus", 1] + job)
    before = set(os.listdir("/tmp"))
    p = subprocess.run([os.path.join(ROOT, "tools", "run_batch8.sh"), "--world", "1"] + [str(a) for a in job],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    script = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][0])
    assert script["world"] == 1 and script["flows_fnv1a"] == line["flows_fnv1a"]
    assert not [n for n in set(os.listdir("/tmp")) - before if n.startswith("flow2d_batch.")]
    stale = tmp_path / "nccl.id"
    stale.write_bytes(b"some-older-run\n" + bytes(128))
    other = run_tool(["--rank", 0, "--world", 1, "--id-file", stale, "--run-id", "fresh"] + job)
    assert other["flows_fnv1a"] == line["flows_fnv1a"] and not stale.exists()


def test_bench_multi_rank_branches_with_two_ranks_on_this_gpu():
    """`bench.py --gpus 2` end to end on the one-GPU box: both ranks use GPU 0 and talk over gloo
    (--rehearse-on-one-gpu), so the world > 1 branches -- shard rule, distinct pairs per rank, parameter broadcast,
    barriers, max over ranks, the batch leg's two gathers -- run against real flow fields.  Not a measurement."""
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--workload",
                        "cfg2_1024_grey", "--steps", "8", "--warmup", "2", "--repeats", "2", "--no-host-entry-leg"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    line = lines[0]
    assert line["n_gpus"] == 2 and "rehearsal" in line and line["value"] > 0 and line["scaling"] == "weak"
    check = line["output_check"]
    assert check["ok"] and check["graph_replay_equals_eager"] and check["oracle"]["first_pair_equals_cpu_oracle"]
    gather = line["batch"]["gather"]
    assert line["batch"]["pairs_total_per_step"] == 16 and line["batch"]["output_check"]["ok"]
    assert gather["own_block_intact"] and gather["every_pair_present"] and gather["root_equals_all_gather"]
