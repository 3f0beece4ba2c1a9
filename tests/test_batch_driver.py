"""The rank logic of the C++ multi-GPU driver flow2d_batch (cuda-flow2d_amd/host/batch_driver.cpp) at world sizes 2, 3
and 8 WITHOUT a GPU: flow2d_batch_selftest runs the same RunBatchRank as the product over an in-process loopback
(threads + shared memory instead of librccl) with planes in host memory and a stamp instead of the flow computation
(u = 2 f0 + 1, v = f1 - f0 + levels).  Pinned here: pair k -> rank k mod world, the padded gather blocks and their byte
offsets, the flow_%04d files rank 0 writes, that only rank 0's parameters count, and that a rank which fails locally
takes every rank out with the same exit code instead of leaving the others blocked in a collective.
(The reference has no counterpart: one context on device 0, src/utils/cuda_utils.cpp:43.)"""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "cuda-flow2d_amd", "host", "flow2d_batch_selftest")


@pytest.fixture(scope="module", autouse=True)
def selftest_binary():
    """Built by `make -C cuda-flow2d_amd/host` (__graft_entry__.build()); a checkout without it builds just this target
    (g++ only: it links neither HIP nor RCCL)."""
    if not os.path.exists(TOOL):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(TOOL), "flow2d_batch_selftest"])
    assert os.path.exists(TOOL)


def run(args, timeout=60):
    return subprocess.run([TOOL] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


def frames_of(k, w, h):
    y, x = np.mgrid[0:h, 0:w]
    f0 = (1000.0 * k + x + 0.5 * y).astype(np.float32)
    f1 = (f0 + np.float32(0.25) * np.float32(k + 1)).astype(np.float32)
    return f0, f1


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("pairs", [5, 64])
def test_pairs_ranks_blocks_and_files(tmp_path, world, pairs):
    w, h, levels = 100, 12, 7  # pitch 512 B: rows are padded, so a tight-for-pitched mix-up shows
    pairs_dir, out_dir = tmp_path / "pairs", tmp_path / "out"
    pairs_dir.mkdir()
    out_dir.mkdir()
    for k in range(pairs):
        f0, f1 = frames_of(k, w, h)
        f0.astype("<f4").tofile(pairs_dir / ("pair_%04d_0.raw" % k))
        f1.astype("<f4").tofile(pairs_dir / ("pair_%04d_1.raw" % k))
    p = run(["--world", world, "--pairs", pairs, "--width", w, "--height", h, "--levels", levels, "--repeat", 2,
             "--pairs-dir", pairs_dir, "--out-dir", out_dir, "--print-layout"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][0])
    pitch = (w * 4 + 255) // 256 * 256
    per_rank = -(-pairs // world)
    plane = pitch * h
    assert line["world"] == world and line["pairs"] == pairs and line["repeat"] == 2
    assert line["pitch_bytes"] == pitch and line["pairs_per_block"] == per_rank
    assert line["gather_bytes_per_rank"] == per_rank * 2 * plane  # ranks with fewer pairs pad their block
    # pair k: rank k mod world, slot k // world of that rank's block
    assert line["layout"] == [[k, k % world, k // world, (k % world) * per_rank * 2 * plane + (k // world) * 2 * plane]
                              for k in range(pairs)]
    digest = 1469598103934665603
    for k in range(pairs):
        f0, f1 = frames_of(k, w, h)
        u = np.fromfile(out_dir / ("flow_%04d_u.raw" % k), "<f4").reshape(h, w)
        v = np.fromfile(out_dir / ("flow_%04d_v.raw" % k), "<f4").reshape(h, w)
        assert np.array_equal(u, np.float32(2) * f0 + np.float32(1)), k
        # `levels` reaches every rank through the broadcast of rank 0's block (the other ranks are started with 1)
        assert np.array_equal(v, f1 - f0 + np.float32(levels)), k
    assert len(os.listdir(out_dir)) == 2 * pairs


def test_digest_does_not_depend_on_the_world_size(tmp_path):
    lines = []
    for world in (1, 2, 5):
        p = run(["--world", world, "--pairs", 11, "--width", 72, "--height", 40])
        assert p.returncode == 0, p.stderr[-2000:]
        lines.append(json.loads(p.stdout.splitlines()[-1]))
    assert len({l["flows_fnv1a"] for l in lines}) == 1


@pytest.mark.parametrize("phase", ["init", "load", "warmup", "pass", "gather-alloc", "broadcast", "gather"])
@pytest.mark.parametrize("rank", [0, 2])
def test_a_failing_rank_takes_every_rank_out(phase, rank):
    """flow2d_batch_selftest returns 99 when the ranks disagree on the exit code, and the timeout catches a rank left
    blocked in a collective (round 3's driver returned from the failing rank alone).  "broadcast" / "gather": the
    collective fails on the rank after it took part (round 4 returned from that rank without an agreement)."""
    p = run(["--world", 3, "--pairs", 7, "--fail-rank", rank, "--fail-phase", phase], timeout=30)
    expected = 0 if (phase == "gather-alloc" and rank != 0) else 1  # only rank 0 allocates the gathered buffer
    assert p.returncode == expected, (p.returncode, p.stderr[-1000:])
    if expected:
        assert "another rank failed" in p.stderr
        assert not [x for x in p.stdout.splitlines() if x.startswith("{")]  # no result line from a failed job


@pytest.mark.parametrize("phase", ["comm-prepare", "comm-connect"])
@pytest.mark.parametrize("rank", [0, 1, 7])
@pytest.mark.parametrize("side", ["threads", "files"])
def test_a_rank_that_cannot_bring_the_communicator_up(tmp_path, phase, rank, side):
    """StartBatchRank: the local prerequisites and the communicator's own rendezvous sit between two agreements of the side
    channel (memory for ranks that are threads, files for ranks that are processes -- here both over threads), so a rank
    that fails there keeps the others out of ncclCommInitRank / makes them abort what they connected; all return 1."""
    extra = ["--file-rendezvous", tmp_path / "id", "--run-id", "r%d" % rank] if side == "files" else []
    p = run(["--world", 8, "--pairs", 9, "--fail-rank", rank, "--fail-phase", phase] + extra, timeout=30)
    assert p.returncode == 1, (p.returncode, p.stderr[-1000:])
    assert ("no rank enters it" if phase == "comm-prepare" else "did not come up") in p.stderr
    assert not [x for x in p.stdout.splitlines() if x.startswith("{")]


@pytest.mark.parametrize("phase", ["broadcast-absent", "allreduce-absent", "gather-absent"])
@pytest.mark.parametrize("rank", [0, 2])
@pytest.mark.parametrize("side", ["threads", "files"])
def test_a_rank_that_never_enters_a_collective_frees_its_peers(tmp_path, phase, rank, side):
    """The rank raises the side channel's flag and leaves; its peers, blocked inside the collective, see the flag (the
    product aborts its communicator, the loopback leaves its hub) and every rank returns 1 within the timeout."""
    extra = ["--file-rendezvous", tmp_path / "id"] if side == "files" else []
    p = run(["--world", 3, "--pairs", 7, "--fail-rank", rank, "--fail-phase", phase] + extra, timeout=30)
    assert p.returncode == 1, (p.returncode, p.stderr[-1000:])
    assert not [x for x in p.stdout.splitlines() if x.startswith("{")]


def test_file_rendezvous_ignores_another_runs_files(tmp_path):
    """Stale side-channel files of an earlier job at the same prefix (another run id, or none) do not count as posts: the
    job still needs -- and gets -- every rank's own post, and removes its files afterwards."""
    prefix = tmp_path / "id"
    for r in range(3):  # an earlier run that was killed: "all fine" posts and a raised flag
        (tmp_path / ("id.s0.r%d" % r)).write_text("old 1")
        (tmp_path / ("id.s1.r%d" % r)).write_text("old 0")
    (tmp_path / "id.abort.old").write_text("old")
    p = run(["--world", 3, "--pairs", 5, "--file-rendezvous", prefix, "--run-id", "new"], timeout=30)
    assert p.returncode == 0, p.stderr[-1000:]
    assert json.loads(p.stdout.splitlines()[-1])["pairs"] == 5
    assert not [n for n in os.listdir(tmp_path) if n.startswith("id.abort.new")]
    # the job's own posts are gone (it overwrote and then removed the stage files of its ranks)
    assert sorted(os.listdir(tmp_path)) == ["id.abort.old"]


def test_a_failed_job_does_not_wedge_the_next_one_at_the_same_prefix(tmp_path):
    """a failed job leaves its abort flag and posts behind on purpose (slower ranks must still read them).  Without
    --run-id both jobs used the id "0" and the healthy job took the leftovers for its own: every rank left at once with "another
    rank failed".  A job whose ranks share a process now makes a fresh id; flow2d_batch refuses --rank with peers and no --run-id."""
    prefix = tmp_path / "id"
    p = run(["--world", 3, "--pairs", 7, "--fail-rank", 2, "--fail-phase", "gather-absent", "--file-rendezvous", prefix], timeout=30)
    assert p.returncode == 1
    assert [n for n in os.listdir(tmp_path) if ".abort." in n]  # the failed job's flag stays
    p = run(["--world", 3, "--pairs", 7, "--file-rendezvous", prefix], timeout=30)
    assert p.returncode == 0, p.stderr[-1000:]
    assert json.loads(p.stdout.splitlines()[-1])["pairs"] == 7


def test_constancy_values_of_the_library_are_accepted():
    for c in (0, 1, 2, 3):  # Grey, Gradient, LogDerivatives, GradientUntiled (data_structs.h)
        assert run(["--world", 2, "--pairs", 2, "--constancy", c], timeout=30).returncode == 0
    assert run(["--world", 2, "--pairs", 2, "--constancy", 4], timeout=30).returncode == 3


def test_exit_codes_of_a_job(tmp_path):
    pairs_dir = tmp_path / "pairs"
    pairs_dir.mkdir()
    w, h = 32, 8
    for k in (0, 1, 3):  # pair 2 (rank 2 of 4) is missing: exit code 2 on every rank
        for j, f in enumerate(frames_of(k, w, h)):
            f.astype("<f4").tofile(pairs_dir / ("pair_%04d_%d.raw" % (k, j)))
    p = run(["--world", 4, "--pairs", 4, "--width", w, "--height", h, "--pairs-dir", pairs_dir], timeout=30)
    assert p.returncode == 2, p.stderr[-1000:]
    # an output directory that does not exist: 255, the reference's code for an output that cannot be written
    p = run(["--world", 2, "--pairs", 3, "--width", w, "--height", h, "--out-dir", tmp_path / "nope"], timeout=30)
    assert p.returncode == 255
    assert run(["--no-such-flag"]).returncode == 3
    # parameters out of range: judged after the broadcast, on rank 0's block, by every rank alike
    for bad in (["--pairs", -1], ["--width", 0], ["--height", 2.5], ["--lanes", 0], ["--scale", 1.5], ["--repeat", 0]):
        p = run(["--world", 3] + bad, timeout=30)
        assert p.returncode == 3 and "out of range" in p.stderr, (bad, p.returncode, p.stderr[-300:])


def test_more_ranks_than_pairs(tmp_path):
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    p = run(["--world", 8, "--pairs", 3, "--width", 40, "--height", 10, "--out-dir", out_dir, "--print-layout"])
    assert p.returncode == 0, p.stderr[-1000:]
    line = json.loads(p.stdout.splitlines()[-1])
    assert line["pairs_per_block"] == 1 and [r for _, r, _, _ in line["layout"]] == [0, 1, 2]
    assert len(os.listdir(out_dir)) == 6
