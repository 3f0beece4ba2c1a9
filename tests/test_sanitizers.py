"""The committed sanitizer recipe (`make san` in oracle/ and cuda-flow2d_amd/host/), run on the CPU: AddressSanitizer +
UndefinedBehaviorSanitizer over the oracle's whole pipelines and over the CPU-only parts of the host layer, and
ThreadSanitizer over the multi-rank driver's loopback (ranks as threads).  GPU sanitizers do not exist on the pool; this
is what can be instrumented.  A sanitizer report makes the binaries exit non-zero (-fno-sanitize-recover=all)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cuda-flow2d_amd", "host")
ORACLE = os.path.join(ROOT, "oracle")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")


def run(cmd, **kw):
    return subprocess.run([str(c) for c in cmd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=ENV, timeout=300, **kw)


def test_oracle_under_address_and_ub_sanitizers():
    assert run(["make", "-s", "-C", ORACLE, "san"]).returncode == 0
    san = run([os.path.join(ORACLE, "_san", "san_check")])
    plain = run([os.path.join(ORACLE, "_san", "san_check_plain")])
    assert san.returncode == 0, san.stderr[-3000:]
    assert "ERROR" not in san.stderr and "runtime error" not in san.stderr, san.stderr[-3000:]
    assert san.stdout.strip() == plain.stdout.strip() and "digest" in san.stdout


def test_host_layer_under_sanitizers(tmp_path):
    p = run(["make", "-s", "-C", HOST, "san"])
    assert p.returncode == 0, p.stderr[-2000:]
    check = run([os.path.join(HOST, "build_san", "host_san_check"), os.path.join(HOST, "settings_rub.xml"), tmp_path])
    assert check.returncode == 0 and "host_san_check ok" in check.stdout, check.stderr[-3000:]
    assert "runtime error" not in check.stderr and "AddressSanitizer" not in check.stderr
    out = tmp_path / "out"
    out.mkdir()
    # the multi-rank driver: a good job, uneven pairs, and a failing rank, under ASan + UBSan and under TSan
    for binary in ("flow2d_batch_selftest", "flow2d_batch_selftest_tsan"):
        tool = os.path.join(HOST, "build_san", binary)
        ok = run([tool, "--world", 5, "--pairs", 13, "--width", 70, "--height", 9, "--out-dir", out])
        assert ok.returncode == 0, (binary, ok.stderr[-3000:])
        assert "WARNING: ThreadSanitizer" not in ok.stderr and "AddressSanitizer" not in ok.stderr, ok.stderr[-3000:]
        # a rank failing in a pass; one that cannot connect the communicator; one that never enters a collective (the
        # side channel's flag frees its peers), over the thread and over the file side channel
        for extra in (["--fail-phase", "pass"], ["--fail-phase", "comm-connect"], ["--fail-phase", "broadcast-absent"],
                      ["--fail-phase", "gather-absent", "--file-rendezvous", tmp_path / ("side_" + binary)]):
            bad = run([tool, "--world", 4, "--pairs", 6, "--fail-rank", 3] + extra)
            assert bad.returncode == 1, (binary, extra, bad.returncode, bad.stderr[-3000:])
            assert "WARNING: ThreadSanitizer" not in bad.stderr and "AddressSanitizer" not in bad.stderr, bad.stderr[-3000:]
