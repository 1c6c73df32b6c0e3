"""bench.py --gpus N without a launcher around it: the parent starts N rank processes itself, relays rank 0's
JSON line, exits with the children's status -- and never loads torch or touches a GPU (CPU-only checks with
stand-in rank commands; the real ranks run on the GPU box, tests/test_bench_rehearsal.py)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD_OK = textwrap.dedent("""
    import json, os, sys
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["CGPS_BENCH_CHILD"] == "1"
    print("noise from rank %d" % r)
    if r == 0:
        print(json.dumps({"n_gpus": w, "rank": r}))
""")
CHILD_FAIL = textwrap.dedent("""
    import os, sys, time
    if int(os.environ["RANK"]) == 1:
        sys.exit(7)
    time.sleep(600)          # a rank stuck in a collective whose peer has died
""")


def _run_parent(code, n=2):
    drv = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        import bench
        t0 = time.time()
        rc = bench.launch_ranks(%d, [], child_cmd=[sys.executable, "-c", %r])
        assert "torch" not in sys.modules, "the launcher must not load torch (no GPU context in the parent)"
        sys.stderr.write("elapsed %%.1f\\n" %% (time.time() - t0))
        sys.exit(rc)
    """) % (ROOT, n, code)
    return subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=120)


def test_launcher_spawns_ranks_and_relays_rank0_line():
    p = _run_parent(CHILD_OK, n=3)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                       # exactly ONE JSON line on stdout
    assert json.loads(lines[0]) == {"n_gpus": 3, "rank": 0}
    assert "noise from rank 1" in p.stderr and "noise from rank 0" in p.stderr


def test_launcher_ends_the_job_when_a_rank_dies():
    p = _run_parent(CHILD_FAIL, n=2)
    assert p.returncode == 7, (p.returncode, p.stderr)
    assert p.stdout.strip() == ""
    el = float([ln for ln in p.stderr.splitlines() if ln.startswith("elapsed")][0].split()[1])
    assert el < 60.0                                        # the surviving rank was ended, not waited for


def test_plain_command_takes_the_launcher_path_without_torch():
    """`python bench.py --gpus 2` with no WORLD_SIZE: main() goes to launch_ranks before anything imports torch
    (checked with a stand-in for the rank command through the same entry point)."""
    drv = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import bench
        seen = {}
        def fake(n, argv, **kw):
            seen["n"], seen["argv"], seen["torch"] = n, list(argv), "torch" in sys.modules
            return 0
        bench.launch_ranks = fake
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "3"]
        import os
        os.environ.pop("WORLD_SIZE", None)
        try:
            bench.main()
        except SystemExit as e:
            assert e.code == 0
        assert seen == {"n": 2, "argv": ["--gpus", "2", "--steps", "3"], "torch": False}, seen
    """) % ROOT
    p = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
