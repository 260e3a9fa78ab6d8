"""
bench.py's rank launcher without a GPU: the children here are small Python programs, so only the watching logic runs
(first failing rank ends the launch, the others are stopped, wall-clock limit).  The GPU twin that goes through
eigd_comm_init is tests/test_gpu_path.py::test_bench_launcher_ends_when_a_rank_dies_before_the_communicator.
"""
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_launcher(tmp_path, child_src, nranks, limit_s):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(child_src))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        bench.__file__ = {str(child)!r}          # the launcher starts `python <this file> <argv>` per rank
        sys.exit(bench.launch_ranks({nranks}, argv=[], limit_s={limit_s}))
    """))
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120)
    return out, time.monotonic() - t0


def test_first_failing_rank_ends_the_launch(tmp_path):
    out, took = _run_launcher(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "2":
            print("rank 2 says: boom", file=sys.stderr)
            sys.exit(7)
        time.sleep(600)          # the other ranks "wait in the collective"
    """, nranks=4, limit_s=300)
    assert out.returncode == 1
    assert "rank 2 of 4 exited with code 7" in out.stderr
    assert "rank 2 says: boom" in out.stderr
    assert took < 30, took


def test_wall_clock_limit(tmp_path):
    out, took = _run_launcher(tmp_path, """
        import time
        time.sleep(600)
    """, nranks=2, limit_s=2)
    assert out.returncode == 1
    assert "exceeded 2 s" in out.stderr
    assert took < 30, took


def test_clean_launch_relays_rank0_stdout(tmp_path):
    out, _ = _run_launcher(tmp_path, """
        import os
        print("line from rank", os.environ["RANK"], "of", os.environ["WORLD_SIZE"], os.environ["EIGD_COMM_DIR"] != "")
    """, nranks=3, limit_s=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "line from rank 0 of 3 True"
