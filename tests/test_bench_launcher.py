"""bench.py's own multi-rank launcher (`python bench.py --gpus N` without torchrun), exercised on the CPU: the parent
starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line, and fails when a
rank fails.  The children here are tests/launcher_child.py (gloo, world 2); on a GPU box they are bench.py itself."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "launcher_child.py")


def run_launcher(code):
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)


def test_launcher_relays_rank0_line_world2():
    r = run_launcher("import sys, bench; sys.exit(bench.launch_ranks(2, [], command=[sys.executable, %r, 'ok']))" % CHILD)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # rank 1's stdout goes to stderr, never to the JSON channel
    rec = json.loads(lines[0])
    assert rec == {"n_gpus": 2, "max": 2, "local_rank": 0, "master": "127.0.0.1"}
    assert "noise from a non-zero rank" in r.stderr


def test_launcher_fails_when_a_rank_fails():
    r = run_launcher("import sys, bench; sys.exit(bench.launch_ranks(2, [], command=[sys.executable, %r, 'fail'], timeout=20))" % CHILD)
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr


def test_launcher_timeout_reports_every_ranks_last_stage():
    """A rank that hangs with the communicator half up (the first real multi-GPU run could): after the launcher's time-out the
    last stage line of EVERY rank is on stderr, the children are gone and the exit code is 124 -- in well under 30 s here."""
    import time
    t0 = time.time()
    r = run_launcher("import sys, bench; sys.exit(bench.launch_ranks(2, [], command=[sys.executable, %r, 'hang'], timeout=6))" % CHILD)
    assert time.time() - t0 < 30
    assert r.returncode == 124, (r.returncode, r.stderr)
    assert "time-out after 6 s; last stage of every rank" in r.stderr
    assert "rank 0 (running): [bench rank 0/2" in r.stderr and "stage first step done" in r.stderr
    assert "rank 1 (running): [bench rank 1/2" in r.stderr and "stage communicator up (nccl_ranks pending)" in r.stderr
    assert r.stdout.strip() == ""


def test_gpus_2_without_two_gpus_exits_nonzero_with_a_message():
    """This container has no GPU (and a 1-GPU box has one): the bare command must refuse, not assert or hang."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    import torch
    if torch.cuda.device_count() >= 2:
        return                                              # a multi-GPU node really runs it; covered by the driver's SCALE run
    assert r.returncode == 2 and "--gpus 2 but this node shows" in r.stderr and r.stdout.strip() == ""


def test_launcher_parent_never_imports_torch():
    """The parent must stay clear of torch / HIP (a process that initialised the GPU must not spawn-and-wait as a launcher
    on this pool, and must never exec).  Checked on the import graph of the launcher path."""
    code = ("import sys, bench\n"
            "bench.parse_args(['--gpus', '2'])\n"
            "bench.free_port()\n"
            "assert 'torch' not in sys.modules and 'cuda_sfm_amd' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)\n")
    r = run_launcher(code)
    assert r.returncode == 0, r.stderr
