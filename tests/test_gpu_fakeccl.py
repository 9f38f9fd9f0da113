"""Two ranks on ONE GPU through the C exchange code (comm.cpp) with a shared-memory stand-in for RCCL (tests/fake_ccl): what
the multi-rank logic of libsfm_amd_rccl.so does with more than one rank, on a box where two RCCL ranks cannot exist."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_ccl", "libsfm_amd_fakeccl.so")


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_one_gpu_through_comm_cpp(world):
    assert os.path.exists(FAKE), "run `make` (the fakeccl target builds tests/fake_ccl/libsfm_amd_fakeccl.so)"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SFM_AMD_COMM_LIB=FAKE)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fakeccl_child.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{err[-3000:]}"
    rec = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert rec["ok"] and rec["world"] == world and rec["ranks"] == world
    for stages in rec["per_rank"]:
        for name in ("communicator_up", "shard_scored", "sharded_step_done", "pipelined_steps_done", "other_scene", "views_sharded_done"):
            assert name in stages, name
