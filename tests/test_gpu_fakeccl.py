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


@pytest.mark.parametrize("world", [2, 3, 8])
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
        for name in ("communicator_up", "shard_scored", "sharded_step_done", "pipelined_steps_done", "other_scene", "views_sharded_done",
                     "injected_failure", "views_sharded_after_failures"):
            assert name in stages, name


def _run_bench_ranks(world, extra_args, timeout=900):
    assert os.path.exists(FAKE)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SFM_AMD_COMM_LIB=FAKE, SFM_BENCH_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra_args, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{err[-3000:]}"
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    assert not any(ln.startswith("{") for o in outs[1:] for ln in o[0].splitlines()), "only rank 0 prints the line"
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", [[], ["--serial"], ["--comm", "torch"]])
def test_bench_multi_rank_flow_on_one_gpu(mode):
    """bench.py exactly as the driver's SCALE run starts it (one process per rank, RANK / WORLD_SIZE in the environment), two
    ranks sharing the one GPU: the pipelined and the serial C exchange and the torch.distributed exchange; the line must say
    two ranks, the ranks must agree on winner, E and mask, and the winner must be the one-GPU winner of the same workload.
    (The numbers mean nothing here: two ranks time-share one GPU.)"""
    rec = _run_bench_ranks(2, ["--steps", "3", "--warmup", "1", "--regions", "2", "--hyps", "65536", "--no-cpu", "--no-variants", "--no-extra"] + mode)
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0 and rec["unit"] == "hypotheses/s"
    mg = rec["result"]["multi_gpu"]
    assert mg["ranks_agree_on_winner_E_mask"] and mg["communicator_spans_all_ranks"] and len(mg["per_rank"]) == 2
    assert rec["config"]["nccl_ranks"] == 2 and len(rec["config"]["per_rank_kernel_ms"]) == 2
    import cuda_sfm_amd as S
    from cuda_sfm_amd import synth
    import numpy as np
    import torch
    dev = torch.device("cuda", 0)
    ctx = S.Context(0, torch.cuda.current_stream().cuda_stream)
    scene = synth.two_view_scene(4096)
    pair = S.ImagePair(ctx, scene["K"], scene["Kinv"], 2, 4096)
    pair.fillXU(torch.from_numpy(scene["sift"].view(np.uint8).reshape(4096, 576)).to(dev))
    # (every timed step samples with its own seed: the line names the seed of the step whose result it reports)
    pair.estimateE(S.default_params(4096, num_hypotheses=65536, seed=rec["result"]["sampler_seed_of_this_result"]))
    assert list(pair.get_best()) == [rec["result"]["best_hypothesis"], rec["result"]["inliers"]]
    assert rec["serial_ms_per_step"] > 0 and (mode != [] or rec["exchange_us"] > 0)


def test_bench_c5_two_ranks_on_one_gpu():
    """configs[4] as the bench job, two ranks on one GPU through sfm_process_views_sharded: 36 dino views dealt over the ranks,
    the count-sized exchange, all 630 pairs; the records must be the one-rank records (the job checks them against the oracle
    chain on rank 0 when asked; here: every pair done, the exchange moved what exists)."""
    rec = _run_bench_ranks(2, ["--config", "c5", "--steps", "1", "--warmup", "1", "--regions", "1", "--no-cpu", "--comm", "rccl"], timeout=1200)
    assert rec["n_gpus"] == 2 and rec["result"]["pairs_done"] == 630
    assert rec["config"]["nccl_ranks"] == 2
    ex = rec["exchange"]
    assert ex["sum_count_x_576"] <= ex["feature_bytes_into_each_rank"] <= 1.1 * ex["sum_count_x_576"]


@pytest.mark.parametrize("nproc", [2, 8])
def test_bench_under_torch_distributed_run_on_one_gpu(nproc):
    """The driver's own command line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` (torchrun sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), with the
    N ranks folded onto the one GPU of this box (N = 2 and N = 8, the driver's largest)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SFM_AMD_COMM_LIB=FAKE, SFM_BENCH_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1", "--regions", "1", "--hyps", "65536", "--no-variants"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == nproc and rec["steps"] == 3 and rec["warmup"] == 1 and rec["higher_is_better"] is True
    assert rec["result"]["multi_gpu"]["ranks_agree_on_winner_E_mask"] and rec["config"]["nccl_ranks"] == nproc
    assert len(rec["result"]["multi_gpu"]["per_rank"]) == nproc
    for key in ("metric", "value", "unit", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key


def test_bench_own_launcher_on_one_gpu():
    """`python bench.py --gpus 2 ...` without a launcher around it: bench.py starts its own ranks (launcher_main)."""
    env = dict(os.environ, SFM_AMD_COMM_LIB=FAKE, SFM_BENCH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--regions", "1", "--hyps", "65536",
                        "--no-variants"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["nccl_ranks"] == 2 and rec["result"]["multi_gpu"]["ranks_agree_on_winner_E_mask"]
