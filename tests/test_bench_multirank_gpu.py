"""bench.py with N > 1 ranks: the control flow the driver runs on the 8-GPU node (barriers, gradient all-reduce inside the
timed steps AND inside the roofline leg, max-over-ranks timing, one JSON line from rank 0), rehearsed with two ranks on the
one GPU of this box over gloo (test hooks JCK_BENCH_ONE_GPU / JCK_BENCH_BACKEND; RCCL needs one device per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_terminates_and_reports_whole_job_rate():
    env = dict(os.environ, JCK_BENCH_ONE_GPU="1", JCK_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--batch", "16"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    assert abs(out["value"] - 2 * 16 / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]      # whole-job images/sec
    # at batch 16 the dominant kernel runs at a fraction of a TFLOP/s: frac (4 decimals) may round to 0
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["achieved"] > 0 and 0 <= out["roofline"]["frac"] < 1
    assert "cpu_baseline" not in out                              # N = 1 only
    d = out["ddp"]                                                # what the first real multi-GPU run is read by
    assert d["rccl_world"] == 2 and d["replicas_in_sync_after_warmup"] and d["replicas_in_sync_at_end"], d
    assert "comm_exposed_ms" in d and d["n1_equivalent_ms"] > 0
    for k in ("loss_d", "loss_g", "gp"):
        v = out["losses_last_step"][k]
        assert v == v and abs(v) < 1e3


def test_one_rank_rccl_bench_reports_the_data_parallel_keys():
    """The nccl branch of bench.py with ONE rank (JCK_BENCH_FORCE_DDP=1): a real RCCL communicator, the overlapped schedule
    (D's all-reduce in two pieces, G's under the next batch's D(real) forward), and the keys the first multi-GPU run will be
    read by - rccl_world, comm_exposed_ms (HIP events around the waits in front of Adam(D) / Adam(G)), n1_equivalent_ms (the same
    step with the reducers off; DESIGN.md section 6's efficiency estimate is n1_equivalent_ms / ms_per_step)."""
    env = dict(os.environ, JCK_BENCH_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "16",
           "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = out["ddp"]
    assert d["rccl_world"] == 1 and d["backend"] == "nccl"
    assert d["replicas_in_sync_after_warmup"] and d["replicas_in_sync_at_end"]
    assert d["comm_exposed_ms"]["d"] >= 0 and d["comm_exposed_ms"]["g"] >= 0
    assert 0 < d["n1_equivalent_ms"] < 50 and d["message_bytes"]["d"] > 1e7 and d["message_bytes"]["g"] > 1e7
    assert "two pieces" in d["mode"]


def test_bare_bench_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE in the environment (the shape of the driver's N = 1 command
    with another N): the parent spawns the two ranks itself, passes rank 0's single JSON line through and exits 0."""
    env = dict(os.environ, JCK_BENCH_ONE_GPU="1", JCK_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "16"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["ddp"]["rccl_world"] == 2
    assert out["ddp"]["replicas_in_sync_at_end"]


def test_a_failing_rank_fails_the_bare_launch():
    """The self-launching parent returns the children's status: a world size the ranks refuse (--gpus 2 with WORLD_SIZE=3 set by
    hand is a usage error in every rank) must not read as success."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
