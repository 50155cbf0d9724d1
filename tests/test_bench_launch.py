"""CPU: `bench.py --gpus N` starts its N ranks itself (reference launcher:
train.py:1178-1185, README.md:130-133 -- world = visible GPUs, one process each).

The mock workload is a toy CPU step over gloo: only the launcher, the process group and the
`comm` report are under test here, never a number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}:\n{out}"
    return json.loads(lines[0])


REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")


def test_the_line_is_small_flat_and_last_on_stdout():
    """the driver keeps a bounded tail of stdout; a 20.8 KB line was not parsed.  The line is one
    JSON object under 4 KB, the LAST line on stdout, with the contract's keys; `config` names the workload in at most
    200 characters; nothing in it is nested deeper than two levels."""
    for gpus in (1, 8):
        r = _run(["--gpus", str(gpus), "--workload", "mock", "--steps", "3", "--warmup", "1", "--batch", "2"],
                 {"MVF_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        last = r.stdout.rstrip("\n").splitlines()[-1]
        assert last.startswith("{") and len(last.encode()) < 4096, len(last)
        d = json.loads(last)
        assert json.loads(json.dumps(d)) == d
        for k in REQUIRED_KEYS:
            assert k in d, k
        assert d["n_gpus"] == gpus and len(d["config"]["workload"]) <= 200
        assert set(d["config"]) == {"workload", "global_batch", "parallelism"}

        def depth(o):
            return 1 + max([depth(v) for v in o.values()] + [0]) if isinstance(o, dict) else 0
        assert depth(d) <= 3, depth(d)


def test_an_oversized_line_is_refused():
    """bench.emit() is the only place the line is printed, and it asserts the size bound."""
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mvf_bench_under_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.LINE_LIMIT == 4096
    with pytest.raises(AssertionError):
        b.emit(json.dumps({"x": "y" * 5000}))


def test_the_default_run_has_no_detail_legs():
    """The driver's default command runs the headline, the hot-path leg, the live counters and the CPU baseline only;
    the child-process legs live in tools/measure_detail.py behind --detail."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert len(src.splitlines()) < 700
    for name in ("def mfma_leg", "def host_leg", "def graph_step_leg", "def other_config_leg", "cpu_baseline_unfused"):
        assert name not in src
    a = __import__("subprocess").run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert "--detail" in a.stdout


def test_gpus_2_spawns_two_ranks_and_reports_the_group():
    r = _run(["--gpus", "2", "--workload", "mock", "--steps", "4", "--warmup", "1", "--batch", "4"],
             {"MVF_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 8
    c = d["comm"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["distinct_processes"] == 2
    assert c["grad_buckets"] >= 2 and c["grad_bucket_bytes"] > 0
    # one exchange per bucket per step
    assert c["collectives_per_step"] == {"grad_all_reduce": float(c["grad_buckets"])}
    assert c["no_overlap"]["ms_per_step"] > 0 and c["overlap_with_backward"] is True
    assert "launching" in r.stderr          # it re-executed itself under torch.distributed.run


def test_gpus_8_spawns_eight_ranks():
    """The driver's largest form: `--gpus 8` starts eight processes, the process group reports eight ranks and
    eight distinct PIDs (gloo, mock step: the launcher and the report are under test, never a number)."""
    r = _run(["--gpus", "8", "--workload", "mock", "--steps", "3", "--warmup", "1", "--batch", "2"],
             {"MVF_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 16
    c = d["comm"]
    assert c["world_size"] == 8 and c["distinct_processes"] == 8 and len(set(c["pids"])) == 8
    assert c["collectives_per_step"] == {"grad_all_reduce": float(c["grad_buckets"])}


def test_reduce_scatter_exchange_is_reported():
    r = _run(["--gpus", "2", "--workload", "mock", "--steps", "3", "--warmup", "1", "--grad-exchange",
              "reduce_scatter", "--no-overlap"], {"MVF_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    c = _line(r.stdout)["comm"]
    nb = float(c["grad_buckets"])
    assert c["grad_exchange"] == "reduce_scatter" and c["overlap_with_backward"] is False
    assert c["collectives_per_step"] == {"grad_all_gather": nb, "grad_reduce_scatter": nb}
    assert "overlapped" in c


def test_more_ranks_than_gpus_fails_loudly():
    """On a box with fewer GPUs than --gpus the benchmark refuses (it used to run ONE rank and
    print n_gpus: 1)."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs" in (r.stderr + r.stdout) and "GPUs" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_must_match_gpus():
    """A torchrun environment whose WORLD_SIZE differs from --gpus is refused, not relabelled."""
    r = _run(["--gpus", "1", "--workload", "mock", "--steps", "1", "--warmup", "0"],
             {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "refusing" in r.stderr


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_train_workload_with_two_ranks_on_the_test_gpu(launcher):
    """The REAL benchmark step (trainer, grouped SyncBatchNorm, bucketed reducer, timing with the maximum
    over ranks, `comm` report with its second issue order) with two ranks -- on the one GPU of the test box
    over gloo (MVF_BENCH_SHARE_GPU=1; RCCL needs a GPU per rank), small shapes.  Both launch forms: the
    script starting its own ranks, and the driver's `python -m torch.distributed.run ... bench.py --gpus 2`."""
    args = ["--gpus", "2", "--workload", "train", "--batch", "2", "--height", "64", "--width", "96", "--steps", "2",
            "--warmup", "1", "--no-cpu-baseline", "--comm-leg-steps", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MVF_DIST_BACKEND="gloo", MVF_BENCH_SHARE_GPU="1")
    if launcher == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    # two ranks SHARE the one GPU of the test box: the line says so (n_gpus = devices that worked, world = group size)
    assert d["n_gpus"] == 1 and d["world"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["roofline"]["launches"] == 2 * 2      # two unit launches per step (6 + 3 units)
    c = d["comm"]
    assert c["world_size"] == 2 and c["distinct_processes"] == 2 and c["backend"] == "gloo" and c["replicas_identical"] is True
    per = c["collectives_per_step"]
    # per step: one exchange per gradient bucket, one statistics collective per BatchNorm layer per grouped
    # call and direction (the CPU/gloo branch of the synchronised batch norm all-reduces in both directions
    # on a CPU tensor; on the GPU it all-gathers forward) -- 40 per direction for ResNet18
    assert per["grad_all_reduce"] == float(c["grad_buckets"])
    assert per.get("bn_all_gather", per.get("bn_all_reduce_fwd")) == 40.0 and per["bn_all_reduce"] == 40.0
    assert c["no_overlap"]["ms_per_step"] > 0
    assert "other_configs" not in d and "cpu_baseline" not in d


@pytest.mark.gpu
def test_eight_rank_dress_rehearsal_on_the_test_gpu():
    """no 8-GPU node exists for this project, so the eight-rank form of the job is rehearsed on the
    one GPU there is -- eight processes on cuda:0 over gloo (MVF_BENCH_SHARE_GPU=1), the REAL training step at a tiny
    shape, `--grad-exchange reduce_scatter`, grouped SyncBatchNorm, one MIOpen find-db copy per process (reference:
    train.py:205-208 DDP + SyncBatchNorm, :692-693).  Held: eight distinct processes in the group, one collective per
    BatchNorm layer per grouped call and direction (40 + 40), one reduce-scatter + one all-gather per gradient bucket,
    identical parameters on all eight ranks after the steps, and a line that reports ONE GPU with a world of eight."""
    args = ["--gpus", "8", "--workload", "train", "--batch", "1", "--height", "64", "--width", "96", "--steps", "2",
            "--warmup", "1", "--no-cpu-baseline", "--comm-leg-steps", "0", "--grad-exchange", "reduce_scatter"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MVF_DIST_BACKEND="gloo", MVF_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and len(last.encode()) < 4096
    d = json.loads(last)
    assert d["n_gpus"] == 1 and d["world"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 8
    c = d["comm"]
    assert c["world_size"] == 8 and c["distinct_processes"] == 8 and len(set(c["pids"])) == 8 and set(c["devices"]) == {0}
    assert c["grad_exchange"] == "reduce_scatter" and c["grad_buckets"] >= 2 and c["replicas_identical"] is True
    per = c["collectives_per_step"]
    nb = float(c["grad_buckets"])
    assert per["grad_reduce_scatter"] == nb and per["grad_all_gather"] == nb and "grad_all_reduce" not in per
    assert per.get("bn_all_gather", per.get("bn_all_reduce_fwd")) == 40.0 and per["bn_all_reduce"] == 40.0
    assert c["exchanges_issued_during_backward"] + c["exchanges_issued_after_backward"] == c["grad_buckets"]
    assert d["roofline"]["launches"] == 2 * 2 and d["value"] > 0


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """On a box with at least two GPUs: `bench.py --gpus 2` starts two ranks on two devices over RCCL (the
    driver's multi-GPU form of the benchmark at a small shape).  Skipped on the 1-GPU test boxes -- there the
    same code runs with two gloo ranks on one GPU (above) and with RCCL in a group of one
    (tests/test_trainer_gpu.py)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r = _run(["--gpus", "2", "--workload", "train", "--batch", "2", "--height", "64", "--width", "96", "--steps", "3",
              "--warmup", "2", "--no-cpu-baseline", "--comm-leg-steps", "2"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    c = d["comm"]
    assert d["n_gpus"] == 2 and c["world_size"] == 2 and c["backend"].startswith("rccl")
    assert sorted(c["devices"]) == [0, 1] and c["distinct_processes"] == 2
    per = c["collectives_per_step"]
    assert per["grad_all_reduce"] == float(c["grad_buckets"]) and per["bn_all_gather"] == 40.0 and per["bn_all_reduce"] == 40.0
    assert c["exchanges_issued_during_backward"] >= c["grad_buckets"] - 1
    assert r.stdout.strip().splitlines()[-1].startswith("{")        # the JSON line is the last line on stdout
