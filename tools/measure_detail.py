"""Detail legs of the benchmark (`python bench.py --detail`): everything that is NOT the driver's line.

`bench.py` prints one flat line (value, roofline of the unit kernel, cpu_baseline, the two hot-path figures).  The
measurements below used to ride in that line and grew it to 20 KB; they now go to
`gpurun_out/bench_detail.json` and are summarised under `profiles/`:

* ``own_kernels``    -- roofline table of this build's own kernels either side of the unit kernel (profile level 2)
* ``unit_launches``  -- the unit kernel per launch kind (single-frame + affine / multi-frame): median, min, max, frac
* ``hotpath_graph``  -- the stand-alone hot-path loop replayed as one HIP graph
* ``other_configs``  -- BASELINE.json configs 3-5 for a few steps each (+ the host cost pinned to a rank's CPUs)
* ``hip_graph_step`` -- the optimisation step replayed as one HIP graph (child process)
* ``host``           -- the step pinned to the CPUs one of eight ranks would have (child processes, eager and graph)
* ``conv_mfma``      -- MFMA / VALU utilisation of the step's kernels by family (child under rocprofv3 --pmc)
* ``cpu_unfused``    -- second CPU baseline: the unit as ~130 ATen operators under autograd (oracle/torch_unfused.py)
* ``fast_mode``      -- ``frac_fast`` / ``us_per_unit_fast``: the hot-path loop through the opt-in fast library (child process
                        with MVF_HOTPATH_LIB), beside the same loop through the shipped one

`run()` gets bench.py's module object (its step classes, timing and roofline helpers) instead of importing it a second
time.  Every leg is optional and must never take the line down: errors are recorded in the leg's entry."""
import collections
import copy
import csv
import gc
import glob
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
LEGS = ("own_kernels", "unit_launches", "hotpath_graph", "other_configs", "hip_graph_step", "host", "conv_mfma", "cpu_unfused",
        "fast_mode")

OTHER_CONFIGS = {
    # BASELINE.json configs[2..4] at their per-GPU shapes (reference: configs/dhrnet/DHRNet_KITTI_MR.txt,
    # configs/litemono/LiteMono_KITTI_HR.txt, configs/dhrnet/DHRNet_CS.txt)
    "C3": dict(backbone="DHRNet", batch=12, height=192, width=640),
    "C4": dict(backbone="LiteMono", batch=8, height=320, width=1024),
    "C5": dict(backbone="DHRNet", batch=12, height=192, width=512),
}

CONV_FAMILIES = (
    ("winograd", ("miopenSp3AsmConv", "Winograd", "winograd")),
    ("igemm_fwd", ("igemm_fwd",)),
    ("igemm_bwd", ("igemm_bwd",)),
    ("igemm_wrw", ("igemm_wrw",)),
    ("ck_conv", ("kernel_grouped_conv", "ck::")),
    ("gemm", ("Cijk_", "rocblas_", "gemv")),
    ("conv_transposes", ("batched_transpose", "transpose_NCHW", "transpose_CNHW", "SubTensorOp")),
    ("batch_norm", ("MIOpenBatchNorm", "batch_norm")),
    ("own_kernels", ("(anonymous namespace)::k_", "k_unit_fb", "k_bias_act", "k_up2cat", "k_reflect", "k_maxpool")),
)


def _child_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    return env


def _shape_flags(a):
    return ["--batch", str(a.batch), "--height", str(a.height), "--width", str(a.width), "--backbone", a.backbone]


def _child_line(cmd, timeout_s, env_extra=None, **kw):
    r = subprocess.run(cmd, env=dict(_child_env(), **(env_extra or {})), capture_output=True, text=True, timeout=timeout_s, **kw)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"child rc {r.returncode}: " + (r.stderr.strip().splitlines() or [""])[-1][:200])
    return json.loads(lines[-1])


def own_kernels(B, step, nat, steps=3):
    """HIP events around every launch of the kernels listed in include/mvf_hotpath.h (MVF_PROF_*) over a few extra
    steps; bytes = algorithmic (every input element read once + every output element written once, stated per
    launcher in csrc/)."""
    nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
    nat.check(nat.lib().mvf_profile_enable(2), "profile_enable")
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        nat.lib().mvf_profile_enable(0)
    out = {}
    for kid in range(nat.PROF_FIRST_GLUE, nat.PROF_COUNT):
        ms, n = nat.profile_read(kid)
        if n == 0:
            continue
        nbytes = nat.profile_read_work(kid)
        ach = nbytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0
        out[nat.profile_name(kid)] = {
            "launches_per_step": round(n / steps, 1), "ms_per_step": round(ms / steps, 4), "avg_us": round(ms / n * 1e3, 2),
            "bytes_per_launch": int(nbytes // n), "achieved_GBs": round(ach, 1), "frac": round(ach / B.HBM_PEAK_GBS, 4)}
    nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
    return {"kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"])),
            "own_glue_ms_per_step": round(sum(v["ms_per_step"] for v in out.values()), 3), "steps": steps}


def unit_launches(B, step, nat, args, steps=10):
    """The launch kinds of a step are different work (single-frame + affine: identity SSIM + hand-over write + mask
    plane; multi-frame: hand-over read): per kind the median / min / max of the recorded launches and its fraction."""
    nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
    nat.check(nat.lib().mvf_profile_enable(1), "profile_enable")
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        nat.lib().mvf_profile_enable(0)
    recs = nat.profile_read_launches(nat.PROF_UNIT_FWDBWD)
    noise_tensor = args.noise != "kernel"
    out = {}
    for tag, name in nat.TAG_NAMES.items():
        sel = [(ms, px) for ms, px, t in recs if t == tag]
        if not sel:
            continue
        med_ms = statistics.median(ms for ms, _ in sel)
        px = statistics.median(p for _, p in sel)
        bpp = B.FB_BYTES_PER_PX + (B.NOISE_BYTES_PER_PX if noise_tensor else 0) + \
            (B.MASK_BYTES_PER_PX if tag == 2 else B.MASK_BYTES_PER_PX / 2 if tag == 3 else 0)
        ach = bpp * px / (med_ms / 1e3) / 1e9
        out[name] = {"launches": len(sel), "median_us": round(med_ms * 1e3, 2), "min_us": round(min(ms for ms, _ in sel) * 1e3, 2),
                     "max_us": round(max(ms for ms, _ in sel) * 1e3, 2), "bytes": int(round(bpp * px)), "bytes_per_px": bpp,
                     "achieved_GBs": round(ach, 1), "frac": round(ach / B.HBM_PEAK_GBS, 4)}
    return out


def hotpath_graph(step, steps=50):
    """The stand-alone hot-path step captured ONCE into a HIP graph and replayed: what the launch-bound loop costs
    without the Python / autograd enqueue time of every step.  (The tie-break noise key is baked into the captured
    launch: a benchmark device, the trainer's graph step draws its noise graph-safely.)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    for u in step.units:
        u["disp"].grad = None
        u["T"].grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"images_per_sec": round(step.images_per_step * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps}


def other_config(B, args, name, rank, world, dev, nat, steps=10, warmup=5):
    from mono_vifi_amd.bench_train import TrainStep
    t_leg = time.perf_counter()
    a = copy.copy(args)
    for k, v in OTHER_CONFIGS[name].items():
        setattr(a, k, v)
    step = TrainStep(a, rank, world, dev)
    for _ in range(warmup):
        step()
    nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
    nat.check(nat.lib().mvf_profile_enable(1), "profile_enable")
    elapsed = B.timed_steps(step, steps, world)
    nat.lib().mvf_profile_enable(0)
    rf = B.unit_roofline(nat, step, a) or {}
    out = {"workload": step.describe()[:160], "images_per_sec": round(a.batch * world * steps / elapsed, 2),
           "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps,
           "host_process_cpu_ms_per_step": B.HOST_CPU.get("process_cpu_ms_per_step"),
           "unit_launch_avg_us": rf.get("avg_us"), "us_per_unit": rf.get("us_per_unit"), "frac": rf.get("frac")}
    del step
    gc.collect()
    torch.cuda.empty_cache()
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    return out


def hip_graph_step(args, steps=20, timeout_s=150):
    """The optimisation step with its device work captured into ONE HIP graph (trainer.py:_StepGraph) and replayed, in
    a CHILD process: a GPU fault or a hang during a replay cannot be caught and must not take the parent down."""
    d = _child_line([sys.executable, BENCH, "--gpus", "1", "--workload", "train", "--hip-graph", "--hip-graph-scope",
                     args.hip_graph_scope, "--steps", str(steps), "--warmup", "6", "--no-cpu-baseline", "--no-hotpath-leg",
                     "--no-pmc-leg"] + _shape_flags(args), timeout_s)
    return {"images_per_sec": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "scope": args.hip_graph_scope}


def host(B, args, steps=10, timeout_s=150):
    """What the host costs when eight ranks share this box: the SAME training step in child processes pinned
    (sched_setaffinity) to the CPUs ONE rank would have with eight ranks on the CPUs this container is granted -- eager,
    and replayed as a HIP graph.  A step that slows down under the pin is host-bound on the 8-GPU node."""
    quota, _ = B.cpu_quota()
    ncpu = max(1, quota // 8)
    cpus = sorted(os.sched_getaffinity(0))[:ncpu]
    out = {"cpus_per_rank": ncpu, "pinned_to": cpus, "steps": steps}
    base = [sys.executable, BENCH, "--gpus", "1", "--workload", "train", "--steps", str(steps), "--warmup", "6",
            "--no-cpu-baseline", "--no-hotpath-leg", "--no-pmc-leg"] + _shape_flags(args)
    for name, extra in (("eager", []), ("hip_graph", ["--hip-graph", "--hip-graph-scope", args.hip_graph_scope])):
        try:
            d = _child_line(base + extra, timeout_s, env_extra={"OMP_NUM_THREADS": str(ncpu)},
                            preexec_fn=lambda: os.sched_setaffinity(0, cpus))
            out[name] = {"images_per_sec": d["value"], "ms_per_step": d["ms_per_step"],
                         "process_cpu_ms_per_step": d.get("host_process_cpu_ms_per_step")}
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def conv_mfma(args, timeout_s=240):
    """north_star's "MFMA utilisation against the chip's peak" for the conv GEMMs: one counter-only `rocprofv3 --pmc
    SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace` pass around a child run of four
    training steps; per kernel family the share of the step's GPU cycles, ms per step, MFMA-pipe and VALU busy.
    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1,024 SIMDs); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs;
    VALU busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / the same.  Measurement only: the kernels are MIOpen's."""
    if not shutil.which("rocprofv3"):
        return {"error": "rocprofv3 not found"}
    warm, timed = 2, 2
    d = tempfile.mkdtemp(prefix="mvf_mfma_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
           "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, BENCH, "--workload", "train",
           "--steps", str(timed), "--warmup", str(warm), "--no-cpu-baseline", "--no-hotpath-leg", "--no-pmc-leg"] + _shape_flags(args)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=_child_env(), capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return {"error": "rocprofv3 child rc %d: %s" % (r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:200])}
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
        dur = collections.defaultdict(float)
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    dur[row["Kernel_Name"]] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6
    finally:
        shutil.rmtree(d, ignore_errors=True)
    steps_all = float(warm + timed)     # the counters cover every step the child ran (warm-up included)

    def family(name):
        for fam, pats in CONV_FAMILIES:
            if any(p_ in name for p_ in pats):
                return fam
        return "other"
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])      # cycles, mfma, valu, ms
    for k, c in acc.items():
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0:
            continue
        a = fam[family(k)]
        a[0] += cyc
        a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0)
        a[3] += dur.get(k, 0.0)
    tot = [sum(v[i] for v in fam.values()) for i in range(4)]
    if tot[0] <= 0:
        return {"error": "no counters collected"}
    per = {k: {"share_of_gpu_cycles": round(v[0] / tot[0], 4), "ms_per_step": round(v[3] / steps_all, 3),
               "mfma_busy": round(v[1] / (v[0] * 1024.0), 4), "valu_busy": round(v[2] / (v[0] * 1024.0), 4)}
           for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
    conv = [v for k, v in fam.items() if k in ("winograd", "igemm_fwd", "igemm_bwd", "igemm_wrw", "ck_conv", "gemm")]
    cc = [sum(v[i] for v in conv) for i in range(4)]
    return {"cycle_weighted_mfma_busy": round(tot[1] / (tot[0] * 1024.0), 4),
            "cycle_weighted_valu_busy": round(tot[2] / (tot[0] * 1024.0), 4),
            "conv_and_gemm_kernels": {"share_of_gpu_cycles": round(cc[0] / tot[0], 4), "ms_per_step": round(cc[3] / steps_all, 3),
                                      "mfma_busy": round(cc[1] / (cc[0] * 1024.0), 4) if cc[0] else None,
                                      "valu_busy": round(cc[2] / (cc[0] * 1024.0), 4) if cc[0] else None},
            "families": per, "kernel_ms_per_step_under_pmc": round(tot[3] / steps_all, 2),
            "peak_note": "MFMA busy 1.0 = the fp32-input MFMA peak of 157.3 TFLOP/s; the Winograd kernels are VALU code"}


def cpu_unfused(B, args):
    """Second CPU baseline: the SAME unit as ~130 separate ATen operators under autograd
    (oracle/torch_unfused.py), i.e. the operator stream the reference's own code issues on its CPU path; torch intra-op
    threads = the CPUs the container is granted; median of three samples."""
    from mono_vifi_amd import synthetic
    from oracle import oracle as O
    from oracle import torch_unfused as U
    Bs = args.batch
    inp = synthetic.unit_inputs(4321, Bs, args.height, args.width, with_mask=True)
    T = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1)) for k in range(2)], 0)
    cores, hw = B.cpu_quota()
    tens = [torch.from_numpy(np.ascontiguousarray(a)) for a in
            (inp["disp"], inp["tgt"], T, inp["K"], inp["inv_K"], inp["noise"], inp["mask_rec"])]
    srcs = [torch.from_numpy(np.ascontiguousarray(a)) for a in inp["src"]]

    def one():
        U.unit(tens[0], tens[1], srcs, tens[2], tens[3], tens[4], tens[5], tens[6], 0)
    old = torch.get_num_threads()
    samples = []
    libc = __import__("ctypes").CDLL(None)          # buffers reused, as in bench.cpu_baseline
    libc.mallopt(-3, 1 << 30), libc.mallopt(-1, (1 << 31) - 1)
    try:
        torch.set_num_threads(cores)
        one()
        for _ in range(3):
            n, t0 = 0, time.perf_counter()
            while True:
                one()
                n += 1
                dt = time.perf_counter() - t0
                if dt >= args.cpu_seconds / 3.0 or n >= 2000:
                    break
            samples.append(Bs / (B.UNITS_PER_STEP * dt / n))
    finally:
        torch.set_num_threads(old)
    return {"value": round(statistics.median(samples), 3), "unit": "images/sec", "cores": cores, "threads": cores, "kind": "port",
            "runs": 3, "min": round(min(samples), 3), "max": round(max(samples), 3), "hardware_threads": hw,
            "sample": "1 unit fwd+bwd per call as ~130 ATen ops under autograd; value = hot-path part of a step (9 units)"}


def fast_mode(args, timeout_s=240):
    """What the reference's evaluation order costs: the stand-alone hot-path loop through the exact
    library and through `make fast`'s (separable shared window sums, contracted formula), each in a child process.  The
    deviation report at the four shapes is tools/fast_mode_report.py -> profiles/r06_fast_mode_report.json."""
    lib = os.path.join(ROOT, "mono-vifi_amd", "lib")
    fast = os.path.join(lib, "libmvf_hotpath_fast.so")
    if not os.path.exists(fast):
        return {"error": "libmvf_hotpath_fast.so not built (make -C mono-vifi_amd/csrc fast)"}
    out = {}
    for mode, path in (("exact", os.path.join(lib, "libmvf_hotpath.so")), ("fast", fast)):
        d = _child_line([sys.executable, BENCH, "--workload", "hotpath", "--steps", "50", "--warmup", "5", "--no-cpu-baseline",
                         "--no-pmc-leg", "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width)],
                        timeout_s, env_extra={"MVF_HOTPATH_LIB": path})
        out[f"frac_{mode}"] = d["roofline"]["frac"]
        out[f"us_per_unit_{mode}"] = d["roofline"]["us_per_unit"]
    out["fast_over_exact_time"] = round(out["us_per_unit_fast"] / out["us_per_unit_exact"], 4)
    return out


def run(B, args, step, nat, rank, world, dev, hp_step):
    want = LEGS if args.detail_legs == "all" else tuple(s.strip() for s in args.detail_legs.split(",") if s.strip())
    headline = (args.workload == "train" and args.backbone == "ResNet18" and not args.hip_graph)
    out = {"legs": list(want)}

    def leg(name, fn, *a, **kw):
        if name not in want:
            return
        t0 = time.perf_counter()
        try:
            out[name] = fn(*a, **kw)
        except Exception as e:      # noqa: BLE001 -- an optional leg never takes the line down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if isinstance(out[name], dict):
            out[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)

    if not args.hip_graph:
        leg("own_kernels", own_kernels, B, step, nat)
        leg("unit_launches", unit_launches, B, step, nat, args)
    if hp_step is not None:
        leg("hotpath_graph", hotpath_graph, hp_step)
    if headline and "other_configs" in want:
        torch.cuda.empty_cache()
        oc = {}
        for n in OTHER_CONFIGS:
            try:
                oc[n] = other_config(B, args, n, rank, world, dev, nat)
            except Exception as e:      # noqa: BLE001
                oc[n] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if "host" in want and "ms_per_step" in oc[n]:
                a = copy.copy(args)
                for k, v in OTHER_CONFIGS[n].items():
                    setattr(a, k, v)
                try:
                    hp = host(B, a)
                    oc[n]["host_pinned"] = hp
                    if hp.get("eager", {}).get("ms_per_step"):
                        oc[n]["pinned_eager_over_unpinned"] = round(hp["eager"]["ms_per_step"] / oc[n]["ms_per_step"], 3)
                except Exception as e:      # noqa: BLE001
                    oc[n]["host_pinned"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        out["other_configs"] = oc
    if headline:
        torch.cuda.empty_cache()
        leg("hip_graph_step", hip_graph_step, args)
        leg("host", host, B, args)
        leg("conv_mfma", conv_mfma, args)
    leg("cpu_unfused", cpu_unfused, B, args)
    leg("fast_mode", fast_mode, args)
    return out
