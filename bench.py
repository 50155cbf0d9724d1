#!/usr/bin/env python3
"""Throughput benchmark of the Mono-ViFI hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload train|hotpath] [--detail]

One process per GPU.  Under ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) each process is one rank; started plainly with ``--gpus N > 1`` the script launches the N ranks ITSELF
(re-executes under ``torch.distributed.run`` on 127.0.0.1) and fails loudly when the box has fewer than N GPUs -- it
never prints a line whose ``n_gpus`` differs from the world size the process group reports (reference:
train.py:1178-1185, README.md:130-133).  W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
synchronize on both sides; the maximum over ranks is used and rank 0 prints ONE flat JSON line, at most 4 KB, as the
LAST line on stdout.

Workloads (``config.workload`` names the one that ran):
* ``train`` (default) -- the whole optimisation step of the drop-in trainer (reference train.py:640-696): networks +
  the 9 hot-path units + backward + gradient exchange + clip + AdamW at BASELINE.json configs[1] (ResNet18, batch 12,
  640x192, 3-frame, fp32), synthetic batch resident in HBM.
* ``hotpath`` -- only the 9 view-synthesis + photometric-loss units of a step (reference train.py:747-883), forward +
  backward, on distinct buffers (531 MB > 256 MiB Infinity Cache).
* ``mock`` -- a toy CPU step over gloo: plumbing test of the launcher and of the line (tests/test_bench_launch.py).
The line: ``value`` (images/sec), ``roofline`` (the unit kernel ``k_unit_fb``: algorithmic bytes over the launch time
from HIP events the library records on the launch stream inside the timed region, live ``rocprofv3 --pmc`` traffic and
VALU counters from short child runs), ``cpu_baseline`` (the C / OpenMP oracle on the host's granted CPUs, median of
three bounded samples; rank 0, N = 1 only), ``hotpath_ms_per_step`` (the stand-alone hot-path loop) and
``hotpath_in_step_ms`` (the hot path inside the timed training steps).  Everything else -- per-kernel table of the
build's own kernels, MFMA utilisation of the convolutions, the other BASELINE configurations, host cost, HIP-graph
steps, the second CPU baseline -- is measured by ``tools/measure_detail.py`` under ``--detail`` (OFF by default) and
written to ``gpurun_out/bench_detail.json``, never into the line.
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# a box with an empty MIOpen cache would also time MIOpen's naive reference solvers during warm-up (hundreds of ms
# each, never selected): warm-up only, the timed region is unaffected (setdefault: an explicit setting wins)
for _d in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _d, "0")

import mono_vifi_amd  # noqa: E402  (importing sets nothing; main() calls its entry-point helpers before the first HIP call)
import numpy as np  # noqa: E402
import torch  # noqa: E402

T_START = time.perf_counter()
LINE_LIMIT = 4096              # bytes: the driver keeps a bounded tail of stdout (a 20.8 KB line was not parsed)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# algorithmic bytes per pixel and unit: disp 4 + target 12 + 2 sources 24 read once, argmin 1 +
# grad_disp 4 written = 45; + 8 when the tie-break noise is a tensor, + 4 mask_rec on the affine units
FB_BYTES_PER_PX, NOISE_BYTES_PER_PX, MASK_BYTES_PER_PX = 45, 8, 4
UNITS_PER_STEP = 9             # reference train.py:747-883 with use_affine
UNIT_KERNEL = "k_unit_fb<2>"
# the reference's own CPU path (imported unmodified, 8 vCPU / 8 threads) on one unit forward +
# backward: samples/s of the loss path; scaled linearly by cores as `reference_cpu_expected`
REFERENCE_CPU_SAMPLES_PER_S = {(4, 192, 640): 15.9, (12, 192, 640): 10.1, (8, 320, 1024): 3.9, (12, 192, 512): 23.8}
REFERENCE_CPU_CORES = 8
# same-core calibration in the build container (tools/cpu_reference_calibration.py, profiles/r06_cpu_baseline_calibration.json): the
# C / OpenMP port runs a unit fwd + bwd this many times faster than the reference's own code on the same 8 cores (2.8-3.0 over runs)
PORT_OVER_REFERENCE = 2.9
HOST_CPU = {}                  # CPU time of the last timed region (this rank)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("MVF_BENCH_WORKLOAD", "train"), choices=["auto", "hotpath", "train", "mock"])
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--backbone", default="ResNet18")
    ap.add_argument("--disp", default="smooth", choices=["smooth", "noise"], help="hotpath: disparity statistics")
    ap.add_argument("--noise", default="kernel", choices=["kernel", "tensor"], help="tie-break noise: in-kernel, or a tensor (+8 B/px)")
    ap.add_argument("--detail", action="store_true", help="also run tools/measure_detail.py -> gpurun_out/bench_detail.json")
    ap.add_argument("--detail-out", dest="detail_out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    ap.add_argument("--detail-legs", dest="detail_legs", default="all", help="comma list of measure_detail.LEGS")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work of the cpu_baseline leg (3 samples)")
    ap.add_argument("--no-hotpath-leg", dest="hotpath_leg", action="store_false")
    ap.add_argument("--no-pmc-leg", dest="pmc_leg", action="store_false")
    ap.add_argument("--time-budget", dest="time_budget", type=float, default=240.0, help="optional legs are skipped after it")
    ap.add_argument("--hip-graph", dest="hip_graph", action="store_true", help="train: the step replayed as ONE HIP graph")
    ap.add_argument("--hip-graph-scope", dest="hip_graph_scope", default="step", choices=["step", "backward"])
    ap.add_argument("--amp-bf16", dest="amp_bf16", action="store_true", help="reduced precision: never the default")
    for flag in ("--channels-last", "--miopen-find", "--no-share-identity", "--no-regroup"):
        ap.add_argument(flag, action="store_true")
    ap.add_argument("--no-batch-units", dest="no_batch_units", action="store_true", help="one launch per unit")
    ap.add_argument("--no-merge-unit-groups", dest="no_merge_unit_groups", action="store_true", help="3 x 3 units, not 6 + 3")
    ap.add_argument("--grad-exchange", dest="grad_exchange", default="all_reduce", choices=["all_reduce", "reduce_scatter"])
    ap.add_argument("--no-overlap", dest="no_overlap", action="store_true", help="gradient exchange AFTER backward")
    ap.add_argument("--force-collectives", dest="force_collectives", action="store_true", help="N = 1: RCCL group of one")
    ap.add_argument("--comm-leg-steps", dest="comm_leg_steps", type=int, default=10, help="N > 1: steps of the other issue order")
    a = ap.parse_args(argv)
    if a.workload == "auto":
        a.workload = "train"
    return a


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: start the N ranks here, one process per GPU
    like the reference's launcher (README.md:130-133); refuses when the box cannot give every rank its own GPU."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import subprocess
    if args.workload != "mock":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        # MVF_BENCH_SHARE_GPU=1: plumbing tests only (gloo ranks on the one GPU of the test box)
        if have < args.gpus and not (have > 0 and os.environ.get("MVF_BENCH_SHARE_GPU") == "1"):
            sys.exit(f"[bench] --gpus {args.gpus} needs {args.gpus} GPUs (one process per GPU); this box has {have}.  "
                     f"Not running {args.gpus} ranks on fewer devices.")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching " + " ".join(cmd), file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL between processes)
    sys.exit(subprocess.call(cmd, env=env))


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    mock = args.workload == "mock"
    if world != args.gpus:
        sys.exit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: "
                 "refusing to report a line whose n_gpus is not the number of ranks")
    if mock:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
        if world > torch.cuda.device_count() and os.environ.get("MVF_BENCH_SHARE_GPU") != "1":
            sys.exit(f"[bench] {world} ranks but {torch.cuda.device_count()} GPUs: one process per GPU")
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    backend = None
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL ("nccl" on ROCm) over xGMI; MVF_DIST_BACKEND=gloo only for dry runs / the mock test
        backend = os.environ.get("MVF_DIST_BACKEND", "gloo" if mock else "nccl")
        kw = {"device_id": dev} if backend == "nccl" else {}
        torch.distributed.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
        if torch.distributed.get_world_size() != args.gpus:
            sys.exit(f"[bench] process group reports world size {torch.distributed.get_world_size()}, --gpus {args.gpus}")
    return world, rank, dev, backend


def reducer_of(step):
    return getattr(getattr(step, "trainer", step), "reducer", None)

def comm_report(args, rank, dev, backend, step, counts_per_step):
    """What the process group itself says about the job (all ranks call this); compact: it rides in the line."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return None
    mine = torch.tensor([torch.cuda.current_device() if dev.type == "cuda" else -1, os.getpid()],
                        dtype=torch.int64, device=dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    devices, pids = [int(p[0]) for p in parts], [int(p[1]) for p in parts]
    if backend == "nccl" and len(set(devices)) != len(devices):
        sys.exit(f"[bench] ranks share a GPU under RCCL: devices {devices}")
    red = reducer_of(step)
    rep = {"backend": "rccl (torch backend 'nccl')" if backend == "nccl" else backend,
           "world_size": dist.get_world_size(), "devices": devices, "distinct_processes": len(set(pids)), "pids": pids,
           "grad_exchange": red.exchange if red else None, "overlap_with_backward": bool(red.overlap) if red else None,
           "grad_buckets": red.num_buckets if red else 0, "grad_bucket_bytes": red.total_bytes if red else 0,
           "collectives_per_step": counts_per_step}
    if red is not None:
        rep["exchanges_issued_during_backward"] = red.issued_from_hook
        rep["exchanges_issued_after_backward"] = red.issued_from_finish
        # data-parallel replicas must hold the same parameters after the same steps: two checksums per rank, compared
        # bit for bit (the same operations on the same values give the same bits)
        with torch.no_grad():
            ps = [p.detach().double() for p in red.params]
            mine = torch.stack([torch.stack([p.sum() for p in ps]).sum(), torch.stack([p.abs().sum() for p in ps]).sum()])
        sums = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(sums, mine)
        rep["replicas_identical"] = all(torch.equal(s_, sums[0]) for s_ in sums)
    return rep


def barrier_sync(world):
    if world > 1:
        torch.distributed.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()

def timed_steps(step, steps, world):
    barrier_sync(world)
    t0 = time.perf_counter()
    c0, th0 = time.process_time(), time.thread_time()
    for _ in range(steps):
        step()
    c1, th1 = time.process_time(), time.thread_time()      # before the closing synchronisation: enqueue cost only
    barrier_sync(world)
    elapsed = time.perf_counter() - t0
    HOST_CPU.update(process_cpu_ms_per_step=round((c1 - c0) / steps * 1e3, 3),
                    main_thread_cpu_ms_per_step=round((th1 - th0) / steps * 1e3, 3))
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


class MockStep:
    """Toy data-parallel step on the CPU over gloo (plumbing test of the launcher and of the line; NOT a benchmark)."""

    def __init__(self, args, rank, world):
        from mono_vifi_amd import parallel
        torch.manual_seed(0)
        self.net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
        self.reducer = parallel.BucketedGradReducer(
            list(self.net.parameters()), world, bucket_mb=0.008, exchange=args.grad_exchange,
            overlap=not args.no_overlap, always_reduce=args.force_collectives)
        self.x = torch.randn(args.batch, 64, generator=torch.Generator().manual_seed(rank))
        self.images_per_step = args.batch

    def describe(self):
        return "mock CPU step (plumbing test of the multi-rank launcher; NOT a benchmark)"

    def __call__(self):
        self.reducer.zero_grad()
        loss = self.net(self.x).pow(2).mean()
        loss.backward()
        self.reducer.finish()
        return loss


def unit_bytes_per_px(noise_tensor, share_identity, use_affine=True):
    """(algorithmic, hand-over) HBM bytes per pixel and unit, mean over the units of a step.  Hand-over
    = this build's own extra traffic (the identity maps a single-frame unit writes and its multi-frame partner reads,
    8 B/px each): in the PMC traffic, NOT part of `roofline.achieved`."""
    groups = 3 if use_affine else 2
    per = [FB_BYTES_PER_PX + (NOISE_BYTES_PER_PX if noise_tensor else 0)] * groups
    if use_affine:
        per[2] += MASK_BYTES_PER_PX
    hand = [8.0 if share_identity else 0.0] * 2 + [0.0] * (groups - 2)
    return sum(per) / groups, sum(hand) / groups


class HotPathStep:
    """The 9 units of a step, forward + backward, as the trainer issues them (reference train.py:747-760, 795-810,
    837-882): single-frame + affine units as one launch of six, then the three multi-frame units, which share target,
    sources and poses with the single-frame ones and take their identity maps.  15 distinct image buffers + 9
    disparities per step (> 256 MiB Infinity Cache at batch 12, 640x192)."""

    def __init__(self, args, rank, dev):
        from types import SimpleNamespace
        from mono_vifi_amd import layers, synthetic
        from mono_vifi_amd.losses import HotPathLosses
        self.l = HotPathLosses()
        self.share = not args.no_share_identity
        self.batched = not args.no_batch_units
        self.merged = self.batched and not args.no_merge_unit_groups
        self.l.opt = SimpleNamespace(min_depth=0.1, max_depth=100.0, no_ssim=False, avg_reprojection=False,
                                     disable_automasking=False, disparity_smoothness=1e-3,
                                     inkernel_noise=args.noise == "kernel", batch_units=self.batched)
        B, H, W = args.batch, args.height, args.width
        self.units = []
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        for u in range(UNITS_PER_STEP):
            affine = u >= 6          # the three affine units carry valid_mask_rec
            inp = synthetic.unit_inputs(1234 + 97 * rank + u, B, H, W, with_mask=affine, disp_mode=args.disp)
            d = dict(disp=t(inp["disp"]).requires_grad_(True))
            if 3 <= u < 6:           # multi-frame unit of target u-3: same target, sources, poses
                d.update({k: self.units[u - 3][k] for k in ("T", "tgt", "src", "K", "inv_K", "mask")})
            else:
                aa, tr = t(inp["axisangle"]), t(inp["translation"])
                T = torch.stack([layers.transformation_from_parameters(aa[k], tr[k], invert=(k == 1))
                                 for k in range(2)], 0).detach()
                d.update(T=T.requires_grad_(True), tgt=t(inp["tgt"]), src=[t(inp["src"][0]), t(inp["src"][1])],
                         K=t(inp["K"]), inv_K=t(inp["inv_K"]), mask=t(inp["mask_rec"]) if affine else None)
            self.units.append(d)
        self.images_per_step = B
        self.bytes_per_px, self.handover_bytes_per_px = unit_bytes_per_px(args.noise != "kernel",
                                                                          self.share and self.batched)
        self.text = (f"{UNITS_PER_STEP} view-synthesis + photometric-loss units fwd+bwd "
                     f"({'6 + 3 units in 2 launches' if self.merged else '3 launches of 3' if self.batched else '9 launches'}), "
                     f"batch {B}/GPU, {W}x{H}, 2 sources/unit, exact mode, {args.disp} disparity")

    def describe(self):
        return self.text

    def __call__(self):
        for u in self.units:
            u["disp"].grad = None
            u["T"].grad = None

        def entries(us, idents=None):
            return [dict(disp_tgt={("disp", 0): u["disp"]}, img_tgt=u["tgt"], poses=u["T"], imgs_src=u["src"],
                         K=u["K"], inv_K=u["inv_K"], mask_rec=u["mask"],
                         ident=(idents[i] if idents is not None else None)) for i, u in enumerate(us)]
        sf, mf, af = self.units[0:3], self.units[3:6], self.units[6:9]
        if self.merged:
            total, ids, _ = self.l.compute_units(entries(sf + af), want_ident=[self.share] * 3 + [False] * 3, want_sum=True)
            total, _, _ = self.l.compute_units(entries(mf, ids[:3] if ids is not None else None), want_sum=True, sum_in=total)
        else:
            total, idents, _ = self.l.compute_units(entries(sf), want_ident=self.share, want_sum=True)
            total, _, _ = self.l.compute_units(entries(mf, idents), want_sum=True, sum_in=total)
            total, _, _ = self.l.compute_units(entries(af), want_sum=True, sum_in=total)
        total.backward()
        return total


def cpu_quota():
    """(CPUs the container may use, hardware threads it sees): cgroup quota, else affinity mask (gpurun boxes: 16 of 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = f.read().strip(), g.read().strip()
        except OSError:
            return n, n
    if q in ("max", "-1"):
        return n, n
    return min(n, max(1, int(round(int(q) / int(per))))), n


def cpu_baseline(args):
    """The oracle (oracle/mvf_oracle.c: the CPU port of the reference's algorithm, held bit-exact to the reference's
    golden vectors) on a bounded sample of the same workload: one unit forward + backward at the benchmark's batch and
    resolution.  Threads = the CPUs the container is granted (no sweep: which oversubscription wins differs box to
    box -- VERDICT r05 item 7); THREE samples of cpu_seconds / 3 each, the median is the value.  The reference's Python
    cannot travel here: `reference_cpu_expected` scales its build-container timing by cores."""
    from mono_vifi_amd import synthetic
    from oracle import oracle as O
    Bs = args.batch
    inp = synthetic.unit_inputs(4321, Bs, args.height, args.width, with_mask=True)
    T = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1)) for k in range(2)], 0)
    cores, hw = cpu_quota()
    threads = O.set_threads(cores)
    # the port's ~100 MB of output arrays per call come from the heap and are reused (M_MMAP_THRESHOLD / M_TRIM_THRESHOLD
    # raised): by default every call page-faults them afresh or not depending on what the process freed before (24 vs 36)
    libc = __import__("ctypes").CDLL(None)
    libc.mallopt(-3, 1 << 30), libc.mallopt(-1, (1 << 31) - 1)

    def one():
        O.unit(inp["disp"], inp["tgt"], inp["src"], T, inp["K"], inp["inv_K"], inp["noise"], inp["mask_rec"], 0,
               want_grads=True)
    one()                                             # warm: page faults, OpenMP team start
    samples, n_tot, t_tot = [], 0, 0.0
    for _ in range(3):
        n, t0 = 0, time.perf_counter()
        while True:
            one()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= args.cpu_seconds / 3.0 or n >= 2000:
                break
        samples.append(Bs / (UNITS_PER_STEP * dt / n))
        n_tot, t_tot = n_tot + n, t_tot + dt
    ref = REFERENCE_CPU_SAMPLES_PER_S.get((Bs, args.height, args.width))
    return {"value": round(statistics.median(samples), 3), "unit": "images/sec", "cores": cores, "threads": threads,
            "kind": "port", "runs": 3, "min": round(min(samples), 3), "max": round(max(samples), 3),
            "hardware_threads": hw,
            # the same quantity from the reference's own measured CPU path: BASELINE.md's timing scaled by cores, and
            # THIS run's value / (port over reference on the same cores)
            "reference_cpu_expected": round(ref / UNITS_PER_STEP * cores / REFERENCE_CPU_CORES, 3) if ref else None,
            "reference_cpu_estimate": round(statistics.median(samples) / PORT_OVER_REFERENCE, 3),
            "sample": f"{n_tot} x (1 unit fwd+bwd, batch {Bs}, {args.width}x{args.height}) in {t_tot:.1f} s, 3 runs, "
                      f"median; value = hot-path part of a step ({UNITS_PER_STEP} units)"}


def isa_cost():
    """Static cost-weighted instruction stream of the shipped unit kernel (tools/isa_cost.py), newest file wins."""
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_cost.json")), reverse=True):
        try:
            with open(p) as f:
                return json.load(f), os.path.relpath(p, ROOT)
        except (OSError, ValueError):
            pass
    return None, None


def unit_roofline(nat, step, args):
    """Roofline object of the unit kernel from the HIP events the library recorded around every launch of it since the
    last profile reset (on the launch stream).  achieved = algorithmic bytes per launch / average launch time; the
    kernel is priced against the HBM roofline although what limits it is VALU issue (`limiter`)."""
    fb_ms, fb_n = nat.profile_read(nat.PROF_UNIT_FWDBWD)
    if fb_n == 0:
        return None
    px_total = nat.profile_read_work(nat.PROF_UNIT_FWDBWD)
    st = getattr(step, "trainer", step)
    o = getattr(st, "opt", None)
    if hasattr(step, "bytes_per_px"):
        bpp, hand = step.bytes_per_px, step.handover_bytes_per_px
    elif o is not None:
        bpp, hand = unit_bytes_per_px(not getattr(o, "inkernel_noise", True),
                                      getattr(o, "share_identity", True) and getattr(o, "batch_units", True)
                                      and getattr(o, "fused_units", True), getattr(o, "use_affine", True))
    else:
        bpp, hand = unit_bytes_per_px(args.noise != "kernel", True)
    px_launch = px_total / fb_n
    avg_s = fb_ms / fb_n / 1e3
    ach = bpp * px_launch / avg_s / 1e9
    px_unit = args.batch * args.height * args.width
    r = {"kernel": UNIT_KERNEL, "bound": "hbm", "limiter": "valu", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "traffic_over_algorithmic": None,
         "avg_us": round(avg_s * 1e6, 2), "launches": fb_n, "us_per_unit": round(avg_s * 1e6 * px_unit / px_launch, 2),
         "units_per_launch": round(px_launch / px_unit, 2), "algorithmic_bytes_per_launch": int(round(bpp * px_launch)),
         "algorithmic_bytes_per_px": round(bpp, 2), "handover_bytes_per_launch": int(round(hand * px_launch)),
         "valu_busy": None, "valu_instr_per_px": None, "valu_pipe_frac": None}
    # the launch kinds of a step are different work: median per kind, and the share of the VALU pipe the executed
    # instruction stream occupies (static cost-weighted ISA / median launch time)
    recs = nat.profile_read_launches(nat.PROF_UNIT_FWDBWD)
    if recs:
        r["median_us"] = round(statistics.median(ms for ms, _, _ in recs) * 1e3, 2)
        cost, src = isa_cost()
        pipe_ns = tot_ns = 0.0
        for tag, name in nat.TAG_NAMES.items():
            sel = [(ms, px) for ms, px, t in recs if t == tag]
            if not sel:
                continue
            med_us = statistics.median(ms for ms, _ in sel) * 1e3
            r["median_us_" + name.replace("+", "_")] = round(med_us, 2)
            if cost:
                cpp = cost["cost_per_px"]
                c = 0.5 * (cpp["single_frame"] + cpp["affine"]) if name == "single_frame+affine" else cpp[name]
                ns = c * statistics.median(px for _, px in sel) / cost["wave"] / cost["simds"] * cost["plain_instruction_ns"]
                pipe_ns += ns * len(sel)
                tot_ns += med_us * 1e3 * len(sel)
        if tot_ns:
            r["valu_pipe_frac"] = round(pipe_ns / tot_ns, 3)
            r["valu_pipe_source"] = src
    return r


def around_unit_launches(step, nat, steps=3):
    """ms per step of the preparing / finishing / gradient-scale launches around the unit launches inside `step`
    (profile level 2: an event pair around every launch of the build's own kernels) over a few extra steps."""
    nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
    nat.check(nat.lib().mvf_profile_enable(2), "profile_enable")
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        nat.lib().mvf_profile_enable(0)
    tot = 0.0
    for kid in range(nat.PROF_FIRST_GLUE, nat.PROF_COUNT):
        if nat.profile_name(kid) in ("k_units_finish", "k_units_prepare", "k_fb_scale", "k_disp_mean"):
            tot += nat.profile_read(kid)[0]
    nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
    return tot / steps


def hotpath_leg(args, rank, dev, nat, steps=20, warmup=5):
    """The hot path alone on the GPU (9 units fwd+bwd per batch, inputs resident in HBM): the quantity cpu_baseline
    measures, in the same unit.  The loop's own time first, without instrumentation; then the same steps again with the
    per-launch events, for the kernel's launch time."""
    step = HotPathStep(args, rank, dev)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
    nat.check(nat.lib().mvf_profile_enable(1), "profile_enable")
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    nat.lib().mvf_profile_enable(0)
    rf = unit_roofline(nat, step, args)
    return {"value": round(step.images_per_step * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "roofline": rf, "step": step}


def pmc_leg(args, timeout_s=120):
    """HBM traffic and VALU counters of the unit kernel, measured NOW: `rocprofv3 --pmc` (counters only; one pass per
    TCC counter as /opt/skills/guides/MI355X_MICROARCH.md prescribes) around short hot-path child runs of this script,
    mean per launch of k_unit_fb.  FETCH_SIZE is doubled (the guide's gfx950 note; calibrated here on k_disp_mean),
    WRITE_SIZE taken 1:1.  None when rocprofv3 is missing or a pass fails."""
    import csv, shutil, subprocess, tempfile  # noqa: E401
    if not shutil.which("rocprofv3"):
        return None
    passes = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
               "GRBM_GUI_ACTIVE"]]
    got, launches = {}, 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    for counters in passes:
        d = tempfile.mkdtemp(prefix="mvf_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--workload", "hotpath", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
               "--no-pmc-leg", "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width),
               "--noise", args.noise]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            acc = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "k_unit_fb" in row["Kernel_Name"]:
                            a = acc.setdefault(row["Counter_Name"], [0.0, 0])
                            a[0] += float(row["Counter_Value"])
                            a[1] += 1
            for c in counters:
                if c not in acc:
                    return None
                got[c] = acc[c][0] / acc[c][1]
                launches = acc[c][1]
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    per_launch = 1.0 if args.no_batch_units else UNITS_PER_STEP / (3.0 if args.no_merge_unit_groups else 2.0)
    px_launch = per_launch * args.batch * args.height * args.width
    quad = got["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0            # SQ counters tick in quad-cycles over 1,024 SIMDs
    return {"traffic": int(round(got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024)),
            "valu_busy": round(got["SQ_ACTIVE_INST_VALU"] / quad, 3),
            "valu_instr_per_px": int(round(got["SQ_INSTS_VALU"] * 64.0 / px_launch)),
            "wave_active": round(got["SQ_ACTIVE_INST_ANY"] / got["SQ_WAVE_CYCLES"], 3),
            "wave_wait_memory_or_barrier": round(got["SQ_WAIT_ANY"] / got["SQ_WAVE_CYCLES"], 3),
            "wave_wait_issue": round(got["SQ_WAIT_INST_ANY"] / got["SQ_WAVE_CYCLES"], 3), "pmc_launches": launches}


def static_pmc():
    """The committed rocprofv3 --pmc summary (profiles/hbm_traffic.json) when no live pass ran; tagged as such."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            j = json.load(f)
        v = j["_valu"][UNIT_KERNEL]
        return {"traffic": j[UNIT_KERNEL], "valu_busy": v["valu_busy"], "valu_instr_per_px": v["valu_instr_per_px"]}
    except (OSError, ValueError, KeyError):
        return None


def over_budget(args, need_s):
    return (time.perf_counter() - T_START) + need_s > args.time_budget

def emit(line):
    """The ONE JSON line is the last thing on stdout: RCCL writes its banner through C stdio, which is block-buffered
    on a pipe and would otherwise land after the line when the process exits."""
    assert len(line.encode()) < LINE_LIMIT, f"bench line is {len(line.encode())} bytes (limit {LINE_LIMIT})"
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stderr.flush()
    print(line, flush=True)


def main():
    args = parse()
    if args.hip_graph:
        mono_vifi_amd.ensure_graph_replay_env()       # before the first HIP call of this process and of its ranks
    launch_ranks_if_needed(args)
    workload = args.workload
    if workload != "mock":
        mono_vifi_amd.use_shipped_miopen_db()         # per rank: after the launcher re-executed, before any convolution
    world, rank, dev, backend = dist_setup(args)
    from mono_vifi_amd import parallel

    nat = None
    if workload != "mock":
        from mono_vifi_amd import _native as nat
        nat.lib()                                     # fail loudly if the HIP library is missing
    if workload == "train":
        torch.backends.cudnn.benchmark = bool(args.miopen_find)
        from mono_vifi_amd.bench_train import TrainStep
        step = TrainStep(args, rank, world, dev)
    elif workload == "mock":
        step = MockStep(args, rank, world)
    else:
        step = HotPathStep(args, rank, dev)

    for _ in range(args.warmup):
        step()
    if workload == "train" and args.hip_graph:
        while step.trainer._step_graph.graph is None:      # eager warm-up + capture stay untimed
            step()
    if nat:
        nat.check(nat.lib().mvf_profile_reset() or nat.lib().mvf_profile_enable(1), "profile_enable")
    parallel.reset_comm_counts()
    red = reducer_of(step)
    elapsed = timed_steps(step, args.steps, world)
    host_cpu = dict(HOST_CPU)
    counts = {k: round(v / args.steps, 2) for k, v in sorted(parallel.comm_counts().items())}
    roofline = (nat.lib().mvf_profile_enable(0), unit_roofline(nat, step, args))[1] if nat else None

    single = world == 1 and rank == 0 and workload != "mock"
    headline = single and not args.hip_graph
    in_step = None
    if headline and workload == "train" and roofline and args.hotpath_leg:
        around = around_unit_launches(step, nat)
        unit_ms = roofline["avg_us"] * roofline["launches"] / args.steps / 1e3
        in_step = (round(unit_ms + around, 4), round(unit_ms, 4))

    # ---- communication report (every rank takes part in its collectives)
    comm = comm_report(args, rank, dev, backend, step, counts)
    if comm is not None and red is not None and args.comm_leg_steps > 0 and not args.hip_graph:
        red.overlap = not red.overlap                  # the same step with the other issue order
        for _ in range(2):
            step()
        t_other = timed_steps(step, args.comm_leg_steps, world)
        red.overlap = not red.overlap
        comm["overlapped" if args.no_overlap else "no_overlap"] = {
            "ms_per_step": round(t_other / args.comm_leg_steps * 1e3, 4), "steps": args.comm_leg_steps}

    # ---- the hot path alone (like-for-like with cpu_baseline), live counters of the unit kernel, CPU baseline
    hp = None
    if single and workload == "train" and args.hotpath_leg and not over_budget(args, 20):
        hp = hotpath_leg(args, rank, dev, nat)
    if roofline is None and hp and hp["roofline"]:
        roofline = dict(hp["roofline"], events_from="hot-path-only leg (no events inside a graph replay)")
    if single and roofline:
        live = pmc_leg(args) if (args.pmc_leg and headline and not over_budget(args, 45)) else None
        pmc = live or (static_pmc() if (args.batch, args.height, args.width) == (12, 192, 640) else None)
        if pmc:
            roofline.update({k: v for k, v in pmc.items() if k != "pmc_launches"})
            roofline["traffic_over_algorithmic"] = round(pmc["traffic"] / roofline["algorithmic_bytes_per_launch"], 3)
            roofline["pmc_source"] = "rocprofv3 --pmc child runs inside this run" if live else "profiles/hbm_traffic.json (static)"

    detail = None
    if args.detail and single:
        import importlib.util
        spec = importlib.util.spec_from_file_location("mvf_measure_detail", os.path.join(ROOT, "tools", "measure_detail.py"))
        measure_detail = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(measure_detail)
        detail = measure_detail.run(sys.modules[__name__], args, step, nat, rank, world, dev,
                                    hp["step"] if hp else (step if workload == "hotpath" else None))

    line = None
    if rank == 0:
        n_gpus = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        assert n_gpus == args.gpus == world
        if comm and dev.type == "cuda" and len(set(comm["devices"])) < world:
            # rehearsal only (MVF_BENCH_SHARE_GPU=1, gloo): several ranks on one device -- the line says how many GPUs
            # really worked, the group size rides beside it
            n_gpus = len(set(comm["devices"]))
        images = step.images_per_step * world * args.steps
        metric = {"train": "training images/sec (640x192, 3-frame)", "mock": "mock plumbing steps/sec x batch (NOT a benchmark)"}.get(
            workload, "hot-path images/sec (9 view-synthesis + photometric-loss units fwd+bwd per batch)")
        out = {"metric": metric, "value": round(images / elapsed, 2), "unit": "images/sec", "n_gpus": n_gpus,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": (f"{workload}: " + (getattr(step, "describe_short", step.describe)() or ""))[:200],
                          "global_batch": args.batch * world, "parallelism": f"dp{world}"},
               "roofline": roofline}
        if n_gpus != world:
            out["world"] = world
        if workload != "mock":
            out["host_process_cpu_ms_per_step"] = host_cpu.get("process_cpu_ms_per_step")
        if in_step:
            out["hotpath_in_step_ms"], out["hotpath_in_step_unit_launches_ms"] = in_step
        if hp:
            out["hotpath_ms_per_step"] = hp["ms_per_step"]
            out["hotpath_images_per_sec"] = hp["value"]
            rf = hp["roofline"] or {}
            if rf.get("avg_us"):
                out["hotpath_unit_launches_ms_per_step"] = round(rf["avg_us"] * rf["launches"] / hp["steps"] / 1e3, 4)
                out["hotpath_frac"] = rf["frac"]
        if comm:
            out["comm"] = comm
        if single and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        if detail is not None:
            os.makedirs(os.path.dirname(args.detail_out), exist_ok=True)
            with open(args.detail_out, "w") as f:
                json.dump(dict(detail, line=out), f, indent=1)
            out["detail_file"] = os.path.relpath(args.detail_out, ROOT)
        out["wall_seconds"] = round(time.perf_counter() - T_START, 1)
        line = json.dumps(out, separators=(",", ":"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if line is not None:
        emit(line)


if __name__ == "__main__":
    main()
