#!/usr/bin/env python3
"""Throughput benchmark of the Mono-ViFI hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload hotpath|train]

One process per GPU.  Under ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment) each process is one rank; started plainly with ``--gpus N > 1``
the script launches the N ranks ITSELF (re-executes under ``torch.distributed.run`` on
127.0.0.1) and fails loudly when the box has fewer than N GPUs -- it never prints a line whose
``n_gpus`` differs from the world size the process group reports (reference: train.py:1178-1185,
README.md:130-133: world = visible GPUs, one process each, ``init_process_group('nccl')``).
W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize on both
sides; the maximum over ranks is used and rank 0 prints ONE JSON line.  For N > 1 (or
``--force-collectives`` on one GPU) the line carries a ``comm`` object: backend, world size and
devices as the process group reports them, gradient buckets, collectives per step by kind, and
the step time with the exchange issued after backward (``no_overlap``) beside the overlapped one.

Workloads (config.workload in the JSON names the one that ran):

* ``hotpath`` -- a step is one pass of the view-synthesis + photometric-loss path over one
  batch: the 9 units ``Trainer.process_batch`` runs per optimisation step with
  ``use_affine`` (reference train.py:747-883), each = 2 x generate_images_pred +
  compute_losses_base, forward AND backward (grad_disp, grad_T), on synthetic
  KITTI-shaped triplets (BASELINE.json configs[1]: batch 12, 640x192, 3-frame) resident in
  HBM.  The 9 units use distinct buffers (531 MB > 256 MiB Infinity Cache).
* ``train`` -- the whole optimisation step of the drop-in trainer (networks + hot path +
  backward + clip + AdamW), see mono-vifi_amd/trainer.py.

``roofline`` is for the dominant hot-path kernel (the unit's forward+backward tile kernel), from HIP events the
library records around every launch of it inside the timed region.  ``cpu_baseline`` times
the CPU oracle (a port of the reference's algorithm, checked bit-exact against it) on the
host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# On a box with an empty MIOpen cache the first call of every convolution shape runs a find
# that also times MIOpen's naive reference solvers (hundreds of ms each; never selected for
# these shapes): 288 s -> 143 s of cold start for this benchmark.  Warm-up only; the timed
# region is unaffected.  (setdefault: an explicit environment setting wins.)
for _d in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _d, "0")

# importing the package sets nothing; main() calls its two entry-point helpers before the first HIP call:
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 only when --hip-graph was asked for, and a per-process
# copy of the shipped MIOpen find-db (the tracked file is never MIOpen's writable user db)
import mono_vifi_amd  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0    # same guide: measured-achievable copy bandwidth (SURVEY.md section 8d quotes both)
FWD_BYTES_PER_PX = 44          # disp 4 + tgt 12 + 2 src 24 read, 4 written
BWD_BYTES_PER_PX = 45          # same reads + argmin 1, grad_disp 4 written
# the forward+backward tile kernel reads its inputs ONCE: disp 4 + tgt 12 + 2 src 24 read,
# argmin 1 + grad_disp 4 written = 45 B/px; + 8 B/px when the tie-break noise is supplied as a
# tensor (2 identity candidates), + 4 B/px when a mask_rec plane is supplied
FB_BYTES_PER_PX = 45
NOISE_BYTES_PER_PX = 8
MASK_BYTES_PER_PX = 4
UNITS_PER_STEP = 9             # reference train.py:747-883 with use_affine
MASKED_UNITS_PER_STEP = 3      # the three affine units carry valid_mask_rec (train.py:830-868)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("MVF_BENCH_WORKLOAD", "auto"),
                    choices=["auto", "hotpath", "train", "mock"],
                    help="mock: a toy CPU step over gloo -- plumbing test of the launcher / comm report "
                         "(tests/test_bench_launch.py), never a benchmark")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--backbone", default="ResNet18")
    ap.add_argument("--disp", default="smooth", choices=["smooth", "noise"],
                    help="hotpath workload: disparity statistics (smooth = like a depth network's "
                         "output; noise = i.i.d. uniform, worst case for gather locality)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hotpath-leg", dest="no_hotpath_leg", action="store_true",
                    help="train workload: skip the extra hot-path-only measurement")
    ap.add_argument("--noise", default="kernel", choices=["kernel", "tensor"],
                    help="auto-mask tie-break noise: drawn inside the tile kernel (counter-based, "
                         "default) or supplied as a torch.randn tensor per unit (+8 B/px)")
    ap.add_argument("--amp-bf16", dest="amp_bf16", action="store_true",
                    help="bf16 autocast for the conv networks (reduced precision: not the default)")
    ap.add_argument("--channels-last", dest="channels_last", action="store_true")
    ap.add_argument("--miopen-find", dest="miopen_find", action="store_true",
                    help="torch.backends.cudnn.benchmark=True (MIOpen exhaustive find in warm-up)")
    ap.add_argument("--hip-graph", dest="hip_graph", action="store_true",
                    help="train workload: the timed steps replay the whole optimisation step as ONE "
                         "HIP graph (trainer --hip_graph); the unit kernel's events cannot be recorded "
                         "inside a graph, so `roofline` then comes from the hot-path-only leg")
    ap.add_argument("--hip-graph-scope", dest="hip_graph_scope", default="step", choices=["step", "backward"],
                    help="what --hip-graph / --graph-leg capture (trainer --hip_graph_scope)")
    ap.add_argument("--no-graph-leg", dest="graph_leg", action="store_false", default=True,
                    help="train workload, N = 1: skip the HIP-graph measurement of the step (`hip_graph_step`, "
                         "run in a child process under a timeout: a fault or hang of a replay cannot be caught "
                         "and must not take the line down)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-batch-units", dest="no_batch_units", action="store_true",
                    help="one launch per unit (round-2 launch structure) instead of one per group of three")
    ap.add_argument("--no-merge-unit-groups", dest="no_merge_unit_groups", action="store_true",
                    help="hot path / training step: single-frame and affine units as two launches of three (round 4) "
                         "instead of one launch of six")
    ap.add_argument("--no-share-identity", dest="no_share_identity", action="store_true",
                    help="multi-frame units re-evaluate the identity candidates instead of taking the maps "
                         "of the single-frame unit of the same target")
    ap.add_argument("--no-regroup", dest="no_regroup", action="store_true",
                    help="train workload: per-group views of the grouped encoder's pyramids re-merged with stack "
                         "(round-3 data flow) instead of one regrouping launch per level")
    ap.add_argument("--grad-exchange", dest="grad_exchange", default="all_reduce",
                    choices=["all_reduce", "reduce_scatter"],
                    help="per gradient bucket: one RCCL all-reduce, or reduce-scatter + all-gather on the "
                         "flat buffer (SURVEY.md 8f-3)")
    ap.add_argument("--no-overlap", dest="no_overlap", action="store_true",
                    help="timed region with the gradient exchange issued AFTER backward (the default "
                         "issues each bucket from a hook during backward)")
    ap.add_argument("--force-collectives", dest="force_collectives", action="store_true",
                    help="N = 1: run the data-parallel collectives through RCCL in a group of one")
    ap.add_argument("--comm-leg-steps", dest="comm_leg_steps", type=int, default=10,
                    help="N > 1: timed steps of the extra no-overlap measurement in `comm` (0 = skip)")
    ap.add_argument("--also-configs", dest="also_configs", default="auto",
                    help="train workload, N = 1: also time BASELINE.json configs 3-5 (DHRNet 640x192, "
                         "Lite-Mono 1024x320, DHRNet 512x192) for a few steps each and report them as "
                         "`other_configs`; auto = on for the default headline run, 'none' = off, or a "
                         "comma list of C3,C4,C5")
    ap.add_argument("--time-budget", dest="time_budget", type=float, default=300.0,
                    help="seconds of wall time after which the OPTIONAL legs that have not started yet (other "
                         "configs, hot-path-only leg, CPU baselines) are skipped and reported as skipped -- the "
                         "line must come out within minutes even on a box whose MIOpen find-db is cold")
    ap.add_argument("--no-replay-leg", dest="replay_leg", action="store_false", default=True,
                    help="hotpath workload: skip the HIP-graph replay measurement (the PMC child runs do)")
    ap.add_argument("--no-pmc-leg", dest="pmc_leg", action="store_false", default=True,
                    help="N = 1 headline run: skip the live PMC passes (rocprofv3 --pmc around short hot-path runs of "
                         "this script as child processes: HBM traffic and VALU instruction counts of the unit "
                         "kernel per launch); roofline.traffic / valu then quote the committed profiles/ summary")
    ap.add_argument("--no-mfma-leg", dest="mfma_leg", action="store_false", default=True,
                    help="N = 1 headline run: skip the live MFMA / VALU utilisation pass over the training step's kernels "
                         "(rocprofv3 --pmc around a child run of four steps; `conv_mfma` in the line)")
    ap.add_argument("--no-kernel-leg", dest="kernel_leg", action="store_false", default=True,
                    help="skip the roofline leg of this build's own glue kernels (`own_kernels` in the line: event "
                         "pairs around every launch for three extra steps)")
    ap.add_argument("--no-host-leg", dest="host_leg", action="store_false", default=True,
                    help="N = 1 headline run: skip the host-cost measurement (the step in child processes pinned to the CPUs "
                         "one of eight ranks would have on this box, eager and as a HIP graph; `host.pinned` in the line)")
    ap.add_argument("--also-steps", dest="also_steps", type=int, default=10)
    ap.add_argument("--also-warmup", dest="also_warmup", type=int, default=5)
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: start the N ranks here.
    One process per GPU like the reference's launcher (README.md:130-133); refuses to run when
    the box cannot give every rank its own GPU."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import subprocess
    if args.workload != "mock":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        # MVF_BENCH_SHARE_GPU=1: plumbing test only (tests/test_bench_launch.py: two gloo ranks on the one
        # GPU of the test box); RCCL needs one GPU per rank and comm_report refuses shared devices under it
        if have < args.gpus and not (have > 0 and os.environ.get("MVF_BENCH_SHARE_GPU") == "1"):
            sys.exit(f"[bench] --gpus {args.gpus} needs {args.gpus} GPUs (one process per GPU); this "
                     f"box has {have}.  Not running {args.gpus} ranks on fewer devices.")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
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
        torch.distributed.init_process_group(backend=backend, init_method="env://",
                                             world_size=world, rank=rank, **kw)
        got = torch.distributed.get_world_size()
        if got != args.gpus:
            sys.exit(f"[bench] process group reports world size {got}, --gpus {args.gpus}")
    return world, rank, dev, backend


def comm_report(args, world, rank, dev, backend, step, counts_per_step):
    """What the process group itself says about the job (all ranks call this)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return None
    mine = torch.tensor([torch.cuda.current_device() if dev.type == "cuda" else -1, os.getpid()],
                        dtype=torch.int64, device=dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    devices = [int(p[0]) for p in parts]
    pids = [int(p[1]) for p in parts]
    if backend == "nccl" and len(set(devices)) != len(devices):
        sys.exit(f"[bench] ranks share a GPU under RCCL: devices {devices}")
    red = getattr(getattr(step, "trainer", step), "reducer", None)
    rep = {"backend": "rccl (torch backend 'nccl')" if backend == "nccl" else backend,
           "world_size": dist.get_world_size(), "devices": devices,
           "distinct_processes": len(set(pids)), "pids": pids,
           "grad_exchange": red.exchange if red else None,
           "overlap_with_backward": bool(red.overlap) if red else None,
           "grad_buckets": red.num_buckets if red else 0,
           "grad_bucket_bytes": red.total_bytes if red else 0,
           "collectives_per_step": counts_per_step}
    if red is not None:
        # of the last step: where its exchanges were issued, and (GPU) how much of the backward pass was still
        # ahead on the compute stream when each bucket went out
        rep["exchanges_issued_during_backward"] = red.issued_from_hook
        rep["exchanges_issued_after_backward"] = red.issued_from_finish
        tl = red.timeline_ms() if hasattr(red, "timeline_ms") else []
        if tl:
            rep["backward_ms_remaining_at_issue"] = tl
    return rep


def barrier_sync(world):
    if world > 1:
        torch.distributed.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class MockStep:
    """Toy data-parallel step on the CPU over gloo (plumbing test of the launcher and of the
    `comm` report; the numbers mean nothing)."""

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


# ----------------------------------------------------------------------------- hot path
def unit_bytes_per_px(noise_tensor, share_identity, use_affine=True):
    """(algorithmic, hand-over) HBM bytes per pixel and unit, mean over the units of a step.
    Algorithmic = what the reference's algorithm must move: every unit reads disp 4 + target 12 + 2 sources 24
    and writes argmin 1 + grad_disp 4 = 45; + 8 tie-break noise when it is supplied as a tensor; + 4 mask_rec on
    the affine units.  Hand-over = this build's own extra traffic: the identity maps a single-frame unit writes
    and its multi-frame partner reads (8 B/px each) -- priced separately, NOT part of `roofline.achieved`."""
    groups = 3 if use_affine else 2
    per = [FB_BYTES_PER_PX + (NOISE_BYTES_PER_PX if noise_tensor else 0)] * groups
    if use_affine:
        per[2] += MASK_BYTES_PER_PX
    hand = [8.0 if share_identity else 0.0, 8.0 if share_identity else 0.0] + [0.0] * (groups - 2)
    return sum(per) / groups, sum(hand) / groups


class HotPathStep:
    """The 9 units of a step, forward + backward, as the trainer issues them: three launches of
    three mutually independent units (single-frame / multi-frame / affine, reference
    train.py:747-760, 795-810, 837-882).  The multi-frame units share target, sources and poses
    with the single-frame ones (train.py:747-749 vs 795-797) and take their identity maps;
    every unit has its own disparity, the affine units their own images: 15 distinct image
    buffers + 9 disparities per step (> 256 MiB Infinity Cache at batch 12, 640x192)."""

    def __init__(self, args, rank, dev):
        from types import SimpleNamespace
        from mono_vifi_amd import synthetic
        from mono_vifi_amd.losses import HotPathLosses

        class L(HotPathLosses):
            pass
        self.l = L()
        self.share = not getattr(args, "no_share_identity", False)
        self.batched = not getattr(args, "no_batch_units", False)
        self.merged = self.batched and not getattr(args, "no_merge_unit_groups", False)
        self.l.opt = SimpleNamespace(min_depth=0.1, max_depth=100.0, no_ssim=False,
                                     avg_reprojection=False, disable_automasking=False,
                                     disparity_smoothness=1e-3, inkernel_noise=args.noise == "kernel",
                                     batch_units=self.batched)
        B, H, W = args.batch, args.height, args.width
        self.units = []
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        from mono_vifi_amd import layers
        for u in range(UNITS_PER_STEP):
            affine = u >= 6          # the three affine units carry valid_mask_rec
            inp = synthetic.unit_inputs(1234 + 97 * rank + u, B, H, W, with_mask=affine,
                                        disp_mode=args.disp)
            d = dict(disp=t(inp["disp"]).requires_grad_(True))
            if 3 <= u < 6:           # multi-frame unit of target u-3: same target, sources, poses
                first = self.units[u - 3]
                d.update({k: first[k] for k in ("T", "tgt", "src", "K", "inv_K", "mask")})
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

    def describe(self):
        return None

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
            # the trainer's issue order (Trainer.process_batch): single-frame + affine units as ONE launch of six,
            # then the multi-frame ones, whose finishing kernel adds the first launch's total
            total, ids, _ = self.l.compute_units(entries(sf + af), want_ident=[self.share] * 3 + [False] * 3,
                                                 want_sum=True)
            idents = ids[:3] if ids is not None else None
            total, _, _ = self.l.compute_units(entries(mf, idents), want_sum=True, sum_in=total)
        else:
            total, idents, _ = self.l.compute_units(entries(sf), want_ident=self.share, want_sum=True)
            total, _, _ = self.l.compute_units(entries(mf, idents), want_sum=True, sum_in=total)
            total, _, _ = self.l.compute_units(entries(af), want_sum=True, sum_in=total)
        total.backward()
        return total


def cpu_quota():
    """CPUs the container may use: the cgroup quota (cpu.max / cfs_quota) if there is one, else the affinity
    mask.  The gpurun boxes show 256 hardware threads and grant 16 CPUs (cpu.max = 1600000 100000): that, not
    the port, is why every thread-count sweep of the CPU baselines peaked at 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            return min(n, max(1, int(round(int(q) / int(per))))), n
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        if q > 0:
            return min(n, max(1, int(round(q / per)))), n
    except (OSError, ValueError):
        pass
    return n, n


def cpu_baseline(args):
    """The oracle (CPU port of the reference's algorithm) on a bounded sample: one unit,
    forward + backward, at the benchmark's batch and resolution (the same per-call work as the GPU's:
    a smaller batch leaves the port's row-parallel regions too little work for a 256-core host),
    thread count from a sweep (OpenMP).  The reference's own CPU path timed in the build container
    (whole step 1.1 images/s on 8 vCPU) cannot travel to this box."""
    from mono_vifi_amd import synthetic
    from oracle import oracle as O
    Bs = args.batch
    inp = synthetic.unit_inputs(4321, Bs, args.height, args.width, with_mask=True)
    T = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1))
                  for k in range(2)], 0)
    quota, host_cores = cpu_quota()

    def one():
        O.unit(inp["disp"], inp["tgt"], inp["src"], T, inp["K"], inp["inv_K"], inp["noise"],
               inp["mask_rec"], 0, want_grads=True)
    # calibrate the thread count (1 warm + 2 timed runs each) and time with the fastest.  The port's
    # parallel regions cover (image x row band) / (plane x row band) tasks since round 3 (its adjoint was
    # 36-way parallel and the smoothness adjoint serial: 16 of 256 cores was the optimum)
    best = (float("inf"), 1)
    # candidates around the CPUs the container is granted (more threads than that only contend)
    for nt in sorted({c for c in (quota // 2, quota, 2 * quota) if 1 <= c <= host_cores}):
        O.set_threads(nt)
        one()
        t0 = time.perf_counter()
        one()
        one()
        best = min(best, ((time.perf_counter() - t0) / 2, nt))
    cores = O.set_threads(best[1])
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or n >= 5000:
            break
    t_unit = dt / n
    # value = images/sec the CPU port sustains on the HOT-PATH part of a step (9 units fwd+bwd);
    # the networks of the full training step are not part of the port
    return {"value": round(Bs / (UNITS_PER_STEP * t_unit), 3), "unit": "images/sec",
            "cores": cores, "kind": "port",
            "sample": f"{n} x (1 unit fwd+bwd, batch {Bs}, {args.width}x{args.height}) in "
                      f"{dt:.1f} s; value = hot-path part of a step ({UNITS_PER_STEP} units); oracle/mvf_oracle.c, OpenMP with "
                      f"{cores} threads (fastest of a sweep around the container's CPU quota: {quota} CPUs granted of "
                      f"{host_cores} hardware threads)",
            "host": {"hardware_threads": host_cores, "cpus_granted": quota}}


def cpu_baseline_unfused(args):
    """Second CPU baseline: the SAME unit as ~130 separate ATen operators
    under autograd (oracle/torch_unfused.py: bmm, grid_sample, reflection pad + avg_pool2d SSIM,
    cat/min, ...), i.e. the operator stream the reference's own code issues on its CPU path,
    on all host cores.  Checked against the reference's golden vectors in
    tests/test_oracle_golden.py; the reference's Python itself cannot travel to this box."""
    from mono_vifi_amd import synthetic
    from oracle import oracle as O
    from oracle import torch_unfused as U
    Bs = args.batch
    inp = synthetic.unit_inputs(4321, Bs, args.height, args.width, with_mask=True)
    T = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1))
                  for k in range(2)], 0)
    quota, host_cores = cpu_quota()
    tens = [torch.from_numpy(np.ascontiguousarray(a)) for a in
            (inp["disp"], inp["tgt"], T, inp["K"], inp["inv_K"], inp["noise"], inp["mask_rec"])]
    srcs = [torch.from_numpy(np.ascontiguousarray(a)) for a in inp["src"]]

    def one():
        U.unit(tens[0], tens[1], srcs, tens[2], tens[3], tens[4], tens[5], tens[6], 0)
    old = torch.get_num_threads()
    best = (float("inf"), 1)
    try:
        for nt in sorted({c for c in (quota // 2, quota, 2 * quota) if 1 <= c <= host_cores}):
            torch.set_num_threads(nt)
            one()
            t0 = time.perf_counter()
            one()
            best = min(best, (time.perf_counter() - t0, nt))
        torch.set_num_threads(best[1])
        n, t0 = 0, time.perf_counter()
        while True:
            one()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= args.cpu_seconds or n >= 5000:
                break
    finally:
        torch.set_num_threads(old)
    t_unit = dt / n
    return {"value": round(Bs / (UNITS_PER_STEP * t_unit), 3), "unit": "images/sec",
            "cores": best[1], "kind": "port",
            "sample": f"{n} x (1 unit fwd+bwd, batch {Bs}, {args.width}x{args.height}) in {dt:.1f} s; "
                      f"value = hot-path part of a step ({UNITS_PER_STEP} units); "
                      f"oracle/torch_unfused.py: the unit as ~130 separate ATen ops under autograd "
                      f"(the operator stream of the reference's CPU path), torch intra-op threads "
                      f"{best[1]} (fastest of a sweep around the container's CPU quota: {quota} CPUs granted of "
                      f"{host_cores} hardware threads)"}


def _profiles_json():
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def traffic_from_profiles(kernel):
    """HBM bytes per launch measured OFFLINE with rocprofv3 --pmc (separate passes,
    FETCH_SIZE doubled per the gfx950 note) and committed under profiles/ -- a static value,
    not measured by this run (PMC counters need rocprofv3 around the process)."""
    return _profiles_json().get(kernel)


def valu_from_profiles(kernel):
    """VALU-pipe utilisation of the kernel from the committed PMC pass (what actually bounds
    a kernel whose HBM fraction is low): {"valu_busy": ..., "valu_instr_per_px": ...} or None.
    Static, like traffic_from_profiles."""
    return _profiles_json().get("_valu", {}).get(kernel)


def profiles_source():
    j = _profiles_json()
    return {"file": "profiles/hbm_traffic.json", "captured": j.get("_captured", "round 1"),
            "note": "traffic / valu are read from this committed rocprofv3 --pmc summary, "
                    "not measured in this run"}


def kernel_rooflines(args, fwd_ms, fwd_n, bwd_ms, bwd_n, fb_ms, fb_n, fb_pixels=0, fb_bytes_px=None,
                     static=True, launches_hint=None):
    px = args.batch * args.height * args.width

    def roof(ms, n, bytes_px, name, pixels=None):
        if n == 0:
            return None
        avg_s = ms / n / 1e3
        px_launch = (pixels / n) if pixels else px         # a launch may carry several units
        ach = bytes_px * px_launch / avg_s / 1e9
        r = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
             "frac_of_measured_achievable": round(ach / HBM_ACHIEVABLE_GBS, 4),
             "traffic": traffic_from_profiles(name) if static else None,
             "valu": valu_from_profiles(name) if static else None,
             "avg_us": round(avg_s * 1e6, 2),
             "launches": n, "images_per_launch": round(px_launch / (args.height * args.width), 2),
             "us_per_unit": round(avg_s * 1e6 * px / px_launch, 2),
             "algorithmic_bytes_per_launch": round(bytes_px * px_launch),
             "algorithmic_bytes_per_px": round(bytes_px, 2)}
        if static:
            r["static_source"] = profiles_source()
        return r

    r_fwd = roof(fwd_ms, fwd_n, FWD_BYTES_PER_PX, "k_photo_fwd<fused>")
    r_bwd = roof(bwd_ms, bwd_n, BWD_BYTES_PER_PX, "k_photo_bwd<fused>")
    # forward+backward of units in one tile kernel (the training path): priced on ITS OWN minimum
    # traffic (inputs read once), plus the optional planes the launches were given
    hand_px = 0.0
    if fb_bytes_px is None:
        st = getattr(launches_hint, "trainer", launches_hint)
        o = getattr(st, "opt", None)
        if hasattr(launches_hint, "bytes_per_px"):
            fb_bytes_px, hand_px = launches_hint.bytes_per_px, getattr(launches_hint, "handover_bytes_per_px", 0.0)
        elif o is not None:
            fb_bytes_px, hand_px = unit_bytes_per_px(
                not getattr(o, "inkernel_noise", True),
                getattr(o, "share_identity", True) and getattr(o, "batch_units", True)
                and getattr(o, "fused_units", True), getattr(o, "use_affine", True))
        else:
            fb_bytes_px, hand_px = unit_bytes_per_px(args.noise != "kernel", True)
    r_fb = roof(fb_ms, fb_n, fb_bytes_px, "k_unit_fb<2>", fb_pixels)
    if r_fb:
        px_launch = (fb_pixels / fb_n) if fb_pixels else px
        avg_s = fb_ms / fb_n / 1e3
        # what bounds this kernel is VALU pipe time, not bandwidth: `frac` stays the fraction of the HBM
        # peak on the algorithmic bytes (the contract's figure); valu_pipe_frac (below, flat) is the bound's own fraction
        r_fb["bound"] = "valu"
        # this build's own extra traffic, NOT in `achieved`: the identity maps handed from the single-frame to
        # the multi-frame units (what the PMC traffic holds beyond the algorithmic bytes)
        r_fb["handover_bytes_per_px"] = round(hand_px, 2)
        r_fb["handover_bytes_per_launch"] = round(hand_px * px_launch)
        r_fb["frac_incl_handover_bytes"] = round((fb_bytes_px + hand_px) * px_launch / avg_s / 1e9 / HBM_PEAK_GBS, 4)
        r_fb["bytes_note"] = (
            f"algorithmic bytes per pixel and unit, mean over the units of a step: {FB_BYTES_PER_PX} B (disp 4 + "
            f"target 12 + 2 sources 24 read once; argmin 1 + grad_disp 4 written), + {NOISE_BYTES_PER_PX} B when the "
            f"tie-break noise is a tensor, + {MASK_BYTES_PER_PX} B mask_rec on the affine units; a launch carries "
            "images_per_launch images (several units).  The 8 B/px identity maps a single-frame unit writes and "
            "its multi-frame partner reads are this build's own traffic (handover_bytes_*): in the PMC traffic, "
            "not in `achieved` / `frac` (frac_incl_handover_bytes prices them too)")
    cands = [(ms, r) for ms, r in ((fwd_ms, r_fwd), (bwd_ms, r_bwd), (fb_ms, r_fb)) if r]
    dominant = max(cands, key=lambda t: t[0])[1] if cands else None
    return {"unit_fwd": r_fwd, "unit_bwd": r_bwd, "unit_fwdbwd": r_fb}, dominant


def unit_launch_types(args, nat, fb_bytes_px_base, noise_tensor):
    """The three kinds of unit launch of a step are different work (single-frame: identity SSIM + hand-over write;
    multi-frame: hand-over read instead; affine: + mask plane): per kind the MEDIAN duration of the recorded launches
    (median of >= 50 runs), its algorithmic bytes and fraction of HBM peak, and the overall median."""
    recs = nat.profile_read_launches(nat.PROF_UNIT_FWDBWD)
    if not recs:
        return None
    import statistics
    out = {}
    for tag, name in nat.TAG_NAMES.items():
        sel = [(ms, px) for ms, px, t in recs if t == tag]
        if not sel:
            continue
        med_ms = statistics.median(ms for ms, _ in sel)
        px = statistics.median(p for _, p in sel)
        # (a mixed launch carries the three single-frame and the three affine units: half of its images bring a mask plane)
        bpp = FB_BYTES_PER_PX + (NOISE_BYTES_PER_PX if noise_tensor else 0) + \
            (MASK_BYTES_PER_PX if tag == 2 else MASK_BYTES_PER_PX / 2 if tag == 3 else 0)
        ach = bpp * px / (med_ms / 1e3) / 1e9
        out[name] = {"launches": len(sel), "median_us": round(med_ms * 1e3, 2),
                     "min_us": round(min(ms for ms, _ in sel) * 1e3, 2), "max_us": round(max(ms for ms, _ in sel) * 1e3, 2),
                     "bytes": int(round(bpp * px)), "bytes_per_px": bpp,
                     "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}
    res = {"by_launch": out, "median_us": round(statistics.median(ms for ms, _, _ in recs) * 1e3, 2),
           "launches_recorded": len(recs)}
    # flat scalars (the driver's parser keeps scalars only): per launch type the median, and the fraction of the VALU
    # pipe -- sum over the executed instructions of their measured issue cost (tools/isa_cost.py on this kernel,
    # profiles/r05_isa_cost.json) over the kernel's time: what a perfect schedule of this instruction stream could gain
    cost = _isa_cost()
    pipe_ns = tot_ns = 0.0
    for name, v in out.items():
        key = name.replace("+", "_")
        res[f"median_us_{key}"] = v["median_us"]
        if cost:
            cpp = cost["cost_per_px"]
            c = {"single_frame": cpp["single_frame"], "multi_frame": cpp["multi_frame"], "affine": cpp["affine"],
                 "single_frame+affine": 0.5 * (cpp["single_frame"] + cpp["affine"])}[name]
            px = v["bytes"] / v["bytes_per_px"]
            ns = c * px / cost["wave"] / cost["simds"] * cost["plain_instruction_ns"]
            res[f"valu_pipe_frac_{key}"] = round(ns / (v["median_us"] * 1e3), 3)
            pipe_ns += ns * v["launches"]
            tot_ns += v["median_us"] * 1e3 * v["launches"]
    if tot_ns:
        res["valu_pipe_frac"] = round(pipe_ns / tot_ns, 3)
        res["valu_pipe_source"] = "profiles/r05_isa_cost.json (static, cost-weighted ISA of this kernel) / median launch time of this run"
    return res


def _isa_cost():
    try:
        with open(os.path.join(ROOT, "profiles", "r05_isa_cost.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def glue_kernel_leg(step, nat, steps=3):
    """Roofline of this build's OWN kernels either side of the unit kernel: a few extra steps
    with the library's event hooks at level 2 (an event pair around every launch of the kernels listed in
    include/mvf_hotpath.h: MVF_PROF_*), outside the timed region.  Per kernel: launches and ms per step,
    algorithmic bytes per step (every input element read once, every output element written once -- stated per
    launcher in csrc/), achieved GB/s over the kernel's own time, fraction of the 8 TB/s HBM peak."""
    try:
        nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
        nat.check(nat.lib().mvf_profile_enable(2), "profile_enable")
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        nat.lib().mvf_profile_enable(0)
        out = {}
        for kid in range(nat.PROF_FIRST_GLUE, nat.PROF_COUNT):
            ms, n = nat.profile_read(kid)
            if n == 0:
                continue
            nbytes = nat.profile_read_work(kid)
            ach = nbytes / (ms / 1e3) / 1e9 if ms > 0 else 0.0
            out[nat.profile_name(kid)] = {
                "launches_per_step": round(n / steps, 1), "ms_per_step": round(ms / steps, 4),
                "avg_us": round(ms / n * 1e3, 2), "bytes_per_step": int(nbytes // steps),
                "bytes_per_launch": int(nbytes // n), "bound": "hbm", "achieved": round(ach, 1), "unit": "GB/s",
                "peak": HBM_PEAK_GBS, "frac": round(ach / HBM_PEAK_GBS, 4)}
        nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
        tot = sum(v["ms_per_step"] for v in out.values())
        return {"kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"])),
                "own_glue_ms_per_step": round(tot, 3), "steps": steps,
                "note": "HIP events around every launch of the listed kernels (profile level 2) over extra steps after "
                        "the timed region; bytes = algorithmic (inputs read once + outputs written once)"}
    except Exception as e:      # noqa: BLE001 -- an optional leg must never take the bench line down
        nat.lib().mvf_profile_enable(0)
        return {"error": f"{type(e).__name__}: {e}"[:300]}


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


def mfma_leg(args, timeout_s=240):
    """north_star's "MFMA utilisation against the chip's peak" for the conv GEMMs, measured NOW:
    one counter-only `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace`
    pass around a child run of two training steps of this script; per kernel family the share of the step's GPU
    cycles, ms per step (kernel-trace durations), MFMA-pipe busy and VALU busy fractions, and the cycle-weighted
    figures over the whole step.  Measurement only: the convolution kernels are MIOpen's (out of scope).
    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1,024 SIMDs); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs;
    VALU busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / the same."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    t0 = time.perf_counter()
    warm, timed = 2, 2
    d = tempfile.mkdtemp(prefix="mvf_mfma_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
           "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
           "--workload", "train", "--steps", str(timed), "--warmup", str(warm), "--no-cpu-baseline", "--no-hotpath-leg",
           "--also-configs", "none", "--no-graph-leg", "--no-pmc-leg", "--no-mfma-leg", "--no-kernel-leg", "--no-host-leg",
           "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width), "--backbone", args.backbone]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
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
    except subprocess.TimeoutExpired:
        return {"error": f"rocprofv3 child did not finish within {timeout_s} s"}
    except (OSError, ValueError, KeyError) as e:
        return {"error": f"{type(e).__name__}: {e}"[:200]}
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
            "peak_note": "MFMA busy 1.0 = the fp32-input MFMA peak of 157.3 TFLOP/s (v_mfma_f32_32x32x2_f32 / 16x16x4_f32: the "
                         "reference's arithmetic is fp32; /opt/skills/guides/MI355X_MICROARCH.md); the Winograd kernels are "
                         "VALU code and issue no MFMA",
            "source": {"measured": "in this run", "how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES "
                       "GRBM_GUI_ACTIVE --kernel-trace (counters only) around a child run of %d training steps" % int(steps_all),
                       "leg_seconds": round(time.perf_counter() - t0, 1)}}


def graph_replay_leg(step, steps=50):
    """The same hot-path step captured ONCE into a HIP graph (forward and backward of the 9 units:
    ~40 launches) and replayed: what the launch-bound loop costs without the Python / autograd
    enqueue time of every step (the eager figure is host-bound on a slow or busy host).  Extra
    information only; `value` and `roofline` always come from the eager, event-timed region.
    The tie-break noise key is baked into the captured launch, so every replay draws the same
    noise -- a benchmark device, the trainer does not run under graphs."""
    try:
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
        return {"value": round(step.images_per_step * steps / dt, 1), "unit": "images/sec",
                "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
                "note": "one HIP graph replay per step (9 units fwd+bwd captured once); extra to the eager figure"}
    except Exception as e:      # noqa: BLE001 -- an optional leg must never take the bench line down
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def hotpath_leg(args, rank, dev, nat, steps=20, warmup=5):
    """The hot path alone on the GPU (9 units fwd+bwd per batch, inputs resident in HBM): the
    quantity the two cpu_baseline legs measure, in the same unit."""
    step = HotPathStep(args, rank, dev)
    for _ in range(warmup):
        step()
    # the loop's own time first, without instrumentation (an event pair around every unit launch is two extra
    # packets per launch on the stream); then the same steps again with the events, for the kernel's launch time
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
    fb_ms, fb_n = nat.profile_read(nat.PROF_UNIT_FWDBWD)
    kernels, dom = kernel_rooflines(args, 0.0, 0, 0.0, 0, fb_ms, fb_n, nat.profile_read_work(nat.PROF_UNIT_FWDBWD),
                                    launches_hint=step)
    return {"value": round(step.images_per_step * steps / dt, 1), "unit": "images/sec",
            "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "workload": f"hot path only: {UNITS_PER_STEP} units fwd+bwd, batch {args.batch}, "
                        f"{args.width}x{args.height}, {args.disp} disparity",
            "roofline": dom, "hip_graph_replay": graph_replay_leg(step)}


def graph_step_leg(args, steps=20, timeout_s=150):
    """The same optimisation step with its device work captured into ONE HIP graph
    (mono-vifi_amd/trainer.py:_StepGraph) and replayed: extra to the eager, event-timed headline.  Runs as a
    CHILD process (this script with --hip-graph): a GPU fault or a hang during a replay cannot be caught and
    must not take the headline line down."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", "train", "--hip-graph",
           "--hip-graph-scope", args.hip_graph_scope, "--steps", str(steps), "--warmup", "6", "--no-cpu-baseline",
           "--no-hotpath-leg", "--also-configs", "none", "--no-graph-leg", "--no-host-leg", "--no-mfma-leg", "--batch",
           str(args.batch), "--height", str(args.height), "--width", str(args.width), "--backbone", args.backbone]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"child rc {r.returncode}: " + (r.stderr.strip().splitlines() or [""])[-1][:200]}
        d = json.loads(lines[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "scope": args.hip_graph_scope, "leg_seconds": round(time.perf_counter() - t0, 1),
                "note": "the optimisation step replayed as one HIP graph per step (networks, 9 units, backward, "
                        "gradient exchange" + (", clipping, capturable AdamW" if args.hip_graph_scope == "step" else
                                               "; clipping + AdamW eager") +
                        "), measured in a child process with the HIP runtime's graph packet capture off "
                        "(DESIGN.md section 7); tie-break noise from torch.randn (graph-safe) instead of the "
                        "in-kernel generator"}
    except subprocess.TimeoutExpired:
        return {"error": f"child did not finish within {timeout_s} s (killed)"}
    except Exception as e:      # noqa: BLE001 -- an optional leg must never take the bench line down
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def host_leg(args, steps=10, timeout_s=150):
    """What the host costs when eight ranks share this box: the SAME training step in child
    processes pinned (sched_setaffinity) to the CPUs ONE rank would have with eight ranks on the CPUs this container
    is granted -- eager, and with the step replayed as a HIP graph.  A step that slows down under the pin is
    host-bound on the 8-GPU node; the graph step is the mitigation."""
    import subprocess
    quota, _ = cpu_quota()
    ncpu = max(1, quota // 8)
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    cpus = allowed[:ncpu]
    out = {"cpus_per_rank": ncpu, "pinned_to": cpus, "steps": steps}
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = str(ncpu)
    base = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", "train", "--steps", str(steps), "--warmup", "6",
            "--no-cpu-baseline", "--no-hotpath-leg", "--also-configs", "none", "--no-graph-leg", "--no-pmc-leg", "--no-mfma-leg",
            "--no-kernel-leg", "--no-host-leg", "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width),
            "--backbone", args.backbone]
    for name, extra in (("eager", []), ("hip_graph", ["--hip-graph", "--hip-graph-scope", args.hip_graph_scope])):
        t0 = time.perf_counter()
        try:
            r = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=timeout_s,
                               preexec_fn=lambda: os.sched_setaffinity(0, cpus))
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"child rc {r.returncode}: " + (r.stderr.strip().splitlines() or [""])[-1][:200]}
                continue
            d = json.loads(lines[-1])
            out[name] = {"value": d["value"], "ms_per_step": d["ms_per_step"],
                         "process_cpu_ms_per_step": (d.get("host") or {}).get("process_cpu_ms_per_step"),
                         "leg_seconds": round(time.perf_counter() - t0, 1)}
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"child did not finish within {timeout_s} s (killed)"}
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def pmc_leg(args, timeout_s=120):
    """HBM traffic and VALU counters of the unit kernel, measured NOW: `rocprofv3 --pmc` (counters only, one
    pass per TCC counter as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not
    fit one pass) around a short hot-path run of this script in a child process, mean per launch of
    k_unit_fb.  FETCH_SIZE is doubled (the guide's gfx950 note; calibrated here on k_disp_mean, whose
    5,898,240 B read 2,894 KB), WRITE_SIZE taken 1:1.  Returns None when rocprofv3 is not there or a pass fails
    (the line then quotes the committed summary, tagged `static_source`)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    passes = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
               "GRBM_GUI_ACTIVE"]]
    got, launches = {}, 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    t0 = time.perf_counter()
    for counters in passes:
        d = tempfile.mkdtemp(prefix="mvf_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--workload", "hotpath", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
               "--no-replay-leg", "--no-kernel-leg", "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width), "--noise", args.noise]
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
    # the child hot-path run issues the step's nine units as two launches (six + three units): mean units per launch
    per_launch = UNITS_PER_STEP / (3.0 if getattr(args, "no_merge_unit_groups", False) or getattr(args, "no_batch_units", False)
                                   else 2.0)
    if getattr(args, "no_batch_units", False):
        per_launch = 1.0
    px_launch = per_launch * args.batch * args.height * args.width
    quad = got["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0            # SQ counters tick in quad-cycles over 1,024 SIMDs
    return {"traffic": int(round(got["FETCH_SIZE"] * 1024 * 2 + got["WRITE_SIZE"] * 1024)),
            "valu": {"valu_busy": round(got["SQ_ACTIVE_INST_VALU"] / quad, 3),
                     "valu_instr_per_px": int(round(got["SQ_INSTS_VALU"] * 64.0 / px_launch)),
                     "wave_active": round(got["SQ_ACTIVE_INST_ANY"] / got["SQ_WAVE_CYCLES"], 3),
                     "wave_wait_memory_or_barrier": round(got["SQ_WAIT_ANY"] / got["SQ_WAVE_CYCLES"], 3),
                     "wave_wait_issue": round(got["SQ_WAIT_INST_ANY"] / got["SQ_WAVE_CYCLES"], 3)},
            "source": {"measured": "in this run", "how": "rocprofv3 --pmc (three counter-only passes) around child runs of "
                       "`bench.py --workload hotpath` (3 units per launch, same shapes); FETCH_SIZE x 2 (gfx950 note of "
                       "MI355X_MICROARCH.md) + WRITE_SIZE; mean over " + str(launches) + " launches of k_unit_fb",
                       "leg_seconds": round(time.perf_counter() - t0, 1)}}


OTHER_CONFIGS = {
    # BASELINE.json configs[2..4] at their per-GPU shapes (reference: configs/dhrnet/DHRNet_KITTI_MR.txt,
    # configs/litemono/LiteMono_KITTI_HR.txt, configs/dhrnet/DHRNet_CS.txt)
    "C3": dict(backbone="DHRNet", batch=12, height=192, width=640),
    "C4": dict(backbone="LiteMono", batch=8, height=320, width=1024),
    "C5": dict(backbone="DHRNet", batch=12, height=192, width=512),
}


HOST_CPU = {}      # CPU time of the last timed region (this rank): process (all threads) and enqueueing thread


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


def other_config_leg(args, name, rank, world, dev, nat):
    """One of BASELINE.json's other training configurations for a few steps (same step function,
    same timing discipline as the headline; fewer steps so the default run stays within minutes)."""
    import copy
    import gc
    from mono_vifi_amd.bench_train import TrainStep
    t_leg = time.perf_counter()
    try:
        a = copy.copy(args)
        for k, v in OTHER_CONFIGS[name].items():
            setattr(a, k, v)
        step = TrainStep(a, rank, world, dev)
        for _ in range(args.also_warmup):
            step()
        nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
        nat.check(nat.lib().mvf_profile_enable(1), "profile_enable")
        elapsed = timed_steps(step, args.also_steps, world)
        nat.lib().mvf_profile_enable(0)
        fb_ms, fb_n = nat.profile_read(nat.PROF_UNIT_FWDBWD)
        _, dom = kernel_rooflines(a, 0.0, 0, 0.0, 0, fb_ms, fb_n, nat.profile_read_work(nat.PROF_UNIT_FWDBWD),
                                  static=False, launches_hint=step)
        out = {"workload": step.describe(), "value": round(a.batch * world * args.also_steps / elapsed, 2),
               "unit": "images/sec", "ms_per_step": round(elapsed / args.also_steps * 1e3, 3),
               "host_process_cpu_ms_per_step": HOST_CPU.get("process_cpu_ms_per_step"),
               "host_main_thread_cpu_ms_per_step": HOST_CPU.get("main_thread_cpu_ms_per_step"),
               "steps": args.also_steps, "warmup": args.also_warmup,
               "unit_launch_avg_us": dom and dom["avg_us"], "us_per_unit": dom and dom["us_per_unit"],
               "frac": dom and dom["frac"], "unit_launches": dom and dom["launches"],
               "images_per_unit_launch": dom and dom.get("images_per_launch")}
        del step
        gc.collect()
        torch.cuda.empty_cache()
    except Exception as e:      # noqa: BLE001 -- an extra leg must never take the headline line down
        out = {"error": f"{type(e).__name__}: {e}"[:300]}
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    return out


T_START = time.perf_counter()


def over_budget(args, need_s):
    """True when an optional leg estimated at `need_s` seconds would end after the time budget."""
    return (time.perf_counter() - T_START) + need_s > args.time_budget


def main():
    args = parse()
    if args.hip_graph:
        mono_vifi_amd.ensure_graph_replay_env()       # before the first HIP call of this process and of its ranks
    launch_ranks_if_needed(args)
    if args.workload != "mock":
        mono_vifi_amd.use_shipped_miopen_db()         # per rank: after the launcher re-executed, before any convolution
    world, rank, dev, backend = dist_setup(args)
    from mono_vifi_amd import parallel

    workload = args.workload
    if workload == "auto":
        workload = "train" if os.path.exists(os.path.join(ROOT, "mono-vifi_amd", "trainer.py")) \
            else "hotpath"
    nat = None
    if workload != "mock":
        from mono_vifi_amd import _native as nat
        nat.lib()   # fail loudly if the HIP library is missing
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
        nat.check(nat.lib().mvf_profile_reset(), "profile_reset")
        nat.check(nat.lib().mvf_profile_enable(1), "profile_enable")
    parallel.reset_comm_counts()
    red0 = getattr(getattr(step, "trainer", step), "reducer", None)
    if red0 is not None and (world > 1 or args.force_collectives):
        red0.timeline = True          # two event records per bucket and step
    elapsed = timed_steps(step, args.steps, world)
    host_cpu = dict(HOST_CPU)
    counts = {k: round(v / args.steps, 2) for k, v in sorted(parallel.comm_counts().items())}
    if nat:
        nat.lib().mvf_profile_enable(0)

    kernels, dominant = {}, None
    if nat:
        fwd_ms, fwd_n = nat.profile_read(nat.PROF_UNIT_FWD)
        bwd_ms, bwd_n = nat.profile_read(nat.PROF_UNIT_BWD)
        fb_ms, fb_n = nat.profile_read(nat.PROF_UNIT_FWDBWD)
        kernels, dominant = kernel_rooflines(args, fwd_ms, fwd_n, bwd_ms, bwd_n, fb_ms, fb_n,
                                             nat.profile_read_work(nat.PROF_UNIT_FWDBWD), launches_hint=step)
        if dominant and dominant.get("kernel") == "k_unit_fb<2>":
            st_ = getattr(step, "trainer", step)
            noise_tensor = not getattr(getattr(st_, "opt", None), "inkernel_noise", args.noise == "kernel") \
                if hasattr(st_, "opt") else args.noise != "kernel"
            types = unit_launch_types(args, nat, FB_BYTES_PER_PX, noise_tensor)
            if types:
                dominant.update(types)

    # ---- this build's own kernels either side of the unit kernel: event pairs around every launch for a few
    # extra steps (outside the timed region; one process only: the extra steps would need every rank)
    own_kernels = None
    if nat and args.kernel_leg and world == 1 and workload in ("train", "hotpath") and not args.hip_graph:
        own_kernels = glue_kernel_leg(step, nat)

    # ---- communication report (every rank takes part in its collectives)
    comm = comm_report(args, world, rank, dev, backend, step, counts)
    red = getattr(getattr(step, "trainer", step), "reducer", None)
    if comm is not None and red is not None and args.comm_leg_steps > 0 and not args.hip_graph:
        # the same step with the other issue order of the gradient exchange
        red.overlap = not red.overlap
        for _ in range(2):
            step()
        t_other = timed_steps(step, args.comm_leg_steps, world)
        red.overlap = not red.overlap
        key = "overlapped" if args.no_overlap else "no_overlap"
        comm[key] = {"ms_per_step": round(t_other / args.comm_leg_steps * 1e3, 4),
                     "steps": args.comm_leg_steps,
                     "note": "gradient buckets reduced after backward" if key == "no_overlap"
                             else "gradient buckets reduced from hooks during backward"}
        comm["timed_region_ms_per_step"] = round(elapsed / args.steps * 1e3, 4)

    # like-for-like figure for the CPU baselines: the hot path alone on the GPU, same units
    hotpath_only = None
    if workload == "train" and rank == 0 and world == 1 and not args.no_hotpath_leg:
        hotpath_only = hotpath_leg(args, rank, dev, nat) if not over_budget(args, 30) else \
            {"skipped": "time budget"}

    graph_leg = None
    if args.hip_graph and workload == "train" and dominant is None and hotpath_only and hotpath_only.get("roofline"):
        dominant = dict(hotpath_only["roofline"])
        dominant["note"] = "measured in the hot-path-only leg of this run (HIP events are not recorded inside a graph replay)"

    # ---- BASELINE.json configs 3-5, a few steps each (N = 1 headline run only)
    also = args.also_configs
    default_headline = (workload == "train" and world == 1 and args.backbone == "ResNet18" and
                        (args.batch, args.height, args.width) == (12, 192, 640) and not args.hip_graph)
    if also == "auto":
        names = list(OTHER_CONFIGS) if default_headline else []
    elif also in ("none", "off", ""):
        names = []
    else:
        names = [n.strip().upper() for n in also.split(",") if n.strip()]
    other = None
    if names and workload == "train":
        describe_headline = step.describe()
        images_headline = step.images_per_step
        del step            # free the headline trainer's activations and MIOpen workspaces
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        other = {}
        for n in names:
            if n not in OTHER_CONFIGS:
                continue
            # 10-15 s with the shipped MIOpen find-db, ~2.5 min when its solver search runs cold
            last = max([v.get("leg_seconds", 0) for v in other.values()] + [20.0])
            other[n] = other_config_leg(args, n, rank, world, dev, nat) if not over_budget(args, 1.2 * last) \
                else {"skipped": "time budget"}
        # the same configurations pinned to the CPUs one of eight ranks would get: BASELINE's
        # 8-GPU configurations are exactly these, and their steps cost the host 1.6-1.8 CPUs unpinned
        if rank == 0 and world == 1 and args.host_leg:
            import copy
            torch.cuda.empty_cache()
            for n in names:
                if n not in OTHER_CONFIGS or "ms_per_step" not in other.get(n, {}):
                    continue
                if over_budget(args, 40):
                    other[n]["host_pinned"] = {"skipped": "time budget"}
                    continue
                a = copy.copy(args)
                for k, v in OTHER_CONFIGS[n].items():
                    setattr(a, k, v)
                hp = host_leg(a, steps=args.also_steps)
                other[n]["host_pinned"] = hp
                e_ = (hp or {}).get("eager", {})
                if e_.get("ms_per_step"):
                    other[n]["pinned_eager_over_unpinned"] = round(e_["ms_per_step"] / other[n]["ms_per_step"], 3)
                    other[n]["host_bound_with_8_ranks"] = bool(e_["ms_per_step"] > 1.05 * other[n]["ms_per_step"])

        class _Done:        # the headline's description outlives its trainer
            images_per_step = images_headline

            @staticmethod
            def describe():
                return describe_headline
        step = _Done()

    # ---- the step under a HIP graph: last GPU leg, in a child process (N = 1 headline run only)
    if default_headline and rank == 0 and args.graph_leg:
        if hasattr(step, "trainer"):
            del step.trainer        # the eager trainer's activations are not needed any more
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        graph_leg = graph_step_leg(args) if not over_budget(args, 60) else {"skipped": "time budget"}

    # ---- live PMC passes of the unit kernel (child processes under rocprofv3; N = 1 headline run only)
    host_pinned = None
    if default_headline and rank == 0 and args.host_leg:
        host_pinned = host_leg(args) if not over_budget(args, 60) else {"skipped": "time budget"}

    conv_mfma = None
    if default_headline and rank == 0 and args.mfma_leg:
        conv_mfma = mfma_leg(args) if not over_budget(args, 150) else {"skipped": "time budget"}

    if default_headline and rank == 0 and args.pmc_leg and dominant and dominant.get("kernel") == "k_unit_fb<2>":
        live = pmc_leg(args) if not over_budget(args, 120) else None
        if live:
            dominant["traffic"], dominant["valu"] = live["traffic"], live["valu"]
            dominant["static_source"] = None
            dominant["pmc_source"] = live["source"]
            for k in ("valu_busy", "valu_instr_per_px", "wave_active", "wave_wait_memory_or_barrier", "wave_wait_issue"):
                if isinstance(live.get("valu"), dict) and k in live["valu"]:
                    dominant[k] = live["valu"][k]
            dominant["traffic_over_algorithmic"] = round(live["traffic"] / dominant["algorithmic_bytes_per_launch"], 3)
            dominant["traffic_over_algorithmic_plus_handover"] = round(
                live["traffic"] / (dominant["algorithmic_bytes_per_launch"] + dominant.get("handover_bytes_per_launch", 0)), 3)

    if rank == 0:
        n_gpus = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        assert n_gpus == args.gpus == world
        images = step.images_per_step * world * args.steps
        metric = {"train": "training images/sec (640x192, 3-frame)",
                  "mock": "mock plumbing steps/sec x batch (NOT a benchmark)"}.get(
            workload, "hot-path images/sec (9 view-synthesis + photometric-loss units "
                      "fwd+bwd per batch, 640x192, 3-frame)")
        out = {
            "metric": metric,
            "value": round(images / elapsed, 2), "unit": "images/sec", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}: " + (getattr(step, "describe", lambda: None)() or (
                f"{UNITS_PER_STEP} units fwd+bwd ({'2 launches of 6 + 3 units' if not getattr(args, 'no_merge_unit_groups', False) else '3 launches of 3 units'}, identity maps handed from the "
                f"single-frame to the multi-frame units), batch {args.batch}/GPU, "
                f"{args.width}x{args.height}, 2 sources/unit, exact mode, {args.disp} disparity")),
                "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "roofline": dominant,
            "kernels": kernels,
        }
        if workload != "mock":
            quota, host_threads = cpu_quota()
            # what a step costs the HOST: CPU time of this process per eager step (the Python
            # thread that enqueues the forward + the autograd thread that enqueues the backward), next to the step time
            # and to the CPUs a rank would have with eight ranks on this box
            out["host"] = dict(host_cpu, cpus_granted=quota, hardware_threads=host_threads,
                               host_cpu_over_step=round(host_cpu.get("process_cpu_ms_per_step", 0.0) /
                                                        (elapsed / args.steps * 1e3), 3),
                               cpus_per_rank_with_8_ranks=round(quota / 8.0, 2),
                               note="process_cpu = CPU time of all threads of this rank per timed step (enqueue only, before "
                                    "the closing synchronisation); a step is host-bound on a node where this exceeds "
                                    "ms_per_step x the CPUs a rank gets")
        if host_pinned and "host" in out:
            out["host"]["pinned"] = host_pinned
            e_, g_ = host_pinned.get("eager", {}), host_pinned.get("hip_graph", {})
            if e_.get("ms_per_step") and g_.get("ms_per_step"):
                out["host"]["host_bound_with_8_ranks"] = bool(e_["ms_per_step"] > 1.05 * out["ms_per_step"])
        if own_kernels:
            out["own_kernels"] = own_kernels
            # the hot path INSIDE the training step: its unit launches (HIP events of the timed region) plus the
            # preparing / finishing / gradient-scale launches around them (own-kernel leg) -- in the trainer the
            # disparity head supplies the mean partials and consumes the raw gradients, so this, not the stand-alone
            # loop of `hotpath_only` (which computes the means and scales the gradients itself), is what a step pays
            kk = own_kernels.get("kernels") or {}
            rf = out.get("roofline") or {}
            if rf.get("avg_us") and rf.get("launches") and out.get("steps"):
                unit_ms = rf["avg_us"] * rf["launches"] / out["steps"] / 1e3
                around = sum((kk.get(k) or {}).get("ms_per_step", 0.0) for k in ("k_units_finish", "k_fb_scale", "k_disp_mean"))
                out["hotpath_in_step_unit_launches_ms"] = round(unit_ms, 4)
                out["hotpath_in_step_ms"] = round(unit_ms + around, 4)
                out["hotpath_in_step_over_unit_launches"] = round((unit_ms + around) / unit_ms, 3)
        if conv_mfma:
            out["conv_mfma"] = conv_mfma
        if comm:
            out["comm"] = comm
        if hotpath_only:
            out["hotpath_only"] = hotpath_only
            # driver-visible scalars of the hot path alone: ms per step of the 9 units forward +
            # backward, eager, and its ratio to the sum of the unit launches inside it
            if hotpath_only.get("ms_per_step"):
                out["hotpath_ms_per_step"] = hotpath_only["ms_per_step"]
                rf = hotpath_only.get("roofline") or {}
                if rf.get("avg_us") and rf.get("launches") and hotpath_only.get("steps"):
                    unit_ms = rf["avg_us"] * rf["launches"] / hotpath_only["steps"] / 1e3
                    out["hotpath_unit_launches_ms_per_step"] = round(unit_ms, 4)
                    out["hotpath_over_unit_launches"] = round(hotpath_only["ms_per_step"] / unit_ms, 3)
                gr_ = hotpath_only.get("hip_graph_replay") or {}
                if gr_.get("ms_per_step"):
                    out["hotpath_graph_replay_ms_per_step"] = gr_["ms_per_step"]
        if graph_leg:
            out["hip_graph_step"] = graph_leg
        if other:
            out["other_configs"] = other
        if workload == "hotpath" and world == 1 and args.replay_leg:
            out["hip_graph_replay"] = graph_replay_leg(step)
        if not args.no_cpu_baseline and world == 1 and workload != "mock":
            # the required baseline always runs (bounded: --cpu-seconds + a thread sweep); the second one
            # only inside the time budget
            out["cpu_baseline"] = cpu_baseline(args)
            out["cpu_baseline_unfused"] = cpu_baseline_unfused(args) if not over_budget(args, 3 * args.cpu_seconds) \
                else {"skipped": "time budget"}
            gpu_hp = (hotpath_only or {}).get("value") if hotpath_only else (out["value"] if workload == "hotpath" else None)
            gr = (hotpath_only or out).get("hip_graph_replay") or {}
            out["like_for_like"] = {
                "unit": "images/sec on the hot-path part of a step (9 units fwd+bwd)",
                "gpu_hotpath_only": gpu_hp,
                "gpu_hotpath_only_hip_graph": gr.get("value"),
                "cpu_port_openmp": out["cpu_baseline"]["value"],
                "cpu_unfused_torch_ops": out["cpu_baseline_unfused"].get("value")}
        out["wall_seconds"] = round(time.perf_counter() - T_START, 1)
        line = json.dumps(out)
    else:
        line = None
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if line is not None:
        # the ONE JSON line is the last thing on stdout: RCCL writes its version banner through C stdio, which
        # is block-buffered on a pipe and would otherwise land after the line when the process exits
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
