"""Child process of tests/test_trainer_gpu.py::test_hip_graph_step_follows_the_eager_step: an eager trainer
and a --hip_graph trainer fed the same batches and tie-break noise; prints one JSON line.  A process of its
own because a GPU memory fault during a graph replay kills the process and cannot be caught."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    backbone, B, H, W, log_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    scope = sys.argv[6] if len(sys.argv) > 6 else "step"
    collectives = len(sys.argv) > 7 and sys.argv[7] == "collectives"
    import mono_vifi_amd
    from mono_vifi_amd import synthetic
    from mono_vifi_amd.options import default_options
    # an entry point that will ask for --hip_graph: packet capture off BEFORE this process's first HIP call
    # (importing the package no longer sets it; the batches below already touch the GPU)
    mono_vifi_amd.ensure_graph_replay_env()
    from mono_vifi_amd.trainer import Trainer, _StepGraph
    dev = torch.device("cuda", 0)
    if collectives:
        # the data-parallel exchanges of a step (bucketed gradient all-reduce from the backward hooks, SyncBatchNorm
        # statistics) through RCCL in a group of one -- captured INTO the graph together with everything else
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 1000), RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0, device_id=dev)
    steps = 7
    batches = []
    for i in range(steps):
        b = synthetic.training_batch(5 + i, B, H, W)
        batches.append({k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in b.items()})
    g = torch.Generator(device=dev).manual_seed(3)
    noise = torch.randn((B, 2, H, W), device=dev, generator=g)
    out = {}
    for graph in (False, True):
        opts = default_options(batch_size=B, height=H, width=W, backbone=backbone, use_affine=True, num_workers=0,
                               synthetic_len=4 * B, log_dir=os.path.join(log_dir, "graph" if graph else "eager"),
                               exp_name="t", log_frequency=10 ** 9, save_frequency=10 ** 9, hip_graph=graph,
                               hip_graph_scope=scope, inkernel_noise=False, lr_sche_type="cos", learning_rate=1e-3,
                               force_collectives=collectives)
        t = Trainer(opts)
        t.set_train()
        t.tie_break_noise = noise
        assert (t._step_graph is not None) == graph
        losses, snap = [], None
        for i, b in enumerate(batches):
            if i == _StepGraph.WARMUP:
                snap = [p.detach().clone() for p in t.parameters_to_train]
            l = t.optimisation_step(dict(b))
            losses.append([float(l[k]) for k in ("loss", "loss_base", "loss_dc")])
        torch.cuda.synchronize()
        delta = torch.cat([(p.detach() - q).flatten() for p, q in zip(t.parameters_to_train, snap)])
        lr = t.model_optimizer.param_groups[0]["lr"]
        out["graph" if graph else "eager"] = dict(
            losses=losses, delta_norm=float(delta.norm()), lr=float(lr), lr_is_tensor=bool(torch.is_tensor(lr)),
            captured=bool(graph and t._step_graph.graph is not None),
            calls=int(t._step_graph.calls) if graph else 0,
            buckets=int(t.reducer.num_buckets), forced=bool(getattr(t.reducer, "always_reduce", False)))
        if graph:
            # a graph-mode checkpoint resumes in an eager trainer
            t.save_model(batch_idx=1)
            o2 = default_options(batch_size=B, height=H, width=W, backbone=backbone, use_affine=True, num_workers=0,
                                 synthetic_len=4 * B, log_dir=os.path.join(log_dir, "graph"), exp_name="t",
                                 log_frequency=10 ** 9, save_frequency=10 ** 9, resume=True, lr_sche_type="cos",
                                 learning_rate=1e-3)
            t2 = Trainer(o2)
            out["resume_equal"] = all(torch.equal(a.detach().cpu(), b.detach().cpu())
                                      for a, b in zip(t.parameters_to_train, t2.parameters_to_train))
            out["resume_lr_is_float"] = isinstance(t2.model_optimizer.param_groups[0]["lr"], float)
            del t2
        del t
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
