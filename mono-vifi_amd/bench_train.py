"""Whole-optimisation-step workload for bench.py (``--workload train``).

A step = ``Trainer.optimisation_step`` on one synthetic KITTI-shaped batch resident in HBM:
3 IFRNet-L teacher passes, 6 PoseNet passes, 8 depth-encoder + 6+3 decoder passes, 3 fusion
passes, the 9 fused hot-path units, backward through all of it with the bucketed RCCL
all-reduce overlapped, grad-norm clipping and AdamW (reference: train.py:654-669, 698-886;
BASELINE.json configs[1]: ResNet18, 640x192, batch 12, use_affine, shared_encoder).
Weights are random-init (no checkpoints / ImageNet weights without network); data loading
is excluded (device-resident batch) and stated in the JSON."""
import tempfile

import numpy as np
import torch

from . import synthetic
from .options import default_options
from .trainer import Trainer


class TrainStep:
    def __init__(self, args, rank, world, dev):
        opts = default_options(
            batch_size=args.batch, height=args.height, width=args.width, backbone=args.backbone,
            use_affine=True, fuse_model_type="shared_encoder", world_size=world, global_rank=rank,
            local_rank=dev.index or 0, log_dir=tempfile.mkdtemp(prefix="mvf_bench_"),
            exp_name=f"bench_r{rank}", num_workers=0, synthetic_len=max(4096, args.batch * world * 4),
            log_frequency=10 ** 9, save_frequency=10 ** 9, learning_rate=1e-4,
            amp_bf16=getattr(args, "amp_bf16", False), channels_last=getattr(args, "channels_last", False),
            inkernel_noise=getattr(args, "noise", "kernel") == "kernel",
            hip_graph=bool(getattr(args, "hip_graph", False)),
            hip_graph_scope=getattr(args, "hip_graph_scope", "step"),
            batch_units=not getattr(args, "no_batch_units", False),
            share_identity=not getattr(args, "no_share_identity", False),
            merge_unit_groups=not getattr(args, "no_merge_unit_groups", False),
            regroup=not getattr(args, "no_regroup", False),
            grad_exchange=getattr(args, "grad_exchange", "all_reduce"),
            no_overlap=bool(getattr(args, "no_overlap", False)),
            force_collectives=bool(getattr(args, "force_collectives", False)))
        self.trainer = Trainer(opts)
        self.trainer.set_train()
        b = synthetic.training_batch(1234 + 7919 * rank, args.batch, args.height, args.width)
        self.batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in b.items()}
        self.images_per_step = args.batch
        self.opts = opts

    def describe(self):
        o = self.opts
        return (f"full optimisation step, {o.backbone} {o.width}x{o.height} 3-frame, batch "
                f"{o.batch_size}/GPU, use_affine, {o.fuse_model_type}, IFRNet-L teacher, 9 fused "
                f"hot-path units (forward+backward tile kernel, "
                f"{('2 launches of 6 + 3 units' if o.merge_unit_groups else '3 launches of 3 units') if o.batch_units else '9 launches'}), "
                f"{'grouped' if o.group_calls else 'one-at-a-time'} network calls with per-call "
                f"BatchNorm statistics, AdamW, random-init weights, device-resident synthetic batch "
                f"(data loading excluded), nets {'bf16 autocast' if o.amp_bf16 else 'fp32'}"
                f"{(', forward + backward + gradient exchange replayed as one HIP graph, clipping + AdamW eager' if o.hip_graph_scope != 'step' else ', device work of the whole step replayed as one HIP graph') if o.hip_graph else ''}")

    def describe_short(self):
        """The same in at most 200 characters (the bench line's config.workload)."""
        o = self.opts
        units = ("6+3 units per launch" if o.merge_unit_groups else "3 units per launch") if o.batch_units else "1 unit per launch"
        return (f"optimisation step (train.py:640-696), {o.backbone} {o.width}x{o.height} 3-frame, batch {o.batch_size}/GPU, "
                f"use_affine, {o.fuse_model_type}, 9 fused units ({units}), AdamW, "
                f"{'bf16 autocast' if o.amp_bf16 else 'fp32'}{', HIP graph' if o.hip_graph else ''}, device-resident batch")

    def __call__(self):
        return self.trainer.optimisation_step(dict(self.batch))
