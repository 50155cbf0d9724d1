"""Drop-in ``Trainer`` for Mono-ViFI on MI355X.

Keeps the reference's ``Trainer`` surface (reference: train.py:27-1176) -- constructor
from an options namespace, ``train() / run_epoch() / process_batch() / predict_poses() /
generate_images_pred() / compute_reprojection_loss() / compute_losses_base() /
compute_SI_log_depth_loss() / compute_depth_consistency_loss_affine() / affine_transform()
/ save_model() / load_ckpt() / load_pretrained_model()`` and the checkpoint file format --
while the view-synthesis + photometric hot path runs in the hand-written gfx950 kernels
(``losses.HotPathLosses``) and the data-parallel exchange is one bucketed RCCL all-reduce
overlapped with backward (``parallel.BucketedGradReducer``).

Known reference quirks that are fixed rather than replicated:
``checkpoint`` is initialised to ``None``; the aliased ``encoder_mf`` parameters are
trained once; the dead ImageNet ``fc`` head is gone; the per-sample Python loops with five
``.item()`` host syncs each in the affine branch (train.py:891-896, 906-911) are batched
tensor ops; the per-step ``dist.barrier()`` (train.py:692-693) is dropped.  The per-epoch
accuracy evaluation (train.py:291-301) needs KITTI ground truth and is skipped.
"""
from __future__ import annotations

import copy
import json
import logging
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.data import DataLoader

from . import datasets, parallel
from .layers import SSIM, BackprojectDepth, Project3D, disp_to_depth, transformation_from_parameters
from .losses import HotPathLosses
from ._native import MAX_UNITS
from .networks import FusionModule, IFRNet, dhrnet, grouped, litemono, monodepth2, posenet


def setup_logging(log_file=None, filemode="w", rank=0):
    level = logging.INFO if rank == 0 else logging.WARNING
    handlers = [logging.StreamHandler()]
    if log_file and rank == 0:
        handlers.append(logging.FileHandler(log_file, mode=filemode))
    logging.basicConfig(level=level, format="%(asctime)s %(message)s", handlers=handlers, force=True)


def sec_to_hm_str(t):
    t = int(t)
    return "{:02d}h{:02d}m{:02d}s".format(t // 3600, (t % 3600) // 60, t % 60)


def split_g(t, groups):
    return grouped.split_groups(t, groups)


class _GroupedPyramids(list):
    """Per input, the feature pyramid of a grouped encoder call (views), plus what they are views of: `bases`
    [level] = the interleaved batch of all `G` inputs (`Trainer._regroup` reads those directly)."""
    bases = None
    G = 0


class Trainer(HotPathLosses):
    def __init__(self, options):
        self.opt = options
        o = self.opt
        # Process-level settings of an entry point, before this process touches the GPU (importing the package
        # sets nothing): whole-step HIP graphs need the runtime's graph packet capture off (read at the first
        # HIP call -- `strict` raises when torch's GPU context already exists without it), and MIOpen gets a
        # per-process copy of the shipped find-db (never the tracked file, never one file for eight ranks).
        from . import ensure_graph_replay_env, use_shipped_miopen_db
        if bool(getattr(o, "hip_graph", False)):
            ensure_graph_replay_env(strict=True)      # (no torch.cuda call before this line: any HIP call fixes the flag)
        use_shipped_miopen_db()
        self.log_path = os.path.join(o.log_dir, o.exp_name)
        if o.global_rank == 0:
            os.makedirs(self.log_path, exist_ok=True)
            self.save_opts()
            resume_log = os.path.exists(os.path.join(self.log_path, "ckpt.pth"))
            setup_logging(os.path.join(self.log_path, "logger.log"), "a" if resume_log else "w",
                          rank=o.global_rank)
            logging.info("Experiment is named: %s", o.exp_name)
            logging.info("GPU numbers: %d", o.world_size)
        else:
            setup_logging(rank=o.global_rank)

        self.device = torch.device("cuda", o.local_rank % torch.cuda.device_count()) \
            if torch.cuda.is_available() else torch.device("cpu")
        if self.device.type == "cuda":
            # reference: train.py sets torch.cuda.set_device(opts.local_rank) before building the
            # Trainer.  Everything that reads "the current device" -- torch's current stream (on
            # which the HIP kernels are enqueued), torch.cuda.synchronize(), RCCL's default
            # device -- must resolve to this rank's GPU, not to cuda:0.
            torch.cuda.set_device(self.device)
        if o.seed > 0:
            self.set_seed(o.seed)

        assert o.height % 32 == 0, "'height' must be a multiple of 32"
        assert o.width % 32 == 0, "'width' must be a multiple of 32"
        assert o.frame_ids[0] == 0, "frame_ids must start with 0"

        self.models = {}
        self.num_input_frames = len(o.frame_ids)
        self.num_pose_frames = 2
        self.ep_start = 0
        self.batch_start = 0
        self.step = 0
        self.epoch = 0
        self.use_pose_net = not (o.use_stereo and o.frame_ids == [0])

        # ---- data (synthetic: no datasets on either box)
        if not o.synthetic:
            raise NotImplementedError(
                "only --synthetic True is available: the KITTI/Cityscapes loaders of the "
                "reference need data and libraries neither box has (DESIGN.md section 9)")
        train_dataset = datasets.SyntheticTripletDataset(o.height, o.width, o.synthetic_len,
                                                         o.use_affine, o.seed,
                                                         device_augment=o.device_augment and
                                                         torch.cuda.is_available())
        self.num_steps_per_epoch = len(train_dataset) // o.world_size // o.batch_size
        self.num_total_steps = self.num_steps_per_epoch * o.num_epochs
        if o.world_size > 1:
            self.sampler = datasets.CustomDistributedSampler(train_dataset, o.seed, o.world_size,
                                                             o.global_rank)
        else:
            self.sampler = datasets.CustomSampler(train_dataset, o.seed)
        self.train_loader = DataLoader(train_dataset, o.batch_size, shuffle=False,
                                       sampler=self.sampler, num_workers=o.num_workers,
                                       pin_memory=torch.cuda.is_available(), drop_last=True)

        # ---- models (reference: train.py:140-190)
        if o.backbone in ("ResNet18", "ResNet50"):
            layers_n = 18 if o.backbone == "ResNet18" else 50
            self.models["encoder"] = monodepth2.DepthEncoder(layers_n, o.weights_init == "pretrained")
            self.models["depth"] = monodepth2.DepthDecoder(self.models["encoder"].num_ch_enc,
                                                           range(o.num_scales))
        elif o.backbone == "DHRNet":
            self.models["encoder"] = dhrnet.DepthEncoder(18, o.weights_init == "pretrained")
            self.models["depth"] = dhrnet.DepthDecoder(self.models["encoder"].num_ch_enc,
                                                       range(o.num_scales))
        elif o.backbone == "LiteMono":
            # reference: train.py:157-167 (lite-mono-pretrain.pth cannot be fetched: random init)
            self.models["encoder"] = litemono.DepthEncoder(model="lite-mono", drop_path_rate=0.2,
                                                           width=o.width, height=o.height)
            self.models["depth"] = litemono.DepthDecoder(self.models["encoder"].num_ch_enc,
                                                         range(o.num_scales))
        else:
            raise ValueError(f"unknown backbone {o.backbone}")
        if o.fuse_model_type == "shared_all":
            self.models["encoder_mf"] = self.models["encoder"]
            self.models["depth_mf"] = self.models["depth"]
        elif o.fuse_model_type == "shared_encoder":
            self.models["encoder_mf"] = self.models["encoder"]
            self.models["depth_mf"] = copy.deepcopy(self.models["depth"])
        else:
            self.models["encoder_mf"] = copy.deepcopy(self.models["encoder"])
            self.models["depth_mf"] = copy.deepcopy(self.models["depth"])
        self.models["fusion_module"] = FusionModule(o, self.models["encoder_mf"].num_ch_enc)
        if self.use_pose_net:
            self.models["pose_encoder"] = posenet.ResnetEncoder(
                o.num_layers, o.weights_init == "pretrained", num_input_images=self.num_pose_frames)
            self.models["pose"] = posenet.PoseDecoder(self.models["pose_encoder"].num_ch_enc,
                                                      num_input_features=1, num_frames_to_predict_for=2)

        if o.pretrained_path and (not o.resume or not os.path.exists(os.path.join(self.log_path, "ckpt.pth"))):
            self.load_pretrained_model()

        self._modules_unique = parallel.unique_modules(self.models)
        for m in self._modules_unique.values():
            m.to(self.device)
            if o.channels_last:
                m.to(memory_format=torch.channels_last)
        self.parameters_to_train = parallel.unique_parameters(self._modules_unique.values())

        checkpoint = self.load_ckpt() if o.resume else None

        if o.group_calls:
            # per-call BatchNorm statistics for interleaved grouped calls; with sync_bn one
            # collective per layer carries the statistics of every call
            for m in self._modules_unique.values():
                grouped.convert_grouped_batchnorm(m, sync=(o.world_size > 1 or o.force_collectives) and o.sync_bn,
                                                  force_sync=o.force_collectives)
        if o.world_size > 1:
            if o.sync_bn and self.device.type == "cuda" and not o.group_calls:
                for k in list(self._modules_unique):
                    conv = nn.SyncBatchNorm.convert_sync_batchnorm(self._modules_unique[k])
                    for name, m in self.models.items():
                        if m is self._modules_unique[k]:
                            self.models[name] = conv
                    self._modules_unique[k] = conv
                self.parameters_to_train = parallel.unique_parameters(self._modules_unique.values())
            parallel.broadcast_module_states(self._modules_unique.values(), src=0)

        # ---- frozen VFI teacher (reference: train.py:210-227)
        self.model_vfi_train = IFRNet(scale="large")
        self.model_vfi_test = IFRNet(scale="small")
        tag = {"kitti": "KITTI", "cityscapes": "CS"}.get(o.dataset)
        for net, sz in ((self.model_vfi_train, "L"), (self.model_vfi_test, "S")):
            path = os.path.join(o.vfi_weights_dir, f"IFRNet_{sz}_{tag}.pth") if tag else None
            if path and os.path.exists(path):
                net.load_state_dict(torch.load(path, map_location="cpu")["VFI"])
            else:
                logging.info("IFRNet_%s weights not found: random-init teacher (synthetic run)", sz)
            net.to(self.device).eval()
            for p in net.parameters():
                p.requires_grad_(False)
        if o.world_size > 1:
            parallel.broadcast_module_states([self.model_vfi_train, self.model_vfi_test], src=0)

        # ---- optimiser (reference: train.py:229-246)
        use_graph = bool(getattr(o, "hip_graph", False)) and self.device.type == "cuda"
        # scope "step" (default) also captures clipping + AdamW, which needs the capturable optimiser
        # (device-resident step counters and learning rate); scope "backward" leaves the update eager with
        # the ordinary optimiser and schedulers
        graph_opt = use_graph and getattr(o, "hip_graph_scope", "step") == "step"
        if graph_opt and o.optimizer not in ("adamw", "adam"):
            raise ValueError("--hip_graph_scope step needs a capturable optimizer (adamw / adam)")
        if use_graph:
            logging.warning("--hip_graph: the optimisation step is captured into a HIP graph and replayed (with the "
                            "HIP runtime's graph packet capture switched off, DESIGN.md section 7); a GPU memory "
                            "fault during a replay cannot be caught")
        # under a HIP graph the step counter and the learning rate live on the device: the
        # schedulers then update the rate in place (fill_) and the replayed launch reads it
        use_graph_sched = graph_opt
        # ONE multi-tensor kernel per chunk of parameters does the whole AdamW update in registers (torch's `fused`
        # implementation) instead of the ~10 passes of the default `foreach` form: same arithmetic, -0.5 ms per eager
        # ResNet18 step, -2.5 ms per graph step (the capturable foreach form is heavier still)
        fused_opt = (self.device.type == "cuda" and o.optimizer in ("adamw", "adam") and
                     getattr(o, "fused_optimizer", True) and os.environ.get("MVF_FUSED_ADAM", "1") != "0")
        gkw = {}
        if graph_opt:
            gkw["capturable"] = True
        if fused_opt:
            gkw["fused"] = True
        elif graph_opt:
            gkw["foreach"] = True
        lr0 = torch.tensor(float(o.learning_rate), device=self.device) if graph_opt else o.learning_rate
        if o.optimizer == "adamw":
            self.model_optimizer = torch.optim.AdamW(self.parameters_to_train, lr=lr0,
                                                     betas=(o.beta1, o.beta2), weight_decay=o.weight_decay,
                                                     **gkw)
        elif o.optimizer == "adam":
            self.model_optimizer = torch.optim.Adam(self.parameters_to_train, lr=lr0,
                                                    betas=(o.beta1, o.beta2), **gkw)
        else:
            self.model_optimizer = torch.optim.SGD(self.parameters_to_train, lr=o.learning_rate,
                                                   momentum=o.momentum)
        # graph mode: the schedule runs on a host-side shadow optimiser (plain float arithmetic) and
        # `_sched_step` pushes the new rate into the device-resident one -- a scheduler bound to a
        # tensor rate would read it back (.item()) on every step
        self._lr_shadow = torch.optim.SGD([torch.zeros(1)], lr=o.learning_rate) if use_graph_sched else None
        sched_opt = self._lr_shadow if use_graph_sched else self.model_optimizer
        if o.lr_sche_type == "cos":
            self.model_lr_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(
                sched_opt, T_max=max(self.num_total_steps, 1), eta_min=o.eta_min)
        else:
            self.model_lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(
                sched_opt, o.decay_step, o.decay_rate)
        if checkpoint:
            # Model weights interchange with the reference's ckpt.pth; optimiser state does not:
            # the reference's optimiser holds the aliased encoder_mf / depth_mf parameters twice
            # and the dead ImageNet fc head (train.py:198-200), this one holds each parameter
            # once.  A foreign optimiser state is therefore dropped (fresh moments, schedule
            # fast-forwarded to the checkpoint's step) instead of raising.
            # A foreign state is recognised by its parameter count, not by catching whatever
            # load_state_dict raises: a truncated / corrupt checkpoint of THIS trainer still fails loudly.
            saved = checkpoint.get("optimizer", {})
            n_saved = sum(len(g.get("params", ())) for g in saved.get("param_groups", ()))
            n_mine = sum(len(g["params"]) for g in self.model_optimizer.param_groups)
            if n_saved == n_mine:
                self.model_optimizer.load_state_dict(saved)
                self.model_lr_scheduler.load_state_dict(checkpoint["lr_scheduler"])
                if o.optimizer in ("adamw", "adam"):
                    # the saved groups carry the settings of the run that wrote them (the reference: foreach,
                    # host-side step counters; this trainer: fused / capturable): restore what THIS optimiser
                    # step needs -- a fused or captured update reads its step counters on the device
                    for g in self.model_optimizer.param_groups:
                        g["capturable"] = bool(graph_opt)
                        g["fused"] = True if fused_opt else None
                        g["foreach"] = None if fused_opt else (True if graph_opt else None)
                        if not graph_opt:
                            # a --hip_graph run stores the rate as a device tensor (its scheduler updates it inside
                            # the capture); the eager update wants plain floats (torch's foreach adam()
                            # rejects a tensor rate, the fused one would read it back every step)
                            for k in ("lr", "initial_lr"):
                                if torch.is_tensor(g.get(k)):
                                    g[k] = float(g[k])
                    for st in self.model_optimizer.state.values():
                        if "step" in st:
                            st["step"] = torch.as_tensor(st["step"], dtype=torch.float32).to(
                                self.device if (fused_opt or graph_opt) else "cpu")
                if self._lr_shadow is not None:
                    self._lr_shadow.param_groups[0]["lr"] = float(self.model_lr_scheduler.get_last_lr()[0])
            else:
                logging.warning("optimizer state of the checkpoint holds %d parameters, this trainer %d (the "
                                "reference's optimiser lists the aliased encoder_mf / depth_mf parameters twice "
                                "and the ImageNet fc head): restarting the optimizer moments, keeping weights, "
                                "epoch and step", n_saved, n_mine)
                sched = checkpoint.get("lr_scheduler", {})
                last = int(sched.get("last_epoch", 0)) if isinstance(sched, dict) else 0
                if o.lr_sche_type == "cos":
                    last = min(last, max(self.num_total_steps, 1))
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")     # scheduler.step() before optimizer.step()
                    for _ in range(last):
                        self.model_lr_scheduler.step()
            del checkpoint
        self._push_lr()

        self.reducer = parallel.BucketedGradReducer(self.parameters_to_train, o.world_size,
                                                    o.bucket_mb, always_reduce=o.force_collectives,
                                                    exchange=getattr(o, "grad_exchange", "all_reduce"),
                                                    overlap=not getattr(o, "no_overlap", False))
        self._step_graph = _StepGraph(self) if use_graph else None

        # ---- hot-path modules (reference: train.py:248-256)
        if not o.no_ssim:
            self.ssim = SSIM().to(self.device)
        self.backproject_depth = BackprojectDepth(o.batch_size, o.height, o.width).to(self.device)
        self.project_3d = Project3D(o.batch_size, o.height, o.width).to(self.device)

        logging.info("There are %d training items (synthetic)", len(train_dataset))
        if o.world_size > 1:
            dist.barrier()

    # ------------------------------------------------------------------ misc
    def set_seed(self, seed=1234):
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def set_train(self):
        for m in self.models.values():
            m.train()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    def save_opts(self):
        os.makedirs(os.path.join(self.log_path, "models"), exist_ok=True)
        with open(os.path.join(self.log_path, "models", "opt.json"), "w") as f:
            json.dump({k: v for k, v in vars(self.opt).items()}, f, indent=2, default=str)

    # ------------------------------------------------------------------ loop
    def train(self):
        """Run the entire training pipeline (reference: train.py:284-303)."""
        for self.epoch in range(self.ep_start, self.opt.num_epochs):
            self.run_epoch()
            if self.opt.lr_sche_type == "step":
                self._sched_step()
            logging.info("per-epoch accuracy evaluation skipped: needs KITTI ground truth")
            if self.opt.global_rank == 0:
                self.save_model(ep_end=True)

    def optimisation_step(self, inputs):
        """process_batch -> backward (bucketed all-reduce overlapped) -> clip -> optimiser
        step (reference: train.py:654-669).  With ``--hip_graph`` the device work of the whole
        step is captured once into a HIP graph and replayed (`_StepGraph`)."""
        if self._step_graph is not None:
            losses = self._step_graph(inputs)
        else:
            losses = self._device_step(inputs)
        if self.opt.lr_sche_type == "cos":
            self._sched_step()
        return losses

    def _sched_step(self):
        if self._lr_shadow is not None:
            self._lr_shadow.step()          # (marks the shadow as stepped: the scheduler's order check)
        self.model_lr_scheduler.step()
        self._push_lr()

    def _push_lr(self):
        """Graph mode: copy the shadow schedule's rate into the optimiser's device-resident one."""
        if self._lr_shadow is None:
            return
        lr = float(self._lr_shadow.param_groups[0]["lr"])
        for g in self.model_optimizer.param_groups:
            if not torch.is_tensor(g["lr"]):     # a loaded optimiser state carries a float rate
                g["lr"] = torch.tensor(lr, device=self.device)
            g["lr"].fill_(lr)

    def _forward_backward(self, inputs):
        """Networks, hot-path units, backward pass and gradient exchange (no host read-back)."""
        _, losses = self.process_batch(inputs)
        self.reducer.zero_grad()
        losses["loss"].backward()
        self.reducer.finish()
        return losses

    def _update(self):
        """Gradient clipping and the optimiser update (reference: train.py:661-667)."""
        if self.opt.clip_grad != -1:
            for group in self.model_optimizer.param_groups:
                nn.utils.clip_grad_norm_(group["params"], max_norm=self.opt.clip_grad)
        self.model_optimizer.step()

    def _device_step(self, inputs):
        """Everything of a step that runs on the device (no host read-back anywhere)."""
        losses = self._forward_backward(inputs)
        self._update()
        return losses

    def run_epoch(self, max_steps=None):
        logging.info("Training epoch %d\n", self.epoch)
        self.sampler.set_epoch(self.epoch)
        if hasattr(self.train_loader.dataset, "set_epoch"):
            self.train_loader.dataset.set_epoch(self.epoch)      # fresh augmentation draws per epoch
        self.sampler.set_start_iter(self.batch_start * self.opt.batch_size)
        self.set_train()
        if self.opt.world_size > 1:
            dist.barrier()
        t_data = time.time()
        for batch_idx, inputs in enumerate(self.train_loader):
            self.step += 1
            t_fp = time.time()
            losses = self.optimisation_step(inputs)
            idx = batch_idx + self.batch_start
            if idx % self.opt.log_frequency == 0:
                if self.opt.world_size > 1:
                    stacked = torch.stack([losses[k].detach().float() for k in sorted(losses)])
                    dist.all_reduce(stacked, op=dist.ReduceOp.SUM)
                    parallel.count_collective("loss_all_reduce")
                    stacked /= self.opt.world_size
                    for i, k in enumerate(sorted(losses)):
                        losses[k] = stacked[i]
                if self.opt.global_rank == 0:
                    if self.device.type == "cuda":
                        torch.cuda.synchronize()
                    self.log_time(idx, t_fp - t_data, time.time() - t_fp, float(losses["loss"]))
                    self.log_tensorboard("train", losses)
            if idx > 0 and idx % self.opt.save_frequency == 0 and self.opt.global_rank == 0:
                self.save_model(batch_idx=idx + 1)
            t_data = time.time()
            if max_steps is not None and batch_idx + 1 >= max_steps:
                break
        self.batch_start = 0

    # ------------------------------------------------------------------ one batch
    def _nets(self, fn):
        if self.opt.amp_bf16 and self.device.type == "cuda":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return fn()
        return fn()

    def _depth(self, decoder, feats):
        """decoder -> {("disp",0): fp32 disparity} (the hot path computes in fp32)."""
        m = self.models[decoder]
        if getattr(m, "takes_depth_range", False):
            # the disparity-head epilogue also emits depth and the per-image mean partials of disp
            out = self._nets(lambda: m(feats, self.opt.min_depth, self.opt.max_depth))
        else:
            out = self._nets(lambda: m(feats))
        # (("disp_head_sink", 0) is not a tensor: ops.HeadSink, passed through)
        return {k: (v.float() if torch.is_tensor(v) else v) for k, v in out.items()}

    def _encode(self, name, img):
        if self.opt.channels_last:
            img = img.contiguous(memory_format=torch.channels_last)
        return self._nets(lambda: self.models[name](img))

    def _unit(self, disp, tgt, poses, srcs, K, inv_K, mask_rec=None):
        """One hot-path unit (reference: e.g. train.py:747-749)."""
        if self.opt.fused_units:
            loss, _ = self.compute_unit(disp, tgt, poses, srcs, K, inv_K, mask_rec)
            return loss
        warped = [self.generate_images_pred(disp, poses[k], srcs[k], K, inv_K) for k in range(len(srcs))]
        loss, _ = self.compute_losses_base(disp, tgt, warped, srcs, mask_rec)
        return loss

    def _units(self, units, want_ident=False, sum_in=None):
        """Mutually independent hot-path units (reference: e.g. train.py:747-760): ONE launch of the
        forward+backward tile kernel for all of them.  units: dicts as for `compute_units`; `want_ident`: one flag
        or one per unit; `sum_in`: the running loss total the launch adds its units' losses to.
        Returns (sum_in + sum of their losses, identity maps per unit | None)."""
        if self.opt.fused_units:
            total, idents, _ = self.compute_units(units, want_ident=want_ident, want_sum=True, sum_in=sum_in)
            return total, idents
        total = sum_in
        for un in units:
            l = self._unit(un["disp_tgt"], un["img_tgt"], un["poses"], un["imgs_src"], un["K"], un["inv_K"],
                           un.get("mask_rec"))
            total = l if total is None else total + l
        return total, None

    def _affine_pose(self, pose, Rc, Rc_inv):
        """Pose of the affine-augmented view: [Rc R Rc^-1 | Rc t] (reference: train.py:820-823)."""
        out = torch.zeros_like(pose)
        out[:, :3, :3] = torch.matmul(Rc, torch.matmul(pose[:, :3, :3], Rc_inv))
        out[:, :3, 3:4] = torch.matmul(Rc, pose[:, :3, 3:4])
        return out

    # ---- the independent invocations of a step, one interleaved batch each (networks/grouped.py)
    def _regroup(self, enc, plans):
        """Per plan (a list of positions in the grouped encoder call `enc` came from), the interleaved batches of
        those inputs' feature pyramids: [plan][level] -- ONE launch per level for all plans (`ops.regroup`)
        instead of a `stack` per plan and level, and one adjoint launch instead of autograd's accumulation of
        the per-group gradients + the stack of `split_groups`' backward.  None when it does not apply."""
        bases = getattr(enc, "bases", None)
        if bases is None or not getattr(self.opt, "regroup", True):
            return None
        if any((not b.is_cuda) or b.dtype != torch.float32 for b in bases):
            return None
        from . import ops
        per_level = [ops.regroup(b, enc.G, plans) for b in bases]
        return [[lvl[k] for lvl in per_level] for k in range(len(plans))]

    def _merge_inputs(self, parts):
        """Input of a grouped call: per group, the tensors whose channel concatenation is that call's input ->
        the interleaved batch.  Images (no gradient) on the device go through one gather launch."""
        flat = [t for p in parts for t in p]
        if (getattr(self.opt, "regroup", True) and len(flat) <= 32 and
                all(t.is_cuda and t.dtype == torch.float32 and not t.requires_grad for t in flat)):
            from . import ops
            return ops.interleave_groups(parts)
        return grouped.merge_groups([p[0] if len(p) == 1 else torch.cat(p, 1) for p in parts])

    def _encode_many(self, name, imgs):
        """Encoder on G independent inputs -> per input, its feature pyramid.  Grouped: one call
        on the interleaved batch with per-call BatchNorm statistics."""
        if not self.opt.group_calls or len(imgs) == 1:
            return [self._encode(name, im) for im in imgs]
        G = len(imgs)
        with grouped.grouped(self.models[name], G):
            feats = self._encode(name, self._merge_inputs([[im] for im in imgs]))
        per_level = [grouped.split_groups(f, G) for f in feats]
        out = _GroupedPyramids([lvl[g] for lvl in per_level] for g in range(G))
        out.bases, out.G = list(feats), G
        return out

    def _depth_many(self, decoder, feats_list, merged=None):
        """Depth decoder on G feature pyramids -> per input {("disp", s): fp32 disparity}.
        `merged`: the interleaved batch of the pyramids per level, when the caller already has it."""
        if not self.opt.group_calls or len(feats_list) == 1:
            return [self._depth(decoder, f) for f in feats_list]
        G = len(feats_list)
        if merged is None:
            merged = [grouped.merge_groups([f[l] for f in feats_list]) for l in range(len(feats_list[0]))]
        # a no-op today (no decoder has BatchNorm); keeps per-call statistics if one ever does
        with grouped.grouped(self.models[decoder], G):
            out = self._depth(decoder, merged)
        split = {k: (grouped.split_groups(v, G) if torch.is_tensor(v) else [v] * G) for k, v in out.items()}
        return [{k: split[k][g] for k in out} for g in range(G)]

    def _fuse_many(self, jobs, merged_feats=None):
        """jobs: (three feature pyramids, two flows, merge mask) per fused frame
        (reference: train.py:788-812) -> per job {("disp", s)} of the multi-frame decoder.
        `merged_feats`: [position][level] interleaved batches over the jobs, when the caller already has them."""
        def one(feats, fl, mask):
            f = self._nets(lambda: self.models["fusion_module"](
                [[t.float() for t in lvl] for lvl in feats], fl, mask))
            return self._depth("depth_mf", f)
        if not self.opt.group_calls or len(jobs) == 1:
            return [one(*j) for j in jobs]
        G, L = len(jobs), len(jobs[0][0][0])
        feats = merged_feats if merged_feats is not None else \
            [[grouped.merge_groups([j[0][pos][l] for j in jobs]) for l in range(L)] for pos in range(3)]
        flows = [grouped.merge_groups([j[1][k] for j in jobs]) for k in range(2)]
        mask = grouped.merge_groups([j[2] for j in jobs])
        with grouped.grouped(self.models["fusion_module"], G), grouped.grouped(self.models["depth_mf"], G):
            out = one(feats, flows, mask)
        split = {k: (grouped.split_groups(v, G) if torch.is_tensor(v) else [v] * G) for k, v in out.items()}
        return [{k: split[k][g] for k in out} for g in range(G)]

    def predict_poses_many(self, pairs):
        """[(img_a, img_b), ...] -> [(pose a->b, inverted), ...] (reference: train.py:724-731)."""
        if not self.opt.group_calls or len(pairs) == 1:
            return [self.predict_poses(a, b) for a, b in pairs]
        G = len(pairs)
        x = self._merge_inputs([[a, b] for a, b in pairs])
        with grouped.grouped(self.models["pose_encoder"], G):
            feats = [self._encode("pose_encoder", x)]
        axisangle, translation = self._nets(lambda: self.models["pose"]([[f.float() for f in feats[0]]]))
        axisangle, translation = axisangle.float(), translation.float()
        pose = split_g(transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=False), G)
        pose_inv = split_g(transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=True), G)
        return list(zip(pose, pose_inv))

    def process_batch(self, inputs):
        """Pass a minibatch through the networks and the 9 hot-path units
        (reference: train.py:698-886).  The data flow is the reference's; mutually independent
        network invocations are issued as one interleaved batch each (`--group_calls`)."""
        o = self.opt
        for key, ipt in inputs.items():
            if torch.is_tensor(ipt):
                inputs[key] = ipt.to(self.device, non_blocking=True)
        if ("color_aug", 0, 0) not in inputs:
            # the loader delivered raw frames + the augmentation draw: flip / ColorJitter / affine
            # views happen here, on the device (reference: datasets/mono_dataset.py:102-184)
            from . import augment
            augment.augment_on_device(inputs, use_affine=o.use_affine)
        B = inputs[("color", 0, 0)].shape[0]
        embt = torch.full((B, 1, 1, 1), 0.5, device=self.device)
        img_n1, img_p1, img_0 = inputs[("color", -1, 0)], inputs[("color", 1, 0)], inputs[("color", 0, 0)]

        with torch.no_grad():
            if o.group_calls:
                # the three teacher passes (train.py:715-722) as one batch: IFRNet has no batch
                # statistics, so plain concatenation is exact; the synthesised frame of the
                # third (flow-only) pair is simply not used
                e3 = torch.cat([embt, embt, embt], 0)
                im, f_a, f_b, mk = self._nets(lambda: self.model_vfi_train(
                    torch.cat([img_n1, img_0, img_n1], 0), torch.cat([img_0, img_p1, img_p1], 0), e3))
                img_nt, img_pt, _ = im.chunk(3)
                flow_nt_n1, flow_pt_0, flow_0_n1 = f_a.chunk(3)
                flow_nt_0, flow_pt_p1, flow_0_p1 = f_b.chunk(3)
                merge_mask_nt, merge_mask_pt, merge_mask_01 = mk.chunk(3)
            else:
                img_nt, flow_nt_n1, flow_nt_0, merge_mask_nt = self._nets(
                    lambda: self.model_vfi_train(img_n1, img_0, embt))
                img_pt, flow_pt_0, flow_pt_p1, merge_mask_pt = self._nets(
                    lambda: self.model_vfi_train(img_0, img_p1, embt))
                flow_0_n1, flow_0_p1, merge_mask_01 = self._nets(
                    lambda: self.model_vfi_train(img_n1, img_p1, embt, onlyFlow=True))
            img_nt, img_pt = img_nt.float(), img_pt.float()
            flows = [t.float() for t in (flow_nt_n1, flow_nt_0, flow_pt_0, flow_pt_p1, flow_0_n1,
                                         flow_0_p1, merge_mask_nt, merge_mask_pt, merge_mask_01)]
            (flow_nt_n1, flow_nt_0, flow_pt_0, flow_pt_p1, flow_0_n1, flow_0_p1, merge_mask_nt,
             merge_mask_pt, merge_mask_01) = flows

        K, inv_K = inputs[("K", 0)], inputs[("inv_K", 0)]
        losses = {"loss_base": torch.zeros((), device=self.device),
                  "loss_dc": torch.zeros((), device=self.device)}

        aug = lambda f: inputs[("color_aug", f, 0)]  # noqa: E731
        ((pose_n1_0, pose_0_n1), (pose_0_p1, pose_p1_0), (pose_n1_nt, pose_nt_n1), (pose_nt_p1, pose_p1_nt),
         (pose_n1_pt, pose_pt_n1), (pose_pt_p1, pose_p1_pt)) = self.predict_poses_many(
            [(aug(-1), aug(0)), (aug(0), aug(1)), (img_n1, img_nt), (img_nt, img_p1), (img_n1, img_pt),
             (img_pt, img_p1)])

        # ---- encoder invocations, in the reference's call order (train.py:745-747, 788-797, 830-868)
        enc_in = [aug(0), img_nt, img_pt]
        if o.fuse_model_type != "separate_all":
            enc_in += [aug(-1), aug(1)]
        if o.use_affine:
            if o.group_calls and img_nt.is_cuda and img_nt.dtype == torch.float32:
                # the two synthesised frames are the first two chunks of the teacher's batched output: ONE launch
                t_nt, t_pt = self.affine_transform(im[:2 * B].float(), inputs).chunk(2)
            else:
                t_nt, t_pt = self.affine_transform(img_nt, inputs), self.affine_transform(img_pt, inputs)
            tgts_a = [inputs[("color_affine", 0, 0)], t_nt, t_pt]
            enc_in += [inputs[("color_affine_aug", 0, 0)], tgts_a[1], tgts_a[2]]
        enc = self._encode_many("encoder", enc_in)
        feats_0, feats_nt, feats_pt = enc[0], enc[1], enc[2]
        feats_aff = enc[-3:] if o.use_affine else []
        # which encoder inputs (positions in enc_in) each consumer takes: the single-frame decoder call, and
        # the three pyramids (previous / centre / next) of the three fusion jobs below
        n_in = len(enc_in)
        dec_plan = [0, 2, 1] + ([n_in - 3, n_in - 2, n_in - 1] if o.use_affine else [])
        fuse_plans = [[3, 3, 0], [0, 1, 2], [4, 0, 4]]       # [feats_n1, feats_n1, feats_0], [feats_0, feats_nt, feats_pt], ...
        pre = pre_mf = None
        if o.group_calls:
            if o.fuse_model_type != "separate_all":
                pre = self._regroup(enc, [dec_plan] + fuse_plans)
                pre_mf = pre[1:] if pre is not None else None
            else:
                pre = self._regroup(enc, [dec_plan])

        # ---- single-frame depths (plain and affine views share the decoder)
        dec = self._depth_many("depth", [feats_0, feats_pt, feats_nt] + feats_aff,
                               merged=pre[0] if pre is not None else None)
        disp_0, disp_pt, disp_nt = dec[0], dec[1], dec[2]
        def to_depth(d):
            # by-product of the decoder's disparity-head epilogue when it ran fused
            dep = d.get(("depth", 0))
            return dep if dep is not None else disp_to_depth(d[("disp", 0)], o.min_depth, o.max_depth)[1]
        depth_0, depth_pt, depth_nt = to_depth(disp_0), to_depth(disp_pt), to_depth(disp_nt)

        srcs = [img_n1, img_p1]

        def unit(disp, tgt, poses, sources=srcs, **kw):
            return dict(disp_tgt=disp, img_tgt=tgt, poses=poses, imgs_src=sources, K=K, inv_K=inv_K, **kw)
        # ---- the affine-augmented views' units (train.py:837-882) need nothing the multi-frame branch produces: their
        # disparities come out of the same decoder call as the single-frame ones
        units_a, todo = [], ()
        if o.use_affine:
            Rc = inputs["Rc"]
            # (the graph step inverts Rc before the replay: a solver call does not belong in a capture)
            # inv_ex = torch.inverse (same LU, bit-identical result) WITHOUT its host-side singularity check: that check
            # reads a device flag and so drained the stream in the middle of every step's forward pass (measured:
            # 92 ms of the host's 120 ms forward enqueue spent waiting there; tools/host_audit.py, tools/inv_probe.py)
            if "Rc_inv" in inputs:
                Rc_inv = inputs["Rc_inv"]
            else:
                Rc_inv, info = torch.linalg.inv_ex(Rc)
                # torch.inverse would have raised on a singular Rc; here the LU's status stays on the device and is
                # folded into a flag the logging step reads when it synchronises anyway
                bad = (info != 0).any()
                self._singular_rc = bad if getattr(self, "_singular_rc", None) is None else (self._singular_rc | bad)
            srcs_a = [inputs[("color_affine", -1, 0)], inputs[("color_affine", 1, 0)]]
            mask_rec = inputs["valid_mask_rec"]
            for (pa, pb), tgt, disp_a in zip(((pose_0_n1, pose_0_p1), (pose_nt_n1, pose_nt_p1), (pose_pt_n1, pose_pt_p1)),
                                              tgts_a, dec[3:]):
                poses_a = [self._affine_pose(pa, Rc, Rc_inv), self._affine_pose(pb, Rc, Rc_inv)]
                units_a.append(unit(disp_a, tgt, poses_a, srcs_a, mask_rec=mask_rec))

        # the three single-frame units (train.py:747-760) AND the three affine ones: mutually independent, ONE launch
        # of six units (round 4: two launches of three).  Each single-frame unit also hands the identity maps of its
        # (target, sources) to the multi-frame unit of the same target below
        share = bool(getattr(o, "share_identity", True)) and not o.disable_automasking
        units_sf = [unit(disp_0, img_0, [pose_0_n1, pose_0_p1]), unit(disp_pt, img_pt, [pose_pt_n1, pose_pt_p1]),
                    unit(disp_nt, img_nt, [pose_nt_n1, pose_nt_p1])]
        merged_launch = bool(units_a) and bool(getattr(o, "merge_unit_groups", True)) and o.fused_units \
            and getattr(o, "batch_units", True) and len(units_sf) + len(units_a) <= MAX_UNITS
        if merged_launch:
            total, idents = self._units(units_sf + units_a, want_ident=[share] * 3 + [False] * len(units_a))
            idents = idents[:3] if idents is not None else None
        else:
            total, idents = self._units(units_sf, want_ident=share)
            if units_a:
                total, _ = self._units(units_a, sum_in=total)
        if share and idents is None and not getattr(self, "_share_warned", False):
            # --share_identity only works through the batched forward+backward launch
            self._share_warned = True
            logging.warning("--share_identity is on but inactive: it needs --fused_units and --batch_units, at most "
                            "two sources per unit and a wanted gradient; the multi-frame units re-evaluate the "
                            "identity candidates")
        id_0, id_pt, id_nt = idents if idents is not None else (None, None, None)

        # ---- multi-frame depths
        if o.fuse_model_type == "separate_all":
            enc_mf = self._encode_many("encoder_mf", [aug(0), img_nt, img_pt, aug(-1), aug(1)])
            feats_0, feats_nt, feats_pt, feats_n1, feats_p1 = enc_mf
            pre_mf = self._regroup(enc_mf, fuse_plans) if o.group_calls else None
        else:
            feats_n1, feats_p1 = enc[3], enc[4]
        fused = self._fuse_many([
            ([feats_n1, feats_0, feats_p1], [flow_0_n1, flow_0_p1], merge_mask_01),
            ([feats_n1, feats_nt, feats_0], [flow_nt_n1, flow_nt_0], merge_mask_nt),
            ([feats_0, feats_pt, feats_p1], [flow_pt_0, flow_pt_p1], merge_mask_pt)], merged_feats=pre_mf)
        disp_0_fuse, disp_nt_fuse, disp_pt_fuse = fused
        depth_0_fuse, depth_nt_fuse, depth_pt_fuse = (to_depth(d) for d in fused)

        # the three multi-frame units (train.py:795-810): one launch, identity maps handed over; its finishing kernel
        # adds the first launch's total (loss_base of train.py:760 / 812 / 882 without an add launch)
        total, _ = self._units([unit(disp_0_fuse, img_0, [pose_0_n1, pose_0_p1], ident=id_0),
                                unit(disp_nt_fuse, img_nt, [pose_nt_n1, pose_nt_p1], ident=id_nt),
                                unit(disp_pt_fuse, img_pt, [pose_pt_n1, pose_pt_p1], ident=id_pt)], sum_in=total)
        losses["loss_base"] = total
        # ---- depth-consistency losses: the three multi-frame / single-frame pairs (train.py:813-815) and, per affine
        # view, the restored depth against both (train.py:868-882) -- nine SI-log evaluations, ONE launch each way
        dc_jobs = [(depth_0, depth_0_fuse, None), (depth_nt, depth_nt_fuse, None), (depth_pt, depth_pt_fuse, None)]
        if o.use_affine:
            from . import ops
            mask_c = inputs["valid_mask_cons"]
            todo = ((depth_0, depth_0_fuse), (depth_nt, depth_nt_fuse), (depth_pt, depth_pt_fuse))
            aff = [to_depth(d) for d in dec[3:]]
            if getattr(o, "batch_silog", True) and all(t.is_cuda and t.dtype == torch.float32 for t in aff):
                rest = ops.affine_restore_many(aff, inputs["angle"], inputs["box"], inputs["ratio_local"])
            else:
                rest = [ops.affine_restore(t, inputs["angle"], inputs["box"], inputs["ratio_local"]) for t in aff]
            for (depth_s, depth_f), restored in zip(todo, rest):
                dc_jobs += [(restored, depth_f, mask_c), (restored, depth_s, mask_c)]
        losses["loss_dc"] = self._silog_sum(dc_jobs)

        losses["loss"] = losses["loss_base"] + o.lamda * losses["loss_dc"]
        return None, losses

    # ------------------------------------------------------------------ loss helpers
    def affine_transform(self, img, inputs):
        """Rotate, crop the box, resize back (reference: train.py:888-902): one launch for the
        batch, angle / box read on the device (`mvf_affine_transform_fwd`)."""
        from . import ops
        return ops.affine_transform(img, inputs["angle"], inputs["box"])

    def compute_depth_consistency_loss_affine(self, depth_affine, depth, depth_fuse, inputs):
        """Scale-aware depth consistency (SADC) (reference: train.py:904-922): resize + paste +
        rotate back + ratio as one launch (`mvf_affine_restore_fwd/bwd`), then two SI-log
        reductions -- no per-sample loop, no .item() syncs."""
        from . import ops
        restored = ops.affine_restore(depth_affine, inputs["angle"], inputs["box"], inputs["ratio_local"])
        mask = inputs["valid_mask_cons"]
        return self.compute_SI_log_depth_loss(restored, depth_fuse, mask) + \
            self.compute_SI_log_depth_loss(restored, depth, mask)

    def _silog_sum(self, jobs, beta=0.5):
        """Sum of compute_SI_log_depth_loss over `jobs` = [(pred, target, mask | None), ...] in list order
        (reference: the `loss_dc +=` lines of train.py:813-815, 868-882): one launch forward, one backward
        (`ops.silog_many`) when every tensor is fp32 on the device, else one call per job."""
        from . import ops
        ok = getattr(self.opt, "batch_silog", True) and 1 <= len(jobs) <= ops.nat.MAX_SILOG_JOBS and all(
            t is None or (t.is_cuda and t.dtype == torch.float32 and t.shape == jobs[0][0].shape and t.dim() == 4
                          and t.shape[1] == 1) for j in jobs for t in j)
        if ok:
            return ops.silog_many(jobs, beta)[0]
        total = torch.zeros((), device=self.device)
        for pred, target, mask in jobs:
            total = total + self.compute_SI_log_depth_loss(pred, target, mask, beta)
        return total

    def compute_SI_log_depth_loss(self, pred, target, mask=None, beta=0.5):
        """Scale-invariant log loss (reference: train.py:924-941): one reduction kernel + one
        element-wise backward kernel instead of ~30 element-wise / reduction launches."""
        from . import ops
        return ops.silog_loss(pred, target, mask, beta)

    def predict_poses(self, img_0, img1):
        """Pose between two frames, forward and inverted (reference: train.py:943-954)."""
        feats = [self._encode("pose_encoder", torch.cat([img_0, img1], 1))]
        axisangle, translation = self._nets(lambda: self.models["pose"]([[f.float() for f in feats[0]]]))
        axisangle, translation = axisangle.float(), translation.float()
        pose = transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=False)
        pose_inv = transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=True)
        return pose, pose_inv

    # ------------------------------------------------------------------ logging / ckpt
    def log_time(self, batch_idx, data_time, batch_time, loss):
        if getattr(self, "_singular_rc", None) is not None:
            bad, self._singular_rc = bool(self._singular_rc), None
            if bad:
                raise RuntimeError("a singular affine matrix Rc reached the affine branch since the last log "
                                   "(torch.inverse would have raised at that step; train.py:820)")
        left = (self.num_total_steps - self.step) * batch_time if self.step > 1 else 0
        lr = float(self.model_optimizer.param_groups[0]["lr"])
        logging.info("epoch: %2d/%d | batch: %4d/%d | data time: %.4f | batch time: %.3f | loss: %.4f | "
                     "lr: %.2e | time left: %s", self.epoch, self.opt.num_epochs - 1, batch_idx,
                     self.num_steps_per_epoch, data_time, batch_time, loss, lr, sec_to_hm_str(left))

    def log_tensorboard(self, mode, losses):
        """Scalars as JSON lines (TensorBoard is absent on both boxes; reference: train.py:1062-1067)."""
        with open(os.path.join(self.log_path, f"scalars_{mode}.jsonl"), "a") as f:
            f.write(json.dumps({"step": self.step, **{k: float(v) for k, v in losses.items()}}) + "\n")

    def _state_dicts(self):
        return {name: m.state_dict() for name, m in self.models.items()}

    def save_model(self, ep_end=False, batch_idx=0):
        """Checkpoint in the reference's format (reference: train.py:1108-1136)."""
        models_dir = os.path.join(self.log_path, "models")
        os.makedirs(models_dir, exist_ok=True)
        to_save = self._state_dicts()
        to_save["height"], to_save["width"] = self.opt.height, self.opt.width
        to_save["use_stereo"] = self.opt.use_stereo
        if ep_end:
            torch.save(to_save, os.path.join(models_dir, "model_{}.pth".format(self.epoch)))
            to_save["epoch"] = self.epoch + 1
        else:
            to_save["epoch"] = self.epoch
        to_save["step_in_total"] = self.step
        to_save["batch_idx"] = batch_idx
        to_save["optimizer"] = self.model_optimizer.state_dict()
        for g in to_save["optimizer"]["param_groups"]:      # graph mode keeps the rate on the device:
            g["lr"] = float(g["lr"])                        # checkpoints always carry a float
        to_save["lr_scheduler"] = self.model_lr_scheduler.state_dict()
        torch.save(to_save, os.path.join(self.log_path, "ckpt.pth"))

    def _load_into_models(self, checkpoint):
        for name, model in self.models.items():
            if name not in checkpoint:
                continue
            own = model.state_dict()
            own.update({k: v for k, v in checkpoint[name].items() if k in own})
            model.load_state_dict(own)

    def load_ckpt(self):
        """Resume (reference: train.py:1138-1159)."""
        path = os.path.join(self.log_path, "ckpt.pth")
        if not os.path.exists(path):
            logging.info("No checkpoint to resume, train from epoch 0.")
            return None
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)
        self._load_into_models(checkpoint)
        self.ep_start = checkpoint["epoch"]
        self.batch_start = checkpoint["batch_idx"]
        self.step = checkpoint["step_in_total"]
        logging.info("Start at epoch %d, batch index %d", self.ep_start, self.batch_start)
        return checkpoint

    def load_pretrained_model(self):
        """Initialise from a weights file (reference: train.py:1161-1176)."""
        path = os.path.abspath(self.opt.pretrained_path)
        assert os.path.exists(path), "Cannot find folder {}".format(path)
        self._load_into_models(torch.load(path, map_location="cpu", weights_only=False))


class _StepGraph:
    """The device work of ``Trainer._device_step`` as ONE HIP graph (``--hip_graph``).

    A training step is several thousand launches (ResNet18: ~2,500, DHRNet: ~13,500 -- at
    26 us of host time per launch the DHRNet step is bound by the enqueueing Python thread, not
    by the GPU).  Shapes are static, nothing in the step reads a value back to the host, the
    gradients are the capture's own allocations (or the reducer's persistent flat buckets when there is an
    exchange: both keep their addresses across replays) and the optimiser is capturable, so
    the whole step -- teacher, encoders, decoders, the nine unit kernels, backward, gradient
    clipping, AdamW -- is captured once and replayed with one launch per step.

    Protocol: the batch is copied into static input tensors; the first ``WARMUP`` steps run
    eagerly on a side stream (MIOpen solver selection, allocator and optimiser state), the next
    call captures, every later call replays.  What stays outside the graph: the host-to-device
    copy of the batch, the inverse of ``Rc`` (a solver call) and the learning-rate schedule
    (updates the device-resident rate in place).  The tie-break noise is drawn with
    ``torch.randn`` (graph-safe Philox offsets), not from the in-kernel generator whose key is a
    launch argument and would be frozen into the capture."""

    WARMUP = 3

    def __init__(self, trainer):
        self.t = trainer
        # "step": the whole device work of a step; "backward": the graph ends after the gradient exchange,
        # clipping + optimiser run eagerly after every replay
        self.scope = getattr(trainer.opt, "hip_graph_scope", "step")
        self.calls = 0
        self.graph = None
        self.static = None
        self.losses = None
        self.stream = torch.cuda.Stream(device=trainer.device)
        trainer.opt.inkernel_noise = False

    def _load(self, inputs):
        dev = self.t.device
        if self.static is None:
            self.static = {k: (v.to(dev, non_blocking=True).clone() if torch.is_tensor(v) else v)
                           for k, v in inputs.items()}
            if "Rc" in self.static:
                self.static["Rc_inv"] = torch.empty_like(self.static["Rc"])
        else:
            for k, v in inputs.items():
                if torch.is_tensor(v):
                    dst = self.static[k]
                    if dst.shape != v.shape:
                        raise RuntimeError(f"--hip_graph needs static shapes: {k} changed from "
                                           f"{tuple(dst.shape)} to {tuple(v.shape)} (drop_last must be on)")
                    dst.copy_(v, non_blocking=True)
        if "Rc" in self.static:
            # LU-based like torch.inverse, without its host-side error check (no synchronisation)
            self.static["Rc_inv"].copy_(torch.linalg.inv_ex(self.static["Rc"])[0])

    def _quiesce_collectives(self):
        """Before the capture: let ProcessGroupNCCL's watchdog retire every collective of the eager warm-up steps.
        The watchdog polls the end events of the works it still lists with hipEventQuery; on ROCm 7.2 a query that
        lands while RCCL's stream is part of an ongoing capture fails with hipErrorCapturedEvent ("operation not
        permitted on an event last recorded in a capturing stream") although the event was recorded before the
        capture began, and the watchdog takes the process down -- measured with the data-parallel collectives forced
        through RCCL: 1-3 of 6 runs died at the capture (tools/r04_graph_rccl_probe.py, profiles/r04_graph_rccl.log).
        Collectives issued DURING the capture are not listed (torch skips them), so an empty list at its start is
        enough: drain the device, then give the watchdog (100 ms poll) time to see the completed works."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
            return
        torch.cuda.synchronize(self.t.device)
        time.sleep(float(os.environ.get("MVF_GRAPH_QUIESCE_S", "1.0")))

    def __call__(self, inputs):
        t = self.t
        self._load(inputs)
        self.calls += 1
        if self.graph is None:
            cur = torch.cuda.current_stream(t.device)
            self.stream.wait_stream(cur)
            if self.calls <= self.WARMUP:
                with torch.cuda.stream(self.stream):
                    losses = t._device_step(dict(self.static))
                cur.wait_stream(self.stream)
                return losses
            self._quiesce_collectives()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                self.losses = (t._device_step if self.scope == "step" else t._forward_backward)(dict(self.static))
            self.graph = g
            logging.info("optimisation step captured into a HIP graph (scope: %s)", self.scope)
        self.graph.replay()
        if self.scope != "step":
            t._update()
        return self.losses


def main(argv=None):
    """``python -m mono_vifi_amd.trainer -c configs/...txt`` (reference: train.py:1178-1185)."""
    from .options import parse_args
    opts = parse_args(argv)
    parallel.init_distributed(opts)
    Trainer(opts).train()


if __name__ == "__main__":
    main()
