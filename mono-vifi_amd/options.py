"""Command-line / config-file options of the trainer.

Same flag names, types, defaults and choices as the reference (options.py:8-226; 49 flags),
read from a ``key = value`` config file given with ``-c`` (the reference's
``configs/**/*.txt`` parse unchanged) plus command-line overrides.  Two differences, both
deliberate: parsing is a function call, not an import side effect
(the reference runs ``parser.parse_args()`` at import, options.py:226), and rank / world
size are also taken from the ``torchrun`` environment (``LOCAL_RANK`` / ``RANK`` /
``WORLD_SIZE``) besides ``--local_rank``.

Build-specific additions are grouped at the end (``--synthetic`` etc.): neither box has the
KITTI data, the ImageNet / IFRNet weights, torchvision or TensorBoard.
"""
import argparse
import os

file_dir = os.path.dirname(__file__)


def _str2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() in ("1", "true", "yes", "y", "on")


def build_parser():
    p = argparse.ArgumentParser(description="Mono-ViFI options (MI355X build)")
    p.add_argument("-c", "--config", default=None, help="config file path (key = value lines)")
    p.add_argument("--local_rank", "--local-rank", dest="local_rank", default=0, type=int)
    p.add_argument("--global_rank", default=0, type=int)
    p.add_argument("--world_size", default=1, type=int)

    # PATHS
    p.add_argument("--data_path", type=str, default=os.path.join(file_dir, "kitti_data"))
    p.add_argument("--data_path_pre", type=str, default=None)
    p.add_argument("--log_dir", type=str, default=os.path.join(os.path.expanduser("~"), "tmp"))

    # TRAINING options
    p.add_argument("--exp_name", type=str, default="mdp")
    p.add_argument("--split", type=str, default="eigen_zhou",
                   choices=["eigen_zhou", "eigen_full", "odom", "benchmark"])
    p.add_argument("--eval_split", type=str, default="eigen",
                   choices=["eigen", "eigen_benchmark", "benchmark", "odom_9", "odom_10"])
    p.add_argument("--num_layers", type=int, default=18, choices=[18, 34, 50, 101, 152])
    p.add_argument("--dataset", type=str, default="kitti",
                   choices=["kitti", "kitti_odom", "kitti_depth", "kitti_test", "nyuv2", "cityscapes"])
    p.add_argument("--jpg", action="store_true")
    p.add_argument("--height", type=int, default=192)
    p.add_argument("--width", type=int, default=640)
    p.add_argument("--disparity_smoothness", type=float, default=1e-3)
    p.add_argument("--num_scales", type=int, default=1)
    p.add_argument("--min_depth", type=float, default=0.1)
    p.add_argument("--max_depth", type=float, default=100.0)
    p.add_argument("--lamda", type=float, default=0.2)
    p.add_argument("--use_stereo", action="store_true")
    p.add_argument("--frame_ids", nargs="+", type=int, default=[0, -1, 1])

    # OPTIMIZATION options
    p.add_argument("--optimizer", type=str, default="adamw", choices=["adamw", "adam", "sgd"])
    p.add_argument("--lr_sche_type", type=str, default="step", choices=["cos", "step"])
    p.add_argument("--eta_min", type=float, default=5e-6)
    p.add_argument("--batch_size", type=int, default=12)
    p.add_argument("--learning_rate", default=0.0001, type=float)
    p.add_argument("--decay_rate", type=float, default=0.1)
    p.add_argument("--decay_step", type=int, nargs="+", default=[15])
    p.add_argument("--weight_decay", type=float, default=0.01)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--beta2", type=float, default=0.999)
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--clip_grad", type=float, default=5)
    p.add_argument("--num_epochs", type=int, default=20)
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--resume", action="store_true")

    # ABLATION options
    p.add_argument("--avg_reprojection", action="store_true")
    p.add_argument("--disable_automasking", action="store_true")
    p.add_argument("--no_ssim", action="store_true")
    p.add_argument("--weights_init", type=str, default="pretrained", choices=["pretrained", "scratch"])
    p.add_argument("--backbone", type=str, default="ResNet18",
                   choices=["ResNet18", "ResNet50", "LiteMono", "DHRNet"])
    p.add_argument("--vfi_scale", type=str, default="small", choices=["large", "small"])
    p.add_argument("--fuse_model_type", type=str, default="shared_encoder",
                   choices=["shared_encoder", "separate_all", "shared_all"])
    p.add_argument("--use_affine", action="store_true")

    # SYSTEM options
    p.add_argument("--num_workers", type=int, default=16)
    p.add_argument("--pretrained_path", type=str, default=None)
    p.add_argument("--log_frequency", type=int, default=500)
    p.add_argument("--save_frequency", type=int, default=500)

    # ---- additions of this build
    p.add_argument("--synthetic", type=_str2bool, default=True,
                   help="KITTI-shaped synthetic triplets generated on the fly (no datasets here)")
    p.add_argument("--synthetic_len", type=int, default=39810,
                   help="virtual training-set length (KITTI eigen_zhou has 39,810 items)")
    p.add_argument("--vfi_weights_dir", type=str, default="./weights",
                   help="IFRNet_{L,S}_{KITTI,CS}.pth; random-init teacher when absent")
    p.add_argument("--sync_bn", type=_str2bool, default=True,
                   help="SyncBatchNorm under world_size > 1 (reference: train.py:207)")
    p.add_argument("--group_calls", type=_str2bool, default=True,
                   help="run the mutually independent encoder / decoder / pose / fusion invocations "
                        "of a step as one interleaved batch each, with per-call BatchNorm statistics "
                        "(networks/grouped.py); False = one call at a time like the reference")
    p.add_argument("--fused_optimizer", type=_str2bool, default=True,
                   help="AdamW / Adam update as torch's single fused multi-tensor kernel (device only; the whole-step HIP "
                        "graph keeps the capturable foreach form)")
    p.add_argument("--regroup", type=_str2bool, default=True,
                   help="with --group_calls: the decoder / fusion calls take their interleaved batches of the "
                        "grouped encoder's feature pyramids from one regrouping launch per level (and one adjoint "
                        "launch) instead of per-group views re-merged with stack and accumulated by autograd")
    p.add_argument("--fused_units", type=_str2bool, default=True,
                   help="fused unit kernels (warped images in LDS) instead of the staged "
                        "generate_images_pred + compute_losses_base pair")
    p.add_argument("--batch_units", type=_str2bool, default=True,
                   help="the three mutually independent hot-path units of each group of a step "
                        "(single-frame / multi-frame / affine) as ONE kernel launch instead of three")
    p.add_argument("--merge_unit_groups", type=_str2bool, default=True,
                   help="with --batch_units: the single-frame and the affine units of a step (mutually independent: "
                        "train.py:747-760, 837-882) as ONE launch of six units, the multi-frame ones second -- two unit "
                        "launches per step instead of three")
    p.add_argument("--batch_silog", type=_str2bool, default=True,
                   help="the nine SI-log depth-consistency losses of a step as one launch forward and one backward")
    p.add_argument("--defer_unit_grads", type=_str2bool, default=True,
                   help="a hot-path unit that reads a disparity head's output in place leaves its RAW disparity gradient "
                        "to the head's adjoint kernel, which applies the per-image shift and the upstream gradient on load "
                        "(no scaling pass, no re-interleaving copy of the group gradients)")
    p.add_argument("--share_identity", type=_str2bool, default=True,
                   help="the multi-frame unit of a target takes the identity-reprojection maps the "
                        "single-frame unit of the same target and sources computed (train.py:747-749 vs "
                        "795-797) instead of re-evaluating them.  Only active together with --batch_units and "
                        "--fused_units, two sources per unit and auto-masking on (the trainer logs once when it "
                        "is requested but inactive)")
    p.add_argument("--device_augment", type=_str2bool, default=True,
                   help="flip / ColorJitter / affine views of a batch on the device (augment.py) instead "
                        "of per item on the host (reference: datasets/mono_dataset.py:102-184)")
    p.add_argument("--inkernel_noise", type=_str2bool, default=True,
                   help="auto-mask tie-break noise (train.py:1023-1024) drawn inside the unit kernel "
                        "from a counter-based generator instead of a torch.randn tensor per unit")
    p.add_argument("--hip_graph", type=_str2bool, default=False,
                   help="capture the device work of an optimisation step (networks, hot-path units, backward, "
                        "clipping, AdamW) once into a HIP graph and replay it: one launch per step instead of "
                        "thousands (needs static shapes; trainer._StepGraph).  Runs with the HIP runtime's graph "
                        "packet capture switched off (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, put into the environment "
                        "when this option is parsed / by the Trainer, before the first HIP call): with it the "
                        "replay faults at the BASELINE shapes on ROCm 7.2 (DESIGN.md section 7).  A GPU memory "
                        "fault during a replay cannot be caught")
    p.add_argument("--hip_graph_scope", type=str, default="step", choices=["step", "backward"],
                   help="--hip_graph: what the captured graph holds.  step (default): everything incl. clipping "
                        "and the capturable AdamW (learning rate resident on the device).  backward: networks, "
                        "hot-path units, backward pass and gradient exchange; clipping and the ordinary optimiser "
                        "stay ~20 eager launches")
    p.add_argument("--bucket_mb", type=float, default=32.0, help="gradient all-reduce bucket size")
    p.add_argument("--grad_exchange", type=str, default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                   help="per gradient bucket: one RCCL all-reduce, or its two halves issued explicitly on "
                        "the flat buffer (reduce_scatter_tensor + all_gather_into_tensor; SURVEY.md 8f-3)")
    p.add_argument("--no_overlap", type=_str2bool, default=False,
                   help="reduce the gradient buckets after backward instead of from the hooks during it "
                        "(measurement aid: the overlapped step beside the serial one)")
    p.add_argument("--force_collectives", type=_str2bool, default=False,
                   help="issue the data-parallel collectives (bucketed gradient all-reduce, "
                        "SyncBatchNorm statistics) even when world_size is 1 -- exercises the RCCL "
                        "code path on a one-GPU box; needs an initialised process group")
    p.add_argument("--channels_last", type=_str2bool, default=False)
    p.add_argument("--amp_bf16", type=_str2bool, default=False,
                   help="bf16 autocast for the conv networks (the hot path stays fp32)")
    return p


_STORE_TRUE = None


def read_config_file(path):
    """``key = value`` lines ('#' comments); the reference's configs use this format
    (e.g. configs/resnet18/ResNet18_KITTI_MR.txt:1-21)."""
    items = []
    with open(path) as f:
        for raw in f:
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            if "=" in line:
                k, v = line.split("=", 1)
            else:
                k, v = line, "True"
            items.append((k.strip().lstrip("-"), v.strip()))
    return items


def parse_args(argv=None):
    parser = build_parser()
    pre, _ = parser.parse_known_args(argv)
    cfg_argv = []
    if pre.config:
        flags = {a.dest: a for a in parser._actions}
        for k, v in read_config_file(pre.config):
            act = flags.get(k)
            if act is None:
                raise SystemExit(f"unknown option '{k}' in {pre.config}")
            if isinstance(act, argparse._StoreTrueAction):
                if _str2bool(v):
                    cfg_argv.append("--" + k)
            elif act.nargs in ("+", "*"):
                cfg_argv += ["--" + k] + v.replace("[", "").replace("]", "").replace(",", " ").split()
            else:
                cfg_argv += ["--" + k, v]
    # command line wins over the file
    opts = parser.parse_args(cfg_argv + list(argv if argv is not None else os.sys.argv[1:]))
    # torchrun / torch.distributed.run environment (reference reads --local_rank only)
    if "LOCAL_RANK" in os.environ:
        opts.local_rank = int(os.environ["LOCAL_RANK"])
    if "RANK" in os.environ:
        opts.global_rank = int(os.environ["RANK"])
    if "WORLD_SIZE" in os.environ:
        opts.world_size = int(os.environ["WORLD_SIZE"])
    _graph_env(opts)
    return opts


def _graph_env(opts):
    """--hip_graph was asked for: the HIP runtime's graph packet capture must be off before the process's first
    HIP call (mono_vifi_amd.ensure_graph_replay_env) -- option parsing is the earliest point an entry point has."""
    if getattr(opts, "hip_graph", False):
        from . import ensure_graph_replay_env
        ensure_graph_replay_env()


def default_options(**overrides):
    """Options namespace with the reference's defaults (for tests and benchmarks)."""
    opts = build_parser().parse_args([])
    for k, v in overrides.items():
        if not hasattr(opts, k):
            raise AttributeError(k)
        setattr(opts, k, v)
    _graph_env(opts)
    return opts
