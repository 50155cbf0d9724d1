"""mono_vifi_amd -- MI355X-native view-synthesis + photometric-loss hot path behind the
Mono-ViFI `layers.py` / `Trainer` API.

Sub-modules are imported lazily; nothing here touches the GPU or loads the HIP library
until an op is called.  Importing the package puts two settings into the environment (an
explicit value in the environment always wins), which is why it should be imported before
the process touches the GPU:

MIOpen find-db: the networks either side of the hot path are MIOpen convolutions, and on a box
with an empty MIOpen cache the first call of every convolution shape runs a solver search
(measured: 144 s before the first ResNet18 640x192 step, 113-140 s for each of the other BASELINE
configurations).  `miopen_db/` holds the user find-db those searches wrote on an MI355X
(MIOpen's own text format, one line per convolution problem: the solvers it timed and their
times) for the four BASELINE.json training shapes; pointing MIOPEN_USER_DB_PATH at it brings the
cold start to 9-10 s and makes the solver choice the same on every box.
MVF_NO_SHIPPED_MIOPEN_DB=1 switches this off.

HIP-graph replay: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, see `ensure_graph_replay_env`."""
import os as _os

__version__ = "0.1.0"

MIOPEN_DB_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")


def use_shipped_miopen_db():
    """Point MIOpen at the find-db this package ships (before the first convolution runs)."""
    if _os.environ.get("MVF_NO_SHIPPED_MIOPEN_DB") == "1" or "MIOPEN_USER_DB_PATH" in _os.environ:
        return _os.environ.get("MIOPEN_USER_DB_PATH")
    try:
        has = any(f.endswith(".ufdb.txt") for f in _os.listdir(MIOPEN_DB_DIR))
    except OSError:
        has = False
    if has:
        _os.environ["MIOPEN_USER_DB_PATH"] = MIOPEN_DB_DIR
        return MIOPEN_DB_DIR
    return None


def ensure_graph_replay_env(strict=False):
    """HIP graphs of a whole optimisation step: switch the HIP runtime's graph *packet capture* off.

    ROCm 7.2's runtime replays kernel nodes as pre-built AQL packets whose kernel arguments live in a
    device-side pool (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default).  With it, the replay of a captured
    training step (~2,200 kernel nodes for ResNet18, ~9,900 for HRNet18) ends in a GPU memory fault or
    a hang at the BASELINE shapes -- within the first replays, with or without the optimiser in the
    graph -- although every stage of the step replays on its own; with the flag at 0 the same graphs
    replayed 60 (ResNet18) / 40 (HRNet18) steps with a synchronisation after each
    (tools/graph_flow_probe.py, profiles/r03_graph_flow_probe.log).  The runtime reads the flag at
    its first HIP call, so it must be in the environment before the process touches the GPU:
    this is called when the package is imported and again when --hip_graph is requested
    (`strict`: raise if the GPU context already exists without the flag)."""
    if _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0":
        return True
    if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in _os.environ and not strict:
        return False            # an explicit user setting wins
    import sys as _sys
    torch = _sys.modules.get("torch")
    if strict and torch is not None and torch.cuda.is_initialized():
        raise RuntimeError(
            "--hip_graph needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the first HIP call "
            "(the replay of a captured optimisation step faults with the HIP runtime's graph packet capture "
            "on ROCm 7.2: DESIGN.md section 7); import mono_vifi_amd before touching the GPU, or export it")
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    return True


use_shipped_miopen_db()
ensure_graph_replay_env()
