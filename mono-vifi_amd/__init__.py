"""mono_vifi_amd -- MI355X-native view-synthesis + photometric-loss hot path behind the
Mono-ViFI `layers.py` / `Trainer` API.

Sub-modules are imported lazily; nothing here touches the GPU or loads the HIP library
until an op is called, and **importing the package changes nothing in the process**: no
environment variable is set on import (round 3 did; a library must not reconfigure every
importer).  Two opt-in helpers exist for the entry points (`Trainer`, `bench.py`, the GPU
tests) to call before the process touches the GPU:

`use_shipped_miopen_db()` -- MIOpen find-db: the networks either side of the hot path are
MIOpen convolutions, and on a box with an empty MIOpen cache the first call of every
convolution shape runs a solver search (measured: 144 s before the first ResNet18 640x192
step, 113-140 s for each of the other BASELINE configurations).  `miopen_db/` holds the user
find-db those searches wrote on an MI355X (MIOpen's own text format, one line per convolution
problem) for the four BASELINE.json training shapes.  The shipped file is READ-ONLY seed data:
the helper copies it into a directory of this process's own and points MIOPEN_USER_DB_PATH
there, so that (a) a run never modifies the tracked file (MIOpen appends every new problem to
its user db), (b) the eight ranks of a node never open one file concurrently, (c) a read-only
install works.  An explicit MIOPEN_USER_DB_PATH in the environment wins;
MVF_NO_SHIPPED_MIOPEN_DB=1 switches the helper off.

`ensure_graph_replay_env()` -- DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for whole-step HIP graphs,
set only by entry points that were asked for `--hip_graph`."""
import os as _os

__version__ = "0.1.0"

MIOPEN_DB_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
_SEED_TAG = "MVF_MIOPEN_DB_SEEDED_BY"     # pid of the process that made the copy MIOPEN_USER_DB_PATH points at


def _remove_seed_dir(path, pid):
    if _os.getpid() != pid:          # a forked child must not remove its parent's directory
        return
    import shutil
    shutil.rmtree(path, ignore_errors=True)


def use_shipped_miopen_db():
    """Seed a per-process MIOpen user find-db from the shipped one (call before the first convolution).

    Returns the directory MIOpen will use, or None when nothing was done.  A MIOPEN_USER_DB_PATH the
    user exported is respected; one inherited from a parent process that seeded ITS copy (bench.py's
    child legs, torchrun ranks started from a seeded launcher) is replaced by a copy of this process's
    own, so no two processes ever share a writable db file."""
    env = _os.environ
    if env.get("MVF_NO_SHIPPED_MIOPEN_DB") == "1":
        return env.get("MIOPEN_USER_DB_PATH")
    mine = str(_os.getpid())
    if "MIOPEN_USER_DB_PATH" in env:
        tag = env.get(_SEED_TAG)
        if tag is None or tag == mine:      # the user's own setting, or this process already seeded
            return env["MIOPEN_USER_DB_PATH"]
    try:
        files = [f for f in _os.listdir(MIOPEN_DB_DIR) if f.endswith(".ufdb.txt")]
    except OSError:
        files = []
    if not files:
        return None
    import atexit
    import shutil
    import tempfile
    rank = env.get("LOCAL_RANK", "0")
    try:
        dst = tempfile.mkdtemp(prefix=f"mvf_miopen_r{rank}_p{mine}_")
        for f in files:
            shutil.copyfile(_os.path.join(MIOPEN_DB_DIR, f), _os.path.join(dst, f))
    except OSError:
        return None                          # no writable temp dir: MIOpen's default location stays
    atexit.register(_remove_seed_dir, dst, _os.getpid())
    env["MIOPEN_USER_DB_PATH"] = dst
    env[_SEED_TAG] = mine
    return dst


_FLAG_STATE = {"late": False}      # True: the flag was asked for when the GPU context already existed


def ensure_graph_replay_env(strict=False):
    """HIP graphs of a whole optimisation step: switch the HIP runtime's graph *packet capture* off.

    ROCm 7.2's runtime replays kernel nodes as pre-built AQL packets whose kernel arguments live in a
    device-side pool (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default).  With it, the replay of a captured
    training step (~2,200 kernel nodes for ResNet18, ~9,900 for HRNet18) ends in a GPU memory fault or
    a hang at the BASELINE shapes -- within the first replays, with or without the optimiser in the
    graph -- although every stage of the step replays on its own; with the flag at 0 the same graphs
    replayed 60 (ResNet18) / 40 (HRNet18) steps with a synchronisation after each
    (tools/graph_flow_probe.py, profiles/r03_graph_flow_probe.log).  The runtime reads the flag at
    its first HIP call, so it must be in the environment before the process touches the GPU.  It is an
    undocumented debug switch of this ROCm release and changes how EVERY graph of the process replays,
    so it is set only where a whole-step graph was asked for: `bench.py --hip-graph`, `Trainer` with
    `--hip_graph` (`strict`: raise if the GPU context already exists without the flag)."""
    import sys as _sys
    torch = _sys.modules.get("torch")
    gpu_up = torch is not None and torch.cuda.is_initialized()
    if _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0":
        # "0" only counts if the runtime can have READ it: inherited from the parent process, or written here before
        # the GPU context existed (a late non-strict call used to write it after the first HIP call, and
        # the strict check then trusted the environment)
        if _FLAG_STATE["late"]:
            if strict:
                raise RuntimeError(
                    "--hip_graph: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was requested after this process had already "
                    "touched the GPU, so the HIP runtime never read it (DESIGN.md section 7): ask for the graph step "
                    "before the first HIP call, or export the variable")
            return False
        return True
    if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in _os.environ and not strict:
        return False            # an explicit user setting wins
    if gpu_up and not strict:
        # too late for the runtime to see it: do not pretend.  Remember, so that a later strict call fails loudly
        # instead of trusting a value written now.
        _FLAG_STATE["late"] = True
        import warnings
        warnings.warn("hip_graph requested after the first HIP call: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 cannot take "
                      "effect in this process (the whole-step graph will refuse to run)")
        return False
    if strict and gpu_up:
        raise RuntimeError(
            "--hip_graph needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the first HIP call "
            "(the replay of a captured optimisation step faults with the HIP runtime's graph packet capture "
            "on ROCm 7.2: DESIGN.md section 7); construct the Trainer (or call "
            "mono_vifi_amd.ensure_graph_replay_env()) before touching the GPU, or export it")
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    return True
