"""mono_vifi_amd -- MI355X-native view-synthesis + photometric-loss hot path behind the
Mono-ViFI `layers.py` / `Trainer` API.

Sub-modules are imported lazily; nothing here touches the GPU or loads the HIP library
until an op is called.

MIOpen find-db: the networks either side of the hot path are MIOpen convolutions, and on a box
with an empty MIOpen cache the first call of every convolution shape runs a solver search
(measured: 144 s before the first ResNet18 640x192 step, 113-140 s for each of the other BASELINE
configurations).  `miopen_db/` holds the user find-db those searches wrote on an MI355X
(MIOpen's own text format, one line per convolution problem: the solvers it timed and their
times) for the four BASELINE.json training shapes; pointing MIOPEN_USER_DB_PATH at it brings the
cold start to 9-10 s and makes the solver choice the same on every box.  An explicit
MIOPEN_USER_DB_PATH in the environment wins; MVF_NO_SHIPPED_MIOPEN_DB=1 switches this off."""
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


use_shipped_miopen_db()
