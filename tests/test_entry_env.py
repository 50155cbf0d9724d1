"""CPU: importing the package configures nothing; the entry-point helpers do.

* `import mono_vifi_amd` leaves the process environment as it found it (round 3 set MIOPEN_USER_DB_PATH to the
  tracked `miopen_db/` directory and DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for every importer);
* `use_shipped_miopen_db()` gives every PROCESS its own writable copy of the shipped find-db -- the tracked
  file is never MIOpen's user db, eight ranks never share one file -- and removes it at exit;
* DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 appears only when --hip_graph is asked for."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "mono-vifi_amd", "miopen_db")
KEYS = ("MIOPEN_USER_DB_PATH", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "MVF_MIOPEN_DB_SEEDED_BY")


def _py(code, env_extra=None, drop=KEYS):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def _db_hash():
    h = hashlib.sha256()
    for f in sorted(os.listdir(DB)):
        with open(os.path.join(DB, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    return h.hexdigest()


def test_import_sets_nothing():
    d = _py("import os, json; before = dict(os.environ); import mono_vifi_amd; "
            "import mono_vifi_amd.options, mono_vifi_amd.layers; "
            "print(json.dumps({k: os.environ.get(k) for k in %r} | {'same': dict(os.environ) == before}))" % (KEYS,))
    assert d["same"] and all(d[k] is None for k in KEYS)


def test_miopen_db_is_seeded_per_process_and_the_tracked_file_stays_untouched():
    before = _db_hash()
    code = r"""
import json, os, subprocess, sys
import mono_vifi_amd as m
p = m.use_shipped_miopen_db()
again = m.use_shipped_miopen_db()
files = sorted(os.listdir(p))
# MIOpen appends to its user db: emulate a new problem being recorded
with open(os.path.join(p, files[0]), "a") as f:
    f.write("appended-by-test\n")
child = subprocess.run([sys.executable, "-c",
    "import os, json, mono_vifi_amd as m; q = m.use_shipped_miopen_db(); "
    "print(json.dumps({'q': q, 'tail': open(os.path.join(q, sorted(os.listdir(q))[0])).read()[-20:]}))"],
    capture_output=True, text=True, env=dict(os.environ))
print(json.dumps({"p": p, "again": again, "files": files, "env": os.environ["MIOPEN_USER_DB_PATH"],
                  "child": json.loads(child.stdout.strip().splitlines()[-1]), "pkg": m.MIOPEN_DB_DIR}))
"""
    d = _py(code, {"LOCAL_RANK": "3"})
    assert d["p"] == d["again"] == d["env"] and d["p"] != d["pkg"] and not d["p"].startswith(ROOT)
    assert "_r3_" in os.path.basename(d["p"])                     # the rank is part of the directory name
    assert d["files"] == sorted(f for f in os.listdir(DB) if f.endswith(".ufdb.txt"))
    # a child that inherits the parent's seeded path makes a copy of its own, from the pristine shipped file
    assert d["child"]["q"] not in (d["p"], d["pkg"]) and "appended-by-test" not in d["child"]["tail"]
    assert not os.path.exists(d["p"]) and not os.path.exists(d["child"]["q"])      # removed at exit
    assert _db_hash() == before                                   # git status stays clean


def test_an_explicit_user_db_path_wins_and_the_opt_out_works(tmp_path):
    d = _py("import os, json, mono_vifi_amd as m; print(json.dumps({'p': m.use_shipped_miopen_db(), "
            "'env': os.environ.get('MIOPEN_USER_DB_PATH')}))", {"MIOPEN_USER_DB_PATH": str(tmp_path)})
    assert d["p"] == d["env"] == str(tmp_path)
    d = _py("import os, json, mono_vifi_amd as m; print(json.dumps({'p': m.use_shipped_miopen_db(), "
            "'env': os.environ.get('MIOPEN_USER_DB_PATH')}))", {"MVF_NO_SHIPPED_MIOPEN_DB": "1"})
    assert d["p"] is None and d["env"] is None


def test_graph_packet_capture_is_switched_off_only_for_hip_graph():
    code = ("import os, json; from mono_vifi_amd.options import default_options, parse_args; "
            "a = default_options(); e0 = os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'); "
            "b = parse_args(['--hip_graph', 'True']); e1 = os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'); "
            "print(json.dumps({'e0': e0, 'e1': e1}))")
    d = _py(code)
    assert d["e0"] is None and d["e1"] == "0"
    # an explicit setting of the user is left alone by the non-strict call
    d = _py("import os, json, mono_vifi_amd as m; r = m.ensure_graph_replay_env(); "
            "print(json.dumps({'r': r, 'e': os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE']}))",
            {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"})
    assert d["r"] is False and d["e"] == "1"


def test_graph_flag_requested_after_the_gpu_context_exists_is_refused():
    """a non-strict request that arrives after the first HIP call must not write the variable (the runtime
    has already read its absence), and the strict call of `Trainer(--hip_graph)` must then fail loudly instead of
    trusting the environment.  The GPU context is simulated (no GPU here)."""
    d = _py("import os, json, warnings, torch, mono_vifi_amd as m\n"
            "torch.cuda.is_initialized = lambda: True\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    r = m.ensure_graph_replay_env()\n"
            "env_after = os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')\n"
            "os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'     # what the old code had done at this point\n"
            "try:\n"
            "    m.ensure_graph_replay_env(strict=True); strict = 'passed'\n"
            "except RuntimeError as e:\n"
            "    strict = 'raised'\n"
            "print(json.dumps({'r': r, 'env_after': env_after, 'warned': len(w) > 0, 'strict': strict}))")
    assert d == {"r": False, "env_after": None, "warned": True, "strict": "raised"}
