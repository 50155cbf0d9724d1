"""One optimisation step per BASELINE.json training configuration AT ITS FULL per-GPU shape
(reference configs/resnet18, configs/dhrnet/DHRNet_KITTI_MR.txt,
configs/litemono/LiteMono_KITTI_HR.txt, configs/dhrnet/DHRNet_CS.txt):

* fused unit kernels (mvf_units_fwdbwd: 3 launches of 3 units) == staged generate_images_pred +
  compute_losses_base on the same weights, batch and tie-break noise: losses to 2e-6, parameter gradients
  to 1e-4 (Lite-Mono 1e-3, see test_other_backbones_step) + three times the staged step's own run-to-run
  deviation;
* then one whole optimisation step (in-kernel noise: the variant bench.py times): finite losses, parameters
  moved.

Each configuration runs in a child process (tests/fullshape_worker.py) with MIOPEN_FIND_MODE=FAST and the
find-db the repo ships (mono-vifi_amd/miopen_db), so the four steps stay inside the driver's time limit on
a box with an empty MIOpen cache."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    "C2-ResNet18-B12-640x192": ("ResNet18", 12, 192, 640),
    "C3-DHRNet-B12-640x192": ("DHRNet", 12, 192, 640),
    "C4-LiteMono-B8-1024x320": ("LiteMono", 8, 320, 1024),
    "C5-DHRNet-B12-512x192": ("DHRNet", 12, 192, 512),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_shape_step(tmp_path, name):
    backbone, B, H, W = CONFIGS[name]
    env = dict(os.environ)
    env.setdefault("MIOPEN_FIND_MODE", "FAST")
    for d in ("FWD", "BWD", "WRW"):
        env.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + d, "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fullshape_worker.py"), backbone,
                        str(B), str(H), str(W), str(tmp_path)], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    r = json.loads(line[-1][len("RESULT "):])
    print(name, json.dumps(r))
    for k in ("loss", "loss_base", "loss_dc"):
        a, b = r[k]["fused"], r[k]["staged"]
        assert a == a and b == b, (k, r)
        assert abs(a - b) <= 2e-6 * abs(b) + 1e-9, (k, a, b)
    assert r["grad_finite"] and r["grad_norm"] > 0
    bar = 1e-3 if backbone == "LiteMono" else 1e-4
    assert r["grad_dev"] <= bar + 3.0 * r["grad_noise"], (r["grad_dev"], r["grad_noise"])
    assert all(v == v and abs(v) < 1e6 for v in r["step"].values()), r["step"]
    assert r["updated"]
