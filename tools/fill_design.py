"""DESIGN.md = tools/DESIGN.md.in with its @TOKEN@ placeholders filled from the round's measured files (profiles/r05_*):
the numbers of the document are the numbers of the committed bench line and rocprofv3 summaries, not retyped ones.  Edit
the text in tools/DESIGN.md.in and re-run.
    python tools/fill_design.py [--check]      (--check: print the substitutions, write nothing)"""
import csv
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles")


def line_of(name):
    with open(os.path.join(P, name)) as f:
        return json.loads([x for x in f if x.startswith("{")][-1])


def train_stats(steps=9):
    out = {}
    with open(os.path.join(P, "r05_train_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            n = r["Name"]
            if "anonymous" in n and "k_" in n:
                key = n.split("::")[1].split("(")[0].split("<")[0]
                out[key] = out.get(key, 0.0) + float(r["TotalDurationNs"]) / steps / 1e6
    return out


def main():
    d = line_of("r05_bench_train_resnet18.json")
    rf, oc, hp = d["roofline"], d["other_configs"], d["hotpath_only"]
    ks = train_stats()
    kk = d["own_kernels"]["kernels"]
    fus_b = sum(v for k, v in ks.items() if k.startswith("k_anc_") or k == "k_fusion_level_bwd_anchor")
    silog = sum(v for k, v in ks.items() if k.startswith("k_silog"))
    affine = sum(v for k, v in ks.items() if k.startswith("k_affine"))
    f8 = fus_b + silog + affine + sum(ks.get(k, 0.0) for k in ("k_fusion_level_fwd", "k_flow_warp_fwd", "k_resize_bilinear_fwd",
                                                                "k_fusion_prep"))
    rows = ["| Kernel | launches / step | ms / step | MB / launch | GB/s | of HBM peak |", "|---|---|---|---|---|---|",
            "| `k_unit_fb<2>` (bound: VALU) | %d | %.3f | %.1f | %.0f | **%.4f** |" % (
                rf["launches"] // d["steps"], rf["avg_us"] * rf["launches"] / d["steps"] / 1e3,
                rf["algorithmic_bytes_per_launch"] / 1e6, rf["achieved"], rf["frac"])]
    for k, v in kk.items():
        rows.append("| `%s` | %g | %.3f | %.1f | %.0f | %.2f |" % (k, v["launches_per_step"], v["ms_per_step"],
                                                                 v["bytes_per_launch"] / 1e6, v["achieved"], v["frac"]))
    ngpu = "?"
    try:
        with open(os.path.join(P, "r05_gputest.log")) as f:
            m = re.findall(r"(\d+) passed", f.read())
            ngpu = m[-1] if m else "?"
    except OSError:
        pass
    pin = lambda v: "%.3f" % v if v is not None else "n/a"  # noqa: E731
    hpin = d["host"].get("pinned", {}).get("eager", {}).get("ms_per_step")
    sub = {
        "STEP_MS": "%.1f" % d["ms_per_step"], "STEP_IPS": "%.1f" % d["value"],
        "C3_MS": "%.1f" % oc["C3"]["ms_per_step"], "C4_MS": "%.1f" % oc["C4"]["ms_per_step"],
        "C5_MS": "%.1f" % oc["C5"]["ms_per_step"],
        "UNIT_US": "%.1f" % rf["us_per_unit"], "FRAC": "%.4f" % rf["frac"], "PIPE": "%.2f" % rf["valu_pipe_frac"],
        "FRAC_CEIL": "%.3f" % (rf["frac"] / rf["valu_pipe_frac"]),
        "PIPE_SFA": "%.2f" % rf["valu_pipe_frac_single_frame_affine"], "PIPE_MF": "%.2f" % rf["valu_pipe_frac_multi_frame"],
        "INSTR": "{:,}".format(int(rf.get("valu_instr_per_px", 0))),
        "HP_MS": "%.3f" % d["hotpath_ms_per_step"], "HP_REPLAY": "%.3f" % d["hotpath_graph_replay_ms_per_step"],
        "HP_RATIO": "%.2f" % d["hotpath_over_unit_launches"],
        "HP_RATIO_R": "%.2f" % (d["hotpath_graph_replay_ms_per_step"] / d["hotpath_unit_launches_ms_per_step"]),
        "HP_UNITS": "%.3f" % d["hotpath_unit_launches_ms_per_step"],
        "HP_IN_STEP": "%.3f" % d.get("hotpath_in_step_ms", 0.0),
        "HP_IN_STEP_RATIO": "%.3f" % d.get("hotpath_in_step_over_unit_launches", 0.0),
        "FUS_BWD": "%.2f" % fus_b, "SILOG": "%.2f" % silog, "AFFINE": "%.2f" % affine, "F8_TOTAL": "%.2f" % f8,
        "TRAFFIC": "%.2f" % rf["traffic_over_algorithmic"], "KERNEL_TABLE": "\n".join(rows), "NGPU": ngpu,
        "PIN_C2": pin(hpin / d["ms_per_step"] if hpin else None), "PIN_C3": pin(oc["C3"].get("pinned_eager_over_unpinned")),
        "PIN_C4": pin(oc["C4"].get("pinned_eager_over_unpinned")), "PIN_C5": pin(oc["C5"].get("pinned_eager_over_unpinned")),
    }
    path = os.path.join(R, "DESIGN.md")
    s = open(os.path.join(R, "tools", "DESIGN.md.in")).read()
    missing = sorted(set(re.findall(r"@([A-Z_0-9]+)@", s)) - set(sub))
    if missing:
        sys.exit("no value for: " + ", ".join(missing))
    if "--check" in sys.argv:
        for k, v in sub.items():
            print(k, "=", v if k != "KERNEL_TABLE" else "(%d rows)" % len(rows))
        return
    for k, v in sub.items():
        s = s.replace("@%s@" % k, v)
    open(path, "w").write(s)


if __name__ == "__main__":
    main()
