"""Refresh profiles/hbm_traffic.json (the static PMC-derived values bench.py attaches to its roofline
object) from a round's PMC summaries.
usage: python tools/update_profiles_json.py gpurun_out/r02/r02_pmc_valu.csv gpurun_out/r02/r02_pmc_fetch_write.csv "round 2" PIXELS"""
import csv
import json
import sys

valu_csv, traffic_csv, tag, px = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
KEY, MATCH = "k_unit_fb<2>", "k_unit_fb<2, false>"
c = {}
for f in (valu_csv, traffic_csv):
    for r in csv.DictReader(open(f)):
        if MATCH in r["kernel"]:
            c[r["counter"]] = float(r["mean_per_launch"])
path = "profiles/hbm_traffic.json"
j = json.load(open(path))
j["_captured"] = tag
j[KEY] = int(round(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024))
simd_quads = c["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
j.setdefault("_valu", {})[KEY] = {
    "valu_busy": round(c["SQ_ACTIVE_INST_VALU"] / simd_quads, 3),
    "valu_instr_per_px": int(round(c["SQ_INSTS_VALU"] * 64.0 / px)),
    "wave_active": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    "wave_wait_memory_or_barrier": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    "wave_wait_issue": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    "lds_bank_conflict_share": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3),
    "source": f"{valu_csv.split('/')[-1]}, {traffic_csv.split('/')[-1]} ({tag})"}
j["_note_" + tag.replace(" ", "")] = (
    f"{KEY}: FETCH_SIZE {c['FETCH_SIZE']:.1f} KB x2 (gfx950 note) + WRITE_SIZE {c['WRITE_SIZE']:.1f} KB per launch at B12 640x192")
json.dump(j, open(path, "w"), indent=1)
print(json.dumps({KEY: j[KEY], "valu": j["_valu"][KEY]}, indent=1))
