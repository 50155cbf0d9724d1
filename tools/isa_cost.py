#!/usr/bin/env python3
"""Cost-weighted static ISA profile of k_unit_fb<2,false> per phase (itemise the kernel).

    python tools/isa_cost.py [--analysis 1|2|3] [--flags "-DX"] [--out profiles/r04_isa_cost_unit_fb.txt]

Builds the analysis form of the kernel (-DMVF_PHASE_MARKERS -DMVF_ANALYSIS=k: the run-time switches frozen to one
hot configuration on an inner tile, so the compiler drops the branches never taken and -- the phases being
straight-line code with unrolled loops -- the static stream is the executed one up to the wave-uniform skips of
idle rounds), cuts it at the phase markers and prices every VALU instruction with the issue cost MEASURED on the
MI355X by tools/valu_ubench.hip (profiles/r04_valu_issue_cost.txt; ns per wave-instruction per SIMD at 4 waves
per SIMD):

    1.0  v_add/sub/mul/fma/fmac_f32, v_mov_b32, v_and/or/xor_b32, v_add/sub_u32, v_lshrrev_b32 (+ literal / abs forms)
    1.7  DPP forms, v_cmp*, v_cndmask, v_min/max (f32, i32), v_med3/max3/min3, v_cvt*, v_floor/fract/rndne,
         v_mul_u32_u24, v_mad_u32_u24, v_mul_lo/hi, v_lshlrev, v_lshl_add, v_add3, v_bfe/bfi, v_and_or, v_div_*,
         v_readfirstlane, anything with an SGPR source operand is NOT re-priced (measured 1.7 in isolation, hidden in mixes)
    1.8 / 2.0 / 2.2   v_pk_add / v_pk_mul / v_pk_fma _f32 (packed fp32 costs two plain instructions: no free lunch)
    3.4  v_rcp/rsq/sqrt/exp/log/sin/cos_f32

Per lane a workgroup of 256 lanes produces 30 x 14 = 420 output pixels: per-pixel figures = per-lane x 256 / 420.
"""
import argparse
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S"]

FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32",
        "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32",
        "v_not_b32", "v_mov_b64", "v_accvgpr")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def cost(m, line):
    if not m.startswith("v_"):
        return 0.0, "non-valu"
    if m.startswith("v_pk_fma"):
        return 2.2, "packed"
    if m.startswith("v_pk_mul"):
        return 2.0, "packed"
    if m.startswith("v_pk_"):
        return 1.8, "packed"
    if any(m.startswith(t) for t in TRANS):
        return 3.4, "transcendental"
    if "dpp" in m or "row_sh" in line or "row_bcast" in line or "quad_perm" in line or "wave_sh" in line:
        return 1.7, "dpp"
    base = re.sub(r"_e(32|64)$", "", m)
    if any(base.startswith(f) for f in FULL):
        return 1.0, "full-rate"
    if base.startswith("v_cmp") or base.startswith("v_cndmask"):
        return 1.7, "compare/select"
    return 1.7, "half-rate other"


def build(flags):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [HIPCC] + BASE + flags + [os.path.join(ROOT, "mono-vifi_amd", "csrc", "mvf_unit_fb.hip"), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_body(asm, kernel):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(kernel), l))
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return body


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--analysis", type=int, default=1)
    ap.add_argument("--flags", default="")
    ap.add_argument("--kernel", default="k_unit_fbILi2ELb0")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    flags = ["-DMVF_PHASE_MARKERS", f"-DMVF_ANALYSIS={a.analysis}"] + a.flags.split()
    body = kernel_body(build(flags), a.kernel)
    phase = "0_prologue"
    per = collections.OrderedDict()
    top = collections.defaultdict(collections.Counter)
    ended = False
    # Blocks the compiler laid out behind the first s_endpgm: the IEEE fallbacks of the guarded fast divides (never run:
    # they hold v_div_scale / v_div_fixup) -- and, since round 6, HOT blocks it moved there by branch weight (the tap
    # adjoint of the second position round of phase 7): those are priced with the phase they belong to.
    end_at = next(i for i, l in enumerate(body) if l.strip().startswith("s_endpgm"))
    block_phase, cur, has_div = {}, None, False
    for i, l in enumerate(body[end_at + 1:], end_at + 1):
        t = l.strip()
        if re.match(r"^\.LBB\S*:", t):
            if cur is not None:
                block_phase[cur] = "cold_out_of_line" if has_div else "7_8_adjoint_smooth"
            cur, has_div = i, False
        if t.startswith(("v_div_scale", "v_div_fixup", "v_div_fmas")):
            has_div = True
    if cur is not None:
        block_phase[cur] = "cold_out_of_line" if has_div else "7_8_adjoint_smooth"
    for i, l in enumerate(body):
        t = l.strip()
        m = re.match(r"; MVF_PHASE (\S+)", t)
        if m:
            phase = m.group(1)
            continue
        if ended and i in block_phase:
            phase = block_phase[i]
        elif ended and phase not in ("cold_out_of_line", "7_8_adjoint_smooth"):
            phase = "cold_out_of_line"
        if t.startswith("s_endpgm"):
            ended = True
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        mn = t.split()[0]
        c, cls = cost(mn, t)
        p = per.setdefault(phase, collections.Counter())
        p["instr"] += 1
        if mn.startswith("v_"):
            p["valu"] += 1
            p["cost"] += c
            p["cls_" + cls] += 1
            p["cost_" + cls] += c
            top[phase][re.sub(r"_e(32|64)$", "", mn)] += 1
        elif mn.startswith("ds_"):
            p["lds"] += 1
        elif mn.startswith(("global_", "buffer_", "flat_")):
            p["vmem"] += 1
        elif mn == "s_barrier":
            p["barrier"] += 1
        elif mn == "s_waitcnt":
            p["waitcnt"] += 1
        elif mn == "s_nop":
            p["nop"] += 1
        elif mn.startswith("s_"):
            p["salu"] += 1
    names = {1: "single-frame launch (identity candidates evaluated, in-kernel noise)", 2: "multi-frame launch (identity maps handed over)",
             3: "affine launch (identity candidates + mask_rec)"}
    lines = [f"k_unit_fb<2,false>, static analysis build -DMVF_ANALYSIS={a.analysis} ({names.get(a.analysis)}), inner tile, flags '{a.flags}'",
             "VALU cost = sum over instructions of the measured issue cost (ns per wave-instruction per SIMD at 4 waves/SIMD, tools/valu_ubench.hip);",
             "per output pixel = per lane x 256 / 420.  Phases are straight-line code executed once per lane except (per-pixel figures scaled):",
             "3_warp x 10/12 and 7_8 x 7/8 (wave-rounds with live positions), 2_identity x 3 (rolled channel loop), cold_out_of_line x 0",
             "(IEEE fallbacks of the guarded fast divides behind s_endpgm).  The compiler moves arithmetic across the markers (they only order",
             "memory operations): 4_ssim_warped's arithmetic shows up under 5a / 5b -- read 4 + 5a + 5b together.",
             "",
             f"{'phase':22s} {'VALU':>6s} {'packed':>7s} {'full':>6s} {'cmp/sel':>8s} {'dpp':>5s} {'trans':>6s} {'other':>6s} {'cost ns':>8s} {'instr/px':>9s} {'cost/px':>8s} {'share':>6s}   LDS VMEM SALU bar"]
    # executed / static: the warp body runs for 10 of 12 wave-rounds (612 positions), the adjoint body for 7 of 8 (420
    # positions); the identity pass (and, when the maps are handed over, the target-statistics pass) is a ROLLED loop
    # over the three channels (#pragma unroll 1): static body x 3; the out-of-line IEEE fallbacks are (almost) never run
    scale = {"3_warp": 10.0 / 12.0, "7_8_adjoint_smooth": 7.0 / 8.0, "2_identity": 3.0, "cold_out_of_line": 0.0}
    tot_cost = sum(p["cost"] * scale.get(k, 1.0) for k, p in per.items())
    tot_valu = sum(p["valu"] * scale.get(k, 1.0) for k, p in per.items())
    for k, p in per.items():
        sc = scale.get(k, 1.0)
        lines.append(f"{k:22s} {p['valu']:6d} {p['cls_packed']:7d} {p['cls_full-rate']:6d} {p['cls_compare/select']:8d} {p['cls_dpp']:5d} "
                     f"{p['cls_transcendental']:6d} {p['cls_half-rate other']:6d} {p['cost']:8.0f} {p['valu'] * sc * 256 / 420:9.0f} "
                     f"{p['cost'] * sc * 256 / 420:8.0f} {100 * p['cost'] * sc / tot_cost:5.1f}%   {p['lds']:3d} {p['vmem']:4d} {p['salu']:4d} {p['barrier']:3d}")
    lines.append(f"{'total (scaled)':22s} {tot_valu:6.0f} {'':7s} {'':6s} {'':8s} {'':5s} {'':6s} {'':6s} {tot_cost:8.0f} {tot_valu * 256 / 420:9.0f} {tot_cost * 256 / 420:8.0f}")
    lines.append("")
    for k in per:
        lines.append(f"{k}: " + ", ".join(f"{m} {n}" for m, n in top[k].most_common(14)))
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
