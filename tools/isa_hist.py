#!/usr/bin/env python3
"""Static ISA histogram of one gfx950 kernel ("commit an ISA histogram of
k_unit_fb<2,false> under profiles/").

    python tools/isa_hist.py [--kernel k_unit_fbILi2ELb0] [--src mono-vifi_amd/csrc/mvf_unit_fb.hip]
                             [--flags "-DX"] [--out profiles/r03_isa_hist_unit_fb.txt] [--blocks]

Compiles the translation unit to device assembly (hipcc --cuda-device-only -S, the flags of the
Makefile), cuts out the kernel's body and counts instructions by class.  Static counts: a rolled
loop body counts once, so this is a map of WHAT the kernel is made of (packed share, moves,
64-bit address arithmetic, selects, hazard nops, LDS and memory instructions), not of how often
each instruction executes -- the dynamic number is SQ_INSTS_VALU of the PMC pass
(profiles/r0x_pmc_valu.csv).  --blocks adds the per-basic-block table (label, instructions,
VALU, packed) so that loop bodies can be weighted by hand.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S"]


def classify(m):
    if m.startswith("v_pk_"):
        return "valu_packed"
    if m.startswith("v_mfma") or m.startswith("v_smfmac"):
        return "mfma"
    if m.startswith("v_"):
        return "valu"
    if m.startswith("s_waitcnt") or m.startswith("s_nop") or m.startswith("s_barrier"):
        return m.split("_e")[0]
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("global_") or m.startswith("buffer_") or m.startswith("flat_") or m.startswith("scratch_"):
        return "vmem"
    return "other"


def kernel_body(asm, kernel):
    lines = asm.splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^(_Z\S*%s\S*):" % re.escape(kernel), l):
            start = i
            name = l.split(":")[0]
            break
    if start is None:
        raise SystemExit(f"kernel matching '{kernel}' not found")
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.strip().startswith(".section"):
            break
        body.append(l)
    return name, body


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_unit_fbILi2ELb0")
    ap.add_argument("--src", default="mono-vifi_amd/csrc/mvf_unit_fb.hip")
    ap.add_argument("--flags", default="")
    ap.add_argument("--out", default=None)
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--asm", default=None, help="use an existing .s instead of compiling")
    a = ap.parse_args()
    if a.asm:
        asm = open(a.asm).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            cmd = [HIPCC] + BASE + a.flags.split() + [os.path.join(ROOT, a.src), "-o", out]
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            asm = open(out).read()
    name, body = kernel_body(asm, a.kernel)
    by_class, by_mn = collections.Counter(), collections.Counter()
    blocks, cur = [], ["entry", collections.Counter()]
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^\.LBB\S+:", t):
                blocks.append(cur)
                cur = [t.split(":")[0], collections.Counter()]
            continue
        m = t.split()[0]
        if m.endswith(":"):
            continue
        base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", m)
        if "dpp" in t and not m.endswith("dpp"):
            pass
        by_class[classify(m)] += 1
        by_mn[base + ("(dpp)" if ("row_sh" in t or "row_bcast" in t or "quad_perm" in t) else "")] += 1
        cur[1][classify(m)] += 1
    blocks.append(cur)
    total = sum(by_class.values())
    valu = by_class["valu"] + by_class["valu_packed"]
    meta = {}
    for key in (".vgpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size"):
        mm = re.search(r"%s:\s+(\d+)\n(?:.*\n)*?\s+\.name:\s+%s" % (re.escape(key), re.escape(name)), asm)
    rows = []
    rows.append(f"kernel {name}")
    rows.append(f"source {a.src}  flags '{a.flags}'")
    rows.append(f"static instructions {total}: VALU {valu} (packed v_pk_* {by_class['valu_packed']} = "
                f"{100.0 * by_class['valu_packed'] / max(valu, 1):.1f} % of VALU), SALU {by_class['salu']}, "
                f"LDS {by_class['lds']}, VMEM {by_class['vmem']}, s_waitcnt {by_class['s_waitcnt']}, "
                f"s_nop {by_class['s_nop']}, s_barrier {by_class['s_barrier']}")
    groups = {
        "v_mov (b32/b64, incl. dpp moves)": lambda k: k.startswith("v_mov") or k.startswith("v_accvgpr"),
        "v_lshl_add_u64 / v_add_co / v_addc (64-bit addresses)": lambda k: k.startswith("v_lshl_add_u64") or
            k.startswith("v_add_co") or k.startswith("v_addc") or k.startswith("v_mad_u64") or k.startswith("v_mad_i64"),
        "v_cndmask": lambda k: k.startswith("v_cndmask"),
        "v_cmp*": lambda k: k.startswith("v_cmp"),
        "integer VALU (add/mul/shift/min/max/and/or, 32-bit)": lambda k: re.match(
            r"v_(add|sub|mul|mad|lshl|lshr|ashr|and|or|xor|min|max|bfe|bfi|perm|alignbit)\w*_(u|i|b)\d+", k) is not None
            or k.startswith("v_add_nc") or k.startswith("v_sub_nc") or k.startswith("v_lshlrev") or k.startswith("v_and_b32"),
        "v_pk_fma_f32": lambda k: k.startswith("v_pk_fma_f32"),
        "v_pk_mul_f32": lambda k: k.startswith("v_pk_mul_f32"),
        "v_pk_add_f32": lambda k: k.startswith("v_pk_add_f32"),
        "v_fma_f32 / v_fmac_f32 (scalar)": lambda k: k.startswith("v_fma_f32") or k.startswith("v_fmac_f32"),
        "v_mul_f32 (scalar)": lambda k: k.startswith("v_mul_f32"),
        "v_add_f32 / v_sub_f32 (scalar)": lambda k: k.startswith("v_add_f32") or k.startswith("v_sub_f32") or k.startswith("v_subrev_f32"),
        "v_rcp / v_exp / v_log / v_sin / v_cos / v_sqrt (transcendental)": lambda k: re.match(
            r"v_(rcp|exp|log|sin|cos|sqrt|rsq)", k) is not None,
        "v_cvt*": lambda k: k.startswith("v_cvt"),
        "ds_read*/ds_load*": lambda k: k.startswith("ds_read") or k.startswith("ds_load"),
        "ds_write*/ds_store*": lambda k: k.startswith("ds_write") or k.startswith("ds_store"),
        "ds_bpermute / ds_swizzle": lambda k: k.startswith("ds_bpermute") or k.startswith("ds_swizzle") or k.startswith("ds_permute"),
        "global_load*": lambda k: k.startswith("global_load") or k.startswith("buffer_load"),
        "global_store*": lambda k: k.startswith("global_store") or k.startswith("buffer_store"),
    }
    rows.append("")
    rows.append(f"{'group':72s} {'count':>6s} {'% of all':>9s}")
    for gname, pred in groups.items():
        n = sum(v for k, v in by_mn.items() if pred(k))
        rows.append(f"{gname:72s} {n:6d} {100.0 * n / total:8.1f}%")
    rows.append("")
    rows.append("top mnemonics:")
    for k, v in by_mn.most_common(40):
        rows.append(f"  {k:40s} {v:6d}")
    if a.blocks:
        rows.append("")
        rows.append("basic blocks (label, instructions, VALU, packed, LDS, VMEM):")
        for lab, c in blocks:
            n = sum(c.values())
            if n >= 8:
                rows.append(f"  {lab:16s} {n:5d} {c['valu'] + c['valu_packed']:5d} {c['valu_packed']:5d} "
                            f"{c['lds']:4d} {c['vmem']:4d}")
    text = "\n".join(rows) + "\n"
    if a.out:
        with open(os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
