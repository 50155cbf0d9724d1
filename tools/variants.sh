# build / measure variants of the unit kernel (only mvf_unit_fb.hip is recompiled per variant).
#   build (here):   bash tools/variants.sh build name1:"-DX -DY" name2:"-DZ" name3@other_source.hip:"-DX" ...
#                   (name@file: the variant is compiled from csrc/file instead of mvf_unit_fb.hip, e.g. the previous
#                    round's kernel written there by `git show REV:mono-vifi_amd/csrc/mvf_unit_fb.hip`)
#   run (GPU box):  bash tools/variants.sh run OUTDIR [parity]   -> OUTDIR/variants.csv
#                   per variant: one rocprofv3 counter pass over 4 hot-path steps (VALU instructions, busy cycles,
#                   GPU cycles per launch of k_unit_fb) and one un-profiled hot-path bench (us per launch by HIP events)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/mono-vifi_amd/lib
C=$R/mono-vifi_amd/csrc
CF="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function"
if [ "$1" = build ]; then
  shift
  rm -rf $L/var_*; mkdir -p $L/varobj
  for f in mvf_geom mvf_photo mvf_fusion mvf_glue mvf_affine; do
    if [ ! -f $L/varobj/$f.o ] || [ $C/$f.hip -nt $L/varobj/$f.o ] || [ $C/mvf_common.hpp -nt $L/varobj/$f.o ] || [ $C/mvf_tile.hpp -nt $L/varobj/$f.o ]; then
      ( cd $C && /opt/rocm/bin/hipcc $CF -c $f.hip -o $L/varobj/$f.o ) &
    fi
  done
  wait
  n=0
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
    src=mvf_unit_fb.hip
    case "$name" in *@*) src=${name#*@}; name=${name%%@*};; esac
    d=$L/var_$name; mkdir -p $d
    ( cd $C && /opt/rocm/bin/hipcc $CF $flags -c $src -o $d/mvf_unit_fb.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libmvf_hotpath.so $d/mvf_unit_fb.o $L/varobj/*.o && rm $d/mvf_unit_fb.o && echo "$flags" > $d/flags.txt ) &
    n=$((n+1)); if [ $((n % 6)) = 0 ]; then wait; fi
  done
  wait
  ls $L/var_*/libmvf_hotpath.so
else
  O=$R/${2:-gpurun_out/var}; mkdir -p $O; : > $O/variants.csv
  cd /tmp && export TMPDIR=/tmp
  echo "variant,flags,us_per_launch_events,valu_instr_per_px,valu_busy,gpu_cycles_per_launch,wave_active,wave_wait_any,wave_wait_inst,lds_idx_active_per_launch,lds_bank_conflict_per_launch,lds_conflict_share,parity" >> $O/variants.csv
  for d in $L/var_*; do
    n=$(basename $d | sed s/var_//); export MVF_HOTPATH_LIB=$d/libmvf_hotpath.so
    par=""
    if [ -n "$3" ]; then par=$(cd $R && python -m pytest tests/test_hip_parity.py -q -x -k "$3" 2>&1 | tail -1 | tr ',' ';'); fi
    us=$(python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline --no-pmc-leg $VBENCH 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_us'])")
    rm -rf $O/pmc_$n
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/bench.py --workload hotpath --steps 4 --warmup 2 --no-cpu-baseline --no-pmc-leg $VBENCH > /dev/null 2>&1 || true
    python - "$O/pmc_$n" "$n" "$(cat $d/flags.txt)" "$us" "$par" >> $O/variants.csv <<'PY'
import csv, glob, sys
d, name, flags, us, par = sys.argv[1:6]
acc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_unit_fb" in row["Kernel_Name"]:
            a = acc.setdefault(row["Counter_Name"], [0.0, 0]); a[0] += float(row["Counter_Value"]); a[1] += 1
g = {k: v[0] / v[1] for k, v in acc.items()}
px = float(__import__('os').environ.get('VUNITS', '4.5')) * 12 * 192 * 640     # mean units per launch of the 6 + 3 step
if g:
    quad = g["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0
    print(f"{name},{flags},{us},{g['SQ_INSTS_VALU']*64/px:.0f},{g['SQ_ACTIVE_INST_VALU']/quad:.3f},{g['GRBM_GUI_ACTIVE']/8:.0f},"
          f"{g['SQ_ACTIVE_INST_ANY']/g['SQ_WAVE_CYCLES']:.3f},{g['SQ_WAIT_ANY']/g['SQ_WAVE_CYCLES']:.3f},{g['SQ_WAIT_INST_ANY']/g['SQ_WAVE_CYCLES']:.3f},{g.get('SQ_LDS_IDX_ACTIVE',0):.0f},{g.get('SQ_LDS_BANK_CONFLICT',0):.0f},{g.get('SQ_LDS_BANK_CONFLICT',0)/max(g.get('SQ_LDS_IDX_ACTIVE',1),1):.3f},{par}")
else:
    print(f"{name},{flags},{us},,,,,,,,,,{par}")
PY
    rm -rf $O/pmc_$n
  done
  cut -c1-220 $O/variants.csv
fi
