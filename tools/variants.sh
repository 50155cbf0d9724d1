# Build kernel variants (different -D switches) into mono-vifi_amd/lib/var_<name>/ and, on the
# GPU box, time each with the hot-path bench (+ a quick parity subset).
#   build (here):  bash tools/variants.sh build name1:"-DX -DY" name2:"-DZ" ...
#   run (GPU box): bash tools/variants.sh run [pytest -k expr]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = build ]; then
  shift
  rm -rf $R/mono-vifi_amd/lib/var_*
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    d=$R/mono-vifi_amd/lib/var_$name; mkdir -p $d
    ( cd $R/mono-vifi_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
        -fvisibility=hidden -Wno-unused-function $flags -o $d/libmvf_hotpath.so mvf_geom.hip mvf_photo.hip mvf_unit_fb.hip mvf_fusion.hip mvf_glue.hip mvf_affine.hip ) &
  done
  wait
  ls $R/mono-vifi_amd/lib/var_*/libmvf_hotpath.so
else
  K=${2:-"fullsize_unit and C2"}
  for d in $R/mono-vifi_amd/lib/var_*; do
    n=$(basename $d)
    export MVF_HOTPATH_LIB=$d/libmvf_hotpath.so
    t=$(python -m pytest $R/tests/test_hip_parity.py -q -x -k "$K" 2>&1 | tail -1)
    for dm in smooth noise; do
      python $R/bench.py --workload hotpath --disp $dm --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']['unit_fwdbwd']; print('$n $dm', k['avg_us'], k['frac'], d['value'], '| $t')"
    done
  done
fi
