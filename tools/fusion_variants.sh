# build variants of mvf_fusion.hip (only that file is recompiled):  bash tools/fusion_variants.sh build name:"-DX=1" ...
# run (GPU box): bash tools/fusion_variants.sh run  -> us of the level-0 anchor adjoint (ResNet18 pyramid, B 36) per variant
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/mono-vifi_amd/lib
C=$R/mono-vifi_amd/csrc
CF="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function"
if [ "$1" = build ]; then
  shift
  rm -rf $L/fvar_*; mkdir -p $L/varobj
  for f in mvf_geom mvf_photo mvf_unit_fb mvf_glue mvf_affine; do
    if [ ! -f $L/varobj/$f.o ] || [ $C/$f.hip -nt $L/varobj/$f.o ] || [ $C/mvf_common.hpp -nt $L/varobj/$f.o ] || [ $C/mvf_tile.hpp -nt $L/varobj/$f.o ] || [ $R/include/mvf_hotpath.h -nt $L/varobj/$f.o ]; then
      ( cd $C && /opt/rocm/bin/hipcc $CF -c $f.hip -o $L/varobj/$f.o ) &
    fi
  done
  wait
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
    d=$L/fvar_$name; mkdir -p $d
    ( cd $C && /opt/rocm/bin/hipcc $CF $flags -c mvf_fusion.hip -o $d/mvf_fusion.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libmvf_hotpath.so $d/mvf_fusion.o $L/varobj/mvf_geom.o $L/varobj/mvf_photo.o $L/varobj/mvf_unit_fb.o $L/varobj/mvf_glue.o $L/varobj/mvf_affine.o && rm $d/mvf_fusion.o && echo "$flags" > $d/flags.txt ) &
  done
  wait
  ls $L/fvar_*/libmvf_hotpath.so
else
  for d in $L/fvar_*; do
    for a in 0.3 6; do
      echo "$(basename $d) [$(cat $d/flags.txt)] $(MVF_HOTPATH_LIB=$d/libmvf_hotpath.so python $R/tools/anchor_probe.py $a 2>/dev/null)"
    done
  done
fi
