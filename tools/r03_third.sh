# round 3, third GPU pass: full GPU suite (incl. the full-shape steps and the exact collective counts),
# hot path after the launch-structure fixes (kernel stats), the default bench line with its wall time,
# MIOpen steering experiments (solver families switched off, own find-db each).
# usage (GPU box): bash tools/r03_third.sh
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
ulimit -c 0
( time python -m pytest tests -m gpu -x -q --durations=12 ) > $O/gputest.log 2>&1; tail -25 $O/gputest.log
s=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? $(( $(date +%s) - s )) s" | tee $O/bench_default.time
python bench.py --workload hotpath --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/hp_smooth.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp -- python $R/bench.py --workload hotpath --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $R
cp $(ls $O/hp/*/*kernel_stats.csv | head -1) $O/r03_hotpath_kernel_stats.csv; rm -rf $O/hp
head -8 $O/r03_hotpath_kernel_stats.csv | cut -c1-150
# MIOpen steering: ResNet18 step with one solver family switched off (fresh find-db per experiment)
steer() { n=$1; shift; mkdir -p /tmp/mdb_$n; s=$(date +%s)
  env MIOPEN_USER_DB_PATH=/tmp/mdb_$n "$@" python bench.py --no-cpu-baseline --no-hotpath-leg --also-configs none --steps 20 --warmup 5 2> $O/steer_$n.err | tail -1 > $O/steer_$n.json
  python -c "import json; d=json.load(open('$O/steer_$n.json')); print('steer $n', d['value'], 'img/s', d['ms_per_step'], 'ms/step', $(( $(date +%s) - s )), 's wall')" | tee -a $O/steer.log; }
s=$(date +%s); python bench.py --no-cpu-baseline --no-hotpath-leg --also-configs none --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/steer_shipped_db.json
python -c "import json; d=json.load(open('$O/steer_shipped_db.json')); print('steer shipped_db', d['value'], 'img/s', d['ms_per_step'], 'ms/step', $(( $(date +%s) - s )), 's wall')" | tee -a $O/steer.log
steer no_winograd MIOPEN_DEBUG_CONV_WINOGRAD=0
steer no_implicit_gemm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0
steer no_winograd_no_direct MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_DIRECT=0
