# time each kernel variant built by `tools/variants.sh build ...` with the hot-path bench (no parity run)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for d in $R/mono-vifi_amd/lib/var_*; do
  n=$(basename $d)
  export MVF_HOTPATH_LIB=$d/libmvf_hotpath.so
  for dm in ${DISPS:-smooth}; do
    python $R/bench.py --workload hotpath --disp $dm --steps 20 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']['unit_fwdbwd']; print('$n $dm', 'launch_us', k['avg_us'], 'us_per_unit', k['us_per_unit'], 'frac', k['frac'], 'img/s', d['value'], 'graph_ms', (d.get('hip_graph_replay') or {}).get('ms_per_step'))"
  done
done
