# GPU box: time each built variant (mono-vifi_amd/lib/var_*) with the hot-path bench, smooth disparity only
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for n in ${@:-base}; do
  d=$R/mono-vifi_amd/lib/var_$n
  MVF_HOTPATH_LIB=$d/libmvf_hotpath.so timeout 120 python bench.py --workload hotpath --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']['unit_fwdbwd']; print('$n', k['avg_us'], k['frac'], (d.get('hip_graph_replay') or {}).get('value'))" | tee -a gpurun_out/variants_sched.log
done
