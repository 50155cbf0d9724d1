# round 3 experiment (negative): ATen native batch-norm kernels below a size threshold in networks/grouped.py::_FoldedBN
# (needed a temporary MVF_BN_NATIVE_BELOW switch there, removed again: MIOpen everywhere stays the fastest)
for thr in 0 2000000 8000000 40000000 1000000000; do
  MVF_BN_NATIVE_BELOW=$thr python bench.py --backbone DHRNet --no-cpu-baseline --no-hotpath-leg --also-configs none --no-graph-leg --no-pmc-leg --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DHRNet native_below=$thr', d['value'], d['ms_per_step'])"
done
for thr in 0 8000000 1000000000; do
  MVF_BN_NATIVE_BELOW=$thr python bench.py --no-cpu-baseline --no-hotpath-leg --also-configs none --no-graph-leg --no-pmc-leg --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ResNet18 native_below=$thr', d['value'], d['ms_per_step'])"
done
