# round 4: roofline of the build's own glue kernels inside the ResNet18 training step (bench.py's kernel leg) + their parity tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/${1:-gpurun_out/r04g}; mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py -x -q -k "${2:-maxpool or bias_act or reflect_pad or decoder or up2cat or fusion}" > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-hotpath-leg --also-configs none --no-graph-leg --no-pmc-leg --no-mfma-leg --no-host-leg > $O/bench.json 2> $O/bench.err
python - "$O/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("step", d["value"], "img/s", d["ms_per_step"], "ms")
ok = d.get("own_kernels", {})
print("own glue ms/step", ok.get("own_glue_ms_per_step"))
for k, v in ok.get("kernels", {}).items():
    print(f"{k:28s} n/step {v['launches_per_step']:6.1f} ms/step {v['ms_per_step']:7.3f} avg_us {v['avg_us']:8.1f} MB/launch {v['bytes_per_launch']/1e6:8.2f} GB/s {v['achieved']:8.1f} frac {v['frac']:.3f}")
PY
