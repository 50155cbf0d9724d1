#!/bin/bash
# Same-box A/B of the training step: tools/ab_step.sh "a:ENV=1" "b:ENV=0 @ --no-regroup" ...
# Each variant = "label:ENV=V ... @ bench flags"; runs every variant ROUNDS times alternating, as HIP graph steps
# (device time, host-independent) and eager.  Output: gpurun_out/ab_step.txt
ROUNDS=${ROUNDS:-2}
OUT=gpurun_out/ab_step.txt
mkdir -p gpurun_out; : > $OUT
COMMON="--no-cpu-baseline --no-hotpath-leg --no-pmc-leg --steps ${STEPS:-30} --warmup 8"
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    label=${v%%:*}; rest=${v#*:}
    envs=${rest%%@*}; flags=""
    case "$rest" in *@*) flags="${rest#*@}";; esac
    for mode in "--hip-graph" ""; do
      line=$(env $envs python bench.py $COMMON $flags $mode 2>/dev/null | tail -1)
      ms=$(python -c "import json,sys; print(json.loads(sys.argv[1])['ms_per_step'])" "$line" 2>/dev/null)
      echo "round $r $label ${mode:-eager} ms_per_step=$ms" | tee -a $OUT
    done
  done
done
