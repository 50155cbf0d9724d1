mkdir -p gpurun_out
( timeout 600 bash tools/variants.sh run "fullsize_unit and C2" ) > gpurun_out/variants.log 2>&1
cat gpurun_out/variants.log | grep -v "^$" | tail -12
timeout 1500 bash tools/prof_backbones.sh
