mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_trainer_gpu.py -k "hip_graph or checkpoint_roundtrip or fused_units" -x -q 2>&1 | grep -v "^20[0-9][0-9]-" | tail -70
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/b_train.err | tail -1 > gpurun_out/b_train.json
python -c "
import json; d=json.load(open('gpurun_out/b_train.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_us'], d.get('hip_graph_step'), d['hotpath_only']['value'])"
tail -3 gpurun_out/b_train.err
timeout 900 python bench.py --backbone DHRNet --hip-graph --no-cpu-baseline --no-hotpath-leg --steps 20 --warmup 5 2> gpurun_out/b_dhr_graph.err | tail -1 > gpurun_out/b_dhr_graph.json
cut -c1-300 gpurun_out/b_dhr_graph.json; tail -3 gpurun_out/b_dhr_graph.err
