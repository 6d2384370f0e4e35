mkdir -p gpurun_out/r32
timeout 150 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r32/pytest.log
timeout 150 python bench.py > gpurun_out/r32/bench_line.json 2> gpurun_out/r32/bench.err
tail -4 gpurun_out/r32/pytest.log; python -c "
import json; d=json.load(open('gpurun_out/r32/bench_line.json')); print(d['value'], d['ms_per_step'], d['score'], d['roofline']['frac'], d['cpu_baseline']['gpu_vs_cpu_identical_rows'])"
