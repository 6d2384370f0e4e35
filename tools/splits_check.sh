mkdir -p gpurun_out/splits
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/splits/pytest.log
timeout 100 python tools/probes/scale_proxy.py ml20m 2>/dev/null | tail -1 > gpurun_out/splits/proxy_ml20m.json
timeout 100 python bench.py --workload ml1m --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/splits/bench_ml1m.json
tail -4 gpurun_out/splits/pytest.log; cat gpurun_out/splits/proxy_ml20m.json; cut -c1-260 gpurun_out/splits/bench_ml1m.json
