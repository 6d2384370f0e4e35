set -x
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final/smoke.log
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/pytest.log
timeout 300 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err
timeout 200 python bench.py --workload ml20m > gpurun_out/final/bench_line_ml20m.json 2>> gpurun_out/final/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/line_under_rocprof.json 2>/dev/null
cd $R
python tools/summarize_rocprof.py /tmp/prof/kt gpurun_out/final/kernel_stats.txt
tail -2 gpurun_out/final/smoke.log; cat gpurun_out/final/pytest.log; cut -c1-300 gpurun_out/final/bench_line.json
