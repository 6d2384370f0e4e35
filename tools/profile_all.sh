# Recipe behind profiles/: run on the GPU box from the repo root (GRAFT_REPO_ROOT set by gpurun); writes gpurun_out/summary/*
set -x
mkdir -p gpurun_out/summary
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/summary/r2_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/summary/r2_smoke.log
timeout 600 python bench.py > gpurun_out/summary/r2_bench_line.json 2> gpurun_out/summary/r2_bench.err
timeout 600 python bench.py --workload ml20m > gpurun_out/summary/r2_bench_line_ml20m.json 2>> gpurun_out/summary/r2_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-prune --steps 3 --warmup 1 > gpurun_out/summary/r2_bench_line_noprune.json 2>> gpurun_out/summary/r2_bench.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/summary/r2_line_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof/fetch -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof/write -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof/sq1 -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/prof/sq2 -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
python tools/summarize_rocprof.py /tmp/prof/kt gpurun_out/summary/r2_kernel_stats.txt
python tools/summarize_rocprof.py /tmp/prof/fetch gpurun_out/summary/r2_pmc_fetch_size.txt
python tools/summarize_rocprof.py /tmp/prof/write gpurun_out/summary/r2_pmc_write_size.txt
python tools/summarize_rocprof.py /tmp/prof/sq1 gpurun_out/summary/r2_pmc_sq1.txt
python tools/summarize_rocprof.py /tmp/prof/sq2 gpurun_out/summary/r2_pmc_sq2.txt
ls -la gpurun_out/summary
timeout 300 python bench.py --no-cpu-baseline --no-norm-order > gpurun_out/summary/r2_bench_line_popularity_order.json 2>> gpurun_out/summary/r2_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/ktm -- python $R/bench.py --no-cpu-baseline --workload ml20m > /dev/null 2>&1
cd $R
python tools/summarize_rocprof.py /tmp/prof/ktm gpurun_out/summary/r2_ml20m_kernel_stats.txt
