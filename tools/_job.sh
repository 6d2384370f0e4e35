export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/probes/krylov_block_probe.py ml20m 50 0 16 32 > gpurun_out/kb4_ml20m.txt 2>&1
python tools/probes/krylov_block_probe.py s1m 50 0 > gpurun_out/kb4_s1m.txt 2>&1
python tools/probes/krylov_block_probe.py ml20m 100 0 32 > gpurun_out/kb4_ml20m_r100.txt 2>&1
for B in 16; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/st$B -- python $R/tools/probes/solver_timeline.py run lanczos $B > $R/gpurun_out/timeline_run_$B.txt 2>&1)
  python tools/probes/solver_timeline.py report /tmp/st$B > gpurun_out/solver_timeline_b$B.txt 2>&1
done
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gputests4.txt 2>&1
tail -15 gpurun_out/gputests4.txt
python bench.py > gpurun_out/bench4.json 2> gpurun_out/bench4.err
tail -c 1500 gpurun_out/bench4.json
