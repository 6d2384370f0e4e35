export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/probes/krylov_block_probe.py ml20m 50 0 16:8:0 > gpurun_out/kb13_ml20m.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/stA -- python $R/tools/probes/solver_timeline.py run lanczos 16 ml20m 0 8 > $R/gpurun_out/timeline_run_lag0c.txt 2>&1)
python tools/probes/solver_timeline.py report /tmp/stA > gpurun_out/solver_timeline_lag0c.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/gputests13.txt 2>&1
tail -5 gpurun_out/gputests13.txt
