# The round-6 evidence run (one gpurun call): GPU tests, bench line + detail, rocprofv3 summaries of the headline and S-1M commands
# (kernel trace + four --pmc passes each), solver timelines, Krylov-block-width tables, cold-path probe, scaling proxies.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
timeout 900 python bench.py --steps 20 --warmup 5 --detail > $O/bench_line.json 2> $O/bench_err.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
timeout 300 python bench.py > $O/bench_line_default_flags.json 2> $O/bench_err_default.txt
timeout 900 bash tools/profile_r06.sh ml20m > $O/prof_ml20m.log 2>&1
timeout 900 bash tools/profile_r06.sh s1m --workload s1m > $O/prof_s1m.log 2>&1
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python $R/tools/probes/solver_timeline.py run lanczos > $O/solver_timeline_run.txt 2>&1 )
python tools/probes/solver_timeline.py report /tmp/st > $O/solver_timeline_lanczos.txt 2>&1
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st0; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st0 -- python $R/tools/probes/solver_timeline.py run lanczos 16 ml20m 0 8 > $O/solver_timeline_sync_run.txt 2>&1 )
python tools/probes/solver_timeline.py report /tmp/st0 > $O/solver_timeline_lanczos_sync_looks.txt 2>&1
timeout 400 python tools/probes/krylov_block_probe.py ml20m 50 64 48 32 24 16 0 > $O/krylov_block_ml20m.txt 2>&1
timeout 400 python tools/probes/krylov_block_probe.py ml20m 100 128 64 32 16 0 > $O/krylov_block_ml20m_r100.txt 2>&1
timeout 600 python tools/probes/krylov_block_probe.py s1m 50 64 32 16 0 > $O/krylov_block_s1m.txt 2>&1
timeout 300 python tools/probes/rounded_step_probe.py ml20m 16 > $O/rounded_step_ml20m_b16.txt 2>&1
timeout 300 python tools/probes/rounded_step_probe.py ml20m 32 > $O/rounded_step_ml20m_b32.txt 2>&1
timeout 300 python tools/probes/rounded_step_probe.py s1m 16 > $O/rounded_step_s1m_b16.txt 2>&1
timeout 300 python tools/probes/cold_probe.py ml20m 50 > $O/cold_probe.txt 2>&1
timeout 300 python tools/probes/reindex_stages.py s1m > $O/reindex_stages_s1m.txt 2>&1
timeout 400 python tools/probes/scale_proxy2.py ml20m > $O/scaling_proxy_ml20m.json 2> $O/proxy_err_ml20m.txt
timeout 600 python tools/probes/scale_proxy2.py s1m > $O/scaling_proxy_s1m.json 2> $O/proxy_err_s1m.txt
cat $O/bench_line_default_flags.json
timeout 600 python tools/probes/shard_pass_modes.py ml20m 2 > $O/shard_pass_modes_ml20m.txt 2>&1
timeout 900 python tools/probes/shard_pass_modes.py s1m 2 > $O/shard_pass_modes_s1m.txt 2>&1
timeout 300 python tools/probes/pass_host_cost.py ml20m 8 > $O/pass_host_cost_ml20m.txt 2>&1
timeout 300 python tools/probes/recorded_handover.py ml20m 8 > $O/recorded_handover_ml20m.txt 2>&1
