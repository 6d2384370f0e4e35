# The round-5 evidence run (one gpurun call): bench line + detail, rocprofv3 summaries of the headline and S-1M commands (kernel
# trace + four --pmc passes each), counters of the fold-in kernels alone, solver timeline, cold-path probe, scaling proxies.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 --detail > $O/bench_line.json 2> $O/bench_err.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
timeout 300 python bench.py > $O/bench_line_default_flags.json 2> $O/bench_err_default.txt
timeout 900 bash tools/profile_r05.sh ml20m > $O/prof_ml20m.log 2>&1
timeout 900 bash tools/profile_r05.sh s1m --workload s1m > $O/prof_s1m.log 2>&1
PK_WARM_UP=0 timeout 600 bash tools/profile_fold_r05.sh ml20m > $O/prof_fold.log 2>&1
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python $R/tools/probes/solver_timeline.py run lanczos > /dev/null 2>&1 )
python tools/probes/solver_timeline.py report /tmp/st > $O/solver_timeline_lanczos.txt 2>&1
timeout 300 python tools/probes/cold_probe.py ml20m 50 > $O/cold_probe.txt 2>&1
timeout 400 python tools/probes/scale_proxy2.py ml20m > $O/scaling_proxy_ml20m.json 2> $O/proxy_err_ml20m.txt
timeout 600 python tools/probes/scale_proxy2.py s1m > $O/scaling_proxy_s1m.json 2> $O/proxy_err_s1m.txt
cat $O/bench_line_default_flags.json
