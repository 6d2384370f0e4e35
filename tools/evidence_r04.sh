# The round-4 evidence run (one gpurun call): bench line + detail, rocprofv3 summaries, solver / fold-in / sweep probes, scaling proxy, GPU suite.
# The diagnostic sweep library is built beforehand (-DPK_SCORE_DIAG on score.hip) as polara_amd/libpolarahip_diag.so.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 --detail > $O/bench_line.json 2> $O/bench_err.txt
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
timeout 900 bash tools/profile_r04.sh ml20m > $O/prof_ml20m.log 2>&1
for a in "ml20m 50" "ml20m 100" "s1m 50" "ml1m 10"; do
  set -- $a
  timeout 300 python tools/probes/solver_methods.py $1 $2 > $O/solver_methods_$1_$2.txt 2>&1
done
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -- python $R/tools/probes/solver_timeline.py run lanczos > /dev/null 2>&1 )
python tools/probes/solver_timeline.py report /tmp/st > $O/solver_timeline_lanczos.txt 2>&1
timeout 300 python tools/probes/fold_head_probe.py ml20m 50 > $O/fold_head_probe_ml20m.txt 2>&1
POLARA_HIP_LIB=$R/polara_amd/libpolarahip_diag.so PK_FLOOR_DIAG=1 timeout 300 python tools/probes/sweep_floor.py ml20m > $O/sweep_floor_diag_ml20m.txt 2>&1
timeout 400 python tools/probes/scale_proxy2.py ml20m > $O/scaling_proxy_ml20m.json 2> $O/proxy_err.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
cat $O/bench_line.json
