# Two more counter passes of the headline command (L1 / TA / L2): writes gpurun_out/prof_r04/r04_ml20m_pmc_tcp.txt and r04_ml20m_pmc_ta_tcc.txt
set -x
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/prof_r04; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_tcp
B="python $R/bench.py --only-headline --no-cpu-baseline --pass-streams 1 --steps 2 --warmup 1"
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d /tmp/prof_tcp/tcp -- $B > /dev/null 2>/tmp/tcp_err.txt; tail -3 /tmp/tcp_err.txt
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_tcp/ta -- $B > /dev/null 2>/tmp/ta_err.txt; tail -3 /tmp/ta_err.txt
cd $R
python tools/summarize_rocprof.py /tmp/prof_tcp/tcp $OUT/r04_ml20m_pmc_tcp.txt > /dev/null
python tools/summarize_rocprof.py /tmp/prof_tcp/ta $OUT/r04_ml20m_pmc_ta_tcc.txt > /dev/null
grep -E "score_candidates_kernel<4, 16, false, true, false>|spmm_csr_groups_kernel<float, 4" $OUT/r04_ml20m_pmc_tcp.txt $OUT/r04_ml20m_pmc_ta_tcc.txt | cut -c1-220
