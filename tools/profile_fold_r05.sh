# Counter passes over the fold-in kernels alone (tools/probes/fold_only.py): writes gpurun_out/prof_r05/r05_fold_<tag>_pmc_*.txt
# usage: bash tools/profile_fold_r05.sh <tag> [workload] [rank]
set -x
TAG=$1; WL=${2:-ml20m}; RANK=${3:-50}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/prof_r05; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_fold_$TAG
B="python $R/tools/probes/fold_only.py $WL $RANK 5"
run() { name=$1; shift; (cd $R && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/prof_fold_$TAG/$name -- $B > /dev/null 2>/tmp/fold_${name}_err.txt); tail -2 /tmp/fold_${name}_err.txt; (cd $R && python tools/summarize_rocprof.py /tmp/prof_fold_$TAG/$name $OUT/r05_fold_${TAG}_pmc_$name.txt > /dev/null); }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run ta TA_TA_BUSY_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
grep -hE "fold_q20_kernel|spmm_csr_groups_kernel<float, (4|16)" $OUT/r05_fold_${TAG}_pmc_*.txt | cut -c1-200
