# Recipe behind profiles/r06_*: run on the GPU box from the repo root (gpurun); writes gpurun_out/prof_r06/*.
# usage: bash tools/profile_r06.sh <tag> [bench.py args...]      e.g.  bash tools/profile_r06.sh ml20m
#        (tag s1m -> add: --workload s1m)
set -x
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
B="python $R/bench.py --only-headline --no-cpu-baseline --pass-streams 1 $*"   # serial passes: per-kernel durations without the overlap of consecutive passes
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/kt -- $B --steps 20 --warmup 5 > $OUT/r06_${TAG}_line_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_$TAG/fetch -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_$TAG/write -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/prof_$TAG/sq1 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/prof_$TAG/sq2 -- $B --steps 2 --warmup 1 > /dev/null 2>&1
cd $R
for k in kt:kernel_stats fetch:pmc_fetch_size write:pmc_write_size sq1:pmc_sq1 sq2:pmc_sq2; do
  python tools/summarize_rocprof.py /tmp/prof_$TAG/${k%%:*} $OUT/r06_${TAG}_${k##*:}.txt > /dev/null
done
python tools/summarize_rocprof.py --cross-check $OUT/r06_${TAG}_kernel_stats.txt $OUT/r06_${TAG}_pmc_sq1.txt
ls -la $OUT
