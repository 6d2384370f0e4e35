export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python tools/probes/krylov_block_probe.py ml20m 50 0 16 32 > gpurun_out/kb5_ml20m.txt 2>&1
python tools/probes/krylov_block_probe.py s1m 50 0 > gpurun_out/kb5_s1m.txt 2>&1
python tools/probes/krylov_block_probe.py ml20m 100 0 32 > gpurun_out/kb5_ml20m_r100.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/gputests5.txt 2>&1
tail -5 gpurun_out/gputests5.txt
python bench.py > gpurun_out/bench5.json 2> gpurun_out/bench5.err
tail -c 600 gpurun_out/bench5.json
