export TMPDIR=/tmp
python tools/probes/krylov_block_probe.py ml20m 100 16:10:0 16:14:0 0 > gpurun_out/kb11_ml20m_r100.txt 2>&1
python tools/probes/krylov_block_probe.py s1m 50 16:8:0 0 > gpurun_out/kb11_s1m.txt 2>&1
python tools/probes/krylov_block_probe.py ml20m 50 16:6:0 16:8:0 16:10:0 > gpurun_out/kb11_ml20m.txt 2>&1
