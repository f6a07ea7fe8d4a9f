#!/bin/bash
# A/B of two builds of one translation unit on the same box (here: the general-shape matmul kernel, K = 100); variants from tools/build_variant.sh
R=/root/repo/gpurun_out/r6_ab4
mkdir -p $R
cd /root/repo
timeout 1200 python -m pytest tests/test_matmul_gen_gpu.py tests/test_matmul_fuzz_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1; tail -1 $R/pytest.txt
cp pyg_lib_amd/libpyg_hip.so /tmp/new.so
for rep in 1 2 3; do
  cp /tmp/new.so pyg_lib_amd/libpyg_hip.so; echo "new $(python tools/gen_time.py 2>/dev/null | tail -1)" | tee -a $R/ab.txt
  cp pyg_lib_amd/libpyg_hip_oldgen.so pyg_lib_amd/libpyg_hip.so; echo "old $(python tools/gen_time.py 2>/dev/null | tail -1)" | tee -a $R/ab.txt
done
cp /tmp/new.so pyg_lib_amd/libpyg_hip.so
