#!/bin/bash
# wide contractions through the general-shape kernel: matmul tests, fuzz, the sweep again
R=/root/repo/gpurun_out/r6_mmd
mkdir -p $R
cd /root/repo
timeout 1500 python -m pytest tests/test_matmul_gpu.py tests/test_matmul_gen_gpu.py tests/test_matmul_fuzz_gpu.py tests/test_stress_gpu.py tests/test_graph_capture_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1; tail -2 $R/pytest.txt
timeout 300 python tools/fuzz_matmul.py 60 21 > $R/fuzz.txt 2>&1; tail -1 $R/fuzz.txt | cut -c1-200
python tools/mm_shape_sweep.py 6000000 bf16 2>&1 | grep -v amdgpu | tee $R/sweep_bf16.txt
python tools/mm_shape_sweep.py 4000000 f32 2>&1 | grep -v amdgpu | tee $R/sweep_f32.txt
