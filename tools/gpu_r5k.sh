#!/bin/bash
R=/root/repo/gpurun_out/r5_k
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_sampler_batched_gpu.py tests/test_sampler_fuzz_gpu.py tests/test_biased_sampler_gpu.py tests/test_dist_helpers_gpu.py tests/test_rgcn_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -6 $R/pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, json, bench_sampler
r = bench_sampler.run(torch.device('cuda:0'), cpu_batches=0)
print(json.dumps({k: r[k] for k in ('ms_per_batch', 'value', 'batched', 'hub')}))
PY
python tools/profile_sampler.py 40 2>&1 | tail -1 | cut -c1-200
cp gpurun_out/gpu_health.txt $R/ 2>/dev/null; true
