#!/bin/bash
# round 5, GPU pass A: the atomic-free weight gradient (tests + bench legs)
R=/root/repo/gpurun_out/r5_a
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_matmul_gpu.py tests/test_matmul_gen_gpu.py tests/test_stress_gpu.py tests/test_rgcn_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -5 $R/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > $R/bench.json 2> $R/bench.err
tail -c 600 $R/bench.err
python - <<'PY'
import json
r = json.loads([l for l in open('/root/repo/gpurun_out/r5_a/bench.json') if l.startswith('{')][-1])
for k in ('segment_matmul_backward', 'segment_k100_backward', 'c4', 'grouped_mixed'):
    print(k, json.dumps(r.get(k))[:600])
print('roofline', r['roofline'])
PY
