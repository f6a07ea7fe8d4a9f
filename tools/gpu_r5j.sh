#!/bin/bash
R=/root/repo/gpurun_out/r5_j
mkdir -p $R
cd /root/repo
timeout 900 python bench.py > $R/bench.json 2> $R/bench.err
tail -c 400 $R/bench.err
python - <<'PY'
import json
r = json.loads([l for l in open('/root/repo/gpurun_out/r5_j/bench.json') if l.startswith('{')][-1])
print('value', r['value'], 'ms_per_step', r['ms_per_step'], 'roofline', r['roofline']['frac'])
s = r['sampler']
print('sampler', s['ms_per_batch'], s['value'], 'batched', s.get('batched'), 'hub', s.get('hub'))
print('c5', {k: v for k, v in r['c5'].items() if k != 'workload'})
print('f32', r['segment_matmul_f32']['exact'])
print('c4', r['c4']['rank0_launch'])
print('bwd', r['segment_matmul_backward'])
print('idx', r['index_sort']['ms'], 'scatter', r['scatter_sum']['ms'], r['scatter_sum']['frac'])
PY
cp gpurun_out/gpu_health.txt $R/ 2>/dev/null
