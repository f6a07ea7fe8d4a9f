#!/bin/bash
# trimmed layer tests + rgcn tests, host timeline of the C3 sampler, c5 leg
R=/root/repo/gpurun_out/r6_u
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_rgcn_csc_gpu.py tests/test_rgcn_grouped_gpu.py tests/test_rgcn_gpu.py tests/test_graph_capture_gpu.py tests/test_deterministic_gpu.py -m gpu -x -q > $R/pytest_rgcn.txt 2>&1
echo "pytest rc=$?" >> $R/pytest_rgcn.txt
tail -8 $R/pytest_rgcn.txt
python tools/trace_sampler_host.py 2> $R/host_trace.txt > /dev/null
tail -12 $R/host_trace.txt
python - > $R/c5.json 2> $R/c5.err <<'PY'
import json, torch, bench_legs
print(json.dumps(bench_legs.leg_c5(torch.device('cuda:0'))))
PY
tail -c 2500 $R/c5.json; tail -3 $R/c5.err
