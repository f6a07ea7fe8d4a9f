#!/bin/bash
# Round-end evidence on the GPU box: bench lines, rocprofv3 kernel stats of the same command, PMC passes (own runs).
#   tools/profile_round.sh <out dir under gpurun_out/>
R=/root/repo/gpurun_out/$1
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
python /root/repo/bench.py > $R/bench_full_1.json 2> $R/bench_full_1.err
python /root/repo/bench.py > $R/bench_full_2.json 2> $R/bench_full_2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_bench -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 > $R/bench_under_rocprof.log 2>&1
# C2 only (no legs): the headline kernel's average must be reproducible from profiles/ alone -- the full bench's stats mix
# the segment_short leg into the same kernel name; same box, same options as the PMC passes below
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_c2 -o c2 -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sampler --no-legs > $R/c2_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/pmc_mm_$c -o p -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sampler --no-legs > $R/pmc_mm_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/pmc_ops_$c -o p -- python /root/repo/tools/pmc_targets.py > $R/pmc_ops_$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_samp -o samp -- python /root/repo/tools/profile_sampler.py 20 > $R/prof_samp.log 2>&1
# K batches per call: throughput (un-profiled) and how much the lanes' kernels overlap (kernel trace)
python /root/repo/tools/bench_sampler_batched.py 8 16 32 > $R/sampler_batched.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/trace_batched -o t -- python /root/repo/tools/bench_sampler_batched.py 16 > $R/trace_batched.log 2>&1
python /root/repo/tools/trace_overlap.py $(find $R/trace_batched -name "*kernel_trace.csv" | head -1) 1600 > $R/sampler_batched_overlap.txt 2>&1
rm -rf $R/trace_batched
python /root/repo/tools/clock_probe.py > $R/clock_probe.txt 2>&1
# C5 layer alone: atomic kernel + zero fill against the atomic-free grouped kernel (wall clock, then the kernels' own times)
PYTHONPATH=/root/repo python /root/repo/tools/rgcn_grouped_probe.py 50 > $R/c5_layer_probe.txt 2>&1
PYTHONPATH=/root/repo rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_layer -o layer -- python /root/repo/tools/rgcn_grouped_probe.py 20 > $R/c5_layer_under_rocprof.log 2>&1
# the same layer at F = 256 and in float32 (kernel times), and the per-workgroup balance of the sample
for a in "256" "128 f32"; do
  PYTHONPATH=/root/repo rocprofv3 --kernel-trace --stats --output-format csv -d "$R/prof_layer_${a// /_}" -o layer -- python /root/repo/tools/rgcn_grouped_probe.py 20 15,10 $a > "$R/c5_layer_${a// /_}_under_rocprof.log" 2>&1
done
PYTHONPATH=/root/repo python /root/repo/tools/rgcn_item_balance.py 768 > $R/c5_item_balance.txt 2>&1
rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showperflevel > $R/rocm_smi.txt 2>&1
cp /root/repo/gpurun_out/gpu_health.txt $R/gpu_health.txt 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 /root/repo/bench.py --gpus 2 --debug-one-device --no-cpu-baseline > $R/bench_2rank_debug.json 2> $R/bench_2rank_debug.err
