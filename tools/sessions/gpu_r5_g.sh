#!/bin/bash
# Round 5, session G: the whole GPU suite on the current tree, the PMC passes, rocprofv3 kernel stats of the headline command, the 13B line and config 3
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/r05_pytest_gpu_summary.txt 2>&1; tail -8 gpurun_out/r05_pytest_gpu_summary.txt
for f in full_depth_llava15_7b full_depth_llava15_13b full_depth_fp32_llava15_7b; do cp gpurun_out/$f.json gpurun_out/r05_$f.json 2>/dev/null; done
bash tools/gpu_prof.sh 2>&1 | grep -E "lmx::|rocprof rc" | head -24
cp gpurun_out/prof/*kernel_stats.csv gpurun_out/r05_rocprofv3_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_bench.json gpurun_out/r05_bench_under_rocprofv3.json 2>/dev/null
bash tools/gpu_pmc_r5.sh 2>&1 | tail -40
timeout 600 python bench.py --model llava15_13b --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection > gpurun_out/r05_bench_13b_tp1.json 2>> gpurun_out/r05_g.err; python tools/bench_brief.py gpurun_out/r05_bench_13b_tp1.json "13B" | head -1
timeout 600 python bench.py --workload config3 --steps 2 --warmup 1 > gpurun_out/r05_bench_config3.json 2>> gpurun_out/r05_g.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_config3.json')); print('config3', d.get('value'), d.get('unit'))"
tail -3 gpurun_out/r05_g.err
