#!/bin/bash
# Round 4, closing session: whole GPU suite, the driver's bench command, rocprofv3 kernel stats of the same command, PMC passes on the final kernels.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) > gpurun_out/r04_pytest_gpu_summary.txt 2>&1; cat gpurun_out/r04_pytest_gpu_summary.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
( time timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_flags.json 2> gpurun_out/r04_bench.err ) 2>&1 | tail -3; tail -2 gpurun_out/r04_bench.err
python tools/bench_brief.py gpurun_out/r04_bench_driver_flags.json "driver flags"
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection --no-batch > $R/gpurun_out/r04_bench_under_rocprofv3.json 2> $R/gpurun_out/r04_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_rocprofv3_kernel_stats.csv && head -14 "$f" | cut -c1-160
SHAPES="qkv gate_up o_proj down flash" bash tools/gpu_pmc_r4.sh 2>&1 | tail -40
