#!/bin/bash
# Round-2 GPU session N: two-pass attention backward (no atomics): slices + step tests, config 5 again (1 x 1024, then 2 x 2048).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_train_step_gpu.py tests/test_train_slices_gpu.py -q -m gpu 2>&1 | tail -15
timeout 900 python bench.py --workload config5 --steps 2 --warmup 1 > $O/bench_config5.json 2> $O/bench_config5.err
tail -c 2500 $O/bench_config5.json; tail -3 $O/bench_config5.err | cut -c1-600
timeout 900 python bench.py --workload config5 --steps 2 --warmup 1 --train-batch 2 --train-seq 2048 > $O/bench_config5_2x2048.json 2> $O/bench_config5_2x2048.err
tail -c 2500 $O/bench_config5_2x2048.json; tail -3 $O/bench_config5_2x2048.err | cut -c1-600
