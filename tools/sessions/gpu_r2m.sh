#!/bin/bash
# Round-2 GPU session M: train-step tests again (ZeRO tolerance), config 5 workload at 7B geometry (1 x 1024 positions, 2 steps).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_train_step_gpu.py tests/test_train_slices_gpu.py -q -m gpu 2>&1 | tail -5
timeout 900 python bench.py --workload config5 --steps 2 --warmup 1 > $O/bench_config5.json 2> $O/bench_config5.err
tail -c 3000 $O/bench_config5.json; tail -5 $O/bench_config5.err | cut -c1-600
