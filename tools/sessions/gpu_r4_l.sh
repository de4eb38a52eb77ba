#!/bin/bash
# Round 4, session L: config 5 at the reference's per-device batch (16 x 2048), config 3, 13B headline.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --workload config5 --train-batch 4 --train-seq 2048 --steps 2 --warmup 1 > gpurun_out/r04_bench_config5_4x2048.json 2> gpurun_out/r04_l.err || tail -5 gpurun_out/r04_l.err
timeout 1200 python bench.py --workload config5 --train-batch 16 --train-seq 2048 --steps 2 --warmup 1 > gpurun_out/r04_bench_config5_16x2048.json 2> gpurun_out/r04_l.err || tail -5 gpurun_out/r04_l.err
timeout 900 python bench.py --workload config3 --steps 3 --warmup 1 > gpurun_out/r04_bench_config3.json 2> gpurun_out/r04_l.err || tail -5 gpurun_out/r04_l.err
python - <<'P'
import json
for f in ("r04_bench_config5_4x2048", "r04_bench_config5_16x2048", "r04_bench_config3"):
    try:
        for l in open(f"gpurun_out/{f}.json"):
            try: d = json.loads(l)
            except Exception: continue
            if "value" in d: print(f, {k: d.get(k) for k in ("value", "ms_per_step", "forward_ms", "forward_backward_ms", "optimizer_ms", "linear_tflops_in_fwd_bwd", "hbm_GB", "prefill_ms")})
    except Exception as e: print(f, "ERR", e)
P
