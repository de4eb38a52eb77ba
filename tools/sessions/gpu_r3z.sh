#!/bin/bash
# Round-3 GPU session Z: M-tail order of the un-split ping-pong GEMM (128-row halves for the last round of full tiles) — parity tests, interleaved A/B on the
# 7B gate|up shape (variants 35 plain / 37 M-tail / 36 K-halves), then the prefill in situ with the automatic rule off and on.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3z; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gemm8p_gpu.py -q -x -k "m_tail or tail_split_order" 2>&1 | tail -5 | tee $O/pytest.txt
timeout 200 python tools/mb_gemm_variants.py "1087,22016,4096;1087,12288,4096;2048,22016,4096" "35,37,36" 7 2>&1 | tee $O/ab.jsonl
for v in 0 1; do
  LMX_GEMM8P_MTAIL=$v timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection 2>&1 | tail -1 > $O/bench_mtail$v.json
  python - <<PY
import json
d=json.load(open("$O/bench_mtail$v.json"))
print("MTAIL=$v", "value", round(d["value"],1), "prefill_ms", round(d["prefill_ms"],2), "frac", round(d["roofline_prefill"]["frac"],4), d["roofline_prefill"]["by_shape_tflops"])
PY
done | tee $O/bench_ab.txt
