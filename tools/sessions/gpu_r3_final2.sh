#!/bin/bash
# Round-3 closing session of the re-entered build: the whole GPU suite, the bench line with the driver's flags, and a dry run of the N > 1 bench path
# (two ranks sharing this GPU: gloo rendezvous, every all-reduce through the P2P kernel) so that the code the driver's multi-GPU run executes after the timed
# loop (max-over-ranks, replicas section, JSON line) has run once on the final tree.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3final2; mkdir -p $O
export TMPDIR=/tmp
timeout 420 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 260 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench_driver_flags.json
python - <<PY
import json
d=json.load(open("$O/bench_driver_flags.json"))
print("value", round(d["value"],1), "prefill_ms", round(d["prefill_ms"],2), "decode_ms", round(d["decode_ms_per_token"],4), "frac", round(d["roofline"]["frac"],4), "prefill frac", round(d["roofline_prefill"]["frac"],4), "cpu", d["cpu_baseline"]["value"])
PY
LMX_BENCH_SHARE_GPU=1 LMX_TP_P2P_ALL=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-pmc > $O/bench_tp2_shared.json 2> $O/bench_tp2_shared.err
echo "tp2 shared rc=$?"; tail -c 600 $O/bench_tp2_shared.json | head -c 600; echo; tail -3 $O/bench_tp2_shared.err
