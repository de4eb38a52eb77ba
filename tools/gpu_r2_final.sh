#!/bin/bash
# Round-2 closing GPU session: whole -m gpu suite, smoke, headline bench (default flags: CPU baseline + live PMC traffic), rocprofv3 kernel stats of the
# same command, config 3 / config 5 workloads, config 4 harness (TP=1 and TP=2 with both ranks on this GPU).  Outputs under gpurun_out/r2final/.
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=gpurun_out/r2final; mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench (default flags)"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json | head -c 400; echo
echo "== rocprofv3 kernel stats of the bench command"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-pmc > $R/$O/prof_bench.json 2> $R/$O/prof.err); echo "rocprof rc=$?"
cp /tmp/prof/*kernel_stats.csv $O/rocprofv3_kernel_stats.csv 2>/dev/null || find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats.csv \;
echo "== config3"
timeout 600 python bench.py --workload config3 --steps 2 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err; tail -c 300 $O/bench_config3.json; echo
echo "== config5"
timeout 600 python bench.py --workload config5 --steps 2 --warmup 1 > $O/bench_config5.json 2> $O/bench_config5.err; tail -c 300 $O/bench_config5.json; echo
echo "== config4 TP=1"
timeout 600 python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 --batch 32 > $O/config4_tp1.log 2> $O/config4_tp1.err; grep '^{' $O/config4_tp1.log | cut -c1-700
echo "== config4 TP=2 (ranks share the GPU)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 tools/config4_harness.py \
    --model llava_plus_v0_13b --requests 32 --batch 32 --shared-gpu > $O/config4_tp2.log 2> $O/config4_tp2.err
grep '^{' $O/config4_tp2.log | cut -c1-700
