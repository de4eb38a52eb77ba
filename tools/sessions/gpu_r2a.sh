#!/bin/bash
# Round-2 GPU session A: box facts, ping-pong GEMM correctness + microbenchmarks, headline bench (with / without the new kernel),
# full-depth parity, then the whole -m gpu suite.  Everything lands under gpurun_out/r2a/.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
{ nproc; free -g | head -2; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocminfo | grep -E "Compute Unit|Max Clock|Marketing" | head -8; } > $O/box.txt 2>&1
cat $O/box.txt
echo "== gemm8p tests"; timeout 600 python -m pytest tests/test_gemm8p_gpu.py -x -q > $O/test_gemm8p.log 2>&1; echo "rc=$?"; tail -15 $O/test_gemm8p.log
echo "== microbench"; timeout 600 python tools/mb_gemm_variants.py "1087,12288,4096;1087,22016,4096;1087,4096,4096;1087,4096,11008;4096,4096,4096;1087,15360,5120;1087,5120,13824" "9,18,7,30,31,32,33,34,35,-1" > $O/mb_gemm.jsonl 2>&1; echo "rc=$?"; cat $O/mb_gemm.jsonl
echo "== bench (new kernels)"; timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -c 1500 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r2a/bench.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token','greedy_ids_identical_across_steps')})
    print('roofline', {k: r['roofline'][k] for k in ('achieved','frac','avg_launch_us','launches')})
    print('roofline_prefill', {k: r['roofline_prefill'][k] for k in ('achieved','frac','avg_launch_us','prefill_end_to_end_frac')})
    print('cpu', r['cpu_baseline'])
    for k, v in r['kernel_breakdown_ms_per_step'].items(): print(f'  {k:26s} {v["ms"]:10.3f} ms  n={v["n"]}')
PY
echo "== bench (LMX_GEMM8P=0)"; LMX_GEMM8P=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch > $O/bench_old.json 2> $O/bench_old.err; echo "rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r2a/bench_old.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s')}, 'roofline_prefill', r['roofline_prefill']['frac'])
PY
echo "== full depth"; timeout 1500 python -m pytest tests/test_full_depth_gpu.py -x -q -s > $O/test_full_depth.log 2>&1; echo "rc=$?"; tail -30 $O/test_full_depth.log
cp gpurun_out/full_depth_*.json $O/ 2>/dev/null
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_full_depth_gpu.py --deselect tests/test_gemm8p_gpu.py > $O/test_gpu.log 2>&1; echo "rc=$?"; tail -8 $O/test_gpu.log
