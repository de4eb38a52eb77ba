#!/bin/bash
# Round 5, session F: the headline bench as the driver runs it (default flags) + rocprofv3 kernel stats of the same command (short)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python bench.py > gpurun_out/r05_bench_driver_flags.json 2> gpurun_out/r05_f.err ) 2>&1 | tail -3
python tools/bench_brief.py gpurun_out/r05_bench_driver_flags.json "driver flags" | head -4
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_bench_driver_flags.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'prefill_ms', 'decode_ms_per_token', 'schema')})
print('roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'avg_launch_us', 'launches', 'traffic')})
print('roofline_prefill', {k: d['roofline_prefill'][k] for k in ('frac', 'gemm_family_frac', 'by_shape_tflops')})
c = d['cpu_baseline']; print('cpu', {k: c.get(k) for k in ('value', 'cores', 'kind', 'partly_priced', 'decode_steps_timed', 'error')})
print('parity', d.get('parity'))
print('serving_batch', d.get('serving_batch'))
PY
bash tools/gpu_prof.sh 2>&1 | tail -32
cp gpurun_out/prof/*kernel_stats.csv gpurun_out/r05_rocprofv3_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_bench.json gpurun_out/r05_bench_under_rocprofv3.json 2>/dev/null
tail -3 gpurun_out/r05_f.err
