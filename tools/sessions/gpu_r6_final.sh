#!/bin/bash
# Round 6, closing session: the whole GPU suite, smoke(), the driver's bench command, rocprofv3 kernel stats of the headline command, PMC passes, the other BASELINE configurations
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
P=r06
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/${P}_pytest_gpu_summary.txt 2>&1; grep -E "passed|failed|real" gpurun_out/${P}_pytest_gpu_summary.txt
for f in full_depth_llava15_7b full_depth_llava15_13b full_depth_fp32_llava15_7b; do cp gpurun_out/$f.json gpurun_out/${P}_$f.json 2>/dev/null; cp gpurun_out/$f.json profiles/${P}_$f.json 2>/dev/null; done   # the bench's parity record reads profiles/: these are this tree's reports
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 )
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${P}_bench_driver_flags.json 2> gpurun_out/${P}_final.err ) 2>&1 | grep real
cp gpurun_out/bench_tp_projection.json gpurun_out/${P}_bench_tp_projection.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_driver_flags.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'prefill_ms', 'decode_ms_per_token', 'schema', 'decode_step_frac_of_hbm_peak', 'prefill_frac_of_mfma_peak')})
r = d['roofline']; print('roofline', {k: r[k] for k in ('achieved', 'frac', 'avg_launch_us', 'launches', 'traffic')}); print('kv_attn', {k: r['kv_attn'][k] for k in ('avg_launch_us', 'achieved', 'frac')}); print('decode_step', r['decode_step'])
print('roofline_prefill', {k: d['roofline_prefill'][k] for k in ('frac', 'gemm_family_frac', 'by_shape_tflops')})
print('traffic', {k: round(v['ratio'], 2) for k, v in (d['roofline_prefill'].get('traffic') or {}).get('by_shape', {}).items()})
c = d['cpu_baseline']; print('cpu', {k: c.get(k) for k in ('value', 'cores', 'kind', 'partly_priced', 'decode_steps_timed', 'prefill_ms', 'decode_tokens_per_s', 'error')})
print('parity', {k: v for k, v in d['parity'].items() if k != 'asserted'})
print('serving_batch', {k: round(v['decode_tokens_per_s']) for k, v in d['serving_batch']['by_batch'].items()})
print('tp_projection', d.get('tp_projection'))
print('kernel_breakdown', {k: round(v['ms'], 3) for k, v in d['kernel_breakdown_ms_per_step'].items()})
PY
bash tools/gpu_prof.sh 2>&1 | grep -E "lmx::|rocprof rc" | head -16
cp gpurun_out/prof/*kernel_stats.csv gpurun_out/${P}_rocprofv3_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_bench.json gpurun_out/${P}_bench_under_rocprofv3.json 2>/dev/null
RND=$P bash tools/gpu_pmc.sh 2>&1 | tail -30
tail -3 gpurun_out/${P}_final.err
( time timeout 500 python bench.py --model llava15_13b --steps 5 --warmup 2 --no-cpu-baseline --no-tp-projection --no-pmc > gpurun_out/${P}_bench_13b.json 2>> gpurun_out/${P}_final.err ) 2>&1 | grep real
python tools/bench_brief.py gpurun_out/${P}_bench_13b.json "13B" | head -3
( time timeout 500 python bench.py --workload config3 --steps 3 --warmup 1 --no-cpu-baseline --no-tp-projection --no-pmc > gpurun_out/${P}_bench_config3.json 2>> gpurun_out/${P}_final.err ) 2>&1 | grep real
python tools/bench_brief.py gpurun_out/${P}_bench_config3.json "config3" | head -3
for r in 1 0; do
  timeout 600 python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 --batch 32 --reuse $r > gpurun_out/${P}_config4_reuse$r.json 2> gpurun_out/${P}_config4.err || tail -5 gpurun_out/${P}_config4.err
  tail -c 600 gpurun_out/${P}_config4_reuse$r.json; echo
done
timeout 900 python bench.py --workload config5 --train-batch 16 --train-seq 2048 --steps 2 --warmup 1 > gpurun_out/${P}_bench_config5_16x2048.json 2>> gpurun_out/${P}_final.err || tail -3 gpurun_out/${P}_final.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench_config5_16x2048.json').read().strip().splitlines()[-1])
print('config5', {k: d[k] for k in ('value', 'ms_per_step', 'forward_ms', 'forward_backward_ms', 'optimizer_ms', 'linear_tflops_in_fwd_bwd', 'loss_first_last', 'hbm_GB') if k in d})
PY
timeout 200 python tools/mb_tp_batch_step.py 8 8 2>/dev/null | tee gpurun_out/${P}_tp_batch_step_8x8.jsonl | cut -c1-400
timeout 200 python tools/mb_tp_batch_step.py 8 32 2>/dev/null | tee gpurun_out/${P}_tp_batch_step_8x32.jsonl | cut -c1-400
( time LMX_BENCH_SHARE_GPU=1 LMX_TP_P2P_ALL=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/${P}_bench_tp2_shared_gpu_dry_run.json 2> gpurun_out/${P}_dry.err ) 2>&1 | grep real; tail -3 gpurun_out/${P}_dry.err
python tools/bench_brief.py gpurun_out/${P}_bench_tp2_shared_gpu_dry_run.json "N=2 dry run" | head -4
timeout 300 python tools/mb_wgrad.py 2>/dev/null | tee gpurun_out/${P}_wgrad_direct.jsonl | cut -c1-250
bash tools/gpu_prof_c5.sh 2>&1 | head -16 | cut -c1-180; cp gpurun_out/prof_c5_kernel_stats.csv gpurun_out/${P}_config5_kernel_stats_final.csv 2>/dev/null
