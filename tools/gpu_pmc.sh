#!/bin/bash
# PMC evidence (rounds 5 and 6; RND=r06 names the outputs) for the prefill kernels + the decode GEMV / attention kernels: one counter SET per rocprofv3 run (--pmc with --kernel-trace
# only: gpurun refuses --pmc next to the hip / hsa / memory trace domains), raw per-dispatch CSVs under gpurun_out/r5pmc/, digested by
# tools/pmc_digest3.py into profiles/r05_pmc_<shape>.csv + profiles/r05_pmc.json.
#   drivers: tools/mb_gemm_one.py <variant> M N K iters   (what the engine launches for that shape: q|k|v / gate|up ping-pong 256x256 (variant 0 picks it),
#                                                           o_proj / down_proj K-sliced ping-pong + launch-boundary reduction = two kernels (variant 30: the
#                                                           single-op entry brings no split-K scratch, so variant 0 would fall back to the 128x128 kernel))
#            tools/mb_flash_one.py 1087 5                  (causal flash prefill, 32 heads x 128: flash_prefill2_kernel)
#            tools/mb_gemv_cold.py                         (decode linears, 32 distinct matrices per shape: gemv2_kernel)
#            tools/mb_kv_attn.py 1150 16                   (split-q decode step's second launch: decode_kv_attn_kernel, + the projection and the attention alone)
set -u
cd "$(dirname "$0")/.."
RND=${RND:-r05}
R=$(pwd); O=$R/gpurun_out/${RND}pmc; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
declare -A DRV
DRV[qkv]="tools/mb_gemm_one.py 0 1087 12288 4096 5"
DRV[gate_up]="tools/mb_gemm_one.py 0 1087 22016 4096 5"
DRV[o_proj]="tools/mb_gemm_one.py 30 1087 4096 4096 5"
DRV[down]="tools/mb_gemm_one.py 30 1087 4096 11008 5"
DRV[flash]="tools/mb_flash_one.py 1087 5"
DRV[gemv]="tools/mb_gemv_cold.py"
DRV[kv_attn]="tools/mb_kv_attn.py 1150 16"
DRV[wgrad]="tools/mb_wgrad_one.py 8192 11008 4096 4"          # gemm8t_kernel: weight gradient from dy and x in their forward layout (LDS transpose reads)
DRV[attn_bwd]="tools/mb_attn_bwd_one.py 2048 3"             # training attention backward of one 2048-row sample, 32 heads x 128 (delta / dq / dkv kernels)
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_COUNT"
      "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
      "FETCH_SIZE"
      "WRITE_SIZE")
for name in ${SHAPES:-qkv gate_up o_proj down flash gemv kv_attn}; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1)); d=/tmp/pmc_${name}_$i; rm -rf $d
    if { [ "$name" = gemv ] || [ "$name" = kv_attn ]; } && [ $i -le 3 ] && [ $i -ne 1 ]; then continue; fi     # the GEMV driver is long: wave / traffic sets only
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python $R/${DRV[$name]} > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then cp $f $O/${RND}_pmc_${name}_set$i.csv; else echo "$name set $i: no csv"; tail -3 $d.log; fi
  done
done
cd $R && python tools/pmc_digest3.py $O $O $RND && cat $O/${RND}_pmc.json | python -c "import json,sys; print(json.dumps(json.load(sys.stdin)['summary'], indent=1))"
