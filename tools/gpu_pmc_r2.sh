#!/bin/bash
# Round-2 PMC evidence for the prefill kernels: one counter SET per rocprofv3 run (--pmc with --kernel-trace only), raw per-dispatch CSVs
# kept under gpurun_out/r2pmc/ (copied to profiles/r02_pmc_*.csv), aggregated into gpurun_out/r2pmc/r02_pmc_gemm.json.
#   drivers: tools/mb_gemm_one.py <variant> M N K iters  (the four 7B prefill linears with the kernel bench.py runs for each)
#            tools/mb_flash_one.py 1087                   (causal flash prefill, 32 heads x 128)
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/r2pmc; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
declare -A DRV
DRV[qkv]="tools/mb_gemm_one.py 30 1087 12288 4096 5"
DRV[gate_up]="tools/mb_gemm_one.py 30 1087 22016 4096 5"
DRV[o_proj]="tools/mb_gemm_one.py 18 1087 4096 4096 5"
DRV[down]="tools/mb_gemm_one.py 30 1087 4096 11008 5"
DRV[flash]="tools/mb_flash_one.py 1087 5"
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_COUNT"
      "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
      "FETCH_SIZE"
      "WRITE_SIZE")
for name in qkv gate_up o_proj down flash; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1)); d=/tmp/pmc_${name}_$i; rm -rf $d
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o p -- python $R/${DRV[$name]} > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then cp $f $O/r02_pmc_${name}_set$i.csv; else echo "$name set $i: no csv"; tail -3 $d.log; fi
  done
done
python - "$O" <<'PY'
import csv, glob, json, os, sys, collections
base = sys.argv[1]
out = {"_how": "rocprofv3 --pmc <set> --kernel-trace --output-format csv, one set per run (tools/gpu_pmc_r2.sh); values are per-dispatch "
               "averages over the timed dispatches of the named kernel (first dispatch dropped); raw rows in profiles/r02_pmc_<shape>_set<i>.csv",
       "kernels": {}}
for f in sorted(glob.glob(os.path.join(base, "r02_pmc_*_set*.csv"))):
    name = os.path.basename(f)[len("r02_pmc_"):].rsplit("_set", 1)[0]
    acc = collections.defaultdict(list); kern = None
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "gemm" not in k and "flash" not in k: continue
        if "swizzle" in k or "interleave" in k: continue
        kern = k.split("(")[0]
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = out["kernels"].setdefault(name, {"kernel": kern})
    for c, v in acc.items():
        v = v[1:] if len(v) > 1 else v
        d[c] = sum(v) / len(v)
for name, d in out["kernels"].items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        # MFMA-busy cycles are summed over the 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE is the kernel's duration in shader cycles
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] * 1024.0)
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["tcc_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1.0)
    if "FETCH_SIZE" in d:
        d["hbm_side_bytes"] = (2.0 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024.0     # gfx950: KiB units, FETCH_SIZE at half rate for wide streams
out["summary"] = {n: {k: d[k] for k in ("mfma_busy_frac", "tcc_hit_rate", "SQ_LDS_BANK_CONFLICT", "hbm_side_bytes") if k in d} for n, d in out["kernels"].items()}
json.dump(out, open(os.path.join(base, "r02_pmc_gemm.json"), "w"), indent=1)
print(json.dumps(out["summary"], indent=1))
PY
