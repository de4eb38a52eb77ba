"""Digest the raw rocprofv3 --pmc CSVs of tools/gpu_pmc_r2.sh (gpurun_out/r2pmc/r02_pmc_<shape>_set<i>.csv) into what profiles/ keeps:
  profiles/r02_pmc_<shape>.csv   one row per (dispatch, counter) of the lmx:: kernel under test: set, dispatch, counter, value, duration_ns
                                 (the raw files also hold torch's RNG / cast kernels with kilobyte-long names: dropped)
  profiles/r02_pmc_gemm.json     per-kernel averages over the timed dispatches (first one dropped) + derived figures.
Derivations (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the 1024 SIMDs (= 32 x MFMAs for 32x32x16 bf16);
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel lasts GRBM_GUI_ACTIVE / 8 shader cycles; FETCH_SIZE / WRITE_SIZE are KiB and
FETCH_SIZE tallies a wide coalesced stream at half its bytes."""
import collections, csv, glob, json, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r2pmc"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
csv.field_size_limit(1 << 30)
out = {"_how": "tools/gpu_pmc_r2.sh: rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python tools/mb_gemm_one.py <variant> M N K 5 (or tools/mb_flash_one.py 1087 5), "
               "one counter set per run; digested by tools/pmc_digest.py", "kernels": {}}
by_shape = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "r02_pmc_*_set*.csv"))):
    shape, setno = os.path.basename(f)[len("r02_pmc_"):-4].rsplit("_set", 1)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "lmx::" not in k or "swizzle" in k or "interleave" in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        by_shape[shape].append((int(setno), int(r["Dispatch_Id"]), short, r["Counter_Name"], float(r["Counter_Value"]),
                                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Grid_Size"])))
for shape, rows in by_shape.items():
    with open(os.path.join(dst, f"r02_pmc_{shape}.csv"), "w") as fo:
        fo.write("set,dispatch,kernel,counter,value,duration_ns,vgpr,lds_bytes,grid\n")
        for r in rows:
            fo.write(",".join(str(x) for x in r) + "\n")
    d = {"kernel": rows[0][2], "vgpr": rows[0][6], "lds_bytes": rows[0][7], "grid": rows[0][8]}
    acc = collections.defaultdict(list); dur = collections.defaultdict(list)
    for setno, disp, _, ctr, val, ns, *_ in rows:
        acc[ctr].append(val); dur[setno].append(ns)
    for ctr, v in acc.items():
        v = v[1:] if len(v) > 1 else v
        d[ctr] = sum(v) / len(v)
    d["duration_us_under_pmc"] = {str(s): round(sum(v[1:]) / max(len(v) - 1, 1) / len(set(c for c in acc)) * 0 + (sum(v) / len(v)) / 1e3, 2) for s, v in dur.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        d["kernel_cycles"] = cyc
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        d["effective_clock_ghz"] = cyc / (d["duration_us_under_pmc"]["1"] * 1e3)
    if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_ANY" in d:
        d["wave_cycle_split"] = {k: d[k] / d["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in d}
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["tcc_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1.0)
    if "FETCH_SIZE" in d:
        d["hbm_side_bytes"] = (2.0 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024.0
    out["kernels"][shape] = d
out["summary"] = {n: {k: (round(d[k], 4) if isinstance(d[k], float) else d[k]) for k in ("kernel", "mfma_busy_frac", "effective_clock_ghz", "tcc_hit_rate", "SQ_LDS_BANK_CONFLICT", "hbm_side_bytes") if k in d}
                  for n, d in out["kernels"].items()}
json.dump(out, open(os.path.join(dst, "r02_pmc_gemm.json"), "w"), indent=1)
print(json.dumps(out["summary"], indent=1))
