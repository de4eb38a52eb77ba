"""Digest the raw rocprofv3 --pmc CSVs of tools/gpu_pmc_r3.sh (<src>/<pfx>_pmc_<shape>_set<i>.csv) into what profiles/ keeps:
  <dst>/<pfx>_pmc_<shape>.csv   one row per (dispatch, counter) of the lmx:: kernels under test: set, dispatch, kernel, counter, value, duration_ns, vgpr, lds, grid
                                 (the raw files also hold torch's RNG / cast kernels with kilobyte-long names: dropped)
  <dst>/<pfx>_pmc.json          per-shape, per-KERNEL averages over the timed dispatches (first one of every kernel dropped) + derived figures; a shape
                                 that runs as two launches (K-sliced GEMM + launch-boundary reduction) lists both, its HBM-side bytes are their sum.
Derivations (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the 1024 SIMDs (= 32 x MFMAs for 32x32x16 bf16);
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel lasts GRBM_GUI_ACTIVE / 8 shader cycles; FETCH_SIZE / WRITE_SIZE are KiB and
FETCH_SIZE tallies a wide coalesced stream at half its bytes."""
import collections, csv, glob, json, os, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3pmc"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
pfx = sys.argv[3] if len(sys.argv) > 3 else "r03"
csv.field_size_limit(1 << 30)
out = {"_how": "tools/gpu_pmc_r3.sh: rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python tools/mb_gemm_one.py 0 M N K 5 | tools/mb_flash_one.py 1087 5 | "
               "tools/mb_gemv_cold.py, one counter set per run; digested by tools/pmc_digest3.py", "shapes": {}}
rows_by_shape = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, f"{pfx}_pmc_*_set*.csv"))):
    shape, setno = os.path.basename(f)[len(pfx) + 5:-4].rsplit("_set", 1)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "lmx::" not in k or "swizzle" in k or "interleave" in k:
            continue
        short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows_by_shape[shape].append((int(setno), int(r["Dispatch_Id"]), short, r["Counter_Name"], float(r["Counter_Value"]),
                                     int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Grid_Size"])))
for shape, rows in rows_by_shape.items():
    with open(os.path.join(dst, f"{pfx}_pmc_{shape}.csv"), "w") as fo:
        fo.write("set,dispatch,kernel,counter,value,duration_ns,vgpr,lds_bytes,grid\n")
        for r in rows:
            fo.write(",".join(str(x) for x in r) + "\n")
    kernels = collections.OrderedDict()
    for setno, disp, kern, ctr, val, ns, vgpr, lds, grid in rows:
        key = f"{kern}|grid{grid}"
        d = kernels.setdefault(key, {"kernel": kern, "vgpr": vgpr, "lds_bytes": lds, "grid": grid, "_v": collections.defaultdict(list), "_ns": collections.defaultdict(list)})
        d["_v"][ctr].append(val); d["_ns"][setno].append(ns)
    parts = {}
    for key, d in kernels.items():
        e = {k: d[k] for k in ("kernel", "vgpr", "lds_bytes", "grid")}
        for ctr, v in d["_v"].items():
            v = v[1:] if len(v) > 1 else v
            e[ctr] = sum(v) / len(v)
        e["duration_us_under_pmc"] = {str(s): round(sum(v[1:] if len(v) > 1 else v) / max(len(v) - 1, 1) / 1e3, 2) for s, v in d["_ns"].items()}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0
            e["kernel_cycles"] = cyc
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
            if "1" in e["duration_us_under_pmc"]:
                e["effective_clock_ghz"] = cyc / (e["duration_us_under_pmc"]["1"] * 1e3)
        if "SQ_WAVE_CYCLES" in e and "SQ_WAIT_ANY" in e and e["SQ_WAVE_CYCLES"] > 0:
            e["wave_cycle_split"] = {k: e[k] / e["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in e}
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["tcc_hit_rate"] = e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1.0)
        if "FETCH_SIZE" in e:
            e["hbm_side_bytes"] = (2.0 * e["FETCH_SIZE"] + e.get("WRITE_SIZE", 0.0)) * 1024.0
        parts[key] = e
    out["shapes"][shape] = parts
summ = {}
for shape, parts in out["shapes"].items():
    main = max(parts.values(), key=lambda e: e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) + 1e-9 * e.get("FETCH_SIZE", 0.0))
    s = {"kernels": [e["kernel"] for e in parts.values()]}
    for k in ("mfma_busy_frac", "effective_clock_ghz", "tcc_hit_rate"):
        if k in main:
            s[k] = round(main[k], 4)
    if any("SQ_LDS_BANK_CONFLICT" in e for e in parts.values()):
        s["SQ_LDS_BANK_CONFLICT"] = sum(e.get("SQ_LDS_BANK_CONFLICT", 0.0) for e in parts.values())
        s["SQ_LDS_IDX_ACTIVE"] = sum(e.get("SQ_LDS_IDX_ACTIVE", 0.0) for e in parts.values())
    if shape != "gemv" and any("hbm_side_bytes" in e for e in parts.values()):
        s["hbm_side_bytes"] = sum(e.get("hbm_side_bytes", 0.0) for e in parts.values())
    if shape == "gemv":
        s["hbm_side_bytes_by_kernel"] = {k: round(e["hbm_side_bytes"]) for k, e in parts.items() if "hbm_side_bytes" in e}
    summ[shape] = s
out["summary"] = summ
json.dump(out, open(os.path.join(dst, f"{pfx}_pmc.json"), "w"), indent=1)
