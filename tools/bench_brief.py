"""Print the few numbers of a bench.py JSON line that an A/B decision needs.  Usage: bench_brief.py file [label]"""
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    if "value" not in r: continue
    kb = r.get("kernel_breakdown_ms_per_step", {})
    pre = {k: round(v["ms"], 3) for k, v in kb.items() if k.startswith(("prefill", "vis", "proj"))}
    dec = {k: round(v["ms"] / max(v["n"], 1) * 1e3, 2) for k, v in kb.items() if k.startswith("decode")}
    print(sys.argv[2] if len(sys.argv) > 2 else "", {k: (round(r[k], 3) if isinstance(r.get(k), float) else r.get(k)) for k in ("value", "prefill_ms", "decode_ms_per_token", "greedy_ids_identical_across_steps")})
    print("  prefill ms:", pre)
    print("  decode us/launch:", dec)
