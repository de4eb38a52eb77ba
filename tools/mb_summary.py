import json, sys
for l in sys.stdin:
    try:
        r = json.loads(l)
    except Exception:
        continue
    if r["kind"] == "gemm":
        print(f"{r['name']:10s} v={str(r['variant']):28s} {r.get('us', 0):9.1f} us {r.get('tflops', 0):8.1f} TF {r.get('error', '')}")
    elif r["kind"] == "gemv":
        print(f"gemv {r['name']:10s} {r['us']:8.1f} us {r['gbps']:8.1f} GB/s")
    else:
        print(r)
