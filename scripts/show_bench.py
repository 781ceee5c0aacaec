"""Dev: the numbers of a bench.py JSON line that one looks at between two changes."""
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(l)
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, {k: v.get("us_per_step") for k, v in (d.get("kernels") or {}).items()})
f = d.get("fresh_batches")
if f:
    print("fed loop", {k: f[k] for k in ("ms_per_step", "ms_per_step_median", "max_over_median")})
for k in ("fp32", "decode", "gta", "gemm"):
    if k in d:
        print(k, {a: b for a, b in d[k].items() if not isinstance(b, (str, dict))})
if d.get("roofline"):
    print("roofline", {k: v for k, v in d["roofline"].items() if k not in ("note", "traffic_source")})
