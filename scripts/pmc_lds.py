"""Summarise a rocprofv3 PMC pass with the LDS counters: which kernels lose LDS cycles to bank conflicts.

  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_lds -o p -- \
      python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras
  python scripts/pmc_lds.py gpurun_out/pmc_lds profiles/rNN_pmc_lds.json

Per kernel name: launches, summed SQ_LDS_BANK_CONFLICT (extra LDS cycles) and SQ_LDS_IDX_ACTIVE (all LDS-array cycles) over all
launches (both summed over the chip's CUs), their ratio, total duration from the kernel trace, and the conflict cycles per CU as a
fraction of the kernel's duration (an upper bound of what removing them could save: LDS cycles overlap with other work)."""
import csv, glob, json, os, re, sys


def main():
    src, dst = sys.argv[1:3]
    cc = [r for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True) for r in csv.DictReader(open(f))]
    kt = {r["Dispatch_Id"]: r for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True) for r in csv.DictReader(open(f))}
    agg = {}
    for r in cc:
        if r["Dispatch_Id"] not in kt:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)[:80]
        a = agg.setdefault(name, {"ids": set(), "conf": 0.0, "act": 0.0, "ns": 0})
        if r["Dispatch_Id"] not in a["ids"]:
            a["ids"].add(r["Dispatch_Id"])
            t = kt[r["Dispatch_Id"]]
            a["ns"] += int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
        if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT":
            a["conf"] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
            a["act"] += float(r["Counter_Value"])
    out = {}
    for k, a in agg.items():
        if a["act"] <= 0:
            continue
        out[k] = {"launches": len(a["ids"]), "duration_ms": a["ns"] / 1e6, "lds_active_cycles": a["act"], "bank_conflict_cycles": a["conf"],
                  "conflict_ratio": a["conf"] / a["act"], "conflict_frac_of_duration": a["conf"] / 256 / (a["ns"] * 2.4)}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["bank_conflict_cycles"]))
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:80s} n={v['launches']:4d} {v['duration_ms']:8.3f} ms  conflict/active {100 * v['conflict_ratio']:5.1f} %  conflict cycles per CU / duration {100 * v['conflict_frac_of_duration']:5.1f} %")


main()
