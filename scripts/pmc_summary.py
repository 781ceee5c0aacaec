"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_pmc_hbm_traffic.json.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 scripts/time_decoder.py --T 40 --iters 1 --prof 0 --overlap 0
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 scripts/time_decoder.py --T 40 --iters 1 --prof 0 --overlap 0
  python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_hbm_traffic.json

Per kernel (short name + grid size): median per-launch FETCH_SIZE / WRITE_SIZE in KB and
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — gfx950 reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv, glob, json, os, re, statistics, sys


def load(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            m = re.search(r"(\w+)_kernel", name)
            short = m.group(1) if m else name[:40]
            if "bf16" in name and "bf16" not in short:
                short += "_bf16"
            key = f"{short}_grid{r['Grid_Size']}"
            if short == "chain_fwd":                     # persistent chains: one key per attention kind (5th template argument)
                targs = re.search(r"chain_fwd_kernel<([^>]*)>", name)
                kind = int(targs.group(1).split(",")[4]) if targs else -1
                dec = targs and int(targs.group(1).split(",")[5]) > 0 if targs and len(targs.group(1).split(",")) > 5 else False
                key = "chain_fwd_" + {0: "lstm", 1: "sma", 2: "lsa"}.get(kind, "x") + ("_decode" if dec else "")
            elif short == "chain_bwd_lstm":
                key = short
            elif short == "chain_bwd_att":              # <MT, kind>: 1 = SMA, 2 = LSA
                targs = re.search(r"chain_bwd_att_kernel<([^>]*)>", name)
                key = "chain_bwd_" + {"1": "sma", "2": "lsa"}.get(targs.group(1).split(",")[1].strip() if targs else "", "x")
            out.setdefault(key, []).append(float(r["Counter_Value"]))
    return out


def main():
    fd, wd, dst = sys.argv[1:4]
    keep = tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("lstm", "attention", "step", "proj", "prenet", "chain")
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # steps one persistent chain launch covers in the profiled command
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(F) | set(W)):
        if not any(t in k for t in keep):
            continue
        f = statistics.median(F.get(k, [0.0])); w = statistics.median(W.get(k, [0.0]))
        res[k] = {"launches": len(F.get(k, [])), "FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w,
                  "hbm_bytes_per_launch": (2 * f + w) * 1024}
        if k.startswith("chain_") and steps:
            res[k]["steps_per_launch"] = steps
            res[k]["hbm_bytes_per_step"] = res[k]["hbm_bytes_per_launch"] / steps
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:50s} n={v['launches']:4d} fetch {v['FETCH_SIZE_KB_median']:10.1f} KB  write {v['WRITE_SIZE_KB_median']:9.1f} KB  hbm {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB")


if __name__ == "__main__":
    main()
