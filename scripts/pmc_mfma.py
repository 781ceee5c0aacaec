"""Summarise a rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES into profiles/rNN_pmc_mfma.json.

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -o p -- \
      python3 scripts/time_decoder.py --T 40 --iters 1 --prof 0 --overlap 0
  python scripts/pmc_mfma.py gpurun_out/pmc_mfma profiles/r01_pmc_mfma.json

Per kernel (short name + grid): launches, median SQ_VALU_MFMA_BUSY_CYCLES per launch (summed over the chip's SIMDs; 32 cycles
per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md per-instruction table), the launch's duration from the kernel trace of the
same run, and mfma_util = busy / (duration * 2.4 GHz * 1024 SIMDs) = fraction of the dense MFMA peak that was issued.
For the bf16 kernels busy/32 * 32768 FLOP reproduces the product's algorithmic FLOPs (checked on the 2560x4096x3072 GEMM)."""
import csv, glob, json, os, re, statistics, sys


def main():
    src, dst = sys.argv[1:3]
    cc = [r for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True) for r in csv.DictReader(open(f))]
    kt = {r["Dispatch_Id"]: r for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True) for r in csv.DictReader(open(f))}
    agg = {}
    for r in cc:
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES" or r["Dispatch_Id"] not in kt:
            continue
        t = kt[r["Dispatch_Id"]]
        m = re.search(r"(\w+)_kernel", r["Kernel_Name"])
        key = (m.group(1) if m else r["Kernel_Name"][:30]) + "_grid" + r["Grid_Size"]
        targs = re.search(r"chain_fwd_kernel<([^>]*)>", r["Kernel_Name"])
        if targs:                                        # persistent chains: one key per (attention kind, decode) instantiation
            a = [x.strip() for x in targs.group(1).split(",")]
            key = "chain_fwd_" + {"0": "lstm", "1": "sma", "2": "lsa"}.get(a[4], "x") + ("_decode" if len(a) > 5 and a[5] != "0" else "")
        agg.setdefault(key, []).append((float(r["Counter_Value"]), int(t["End_Timestamp"]) - int(t["Start_Timestamp"])))
    out = {}
    for k, v in agg.items():
        busy, dur = statistics.median(x[0] for x in v), statistics.median(x[1] for x in v)
        if busy > 0:
            out[k] = {"launches": len(v), "mfma_busy_cycles_median": busy, "duration_us_median": dur / 1e3,
                      "mfma_util": busy / (dur * 2.4 * 1024)}
    json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_util"])), open(dst, "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_util"]):
        print(f"{k:46s} n={v['launches']:3d} busy {v['mfma_busy_cycles_median']:11.0f} dur {v['duration_us_median']:8.1f} us  util {100 * v['mfma_util']:5.1f} %")


if __name__ == "__main__":
    main()
