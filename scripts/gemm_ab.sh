#!/bin/bash
# Dev tool (run through gpurun from the repo root): kernel-only durations of the 256-tile GEMM for several library builds.
# usage: gemm_ab.sh outdir lib1 lib2 ...   ("default" = the in-tree library)
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset T2AMD_LIB; else export T2AMD_LIB=$R/variants/lib_$v.so; fi
  timeout -k 10 100 rocprofv3 --kernel-trace --output-format csv -d $O/$v -o g -- python3 $R/scripts/check_gemm256.py --notest --reps 1 > $O/log_$v.txt 2>&1 < /dev/null || exit 1
done
python3 - "$O" "$@" <<'PY'
import csv, collections, sys
O = sys.argv[1]
for v in sys.argv[2:]:
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f"{O}/{v}/g_kernel_trace.csv")):
        if "256" not in r["Kernel_Name"]: continue
        agg.setdefault((r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"{v:12s}", " ".join(f"{k[0]}x{k[1]}x{k[2]}:{min(d):.1f}" for k, d in agg.items()))
PY
