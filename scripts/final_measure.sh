#!/bin/bash
# Round-end measurement set (run through gpurun from the repo root): bench lines + rocprofv3 summaries -> gpurun_out/final/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
: > $O/bench_lines.jsonl
timeout -k 10 300 python bench.py                             2> $O/err_train.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline 2> $O/err_f32.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --attention lsa --no-cpu-baseline 2> $O/err_lsa.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload infer           2> $O/err_infer.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload infer --attention lsa 2> $O/err_infer_lsa.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload gta             2> $O/err_gta.log | tail -1 >> $O/bench_lines.jsonl
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bp -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1 < /dev/null
echo "bench profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec -o dp -- python3 $R/scripts/time_decoder.py --T 100 --iters 2 --prof 0 > $O/prof_dec.log 2>&1 < /dev/null
echo "decoder profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o pm -- python3 $R/scripts/time_decoder.py --T 40 --iters 1 --prof 0 --overlap 0 > $O/pmc_mfma.log 2>&1 < /dev/null
echo "pmc done"
