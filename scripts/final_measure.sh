#!/bin/bash
# Round-end measurement set (run through gpurun from the repo root): bench lines + rocprofv3 summaries -> gpurun_out/final/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
: > $O/bench_lines.jsonl
timeout -k 10 400 python bench.py                             2> $O/err_train.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --no-extras 2> $O/err_f32.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --attention lsa --no-cpu-baseline --no-extras 2> $O/err_lsa.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload infer           2> $O/err_infer.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload infer --attention lsa 2> $O/err_infer_lsa.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload gta             2> $O/err_gta.log | tail -1 >> $O/bench_lines.jsonl
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bp -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1 < /dev/null
echo "bench profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -o ip -- python3 $R/bench.py --workload infer --steps 2 --warmup 1 > $O/prof_infer.log 2>&1 < /dev/null
echo "infer profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lsa -o lp -- python3 $R/bench.py --attention lsa --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_lsa.log 2>&1 < /dev/null
echo "lsa profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pf -- python3 $R/scripts/time_decoder.py --T 400 --iters 1 --prof 0 > $O/pmc_fetch.log 2>&1 < /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pw -- python3 $R/scripts/time_decoder.py --T 400 --iters 1 --prof 0 > $O/pmc_write.log 2>&1 < /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_lsa -o pf -- python3 $R/scripts/time_decoder.py --T 400 --iters 1 --prof 0 --att lsa > $O/pmc_fetch_lsa.log 2>&1 < /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_lsa -o pw -- python3 $R/scripts/time_decoder.py --T 400 --iters 1 --prof 0 --att lsa > $O/pmc_write_lsa.log 2>&1 < /dev/null
echo "pmc traffic done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o pm -- python3 $R/scripts/time_decoder.py --T 400 --iters 1 --prof 0 > $O/pmc_mfma.log 2>&1 < /dev/null
echo "pmc done"
# GEMM (256-tile LDS-DMA kernel): per-shape kernel durations, MFMA issue, fabric-side traffic
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gemm -o gp -- python3 $R/scripts/check_gemm256.py --notest --reps 1 > $O/prof_gemm.log 2>&1 < /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gemm_mfma -o gm -- python3 $R/scripts/check_gemm256.py --notest --reps 1 > $O/pmc_gemm_mfma.log 2>&1 < /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gemm_fetch -o gf -- python3 $R/scripts/check_gemm256.py --notest --reps 1 > $O/pmc_gemm_fetch.log 2>&1 < /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gemm_write -o gw -- python3 $R/scripts/check_gemm256.py --notest --reps 1 > $O/pmc_gemm_write.log 2>&1 < /dev/null
echo "gemm profiles done"
# summaries (copy into profiles/ by hand after looking at them)
python3 $R/scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json lstm,attention,step,proj,prenet,chain 400 > $O/pmc_hbm_traffic.txt 2>&1 || true
python3 $R/scripts/pmc_summary.py $O/pmc_fetch_lsa $O/pmc_write_lsa $O/pmc_hbm_traffic_lsa.json lstm,attention,step,proj,prenet,chain 400 > $O/pmc_hbm_traffic_lsa.txt 2>&1 || true
python3 $R/scripts/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1 || true
python3 $R/scripts/pmc_summary.py $O/pmc_gemm_fetch $O/pmc_gemm_write $O/pmc_gemm_traffic.json gemm > $O/pmc_gemm_traffic.txt 2>&1 || true
python3 $R/scripts/pmc_mfma.py $O/pmc_gemm_mfma $O/pmc_gemm_mfma.json > $O/pmc_gemm_mfma.txt 2>&1 || true
echo "summaries done"
