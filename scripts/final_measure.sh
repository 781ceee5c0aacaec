#!/bin/bash
# Round-end measurement set (run through gpurun from the repo root): bench lines + rocprofv3 summaries -> gpurun_out/final/
# usage: final_measure.sh [all|lines|prof]   (two gpurun calls — lines, then prof — keep each under the 20-minute limit)
set -e
STAGE=${1:-all}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
if [ "$STAGE" != "prof" ]; then
: > $O/bench_lines.jsonl
timeout -k 10 400 python bench.py                             2> $O/err_train.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --no-extras 2> $O/err_f32.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --attention lsa --no-cpu-baseline --no-extras 2> $O/err_lsa.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload infer           2> $O/err_infer.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload infer --attention lsa 2> $O/err_infer_lsa.log | tail -1 >> $O/bench_lines.jsonl
timeout -k 10 300 python bench.py --workload gta             2> $O/err_gta.log | tail -1 >> $O/bench_lines.jsonl
echo "bench lines done"
fi
[ "$STAGE" = "lines" ] && exit 0
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
# LDS bank conflicts per kernel (SMA and LSA iterations), one iteration's timeline, per-segment stamps of the chains (variants built beforehand:
# scripts/build_variant.sh c_stamps chain.hip -DT2_STAMPS=1; scripts/build_variant.sh cb_stamps chain_bwd.hip -DT2_STAMPS=1)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds -o pl -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_lds.log 2>&1 < /dev/null
python3 $R/scripts/pmc_lds.py $O/pmc_lds $O/pmc_lds.json > $O/pmc_lds.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds_lsa -o pl -- python3 $R/bench.py --attention lsa --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_lds_lsa.log 2>&1 < /dev/null
python3 $R/scripts/pmc_lds.py $O/pmc_lds_lsa $O/pmc_lds_lsa.json > $O/pmc_lds_lsa.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/tl.log 2>&1 < /dev/null
python3 $R/scripts/iter_timeline.py "$(find $O/tl -name '*kernel_trace.csv' | head -1)" 1 5 > $O/iter_timeline.txt 2>&1 || true
cd $R
if [ -f variants/lib_c_stamps.so ] && [ -f variants/lib_cb_stamps.so ]; then
  : > $O/chain_stamps.txt
  for att in sma lsa; do
    echo "== fwd $att" >> $O/chain_stamps.txt; T2AMD_LIB=variants/lib_c_stamps.so timeout -k 10 200 python scripts/chain_stamps.py --att $att 2>/dev/null | grep -v amdgpu.ids >> $O/chain_stamps.txt || true
    echo "== bwd $att" >> $O/chain_stamps.txt; T2AMD_LIB=variants/lib_cb_stamps.so timeout -k 10 200 python scripts/chain_bwd_stamps.py --att $att 2>/dev/null | grep -v amdgpu.ids >> $O/chain_stamps.txt || true
  done
fi
cd /tmp
echo "lds / timeline / stamps done"
# summaries (copy into profiles/ by hand after looking at them)
python3 $R/scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json lstm,attention,step,proj,prenet,chain 400 > $O/pmc_hbm_traffic.txt 2>&1 || true
python3 $R/scripts/pmc_summary.py $O/pmc_fetch_lsa $O/pmc_write_lsa $O/pmc_hbm_traffic_lsa.json lstm,attention,step,proj,prenet,chain 400 > $O/pmc_hbm_traffic_lsa.txt 2>&1 || true
python3 $R/scripts/pmc_mfma.py $O/pmc_mfma $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1 || true
python3 $R/scripts/pmc_summary.py $O/pmc_gemm_fetch $O/pmc_gemm_write $O/pmc_gemm_traffic.json gemm > $O/pmc_gemm_traffic.txt 2>&1 || true
python3 $R/scripts/pmc_mfma.py $O/pmc_gemm_mfma $O/pmc_gemm_mfma.json > $O/pmc_gemm_mfma.txt 2>&1 || true
rm -rf $O/pmc_lds $O/pmc_lds_lsa $O/tl
echo "summaries done"
