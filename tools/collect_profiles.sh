#!/bin/bash
# Round evidence in one go (run on the GPU box from the repo root): tools/collect_profiles.sh <tag>   e.g. r02
# Writes under gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RT_OVERLAP_TOWER=0    # rocprof passes: every kernel alone on the chip (the bench lines of step 6 run the default, tower beside the transformer)
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph"
echo "[1] kernel trace + stats (bf16)"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bf16 -- $B > $OUT/bench_under_rocprof_bf16.json 2> $OUT/stats_bf16.err
echo "[2] kernel trace + stats (fp8)";  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fp8 -- $B --precision fp8 > $OUT/bench_under_rocprof_fp8.json 2> $OUT/stats_fp8.err
echo "[2b] kernel trace + stats (fp8-mx)";  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fp8mx -- $B --precision fp8-mx > $OUT/bench_under_rocprof_fp8mx.json 2> $OUT/stats_fp8mx.err
echo "[3] traffic passes (bf16)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_bf16 -- $B --no-roofline-pass > /dev/null 2> $OUT/fetch_bf16.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_bf16 -- $B --no-roofline-pass > /dev/null 2> $OUT/write_bf16.err
echo "[4] traffic passes (fp8)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_fp8 -- $B --no-roofline-pass --precision fp8 > /dev/null 2> $OUT/fetch_fp8.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_fp8 -- $B --no-roofline-pass --precision fp8 > /dev/null 2> $OUT/write_fp8.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_fp8mx -- $B --no-roofline-pass --precision fp8-mx > /dev/null 2> $OUT/fetch_fp8mx.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_fp8mx -- $B --no-roofline-pass --precision fp8-mx > /dev/null 2> $OUT/write_fp8mx.err
echo "[5] MFMA / LDS / wave-time counters, attention and GEMM alone"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
P3="GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
for K in attn gemm; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_${K}_p$i -- python3 $R/tools/prof_one.py $K > /dev/null 2> $OUT/pmc_${K}_p$i.err
  done
done
# the 2-waves-per-SIMD attention kernel alone, for comparison (round 3: attention_v3 serves S >= 1536 by default)
export RT_ATTN_V3=0
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc_attnold_p$i -- python3 $R/tools/prof_one.py attn > /dev/null 2> $OUT/pmc_attnold_p$i.err
done
unset RT_ATTN_V3
echo "[5b] VAE decode alone: kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_vae -- python3 $R/tools/prof_vae.py > /dev/null 2> $OUT/stats_vae.err
for d in fetch_bf16 write_bf16 fetch_fp8 write_fp8 fetch_fp8mx write_fp8mx pmc_attn_p1 pmc_attn_p2 pmc_attn_p3 pmc_attn_p4 pmc_attn_p5 pmc_gemm_p1 pmc_gemm_p2 pmc_gemm_p3 pmc_gemm_p4 pmc_gemm_p5 pmc_attnold_p1 pmc_attnold_p2 pmc_attnold_p3 pmc_attnold_p4 pmc_attnold_p5; do
  python3 $R/tools/pmc_summary.py $OUT/$d > $OUT/$d.json
  rm -rf $OUT/$d                       # the raw per-dispatch CSVs are large; the per-kernel means are what is kept
done
for d in stats_bf16 stats_fp8 stats_fp8mx stats_vae; do
  cp $OUT/$d/*/*kernel_stats.csv $OUT/${d}_kernel_stats.csv
  rm -rf $OUT/$d
done
echo "[6] bench lines"
cd $R
unset RT_OVERLAP_TOWER
python3 bench.py > $OUT/bench_c2_bf16.json 2> $OUT/bench_c2_bf16.err
python3 bench.py --no-cpu-baseline --precision fp8 > $OUT/bench_c2_fp8.json 2>/dev/null
python3 bench.py --no-cpu-baseline --batch-per-gpu 4 --steps 2 --warmup 1 > $OUT/bench_c3_batch4_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --height 1536 --width 1536 --steps 2 --warmup 1 > $OUT/bench_c5shape_1536_bf16.json 2>/dev/null
python3 bench.py --no-cpu-baseline --height 1536 --width 1536 --steps 2 --warmup 1 --precision fp8 > $OUT/bench_c5_1536_fp8.json 2>/dev/null
python3 bench.py --no-cpu-baseline --precision fp8-mx > $OUT/bench_c2_fp8mx.json 2>/dev/null
python3 bench.py --no-cpu-baseline --height 1536 --width 1536 --steps 2 --warmup 1 --precision fp8-mx > $OUT/bench_c5_1536_fp8mx.json 2>/dev/null
python3 tools/bench_inpaint.py > $OUT/bench_c4_inpaint.json 2> $OUT/bench_c4_inpaint.err
python3 tools/mx_dev/time_gemm.py > $OUT/mx_gemm_microbench.txt 2>/dev/null
python3 tools/bench_kernels.py all > $OUT/kernel_microbench.txt 2>/dev/null
ls $OUT
