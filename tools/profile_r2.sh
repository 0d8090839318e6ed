# Runs on the GPU box (via gpurun): round-2 profiles.  Output under gpurun_out/prof_r2/;
# tools/make_profiles_r2.py condenses it into profiles/round2_*.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
# 1. headline: kernel trace + stats, HBM traffic passes, SQ counters (program right after --)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extra > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $OUT/bench_sq.json 2> $OUT/sq.log
python3 $R/bench.py --steps 20 --warmup 3 --no-extra > $OUT/bench_plain.json 2> $OUT/plain.log
# 2. every other kernel family: kernel trace + stats and the traffic passes of the tour
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tour_trace -- python3 $R/tools/kernel_tour.py > $OUT/tour_trace.jsonl 2> $OUT/tour_trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/tour_fetch -- python3 $R/tools/kernel_tour.py > $OUT/tour_fetch.jsonl 2> $OUT/tour_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/tour_write -- python3 $R/tools/kernel_tour.py > $OUT/tour_write.jsonl 2> $OUT/tour_write.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/tour_sq -- python3 $R/tools/kernel_tour.py > $OUT/tour_sq.jsonl 2> $OUT/tour_sq.log
python3 $R/tools/kernel_tour.py > $OUT/tour_plain.jsonl 2> $OUT/tour_plain.log
# 3. BASELINE configs[1..4] with their selectivity sweeps (event timing, property checks)
python3 $R/tools/configs_bench.py > $OUT/configs_bench.jsonl 2> $OUT/configs_bench.log
# condense on the box (the raw csv files are too large to travel): summaries only
cd $R
python3 tools/make_profiles_r2.py > $OUT/make_profiles.log 2>&1
find $OUT -name "*.csv" -size +2M -delete
ls -R $OUT | head -60
