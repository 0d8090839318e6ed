# Runs on the GPU box (via gpurun): round-3 profiles.  Output under gpurun_out/prof_r3/;
# tools/make_profiles_r2.py (IPS_PROF_TAG=round3) condenses it into profiles/round3_*.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r3
rm -rf $OUT; mkdir -p $OUT
cd /tmp
step() { echo "[profile_r3] $1 $(date +%T)"; }
# 1. headline: kernel trace + stats, HBM traffic passes, SQ counters (program right after --)
step 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extra > $OUT/bench_trace.json 2> $OUT/trace.log
step 2
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $OUT/bench_fetch.json 2> $OUT/fetch.log
step 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $OUT/bench_write.json 2> $OUT/write.log
step 4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extra > $OUT/bench_sq.json 2> $OUT/sq.log
step 5
python3 $R/bench.py --steps 20 --warmup 3 --no-extra > $OUT/bench_plain.json 2> $OUT/plain.log
# 2. every other kernel family: kernel trace + stats and the traffic passes of the tour
step 6
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tour_trace -- python3 $R/tools/kernel_tour.py > $OUT/tour_trace.jsonl 2> $OUT/tour_trace.log
step 7
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/tour_fetch -- python3 $R/tools/kernel_tour.py > $OUT/tour_fetch.jsonl 2> $OUT/tour_fetch.log
step 8
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/tour_write -- python3 $R/tools/kernel_tour.py > $OUT/tour_write.jsonl 2> $OUT/tour_write.log
step 9
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/tour_sq -- python3 $R/tools/kernel_tour.py > $OUT/tour_sq.jsonl 2> $OUT/tour_sq.log
step 10
python3 $R/tools/kernel_tour.py > $OUT/tour_plain.jsonl 2> $OUT/tour_plain.log
# 2b. the page-list calls against the contiguous buffers (kernel trace + event timing), the sharded step
# on a one-rank communicator piece by piece (no profiler around RCCL)
step 11
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/chunks_trace -- python3 $R/tools/chunk_bench.py > $OUT/chunks_trace.jsonl 2> $OUT/chunks_trace.log
step 12
python3 $R/tools/chunk_bench.py > $OUT/chunk_bench.jsonl 2> $OUT/chunk_bench.log
step 13
timeout -k 10 120 python3 $R/tools/gather_probe.py > $OUT/gather_probe.log 2>&1
# 3. BASELINE configs[1..4] with their selectivity sweeps (event timing, property checks)
step 14
python3 $R/tools/configs_bench.py > $OUT/configs_bench.jsonl 2> $OUT/configs_bench.log
# condense on the box (the raw csv files are too large to travel): summaries only
cd $R
IPS_PROF_DIR=prof_r3 IPS_PROF_TAG=round3 python3 tools/make_profiles_r2.py > $OUT/make_profiles.log 2>&1
find $OUT -name "*.csv" -size +2M -delete
ls -R $OUT | head -60
