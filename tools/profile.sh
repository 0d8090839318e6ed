# Runs on the GPU box (via gpurun): kernel-trace stats + the two HBM traffic PMC passes of the
# exact bench.py command.  Output under gpurun_out/prof_r1/ ; tools/make_profiles.py condenses it.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r1
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_sq.json 2> $OUT/sq.log
python3 $R/bench.py --steps 20 --warmup 3 > $OUT/bench_plain.json 2> $OUT/plain.log
ls -R $OUT | head -40
