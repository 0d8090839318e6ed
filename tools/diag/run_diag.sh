set -x
mkdir -p gpurun_out/r2
cd tools/diag
for m in 0 1 2 3; do timeout -k 5 60 ./async_min $m > ../../gpurun_out/r2/async_min_$m.txt 2>&1; echo "mode $m rc=$?" >> ../../gpurun_out/r2/async_min_rc.txt; done
LD_LIBRARY_PATH=$PWD timeout -k 5 120 ./facade_test_async > ../../gpurun_out/r2/facade_async.txt 2>&1; echo "facade_async rc=$?" >> ../../gpurun_out/r2/async_min_rc.txt
cd ../..
python bench.py > gpurun_out/r2/bench_base.json 2> gpurun_out/r2/bench_base.err
python tools/kbench.py --bw 32,16,12,8,4 --what pred,scan --reps 20 > gpurun_out/r2/kbench_base.txt 2>&1
python tools/aux_bench.py > gpurun_out/r2/aux_base.txt 2>&1
echo done
