# A/B of library variants on one box: usage  bash tools/diag/ab.sh "<suffixes>" "<kbench args>"
set -x
O=gpurun_out/r2
mkdir -p $O
for rep in 1 2; do
for v in $1; do
  s=${v#base}
  IPS_LIB=$PWD/impala-avx2-parquet-scanner_amd/libips_hip$s.so python tools/kbench.py $2 >> $O/ab_$v.txt 2>&1
done
done
echo done
