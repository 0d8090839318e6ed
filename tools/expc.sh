P=impala-avx2-parquet-scanner_amd
for rep in 1 2; do
for lib in libips_hip.so libips_NO_BM.so libips_NO_CNT.so libips_NO_VAL.so libips_NO_ALL.so; do
  echo "== $lib"
  IPS_LIB=$PWD/$P/$lib timeout -k 5 200 python tools/kbench.py --bw 32 --what scan --sel 0.1 --reps 25 2>&1 | grep "w="
done
done
