P=impala-avx2-parquet-scanner_amd
for lib in libips_hip.so libips_hip_v1.so; do
  echo "== $lib"
  IPS_LIB=$PWD/$P/$lib timeout -k 5 200 python tools/kbench.py --bw 32,8 --what scan --sel 0.01,0.05,0.1,0.15 --reps 15 2>&1 | grep "w="
done
