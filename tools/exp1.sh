P=impala-avx2-parquet-scanner_amd
for lib in libips_hip.so libips_hip_w4.so libips_hip_w2.so; do
 for gm in 1 2 8; do
  echo "== $lib grid_mult=$gm"
  IPS_LIB=$PWD/$P/$lib IPS_GRID_MULT=$gm timeout -k 5 120 python tools/kbench.py --bw 32 --what scan,pred,decode --sel 0.1 --reps 15 2>&1 | grep "w="
 done
done
