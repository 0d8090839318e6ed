for gm in 1 2 4 8 16; do
  echo "== grid_mult=$gm"
  IPS_GRID_MULT=$gm timeout -k 5 200 python tools/kbench.py --bw 32,16,8,4 --what scan,pred --sel 0.1 --reps 15 2>&1 | grep "w="
done
