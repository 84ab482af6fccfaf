#!/bin/bash
# sweep of the factorisation plan's knobs on the KITTI-map-sized BA problem
for st in 0 32 64 128 256; do for wg in 256 512; do
  echo "== subtree $st wg_sub $wg"
  SIM3OPT_BA_SUBTREE=$st SIM3OPT_BA_WG_SUB=$wg timeout -k 10 120 python scripts/gpu_ba_scale.py 5 --no-cpu 2>&1 | grep GPU || exit 1
done; done
