#!/bin/bash
# GPU box: per-class chunk times of alternative wgrad builds (build_exp/lib_e*.so), one block per variant.
for f in "$@"; do
  echo "### $f"
  DMNERF_DIAG_LIB=$f python scripts/diag_wgrad.py 2>&1 | grep -E "==|cls" | awk '{k=$6 $7; if (!(k in s)) {s[k]=1; print}}'
done
