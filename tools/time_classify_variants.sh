#!/bin/bash
# usage: tools/time_classify_variants.sh <lib|-> ...: the classification alone at 512^3 for each library variant
for L in "$@"; do
  if [ "$L" = "-" ]; then unset CFX_LIB; else export CFX_LIB=$L; fi
  echo "== $L"; timeout -k 10 200 python3 tools/time_classify.py 512 2>&1 | tail -4
done
