#!/bin/bash
# Developer tool: tile walk (OLAP_XY_SUPER x OLAP_XY_SUPER blocks of tiles per group of consecutive workgroups, X- or
# Y-fastest) and store policy (streaming | cached) of transpose_xy_kernel on the shapes of tools/transpose_probe.py.
for cfg in "1 x 0" "1 x 1" "1 y 0" "1 y 1" "4 x 0" "4 x 1" "4 y 1" "8 y 1"; do
  set -- $cfg
  echo "super $1 order $2 cached_stores $3"
  if [ "$3" = 1 ]; then export OLAP_XY_CACHED_STORES=1; else unset OLAP_XY_CACHED_STORES; fi
  OLAP_XY_SUPER=$1 OLAP_XY_ORDER=$2 timeout -k 10 200 python tools/transpose_probe.py 2>&1 | grep -E "^\[100000, 1000\]|^\[3652, 27400\]|^\[10000, 10000\]|^\[100, 1000000\]|^\[1000, 100000\]|^\[31623" | cut -c1-80
done
