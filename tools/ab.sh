#!/bin/bash
# Developer tool: same-box A/B of two builds of libolapgpu (run-to-run and box-to-box spread is several per cent,
# so two builds are only comparable inside one gpurun call).  The other build goes to olap-in-memory_amd/lib_prev/
# (git-ignored; e.g. built from a worktree of the previous commit) and is selected with OLAP_LIBOLAPGPU.
#   bash tools/ab.sh [grep pattern]
P=$PWD/olap-in-memory_amd/lib_prev/libolapgpu.so
O=gpurun_out/ab
mkdir -p $O
for i in 1 2; do
  for t in sweep sweep2 odd; do
    python tools/$t.py > $O/${t}_new_$i.txt 2>&1
    OLAP_LIBOLAPGPU=$P python tools/$t.py > $O/${t}_old_$i.txt 2>&1
  done
done
for t in sweep sweep2 odd; do
  paste -d"\n" $O/${t}_new_1.txt $O/${t}_old_1.txt $O/${t}_new_2.txt $O/${t}_old_2.txt | grep -v amdgpu.ids | awk '{printf "%s %s\n", (NR%2==1?"new":"old"), $0}' | cut -c1-120
done > $O/ab.txt
grep -E "${1:-.}" $O/ab.txt
