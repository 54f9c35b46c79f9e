#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
L="tools/_ab/pack.so tools/_ab/laync.so tools/_ab/laync2.so tools/_ab/sort.so"
python tools/ab_bench.py $L --rounds 9 > $O/r04_ab4_full.txt 2>&1; cat $O/r04_ab4_full.txt
python tools/ab_bench.py $L --rounds 9 --tile 8 > $O/r04_ab4_tile8.txt 2>&1; cat $O/r04_ab4_tile8.txt
