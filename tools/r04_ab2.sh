#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
L="tools/_ab/pack.so tools/_ab/g.so tools/_ab/gc.so tools/_ab/lay.so tools/_ab/laync.so"
python tools/ab_bench.py $L --rounds 9 > $O/r04_ab3_full.txt 2>&1; cat $O/r04_ab3_full.txt
python tools/ab_bench.py $L --rounds 9 --tile 8 > $O/r04_ab3_tile8.txt 2>&1; cat $O/r04_ab3_tile8.txt
