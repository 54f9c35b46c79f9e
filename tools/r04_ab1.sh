#!/bin/bash
# round 4, first GPU call: parity suite on the in-tree build, then interleaved A/B of the step-by-step builds
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_gpu_tests_1.txt 2>&1 || { tail -30 $O/r04_gpu_tests_1.txt; exit 1; }
tail -3 $O/r04_gpu_tests_1.txt
L="tools/_ab/r03.so tools/_ab/lean.so tools/_ab/pack.so tools/_ab/stat.so"
python tools/ab_bench.py $L --rounds 9 > $O/r04_ab1_full.txt 2>&1; cat $O/r04_ab1_full.txt
python tools/ab_bench.py $L --rounds 9 --tile 8 > $O/r04_ab1_tile8.txt 2>&1; cat $O/r04_ab1_tile8.txt
python tools/ab_bench.py $L --rounds 7 --spp 1 > $O/r04_ab1_1spp.txt 2>&1; cat $O/r04_ab1_1spp.txt
python tools/ab_bench.py $L --rounds 3 --kernels 2 > $O/r04_ab1_flat.txt 2>&1; cat $O/r04_ab1_flat.txt
