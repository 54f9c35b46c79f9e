#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_gpu_tests_2.txt 2>&1 || { tail -40 $O/r04_gpu_tests_2.txt; exit 1; }
tail -3 $O/r04_gpu_tests_2.txt
python bench.py > $O/r04_bench_default_a.json 2> $O/r04_bench_default_a.err || { tail -20 $O/r04_bench_default_a.err; exit 1; }
cat $O/r04_bench_default_a.json
python tools/ab_bench.py tools/_ab/laync2.so tools/_ab/cur.so --rounds 9 > $O/r04_ab5_full.txt 2>&1; cat $O/r04_ab5_full.txt
