#!/bin/bash
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r04_gpu_tests_3.txt 2>&1 || { tail -40 $O/r04_gpu_tests_3.txt; exit 1; }
tail -3 $O/r04_gpu_tests_3.txt
python tools/all_configs.py > $O/r04_all_configs_b.txt 2>&1; cat $O/r04_all_configs_b.txt
python tools/tile_timing.py > $O/r04_tile_timing_b.txt 2>&1; cat $O/r04_tile_timing_b.txt
python tools/ab_bench.py tools/_ab/cur.so tools/_ab/pool9.so --rounds 9 > $O/r04_ab6_full.txt 2>&1; cat $O/r04_ab6_full.txt
bash tools/pmc_traffic.sh t_pool9 vulkan-rtiow_amd/librtiow_hip.so "--steps 5" > $O/r04_traffic_pool9.txt 2>&1; cat $O/r04_traffic_pool9.txt
