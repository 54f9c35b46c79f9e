#!/bin/bash
# The round's block-counter profiles in one GPU call.  First, HERE (no GPU needed): tools/r05_blockprof.sh build
#   -> tools/_ab/blk_{c3,c3_lanes,c2,c5_lanes}.so + their maps; then on the box: tools/r05_blockprof.sh run  -> gpurun_out/blk_*.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$(pwd)
if [ "$1" = build ]; then
  set -e
  mkdir -p tools/_ab
  bash tools/blockprof/build.sh small > /dev/null;                     cp vulkan-rtiow_amd/librtiow_hip_blk.so tools/_ab/blk_c3.so;       cp tools/blockprof/_build/small_map.json tools/_ab/blk_c3_map.json
  BLOCKPROF_LANES=1 bash tools/blockprof/build.sh small > /dev/null;   cp vulkan-rtiow_amd/librtiow_hip_blk.so tools/_ab/blk_c3_lanes.so; cp tools/blockprof/_build/small_map.json tools/_ab/blk_c3_lanes_map.json
  bash tools/blockprof/build.sh flat > /dev/null;                      cp vulkan-rtiow_amd/librtiow_hip_blk.so tools/_ab/blk_c2.so;       cp tools/blockprof/_build/flat_map.json tools/_ab/blk_c2_map.json
  BLOCKPROF_LANES=1 bash tools/blockprof/build.sh compact > /dev/null; cp vulkan-rtiow_amd/librtiow_hip_blk.so tools/_ab/blk_c5_lanes.so; cp tools/blockprof/_build/compact_map.json tools/_ab/blk_c5_lanes_map.json
  ls -la tools/_ab/blk_*.so
  exit 0
fi
O=gpurun_out
mkdir -p $O
run() {  # tag workload spp
  BLOCKPROF_LIB=$R/tools/_ab/blk_$1.so RTIOW_BLOCK_DUMP=$O/blk_$1.txt timeout -k 10 400 python tools/blockprof/run.py $2 $3 3 1 2>&1 | grep -v amdgpu.ids && cp tools/_ab/blk_$1_map.json $O/blk_$1_map.json
}
run c3 cover 100 && run c3_lanes cover 100 && run c2 three 100 && run c5_lanes cover4096 64
