#!/bin/bash
# where do the path kernel's HBM writes come from?  WRITE_SIZE / FETCH_SIZE of the cover frame under the knobs build with
# one knob turned at a time (traffic <= 3x the 3.84 MB of the frame)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=vulkan-rtiow_amd/librtiow_hip_knobs.so
bash tools/pmc_traffic.sh t_default $L
RTIOW_DEBUG_CHUNK_UNTIL=0 bash tools/pmc_traffic.sh t_allwhole $L
RTIOW_DEBUG_CHUNK_UNTIL=4000000000 bash tools/pmc_traffic.sh t_nowhole $L
RTIOW_DEBUG_THREADS=256 bash tools/pmc_traffic.sh t_256 $L
RTIOW_DEBUG_NO_ORDER=1 bash tools/pmc_traffic.sh t_noorder $L
RTIOW_DEBUG_POOL_PIX=16 bash tools/pmc_traffic.sh t_pool16 $L
