#!/bin/bash
# Round 5, first call: the kernels of round 4's HEAD under the new instruction-class passes (C3), C2's kernel under every pass
# (it had no profile since round 1), and the tile / dbg figures of this box as the A/B baseline.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
bash tools/profile.sh r05_base "" noch > $O/r05_base_profile.log 2>&1; tail -3 $O/r05_base_profile.log
bash tools/profile.sh r05_c2_base "--workload three_400x225_100spp" noch > $O/r05_c2_base_profile.log 2>&1; tail -3 $O/r05_c2_base_profile.log
python tools/tile_timing.py > $O/r05_base_tile_timing.txt 2>&1; tail -12 $O/r05_base_tile_timing.txt
RTIOW_LIB=$PWD/vulkan-rtiow_amd/librtiow_hip_dbg.so RTIOW_DEBUG_HIST=1 python tools/dbg_counters.py 100 > $O/r05_base_dbg_counters.txt 2>&1; grep -v "waves d" $O/r05_base_dbg_counters.txt | cut -c1-400
