#!/bin/bash
# Round 5: rocprofv3 kernel stats + PMC passes of the final kernels (C3 with the reference's shaders' leg, C2, C5 at 64 spp)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
bash tools/profile.sh r05 "" > $O/r05_profile.log 2>&1; tail -2 $O/r05_profile.log | cut -c1-300
bash tools/profile.sh r05_c2 "--workload three_400x225_100spp" noch > $O/r05_c2_profile.log 2>&1; tail -2 $O/r05_c2_profile.log | cut -c1-300
bash tools/profile.sh r05_c5 "--workload cover4096_3840x2160_64spp" noch > $O/r05_c5_profile.log 2>&1; tail -2 $O/r05_c5_profile.log | cut -c1-300
