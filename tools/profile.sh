#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box):
#   1. --kernel-trace --stats        per-kernel durations of the exact bench command
#   2. --pmc FETCH_SIZE / WRITE_SIZE HBM traffic of each kernel (separate passes: TCC slots)
#   3. --pmc SQ_*                    VALU / LDS / wait mix of the path kernel (+ 4. lane occupancy, 5. ch_kernel)
# Raw output goes to gpurun_out/prof_<tag>_*/ ; tools/profile_summary.py condenses it into profiles/.
set -e
TAG=${1:-r01}
EXTRA=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -- $BENCH > $O/prof_${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_fetch -- $BENCH > $O/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}_write -- $BENCH > $O/prof_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d $O/prof_${TAG}_sq -- $BENCH > $O/prof_${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE \
    --output-format csv -d $O/prof_${TAG}_lds -- $BENCH > $O/prof_${TAG}_lds.log 2>&1 || true
#   4. lane occupancy of the vector instructions: SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU), and the rest of the mix
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC \
    --output-format csv -d $O/prof_${TAG}_lane -- $BENCH > $O/prof_${TAG}_lane.log 2>&1 || true
#   4b. the executed instruction mix by class (round 5: tools/instruction_mix.py reads it): the vector classes the SQ counts apart, then the rest
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 \
    --output-format csv -d $O/prof_${TAG}_mixa -- $BENCH > $O/prof_${TAG}_mixa.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM \
    --output-format csv -d $O/prof_${TAG}_mixb -- $BENCH > $O/prof_${TAG}_mixb.log 2>&1 || true
cd $R && python3 tools/profile_summary.py $TAG
#   5. the reference's own kernels at sizes where bytes matter: kernel stats + PMC passes of tools/ch_bandwidth.py
#      (third argument "noch": not for this tag -- e.g. the C5 leg: profile.sh r04_c5 "--workload cover4096_3840x2160_64spp" noch)
if [ "${3:-}" != "noch" ]; then bash $R/tools/pmc_ch.sh $TAG || true; fi
