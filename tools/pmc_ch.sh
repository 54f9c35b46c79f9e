#!/bin/bash
# PMC pass for the reference's own kernels at sizes where bytes matter (tools/ch_bandwidth.py: CH06 at 800x608 ...
# 16384^2): instruction mix, waits and HBM traffic of ch_kernel_rows.  Raw csv under gpurun_out/prof_<tag>_ch*/,
# condensed by tools/pmc_ch_summary.py into profiles/<tag>_ch_pmc.json.  Separate passes per counter group.
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python3 $R/tools/ch_bandwidth.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_ch -- $CMD > $O/prof_${TAG}_ch.log 2>&1
export CH_BW_NO_CH05=1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d $O/prof_${TAG}_ch_sq -- $CMD > $O/prof_${TAG}_ch_sq.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_BRANCH SQ_WAVES \
    --output-format csv -d $O/prof_${TAG}_ch_mix -- $CMD > $O/prof_${TAG}_ch_mix.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_ch_fetch -- $CMD > $O/prof_${TAG}_ch_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}_ch_write -- $CMD > $O/prof_${TAG}_ch_write.log 2>&1
cd $R && python3 tools/pmc_ch_summary.py $TAG
