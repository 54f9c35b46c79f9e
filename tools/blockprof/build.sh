#!/bin/bash
# Builds vulkan-rtiow_amd/librtiow_hip_blk.so: the library with basic-block execution counters in ONE family of the persistent
# kernels (tools/blockprof/instrument.py).  usage: [BLOCKPROF_LANES=1] build.sh small|large|flat [kernel-name-substring]
#   BLOCKPROF_LANES=1: the counters also sum the active lanes (blocks cut at every write of EXEC): lane occupancy by region
#   small: path_persistent_kernel<true,true,true>   (C3's default kernel; its own compilation pass, SMALL_FLAGS)
#   large: path_persistent_kernel<false,true,true>  (C5's)
#   flat : path_persistent_kernel<true,false,false> (C2's; the whole-file pass)
#   compact: path_persistent_kernel<false,true,true,true> (C5's since round 5: four waves per SIMD; the whole-file pass)
#   (a kernel that fills its waves' register budget -- large, compact -- lends the counters the unused lanes of its scalar-spill register)
set -e
WHICH=${1:-small}
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vulkan-rtiow_amd/csrc
B=$R/tools/blockprof/_build
LLVM=/opt/rocm/lib/llvm/bin
N=3072
mkdir -p $B
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-parameter -DRTIOW_DEBUG_KNOBS -DRTIOW_BLOCK_COUNTERS=$N"
case $WHICH in
  small) PASS="-DRTIOW_TU_SMALL_CLUSTERED -DRTIOW_SMALL_WAVES_PER_EU=4 -Wno-unused-function -Wno-unused-const-variable"; OBJ=rtiow_kernels_small; KEY=${2:-path_persistent_kernelILb1ELb1ELb1E};;
  large) PASS="-DRTIOW_TU_LARGE_CLUSTERED -mllvm -amdgpu-sched-strategy=iterative-ilp -Wno-unused-function -Wno-unused-const-variable"; OBJ=rtiow_kernels_large; KEY=${2:-path_persistent_kernelILb0ELb1ELb1E};;
  flat)  PASS=""; OBJ=rtiow_kernels; KEY=${2:-path_persistent_kernelILb1ELb0ELb0E};;
  compact) PASS=""; OBJ=rtiow_kernels; KEY=${2:-path_persistent_kernelILb0ELb1ELb1ELb1E};;
esac
# 1. the whole library with the extended counter block (host side: allocation, zeroing, dump)
make -s -C $C OUT=../librtiow_hip_blk.so EXTRA="-DRTIOW_DEBUG_KNOBS -DRTIOW_BLOCK_COUNTERS=$N" -j8 > $B/make.log 2>&1 || { tail -20 $B/make.log; exit 1; }
# 2. offsets the inserted code needs
cat > $B/offsets.cpp <<EOC
#include <cstddef>
#include <cstdio>
#include "$C/rtiow_device.h"
int main() { printf("%zu %zu %zu\n", offsetof(rtiow::PathArgs, counters), offsetof(rtiow::Counters, block_counts), sizeof(rtiow::Counters)); }
EOC
/opt/rocm/bin/hipcc -std=c++17 --offload-arch=gfx950 -DRTIOW_BLOCK_COUNTERS=$N -o $B/offsets $B/offsets.cpp 2> /dev/null
read OFF_COUNTERS OFF_BLOCKS SIZE < <($B/offsets)
# 3. the device assembly of that pass (with line tables: report.py attributes instructions to source lines), instrumented
/opt/rocm/bin/hipcc $FLAGS $PASS -gline-tables-only --cuda-device-only -S -o $B/$OBJ.s $C/rtiow_kernels.hip
python3 $R/tools/blockprof/instrument.py $B/$OBJ.s $B/${OBJ}_blk.s $B/${WHICH}_map.json $KEY $OFF_COUNTERS $OFF_BLOCKS ${BLOCKPROF_LANES:+lanes}
# 4. assemble -> code object -> fat binary -> the host half of the same translation unit around it
python3 $R/tools/blockprof/relax.py $B/${OBJ}_blk.s $B/${OBJ}_blk.dev.o   # (assembles; trampolines for branches the counters pushed out of reach)
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/${OBJ}_blk.hsaco $B/${OBJ}_blk.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input=$B/${OBJ}_blk.hsaco -output=$B/${OBJ}_blk.hipfb
/opt/rocm/bin/hipcc $FLAGS $PASS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/${OBJ}_blk.hipfb -c -o $R/vulkan-rtiow_amd/librtiow_hip_blk_obj/$OBJ.o $C/rtiow_kernels.hip
# 5. relink
OBJS=$(ls $R/vulkan-rtiow_amd/librtiow_hip_blk_obj/*.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=$C/rtiow.map -o $R/vulkan-rtiow_amd/librtiow_hip_blk.so $OBJS -ldl
cp $B/${WHICH}_map.json $R/tools/blockprof/_build/current_map.json
echo "built librtiow_hip_blk.so ($WHICH: $KEY), counters at kernarg+$OFF_COUNTERS -> +$OFF_BLOCKS, Counters $SIZE bytes"
