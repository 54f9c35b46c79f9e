for v in 0 1 0 1; do
  echo "== exact lean kernel: $v"
  if [ $v = 1 ]; then export RTIOW_DEBUG_CH_LEAN=1; else unset RTIOW_DEBUG_CH_LEAN; fi
  RTIOW_LIB=vulkan-rtiow_amd/librtiow_hip_knobs.so python tools/ch_bandwidth.py 2>&1 | grep CH06
done | tee gpurun_out/ch_variants.txt
