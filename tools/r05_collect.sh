#!/bin/bash
# After tools/r05_final.sh, tools/r05_blockprof.sh run and tools/r05_profiles.sh (GPU box): gpurun_out/ -> profiles/ (here, no GPU)
cd /root/repo
O=gpurun_out; P=profiles
for f in gpu_tests all_configs tile_timing dbg_counters timeline inflight scene_size_sweep ch_bandwidth_plain; do cp $O/r05_$f.txt $P/r05_$f.txt; done
for f in gpu_tests all_configs tile_timing dbg_counters timeline inflight scene_size_sweep ch_bandwidth_plain; do sed -i '/amdgpu.ids/d' $P/r05_$f.txt; done
sed -i '1i cover scenes of growing size at 1200x800x32spp (tools/grid_stats.py, the knobs build for RTIOW_DEBUG_LAUNCH: what launch_path chose)' $P/r05_scene_size_sweep.txt
for f in default three_400x225_100spp cover_1200x800_500spp cover4096_3840x2160_1024spp; do cp $O/r05_bench_$f.json $P/r05_bench_$f.json; done
grep '^{"metric"' $O/r05_bench_2rank_gloo_rehearsal_one_gpu.json > $P/r05_bench_2rank_gloo_rehearsal_one_gpu.json
for t in r05 r05_c2 r05_c5; do python3 tools/profile_summary.py $t | tail -1 | cut -c1-200; done
python3 tools/pmc_ch_summary.py r05 | tail -1 | cut -c1-200
R="python3 tools/blockprof/report.py"
$R $O/blk_c3_map.json $O/blk_c3.txt $P/r05_pmc.json "path_persistent_kernel<true,true,true,false>" --lines 40 > $P/r05_instruction_mix.txt
$R $O/blk_c3_lanes_map.json $O/blk_c3_lanes.txt $P/r05_pmc.json "path_persistent_kernel<true,true,true,false>" --lines 40 > $P/r05_lane_occupancy.txt
$R $O/blk_c2_map.json $O/blk_c2.txt $P/r05_c2_pmc.json "path_persistent_kernel<true,false,false,false>" --lines 30 > $P/r05_c2_instruction_mix.txt
$R $O/blk_c5_lanes_map.json $O/blk_c5_lanes.txt $P/r05_c5_pmc.json "path_persistent_kernel<false,true,true,true>" --lines 40 > $P/r05_c5_lane_occupancy.txt
for f in instruction_mix lane_occupancy c2_instruction_mix c5_lane_occupancy; do sed -n 5,6p $P/r05_$f.txt; grep "as the SQ counters take them" $P/r05_$f.txt | cut -c1-200; done
git status --short | head -60
