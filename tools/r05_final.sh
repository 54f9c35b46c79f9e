#!/bin/bash
# Round 5, the measurement run on the final kernels (one gpurun call): the GPU suite, the bench lines of the BASELINE configurations, the
# table of DESIGN 5, the tiles, the diagnostic build's stage shares, the timeline of a small frame.  (tools/profile.sh -- kernel stats and PMC
# passes -- and tools/blockprof are calls of their own.)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r05_gpu_tests.txt 2>&1 || { tail -40 $O/r05_gpu_tests.txt; exit 1; }
tail -2 $O/r05_gpu_tests.txt
python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err || { tail -20 $O/r05_bench_default.err; exit 1; }
python -c "import json; d=json.load(open('$O/r05_bench_default.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['frame_check'], d['config'].get('first_frame_ms'), d['config'].get('moving_camera_penalty'))"
for W in three_400x225_100spp cover_1200x800_500spp; do python bench.py --workload $W --no-cpu-baseline > $O/r05_bench_$W.json 2>> $O/r05_bench_default.err; done
python bench.py --workload cover4096_3840x2160_1024spp --steps 3 --warmup 1 --no-cpu-baseline > $O/r05_bench_cover4096_3840x2160_1024spp.json 2>> $O/r05_bench_default.err
for W in three_400x225_100spp cover_1200x800_500spp cover4096_3840x2160_1024spp; do python -c "import json; d=json.load(open('$O/r05_bench_$W.json')); print('$W', d['ms_per_step'], d['roofline']['frac'], d['frame_check'])"; done
python tools/all_configs.py > $O/r05_all_configs.txt 2>&1; cat $O/r05_all_configs.txt
python tools/tile_timing.py > $O/r05_tile_timing.txt 2>&1; cat $O/r05_tile_timing.txt
RTIOW_LIB=$PWD/vulkan-rtiow_amd/librtiow_hip_dbg.so RTIOW_DEBUG_HIST=1 python tools/dbg_counters.py 100 > $O/r05_dbg_counters.txt 2>&1; grep -v "waves d" $O/r05_dbg_counters.txt | cut -c1-300
RTIOW_DEBUG_HIST=1 RTIOW_LIB=$PWD/vulkan-rtiow_amd/librtiow_hip_tl.so python tools/timeline.py 8 1 > $O/r05_timeline.txt 2>&1; grep -E "iterations after|queue 0 head|G=" $O/r05_timeline.txt | cut -c1-200
python tools/inflight.py > $O/r05_inflight.txt 2>&1; tail -8 $O/r05_inflight.txt
BENCH_ONE_DEVICE=1 BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $O/r05_bench_2rank_gloo_rehearsal_one_gpu.json 2> $O/r05_bench_2rank.err; grep "^{\"metric" $O/r05_bench_2rank_gloo_rehearsal_one_gpu.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"2-rank gloo rehearsal\", d[\"ms_per_step\"], d[\"config\"][\"gathered_frame_vs_single_gpu_frame\"], d[\"frame_check\"])" || tail -5 $O/r05_bench_2rank.err
# (the knobs build -- the shipped kernels -- for RTIOW_DEBUG_LAUNCH: what launch_path chose for each scene)
RTIOW_LIB=$PWD/vulkan-rtiow_amd/librtiow_hip_knobs.so RTIOW_DEBUG_LAUNCH=1 GRIDS=11,12,14,16,20,22,24,28,30,32,34,36,38,39 python tools/grid_stats.py 2>&1 | grep "grid\|launch" | uniq > $O/r05_scene_size_sweep.txt; cat $O/r05_scene_size_sweep.txt
python tools/ch_bandwidth.py 800x608 4096x4096 16384x8192 16384x16384 2> /dev/null | grep "^CH0" > $O/r05_ch_bandwidth_plain.txt; cat $O/r05_ch_bandwidth_plain.txt
