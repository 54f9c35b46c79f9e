#!/bin/bash
# usage: r05_ab_c5.sh libA.so libB.so ...  -- interleaved A/B on C5's scene (4099 spheres) at 1200x800x64spp, and on C2
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/ab_bench.py "$@" --rounds 5 --grid 32 --spp 64 2>&1 | grep median
python tools/ab_three.py "$@" 2>&1 | tail -6
