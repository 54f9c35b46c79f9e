#!/bin/bash
# usage: r05_ab.sh libA.so libB.so ...   -- interleaved A/B of the cover frame, its 1/8 tile, C2 and a parity smoke, on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/ab_bench.py "$@" --rounds 7 2>&1 | grep median
python tools/ab_bench.py "$@" --rounds 9 --tile 8 2>&1 | grep median
python tools/ab_three.py "$@" 2>&1 | tail -4
