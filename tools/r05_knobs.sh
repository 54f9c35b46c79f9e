#!/bin/bash
# Round 5: the launch constants that rounds 3-4 tuned for two 512-thread groups per CU, looked at again for one 1024-thread group with 64 pass records
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== 1/8 tile (tile 0 of 8), pool sizes and the fine head =="
python tools/knob_ab.py --tile 8 --rounds 9 default POOL_PIX=1 POOL_PIX=2 POOL_PIX=3 POOL_PIX=6 FINE_DIV=0 FINE_DIV=4 FINE_DIV=16 2>&1 | grep median
echo "== 1/8 tile, tile 5 of 8 (the rim of the glass ball) =="
python tools/knob_ab.py --tile 8 --rank 5 --rounds 9 default POOL_PIX=2 FINE_DIV=4 FINE_DIV=16 2>&1 | grep median
echo "== whole frame, pool sizes =="
python tools/knob_ab.py --rounds 7 default POOL_PIX=6 POOL_PIX=12 POOL_PIX=18 POOL_SHARE=4 POOL_SHARE=16 2>&1 | grep median
