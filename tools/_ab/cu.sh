for u in 1000000 40000 20000 12000 9272 6000; do echo "chunk_until $u"; RTIOW_DEBUG_CHUNK_UNTIL=$u python tools/ab_bench.py tools/_ab/lines.so --rounds 7 2>&1 | grep -v amdgpu.ids; done
