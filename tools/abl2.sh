for d in 0 16; do echo "== RICK_CONV_DEBUG=$d"; RICK_CONV_DEBUG=$d timeout 300 python tools/bench_conv.py fprop 2>&1 | grep -E "512 @ 64|128 @256|@ 32" | head -4; done
