for d in 0 1 2 4 6 7; do echo "== RICK_CONV_DEBUG=$d"; RICK_CONV_DEBUG=$d timeout 300 python tools/bench_conv.py fprop 2>&1 | grep -E "512 @ 64|128 @256" | head -2; done
