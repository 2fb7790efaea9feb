for d in 0 4 8 2; do echo "== RICK_CONV_DEBUG=$d"; RICK_CONV_DEBUG=$d timeout 300 python tools/bench_conv.py wgrad 2>&1 | grep -E "512 @ 64|@256" | head -4; done
