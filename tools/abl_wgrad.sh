# wgrad ablations (RICK_CONV_DEBUG bits: 1 = no MFMA phase, 2|4 = convert/LDS-store only for the first tile, 8|2 = no prefetch loads)
for d in 0 1 6 10 14; do echo "== RICK_CONV_DEBUG=$d"; RICK_CONV_DEBUG=$d timeout 300 python tools/bench_conv.py wgrad 2>&1 | grep -E "512 @ 64|128 @256" | head -2; done
