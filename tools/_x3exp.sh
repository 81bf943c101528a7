for d in 0 32 64 96 128 40 72 104; do echo "dbg=$d"; HTD_CONV_DBG=$d python tools/sweep_conv_tiles.py "fpn P2" 2>/dev/null | tail -1; done
