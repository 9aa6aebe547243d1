#!/bin/bash
# geometry sweep of the strided NTT passes (SP_NTT_TILE_BIG / SP_NTT_TILE_SMALL = log2 of the minimum tile)
for cfg in "10 9" "9 9" "10 10" "8 8"; do
  set -- $cfg
  export SP_NTT_TILE_BIG=$1 SP_NTT_TILE_SMALL=$2
  echo "== big=$1 small=$2"
  python tools/ntt_batch_bench.py 22 1 2>&1 | tail -1
  python tools/ntt_batch_bench.py 22 34 2>&1 | tail -1
  python tools/prove_bench.py 149000 8 80 20 2>&1 | tail -1
done
