#!/bin/bash
# Per-shape kernel / tile choice for the ResNet-50 convolutions at several per-GPU batch sizes, one MI355X:
#   bench/tile_table.sh OUT_DIR [BATCHES="256 512 1024"]
# For every batch size each forward / data-gradient shape is timed with the automatic choice, with each tile of the
# LDS-DMA kernel forced, and on the register-staged implicit-GEMM kernel (PDT_CONV_DMA=0); weight gradients with the
# automatic choice and each forced tile.  bench/tile_table.py joins the CSV files into one table.
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/tiles}"
BATCHES="${2:-256 512 1024}"
mkdir -p "$OUT"
cd "$ROOT"
for b in $BATCHES; do
  for t in 0 -3 -4; do
    timeout -k 10 170 python bench/conv_layers.py --batch "$b" --ops fprop,dgrad --miopen 0 --tile "$t" --csv "$OUT/b${b}_dma_t${t#-}.csv" > "$OUT/b${b}_dma_t${t#-}.txt"
    echo "batch $b dma tile $t done"
  done
  PDT_CONV_DMA=0 timeout -k 10 170 python bench/conv_layers.py --batch "$b" --ops fprop,dgrad --miopen 0 --csv "$OUT/b${b}_igemm.csv" > "$OUT/b${b}_igemm.txt"
  echo "batch $b igemm done"
  for w in 0 1 2 4; do
    timeout -k 10 170 python bench/conv_layers.py --batch "$b" --ops wgrad --miopen 0 --wtile "$w" --layers a1,a3,a4,b1,b3,b4,b5,b6,c1,c2,c3,c4,c5,c6,d1,d2,d3,d4,d5,d6 --csv "$OUT/b${b}_wgrad_t${w}.csv" > "$OUT/b${b}_wgrad_t${w}.txt"
    echo "batch $b wgrad tile $w done"
  done
done
python bench/tile_table.py "$OUT" $BATCHES | tee "$OUT/tile_table.txt"
