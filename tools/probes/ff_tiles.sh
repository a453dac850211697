mkdir -p gpurun_out/ff
for s in "8192 2560 320 g" "2048 5120 640 g" "512 10240 1280 g" "8192 320 1280" "2048 640 2560" "512 1280 5120" "512 1280 1280" "8192 320 320"; do
  timeout -k 10 120 python tools/gemm_tiles.py $s >> gpurun_out/ff/tiles.txt 2>&1 || exit 1
done
