for K in "ast 8" "vit 128"; do
  for F in 1 0; do
    echo "== $K fused_planes=$F"
    EAV_FUSED_PLANES=$F SPLIT_TIMES=1 python3 tools/encoder_step_bench.py $K split 2>&1 | grep -E "ms/step|forward"
  done
done
