for s in 0.5 0.58 0.64 0.70 0.76; do
  export PINN_RNET_SHARE=$s
  bash tools/trace_small.sh rnet_share_$s --arch wide64 --steps 200 --warmup 20 2>&1 | grep "ms_per_step\|k_rnet_mfma_bwd" | tr '\n' ' '; echo " <- rnet share $s"
done
