# four blocks per CU: kernel time against the geometric share ratio q (classes take 1 : q : q^2 : q^3 of the tiles)
#   k_trunk16 forward-only sweep (fine-tune): PINN_SHARE_Q, HIP events; R-net forward kernel: PINN_RNET_Q, rocprofv3 averages
for q in ${@:-1.0 0.8 0.65 0.5 0.4 0.3}; do
  PINN_SHARE_Q=$q python bench.py --arch ref16ft --points 1000000 --steps 300 --warmup 100 --no-secondary --no-cpu-baseline 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ref16ft q $q: kernel %.4f ms  step %.4f ms' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
done
for q in ${@:-1.0 0.8 0.65 0.5 0.4 0.3}; do
  export PINN_RNET_Q=$q
  bash tools/trace_small.sh rnet_q_$q --arch wide64 --steps 200 --warmup 20 2>&1 | grep "ms_per_step\|k_rnet_mfma_fwd" | tr '\n' ' '; echo " <- rnet q $q"
done
