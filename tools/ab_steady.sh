# steady-state A/B of two PINN_MFMA_VARIANT values on ONE box: separate processes, each long enough for the chip's clock
# control to settle on that variant alone (the interleaved rounds of tools/kbench.py can flatter a variant that stalls less:
# MI355X_MICROARCH.md "DVFS give-back").   bash tools/ab_steady.sh <arch> <variantA> <variantB> [points]
ARCH=$1; VA=$2; VB=$3; PTS=${4:-1000000}
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
  for v in $VA $VB; do
    PINN_MFMA_VARIANT=$v python bench.py --arch $ARCH --points $PTS --steps 300 --warmup 100 --no-secondary --no-cpu-baseline 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v rep $rep: kernel %.4f ms  step %.4f ms  frac %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
  done
done
