# kernel time of one arch against PINN_OLDER_SHARE (the first grid half's share of the tiles): bash tools/share_sweep.sh <arch> [points] [shares ...]
ARCH=${1:-wide64}; PTS=${2:-1000000}; shift 2
SH=${@:-0.5 0.58 0.60 0.62 0.64 0.66 0.68}
for s in $SH; do
  PINN_OLDER_SHARE=$s python bench.py --arch $ARCH --points $PTS --steps 300 --warmup 100 --no-secondary --no-cpu-baseline 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ARCH share $s: kernel %.4f ms  step %.4f ms  frac %.4f' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac']))"
done
