# steady-state A/B of library builds and/or PINN_MFMA_VARIANT values on ONE box, separate processes (see ab_steady.sh):
#   bash tools/ab_libs.sh <arch> <points> <spec> [<spec> ...]      spec = LIB[:VARIANT], LIB = "tree" or a path under the repo
ARCH=$1; PTS=$2; shift 2
for rep in 1 2 3; do
  for spec in "$@"; do
    LIB=${spec%%:*}; V=0; case "$spec" in *:*) V=${spec##*:};; esac
    if [ "$LIB" = tree ]; then unset PINN_HIP_LIB; else export PINN_HIP_LIB=$PWD/$LIB; fi
    PINN_MFMA_VARIANT=$V python bench.py --arch $ARCH --points $PTS --steps 300 --warmup 100 --no-secondary --no-cpu-baseline 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$spec rep $rep: kernel %.4f ms  step %.4f ms  frac %.4f  parity %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d.get('parity_observed', {}).get('all_points', {}))[:120]))"
  done
done
