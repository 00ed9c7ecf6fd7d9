# ab_trace.sh NAME LIB [LIB ...] -- per-kernel average durations (rocprofv3 kernel trace) of the default bench.py step with each
# build of the library (PINN_HIP_LIB; "tree" = the working tree's), one after the other on the same box, twice (A B A B):
#   bash tools/ab_trace.sh r4_rnet build_ab/libpinn_hip_base.so tree
NAME=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for LIB in "$@"; do
    if [ "$LIB" = tree ]; then unset PINN_HIP_LIB; TAG=tree; else export PINN_HIP_LIB=$R/$LIB; TAG=$(basename $LIB .so); fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_$round -o t -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-prewarm > $O/${TAG}_$round.log 2>&1 || exit 1
    echo "== $TAG round $round: $(tail -1 $O/${TAG}_$round.log | python3 -c 'import sys, json; d = json.loads(sys.stdin.readline()); print("ms_per_step", d["ms_per_step"])')"
    python3 - $O/${TAG}_$round <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("   %-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
