# per-kernel rocprofv3 averages of one bench.py configuration: bash tools/trace_small.sh <name> [bench.py arguments ...]
NAME=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-secondary "$@" > $O/trace.log 2>&1 || exit 1
grep -h '^{' $O/trace.log | python3 -c 'import sys, json; d = json.loads(sys.stdin.readline()); print("ms_per_step", d["ms_per_step"], "value", d["value"])'
python3 - $O/trace <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("   %-90s calls %6s avg %9.2f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
