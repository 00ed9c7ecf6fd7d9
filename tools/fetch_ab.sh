# HBM-side traffic of one trunk kernel with several library builds: bash tools/fetch_ab.sh <arch> <points> <lib|tree> ...
ARCH=$1; PTS=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  if [ "$LIB" = tree ]; then unset PINN_HIP_LIB; TAG=tree; else export PINN_HIP_LIB=$R/$LIB; TAG=$(basename $LIB .so); fi
  for C in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/fetch_$TAG/$C; mkdir -p $O
    rocprofv3 --pmc $C --output-format csv -d $O -o p -- python3 $R/tools/kbench.py --arch $ARCH --variants 0 --rounds 1 --steps 5 --points $PTS > $O.log 2>&1 || exit 1
    python3 - $O $C $TAG <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trunk" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("%-22s %-12s %-60s mean %.4g KB per launch" % (sys.argv[3], sys.argv[2], k, sum(v) / len(v)))
PY
  done
done
