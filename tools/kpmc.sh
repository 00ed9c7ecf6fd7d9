# PMC passes for ONE trunk-kernel variant (tools/kbench.py, PINN_MFMA_VARIANT): bash tools/kpmc.sh <arch> <variant> <name> [points]
# raw output under gpurun_out/<name>/, per-kernel means printed.  The program after `--` is python3 itself.
ARCH=$1; VAR=$2; NAME=$3; PTS=${4:-1000000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_a -o p -- python3 $R/tools/kbench.py --arch $ARCH --variants $VAR --rounds 1 --steps 5 --points $PTS > $O/pmc_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_b -o p -- python3 $R/tools/kbench.py --arch $ARCH --variants $VAR --rounds 1 --steps 5 --points $PTS > $O/pmc_b.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O/pmc_c -o p -- python3 $R/tools/kbench.py --arch $ARCH --variants $VAR --rounds 1 --steps 5 --points $PTS > $O/pmc_c.log 2>&1 || echo "pass c failed (counter names?)"
cd $R && python3 - <<PY
import csv, glob, collections
for tag in ("a", "b", "c"):
    for f in glob.glob("gpurun_out/$NAME/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "trunk" in k:
                for c, val in sorted(v.items()):
                    print("%-72s %-28s %.4g" % (k, c, val / cnt[(k, c)]))
PY
