cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
G=${1:-4}
export PINN_MFMA_GEN=$G
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_a_g$G -o pmc --output-format csv -- python3 $R/bench.py --arch wide64 --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_a_g$G.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_b_g$G -o pmc --output-format csv -- python3 $R/bench.py --arch wide64 --no-secondary --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_b_g$G.log 2>&1 || exit 1
cd $R && python3 - <<PY
import csv, glob, collections
for tag in ("a","b"):
    for f in glob.glob("gpurun_out/pmc_%s_g$G/**/*counter_collection.csv"%tag, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            cnt[(k,r["Counter_Name"])]+=1
        for k,v in acc.items():
            if "trunk" in k:
                print(tag,k)
                for c,val in sorted(v.items()): print("   %-28s %.4g (per launch, %d launches)"%(c,val/cnt[(k,c)],cnt[(k,c)]))
PY
