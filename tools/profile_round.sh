# rocprofv3 passes for one bench.py configuration (each PMC set is its own run, no trace options combined
# with --pmc); raw output under gpurun_out/<name>/, summary -> gpurun_out/<name>/summary.md
#   bash tools/profile_round.sh <name> [bench.py arguments ...]
# The program after `--` is python3 itself (no env/bash hop: the profiler's preload initialises the GPU).
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$NAME
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $O/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_a -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prewarm "$@" > $O/pmc_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_b -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prewarm "$@" > $O/pmc_b.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prewarm "$@" > $O/pmc_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prewarm "$@" > $O/pmc_w.log 2>&1 || exit 1
cd $R && python3 tools/summarize_prof.py $O > $O/summary.md && tail -5 $O/summary.md
