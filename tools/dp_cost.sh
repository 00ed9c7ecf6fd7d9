# what the data-parallel step costs with ONE rank on one GPU (launch-only cost of the exchange path; no second device here):
# plain step, RCCL all-reduce forced, one-shot exchange, one-shot exchange inside a captured graph.   bash tools/dp_cost.sh
mkdir -p gpurun_out
B="python bench.py --steps 300 --warmup 100 --no-secondary --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); print("%-34s step %.4f ms  kernel %.4f ms  exchange %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"] if d["roofline"]["kernel_ms"]==d["roofline"]["kernel_ms"] else float("nan"), d["config"]["exchange"]))'
for rep in 1 2; do
  $B 2>/dev/null | python -c "$P" "plain"
  PINN_FORCE_DIST=1 $B 2>/dev/null | python -c "$P" "RCCL all-reduce (1 rank)"
  PINN_FORCE_DIST=1 $B --one-shot 2>/dev/null | python -c "$P" "one-shot exchange (1 rank)"
  PINN_FORCE_DIST=1 $B --one-shot --graph 2>/dev/null | python -c "$P" "one-shot exchange, captured graph"
  $B --graph 2>/dev/null | python -c "$P" "plain, captured graph"
done
