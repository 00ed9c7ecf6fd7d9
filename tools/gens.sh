for g in 1 4 5 6; do
  PINN_MFMA_GEN=$g timeout -k 10 120 python bench.py --arch wide64 --no-cpu-baseline > gpurun_out/b_w64_g$g.log 2>&1 || exit 1
done
PINN_MFMA_GEN=5 timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "mfma" > gpurun_out/t_mfma_g5.log 2>&1; tail -2 gpurun_out/t_mfma_g5.log
python - <<PY
import json
for g in (1,4,5,6):
    d=json.loads(open("gpurun_out/b_w64_g%d.log"%g).read().strip().splitlines()[-1])
    print(g, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["loss"]["Ltot"])
PY
