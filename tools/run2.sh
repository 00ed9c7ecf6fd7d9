timeout -k 10 500 python -m pytest tests -m gpu -q -x > gpurun_out/t_all.log 2>&1; tail -3 gpurun_out/t_all.log
timeout -k 10 200 python bench.py --arch wide64 --no-cpu-baseline > gpurun_out/b_w64.log 2>&1 || exit 1
PINN_MFMA_GEN=4 timeout -k 10 200 python bench.py --arch wide64x3 --no-secondary --no-cpu-baseline > gpurun_out/b_w64x3_g4.log 2>&1 || exit 1
timeout -k 10 200 python bench.py --arch wide128x3 --no-secondary --no-cpu-baseline > gpurun_out/b_w128x3.log 2>&1 || exit 1
PINN_MFMA_GEN=2 timeout -k 10 200 python bench.py --arch wide128x3 --no-secondary --no-cpu-baseline > gpurun_out/b_w128x3_g2.log 2>&1 || exit 1
PINN_MFMA_GEN=5 timeout -k 10 200 python bench.py --arch wide64 --no-secondary --no-cpu-baseline > gpurun_out/b_w64_g5.log 2>&1 || exit 1
python - <<PY
import json
for f in ("b_w64","b_w64x3_g4","b_w128x3","b_w128x3_g2","b_w64_g5"):
    d=json.loads(open("gpurun_out/%s.log"%f).read().strip().splitlines()[-1])
    print(f, d["config"]["arch"], round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4), round(d["roofline"]["frac"],4), [(s["arch"],round(s["ms_per_step"],3),round(s["roofline"]["kernel_ms"],3)) for s in d.get("secondary",[])])
PY
