timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "mfma" > gpurun_out/t_mfma.log 2>&1; tail -3 gpurun_out/t_mfma.log
timeout -k 10 200 python bench.py --arch wide4x128 --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/b_w4x128.log 2>&1 || exit 1
PINN_MFMA_GEN=2 timeout -k 10 200 python bench.py --arch wide4x128 --no-secondary --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/b_w4x128_g2.log 2>&1 || exit 1
python - <<PY
import json
for f in ("b_w4x128","b_w4x128_g2"):
    d=json.loads(open("gpurun_out/%s.log"%f).read().strip().splitlines()[-1])
    print(f, d["config"]["arch"], round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4), round(d["roofline"]["frac"],4), d["loss"]["Ltot"])
PY
