timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "mfma or graph" > gpurun_out/t_mfma.log 2>&1; tail -3 gpurun_out/t_mfma.log
timeout -k 10 200 python bench.py --arch wide64 --no-cpu-baseline > gpurun_out/b_w64.log 2>&1 || exit 1
python - <<PY
import json
for f in ("b_w64",):
    d=json.loads(open("gpurun_out/%s.log"%f).read().strip().splitlines()[-1])
    print(f, d["config"]["arch"], round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4), round(d["roofline"]["frac"],4), [(s["arch"],round(s["ms_per_step"],3),round(s["roofline"]["kernel_ms"],3)) for s in d.get("secondary",[])])
PY
