timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/t_all.log 2>&1; tail -2 gpurun_out/t_all.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b_new.log 2>&1 || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/b_new.log").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4), [(s["arch"],round(s["ms_per_step"],4),round(s["roofline"]["kernel_ms"],4)) for s in d["secondary"]])
PY
