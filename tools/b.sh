timeout -k 10 300 python bench.py --arch ref16 --no-cpu-baseline > gpurun_out/b_new.log 2>&1 || exit 1
PINN_TRUNK16_MFMA=0 timeout -k 10 300 python bench.py --arch ref16 --no-cpu-baseline > gpurun_out/b_old.log 2>&1 || exit 1
python - <<PY
import json
for f in ("b_new","b_old"):
    d=json.loads(open("gpurun_out/%s.log"%f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4), round(d["roofline"]["frac"],3), [(s["arch"],round(s["ms_per_step"],4),round(s["roofline"]["kernel_ms"],4)) for s in d["secondary"]])
PY
for n in 4000 100000; do timeout -k 10 100 python bench.py --arch ref16 --points $n --no-secondary --no-cpu-baseline > gpurun_out/b_n$n.log 2>&1; python -c "
import json; d=json.loads(open('gpurun_out/b_n$n.log').read().strip().splitlines()[-1]); print($n, d['ms_per_step'], d['value'])"; done
