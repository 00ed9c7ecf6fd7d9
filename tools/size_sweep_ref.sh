for arch in ref16 ref16ft; do
for n in 1000 4000 16000 100000; do
  python bench.py --arch $arch --points $n --steps 500 --warmup 50 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$arch n=$n: trunk kernel %.2f us  step %.2f us  value %.3e' % (1e3*d['roofline']['kernel_ms'], 1e3*d['ms_per_step'], d['value']))"
done; done
