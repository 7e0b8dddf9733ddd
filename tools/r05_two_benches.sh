mkdir -p gpurun_out/r5
python bench.py --steps 30 --warmup 10 --no-legs --no-micro --no-cpu-baseline > gpurun_out/r5/bench_quick.log 2> gpurun_out/r5/bench_quick.err; grep "timed region" gpurun_out/r5/bench_quick.err
python bench.py --model frcnn --steps 10 --warmup 4 --no-micro --no-cpu-baseline > gpurun_out/r5/bench_frcnn.log 2> gpurun_out/r5/bench_frcnn.err; grep "timed region" gpurun_out/r5/bench_frcnn.err
python - <<'PY'
import json
d=json.load(open('bench_detail.json'))
pk=d['roofline'].get('per_kernel',{})
print(d['ms_per_step'], {k:(v.get('sum_ms'),v.get('calls')) for k,v in pk.items() if 'conv3x3' in k})
PY
