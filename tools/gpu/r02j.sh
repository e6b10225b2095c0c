mkdir -p gpurun_out/r02j
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --no-cpu-baseline > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err
tail -2 gpurun_out/r02j/bench.err; cut -c1-330 gpurun_out/r02j/bench.json
python bench.py --no-cpu-baseline --no-graph --steps 5 --warmup 2 2>&1 | grep -v amdgpu | cut -c80-330
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02j/bench.json'))
print({k: d['roofline'][k] for k in ('achieved', 'frac', 'launches_per_step', 'avg_launch_us', 'gemm_ms_per_step', 'traffic')})
PY
