set -e
timeout -k 10 300 python -m pytest tests/test_gpu_theta_fixed.py -q -m gpu -x 2>&1 | tail -15
python bench.py --no-reference-rng --steps 60 --warmup 10 > gpurun_out/bench_tf_on.json 2> gpurun_out/bench_tf_on.err
GPIRT_THETA_FIXED=2 python bench.py --no-reference-rng --steps 60 --warmup 10 > gpurun_out/bench_tf_off.json 2> gpurun_out/bench_tf_off.err
python - <<'PY'
import json
for t in ("on","off"):
    d=json.loads(open(f"gpurun_out/bench_tf_{t}.json").read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["config"].get("stage_ms"))
PY
