set -e
python -m pytest tests/test_ll_fast.py -q -m gpu -x 2>&1 | tail -5
python bench.py --no-reference-rng --steps 60 --warmup 10 > gpurun_out/bench_screen_on.json 2> gpurun_out/bench_screen_on.err
GPIRT_ESS_SCREEN=2 python bench.py --no-reference-rng --steps 60 --warmup 10 > gpurun_out/bench_screen_off.json 2> gpurun_out/bench_screen_off.err
python - <<'PY'
import json
for t in ("on","off"):
    d=json.loads(open(f"gpurun_out/bench_screen_{t}.json").read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["config"].get("stage_ms"))
PY
