mkdir -p gpurun_out/r3y
O=gpurun_out/r3y/sweep.txt; : > $O
timeout -k 10 120 python tools/iter_time.py 4 10 >> $O 2>&1
for v in "GPIRT_NBP=384" "GPIRT_NBP=448" "GPIRT_NBP=576" "GPIRT_NBP=640" "GPIRT_DEFER=2" "GPIRT_DEFER_AHEAD=2" "GPIRT_DEFER_SPLIT=1" "GPIRT_BG128_MIN=320 GPIRT_TRAIL128_MIN=320" "GPIRT_HOLD_REST=3" "GPIRT_AUX_PRIO=2"; do
  env $v timeout -k 10 120 python tools/iter_time.py 4 10 >> $O 2>&1
done
grep -v amdgpu $O | cut -c1-60,170-
