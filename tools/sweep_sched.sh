mkdir -p gpurun_out/r3f
O=gpurun_out/r3f/sweep.txt
: > $O
for S in 1 2; do
 for BG in 448 320 224 160; do
  GPIRT_SCHED=$S GPIRT_BG128_MIN=$BG GPIRT_TRAIL128_MIN=$BG timeout -k 10 100 python tools/factor_time.py 8192 15 >> $O 2>&1
 done
done
GPIRT_SCHED=2 GPIRT_WIN_PRE=0 timeout -k 10 100 python tools/factor_time.py 8192 15 >> $O 2>&1
GPIRT_SCHED=2 GPIRT_CHAIN_SPLITK=1 timeout -k 10 100 python tools/factor_time.py 8192 15 >> $O 2>&1
GPIRT_SCHED=1 GPIRT_DEFER=2 timeout -k 10 100 python tools/factor_time.py 8192 15 >> $O 2>&1
GPIRT_SCHED=1 GPIRT_DEFER=2 GPIRT_TRAIL128_MIN=192 timeout -k 10 100 python tools/factor_time.py 8192 15 >> $O 2>&1
GPIRT_SCHED=1 GPIRT_DEFER_AHEAD=2 timeout -k 10 100 python tools/factor_time.py 8192 15 >> $O 2>&1
grep factor $O
