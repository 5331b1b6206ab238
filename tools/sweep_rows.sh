mkdir -p gpurun_out/r3i
O=gpurun_out/r3i/rows.txt
: > $O
timeout -k 10 100 python tools/factor_time.py 8192 20 >> $O 2>&1
for W in 0 512 1024; do
  GPIRT_ROWS=1 GPIRT_ROWS_WINDOW=$W timeout -k 10 100 python tools/factor_time.py 8192 20 >> $O 2>&1
done
GPIRT_ROWS=1 GPIRT_ROWS_WINDOW=512 GPIRT_DEFER=2 timeout -k 10 100 python tools/factor_time.py 8192 20 >> $O 2>&1
GPIRT_ROWS=1 GPIRT_ROWS_WINDOW=512 GPIRT_BG128_MIN=224 GPIRT_TRAIL128_MIN=224 timeout -k 10 100 python tools/factor_time.py 8192 20 >> $O 2>&1
GPIRT_ROWS=1 GPIRT_ROWS_WINDOW=512 timeout -k 10 100 python tools/factor_time.py 12288 10 >> $O 2>&1
timeout -k 10 100 python tools/factor_time.py 12288 10 >> $O 2>&1
GPIRT_ROWS=1 GPIRT_ROWS_WINDOW=512 timeout -k 10 100 python tools/factor_time.py 16384 6 >> $O 2>&1
timeout -k 10 100 python tools/factor_time.py 16384 6 >> $O 2>&1
grep factor $O
