# heavy-first dispatch (pct_order_kernel): PCT_ORDER=0 (off) against the default (on where the launch outgrows the chip)
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_numpy_stream.py -m gpu -x -q -k "u64" 2>&1 | grep -E "Error|error|passed|failed" | head -12 > gpurun_out/strict_u64.txt
cat gpurun_out/strict_u64.txt
run() {  # workload envs steps warmup
  timeout 250 python bench.py --workload $1 --envs-per-gpu $2 --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PCT_ORDER=${PCT_ORDER:-default} $1@$2 %.3f M/s ms/step %.4f kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_us']))"
}
for o in default 0; do
  if [ $o = default ]; then unset PCT_ORDER; else export PCT_ORDER=$o; fi
  run c1 4096 300 100
  run c3s1 4096 200 80
  run c3 4096 500 100
  run c5 2048 100 60
  run c2 16384 500 100
  run c2 8192 500 100
done 2>&1 | tee gpurun_out/order_check.txt
unset PCT_ORDER
PCT_ORDER=1 run c2 4096 1000 200 | tee -a gpurun_out/order_check.txt
run c2 4096 1000 200 | tee -a gpurun_out/order_check.txt
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_scale.py -m gpu -x -q -k "heavy_first or c2_full or known_answer or discrete_s2_10_80_50 or overflow or every_step or continuous_s1 or discrete_s1" 2>&1 | tail -3 | tee -a gpurun_out/order_check.txt
