# heavy-first dispatch, final kernels: off / last-step key / smoothed key
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_numpy_stream.py -m gpu -x -q -k "u64" 2>&1 | tail -2 | tee gpurun_out/strict_u64.txt
run() {  # workload envs steps warmup
  timeout 250 python bench.py --workload $1 --envs-per-gpu $2 --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ORDER=${PCT_ORDER:-default} SMOOTH=${PCT_ORDER_SMOOTH:-0} $1@$2 %.3f M/s ms/step %.4f kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_us']))"
}
{
run c2 4096 1000 200
PCT_ORDER=1 run c2 4096 1000 200
for sm in 0 1; do
  export PCT_ORDER_SMOOTH=$sm
  run c1 4096 300 100
  run c3s1 4096 200 80
  run c3 4096 500 100
  run c5 2048 100 60
  run c2 16384 500 100
done
unset PCT_ORDER_SMOOTH
} 2>&1 | tee gpurun_out/order_check2.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy_first" 2>&1 | tail -2 | tee -a gpurun_out/order_check2.txt
