run() {
  timeout 250 python bench.py --workload $1 --envs-per-gpu $2 --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('AHEAD=${PCT_ORDER_AHEAD:-1} $1@$2 %.3f M/s ms/step %.4f kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_us']))"
}
for a in 1 0; do export PCT_ORDER_AHEAD=$a
  run c1 4096 300 100
  run c3s1 4096 200 80
  run c3 4096 500 100
  run c5 2048 100 60
  run c2 16384 500 100
  run c2 8192 500 100
done 2>&1 | tee gpurun_out/ahead_check.txt
unset PCT_ORDER_AHEAD
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy_first or macs or MACS or 64bit or heuristics" 2>&1 | tail -2 | tee -a gpurun_out/ahead_check.txt
