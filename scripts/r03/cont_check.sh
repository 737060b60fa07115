run() {
  timeout 250 python bench.py --workload $1 --envs-per-gpu $2 --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1@$2 %.3f M/s ms/step %.4f kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_us']))"
}
run c3 4096 1000 200
run c5 2048 200 100
run c3s1 4096 400 200
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_scale.py tests/test_numpy_stream.py -m gpu -x -q -k "continuous or c5 or C5" 2>&1 | tail -1
