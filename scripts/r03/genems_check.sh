mkdir -p gpurun_out
run() {
  timeout 250 python bench.py --workload $1 --envs-per-gpu $2 --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1@$2 %.3f M/s ms/step %.4f kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_us']))"
}
{
run c3 4096 500 100
run c5 2048 100 60
run c3s1 4096 200 80
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_scale.py tests/test_numpy_stream.py -m gpu -x -q -k "continuous or c5 or C5" 2>&1 | tail -2
timeout 300 python scripts/step_profile.py 4096 40 c1 2>/dev/null | tail -8
timeout 300 python scripts/step_profile.py 4096 40 c3 2>/dev/null | grep -v "^    [a-z].*cycles  \|0.0   (all" | tail -30
timeout 300 python scripts/step_profile.py 2048 20 c5 2>/dev/null | grep -v "0.0   (all" | tail -34
} 2>&1 | tee gpurun_out/genems_check.txt
