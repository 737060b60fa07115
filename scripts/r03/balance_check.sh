for nb in 0 1; do
  for w in "c2 4096" "c2 8192" "c2 16384" "c1 4096"; do set -- $w
    if [ $nb = 1 ]; then export PCT_NO_BALANCE=1; else unset PCT_NO_BALANCE; fi
    timeout 200 python bench.py --workload $1 --envs-per-gpu $2 --steps 1000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('no_balance=$nb $1@$2 %.3f M/s ms/step %.4f kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_us']))"
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_scale.py -m gpu -x -q -k "c2_full or known_answer or discrete_s2_10_80_50 or overflow or every_step" 2>&1 | tail -3
