# wave-priority thresholds again, on the round-3 kernels (PCT_WAVE_PRIO: tuning knob, no rebuild)
run() {
  timeout 250 python bench.py --workload $1 --envs-per-gpu $2 --steps $3 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('PRIO=${PCT_WAVE_PRIO:-default} $1@$2 %.3f M/s ms/step %.4f kernel %.1f us' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_us']))"
}
for s in default 0 12,18,24 20,26,32 16,24,32 10,16,22 18,22,26; do
  if [ $s = default ]; then unset PCT_WAVE_PRIO; else export PCT_WAVE_PRIO=$s; fi
  run c2 4096 1500
done 2>&1 | tee gpurun_out/prio_sweep2.txt
for s in default 0 30,45,60 18,28,40; do
  if [ $s = default ]; then unset PCT_WAVE_PRIO; else export PCT_WAVE_PRIO=$s; fi
  run c3 4096 600
done 2>&1 | tee -a gpurun_out/prio_sweep2.txt
