# (needs tools/experiments/wave_tpc.patch applied to csrc/de_kernels.hip: the DE_WAVE_TPC knob is not in the shipped library.
#  Measured, round 6: C5N 1.017 (default, 15) / 1.073 (8) / 1.053 (12) / 1.004 (20) / 1.002 (24) / 1.012 (32) / 0.998 (48) / 0.997 ms (63);
#  C5 7.33 / 7.40 / 7.33 / 7.31 / 7.33 / 7.33 / 7.31 / 7.31 ms: flat within 2 % — the default stays 63 / W.)
# trees per wave of a wave group (default 63 / W): C5N (8 per-sample parameters, W = 4) and C5 per value of DE_WAVE_TPC
for wl in C5N C5; do
  for v in 0 8 12 15 20 24 32 48 63; do
    DE_WAVE_TPC=$v timeout 200 python bench.py --workload $wl --steps 20 --warmup 50 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('DE_WAVE_TPC=$v', '$wl', round(d['ms_per_step'],3))"
  done
done
