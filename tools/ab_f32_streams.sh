P=$PWD/rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/lib
for rep in 1 2; do
for k in "4=0" "4=1"; do
  printf "knob %s  " $k
  ROVIT_HIP_LIB=$P/librovit_hip_dev.so ROVIT_DEV_KNOBS=$k python tools/time_fp32.py 16 32 64 128 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v['ms_per_forward'] for k, v in d.items()})"
done
done
