# the constant fill beside the index build (MASHGPU_FILL_ASIDE = workgroups,naps): the per-table step of a leg over a grid of paces
LEG=${LEG:-c3}
for w in ${WGS:-32 64 96 128 160 192}; do
  for z in ${NAPS:-0 4 8 12 16 32}; do
    echo -n "ASIDE=$w,$z  "; MASHGPU_FILL_ASIDE=$w,$z python tools/prof_leg.py --leg $LEG --steps ${STEPS:-8} --cold 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
  done
done
echo -n "ASIDE=0  "; MASHGPU_FILL_ASIDE=0 python tools/prof_leg.py --leg $LEG --steps ${STEPS:-8} --cold 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
