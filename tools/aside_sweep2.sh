for v in 48,0 64,0 64,1 64,2 80,0 80,2 96,0 96,2 96,4 256,0,64 256,2,64 384,2,64 384,4,64 512,4,64 512,8,64 128,0,128 128,2,128 192,2,128 0; do
  echo -n "ASIDE=$v  "; MASHGPU_FILL_ASIDE=$v python tools/prof_leg.py --leg c3 --steps 6 --cold | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done
