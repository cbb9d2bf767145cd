#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for w in 0 2 4 8 16; do
  echo "=== MASHGPU_FILL_OVERLAP=$w"
  MASHGPU_FILL_OVERLAP=$w timeout 600 python bench.py --steps 10 --warmup 3 --no-sketch --no-screen --no-h2h --no-cli --no-cpu --no-brackets --detail /tmp/d.json 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step',d['ms_per_step'],'phases',d['roofline']['pass']['phases_ms'],'c5',d.get('c5_pairs_s'),'checksum',d['config']['output_checksum'])"
done
