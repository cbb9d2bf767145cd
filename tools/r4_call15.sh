#!/bin/bash
MASHGPU_DENSE_STREAM=1 bash tools/r4_prof.sh c3 | grep -E "rc=|dn_pairs|sp_fill_value"
bash tools/r4_prof.sh c3 | grep -E "rc=|dn_pairs|sp_fill_value"
MASHGPU_DENSE_STREAM=1 bash tools/r4_prof.sh one_clade | grep -E "rc=|dn_pairs"
