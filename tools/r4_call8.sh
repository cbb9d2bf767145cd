#!/bin/bash
bash tools/r4_prof.sh c3 cold | head -60
bash tools/r4_prof.sh c5 cold | head -45
