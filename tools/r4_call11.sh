#!/bin/bash
bash tools/r4_prof.sh c5 cold | head -40
bash tools/r4_prof.sh c3 cold | head -32
