#!/bin/bash
set -u
bash tools/profile.sh r02_h 2>&1 | tail -2
bash tools/profile.sh r02_h_m2 --config 3 2>&1 | tail -2
bash tools/profile.sh r02_h_tags --config 4 --sentences 300000 2>&1 | tail -2
