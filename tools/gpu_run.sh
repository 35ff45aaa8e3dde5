#!/bin/bash
# Runs one recipe of tools/recipes/ on the GPU box:  gpurun --timeout N -- './tools/gpu_run.sh <recipe> [args]'
# A recipe is a committed, named script (one per measurement, never rewritten): what produced a file under profiles/ can be read there.
set -u
R=${1:?usage: tools/gpu_run.sh <recipe> [args]}; shift
export TMPDIR=/tmp
exec bash "$(dirname "$0")/recipes/$R.sh" "$@"
