#!/bin/bash
set -u
O=gpurun_out/c43; mkdir -p $O
run() { n=$1; shift; env "$@" timeout 300 python tools/tokenize_bench.py $EXTRA > $O/t_$n.json 2> $O/t_$n.err; echo "$n $(cut -c1-200 $O/t_$n.json)"; }
EXTRA="--repeat 10 --iters 6"
run big_one VPT_TOKENIZE_CHUNK_BYTES=9999999999
run big_c12m VPT_TOKENIZE_CHUNK_BYTES=12582912
run big_c24m VPT_TOKENIZE_CHUNK_BYTES=25165824
run big_c48m VPT_TOKENIZE_CHUNK_BYTES=50331648
EXTRA="--config 4 --repeat 4 --iters 6"
run tagbig_one VPT_TOKENIZE_CHUNK_BYTES=9999999999
run tagbig_c24m VPT_TOKENIZE_CHUNK_BYTES=25165824
