# round 6: the lane-pair loads for the 64-byte bigram nodes of row windows 4 .. 8 (two requests a node instead of three or four): charw4 (model kind 5), and M1 again
O=gpurun_out/r06_n; mkdir -p $O
python tools/ab_bench.py --variants new,nopair --rounds 3 --model-kind 5 > $O/ab_charw4.jsonl 2> $O/ab_charw4.err; cut -c1-260 $O/ab_charw4.jsonl
python tools/ab_bench.py --variants new,nopair --rounds 2 > $O/ab_m1.jsonl 2> $O/ab_m1.err; cut -c1-260 $O/ab_m1.jsonl
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "window or wide or random_models" 2>&1 | tail -2 ) | tee $O/gpu_tests_windows.log
