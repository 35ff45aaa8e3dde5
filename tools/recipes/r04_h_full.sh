# round 4, eighth call: the whole parity suite and every workload of the bench on the sources with the flat decode kernel
O=gpurun_out/r04_h; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
python bench.py --steps 20 --warmup 3 > $O/bench_all.json 2> $O/bench_all.err; tail -2 $O/bench_all.err | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
