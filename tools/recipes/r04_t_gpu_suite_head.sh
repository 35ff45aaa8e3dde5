# round 4: the parity suite on the round's last tree (adds the grapheme-cluster filter's tokenize test) and the smoke entry
O=gpurun_out/r04_t; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/smoke.log; cat $O/smoke.log
