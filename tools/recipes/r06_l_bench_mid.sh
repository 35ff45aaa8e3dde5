# round 6, mid-round evidence: the whole GPU suite and the default bench on the sources after the tag records, the removal of phase D and the new baseline / gather legs
O=gpurun_out/r06_l; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
SECONDS=0; python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench.py wall clock: $SECONDS s" | tee $O/bench_wall.txt; tail -5 $O/bench.err
python -c "
import json; l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(len(json.dumps(l)), l['value'], l['roofline']['frac'], l['parity']); print(json.dumps(l['roofline'].get('gather'))[:1500]); print(json.dumps(l['cpu_baseline'])[:1200]); print([(w['name'], w.get('frac'), w.get('parity'), w.get('tags_ms'), w.get('tags_dense_ms'), w.get('emit_ms'), w.get('tokenize_ms')) for w in l['workloads']])"
