# round 5, last call: what the driver does at round end, on the final tree -- build() is a no-op on the shipped library, smoke(), the default bench (its line
# and wall clock), and the 2-rank launch through torch.distributed.run with a collective that fails after init (the new guard in bench.py: rank 0 falls back)
O=gpurun_out/r05_s; mkdir -p $O
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
SECONDS=0; python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench.py wall clock: $SECONDS s" | tee $O/bench_wall.txt
python -c "
import json; l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(len(json.dumps(l)), l['value'], l['roofline']['frac'], l['roofline']['traffic'], l['parity'], [w['frac'] for w in l['workloads']])"
VPT_BENCH_ONE_DEVICE=1 VPT_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --config 2 --sentences 200000 --model-scale 0.05 --steps 3 --warmup 1 --no-emit --no-e2e > $O/two_ranks.json 2> $O/two_ranks.err; python -c "
import json; l=json.loads([x for x in open('$O/two_ranks.json').read().splitlines() if x.startswith('{')][-1]); print('2 ranks:', l['n_gpus'], l['parity'], l['config']['tables_broadcast'], l['config']['launch'][:60])"
