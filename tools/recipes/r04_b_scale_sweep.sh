# round 4, second call: parity suite (with the dry-scale test), same-process A/B of the round-3 final library against the generalised
# kernel (M1, M2, ragged), the text-realism sweep with node-read counters, the dry run of the scaling job on the real configs[2]
O=gpurun_out/r04_b; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/gpu_tests.log
python tools/ab_bench.py --variants new,r03f --rounds 3 2>&1 | grep -v amdgpu.ids > $O/ab_m1.jsonl
python tools/ab_bench.py --variants new,r03f --model-kind 2 --rounds 2 2>&1 | grep -v amdgpu.ids > $O/ab_m2.jsonl
python tools/ab_bench.py --variants new,r03f --min-len 8 --max-len 512 --rounds 2 2>&1 | grep -v amdgpu.ids > $O/ab_ragged.jsonl
python tools/hit_share_sweep.py 2>&1 | grep -v amdgpu.ids > $O/hit_share_sweep.jsonl
python tools/hit_share_sweep.py --model-kind 2 --shares 0.3,0.7 2>&1 | grep -v amdgpu.ids > $O/hit_share_sweep_m2.jsonl
python bench.py --dry-scale --steps 10 --warmup 2 --no-e2e --no-emit > $O/dry_scale.jsonl 2> $O/dry_scale.err
tail -3 $O/dry_scale.err
