#!/bin/bash
# A/B on the GPU box: rebuilds the library with a sed-patched kernel source and benches each variant.
bench() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
echo "== as committed"; bench; bench
for V in "$@"; do
  cp vaporetto_amd/csrc/kernels_fast.hip /tmp/kf_backup.hip
  echo "== variant: $V"; sed -i "$V" vaporetto_amd/csrc/kernels_fast.hip
  python -m vaporetto_amd.build --force > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; }
  bench; bench
  cp /tmp/kf_backup.hip vaporetto_amd/csrc/kernels_fast.hip
done
python -m vaporetto_amd.build --force > /dev/null 2>&1
