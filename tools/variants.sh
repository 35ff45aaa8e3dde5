#!/bin/bash
# A/B on the GPU box: each argument is a '@@'-separated list of FILE::SEDEXPR edits (FILE relative to csrc/);
# the library is rebuilt with them, benched twice, and the sources restored.
bench() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step %.4f kernel_ms %.4f tiles %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['tiles']))"; }
C=vaporetto_amd/csrc
echo "== as committed"; bench; bench
for V in "$@"; do
  rm -rf /tmp/csrc_backup; cp -r $C /tmp/csrc_backup
  echo "== variant: $V"
  echo "$V" | sed 's/@@/\n/g' | while IFS= read -r E; do F="${E%%::*}"; X="${E#*::}"; sed -i "$X" "$C/$F"; done
  if python -m vaporetto_amd.build --force > /tmp/build.log 2>&1; then bench; bench; else echo "  build failed:"; grep -m3 error /tmp/build.log; fi
  cp /tmp/csrc_backup/*.h* /tmp/csrc_backup/*.cpp $C/ 2>/dev/null
done
python -m vaporetto_amd.build --force > /dev/null 2>&1
