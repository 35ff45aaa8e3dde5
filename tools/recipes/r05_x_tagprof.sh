# round 5: where a wave of fill_tags' flat front end spends its time (a -DVPT_TAG_PROFILE build: shader-clock ticks per part of a step, summed over the waves)
O=gpurun_out/r05_x; mkdir -p $O
cp tools/prebuilt/libvaporetto_${1:-t6up}.so vaporetto_amd/lib/libvaporetto_hip.so
python bench.py --config 4 --quick --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-emit > $O/bench.json 2> $O/bench.err
grep "tag front profile" $O/bench.err | tail -4 | tee $O/tag_front_profile_${1:-t6up}.txt
python -c "
import json; l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('configs[4]', l['ms_per_step'], l['tags']['ms_per_step'])"
