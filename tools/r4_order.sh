#!/bin/bash
# Round 4: heaviest-tile-first item order under the rebuilt hand-over, A/B (development build).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; TAG=${TAG:-o}; mkdir -p $O; cd $R
export EG_DEV_SWITCHES=1
b() { env $2 timeout 400 python bench.py $3 --no-cpu-baseline --no-extra --no-traffic 2>$O/bench_$1_$TAG.err | tail -1 > $O/bench_$1_$TAG.json; }
b c2s "X=1" "--config config2"
b c2s_heavy_tilemajor "EG_TILE_ORDER=1 EG_FRONT_SLICES=0" "--config config2"
b c2s_heavy_classes "EG_TILE_ORDER=1" "--config config2"
b c2s_itemorder "EG_FRONT_SLICES=0" "--config config2"
b c2i "X=1" "--config config2 --init-opacity"
b c2i_heavy_tilemajor "EG_TILE_ORDER=1 EG_FRONT_SLICES=0" "--config config2 --init-opacity"
b c1 "X=1" "--config config1"
b c1_heavy_tilemajor "EG_TILE_ORDER=1 EG_FRONT_SLICES=0" "--config config1"
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*_$TAG.json")):
    try:
        d=json.loads(open(f).read())
        print(f.split('/')[-1], round(d['ms_per_step']*1e3,1),'us', {k:round(v,1) for k,v in d.get('stages_us',{}).items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
