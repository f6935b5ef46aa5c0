# development: time all reference configurations over hierarchy-build knobs (tools/ab_scenes.py specs)
specs="cur"
for b in 8 10 12 13 15; do for o in 5 7 1; do specs="$specs cur:RT_AMD_BVH_BINS=$b:RT_AMD_BVH_ORDER=$o"; done; done
python3 tools/ab_scenes.py $specs
