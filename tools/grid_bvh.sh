# development: time the headline frame over hierarchy-build knobs (tools/ab.py specs; tools/ab_scenes.py for all configurations)
specs=""
for am in 0.2 0.5 0.8; do for b in 8 12 16 20 24 32; do specs="$specs cur:RT_AMD_BVH_AXIS_MIN=$am:RT_AMD_BVH_BINS=$b"; done; done
python3 tools/ab.py $specs
