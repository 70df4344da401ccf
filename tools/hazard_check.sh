# The packed-FP32 / side-stream hazard of round 4 (profiles/NOTES.md), through gpurun:
#   gpurun --timeout 900 -- 'bash tools/hazard_check.sh'
# Expected on the tree's own build: every line of every placement identical (same sample counts), chunk stress 0 mismatches.
# To see the hazard again: drop NO_SLP for ren_pose.hip / ren_jvp.hip in robust_e_nerf_amd/build.py, rebuild, re-run: the
# "early" placement and the unordered prefetch then differ in ~1 three-step run out of 4.
python -m robust_e_nerf_amd.build --check
for p in 1 2 3 4; do timeout 600 python tools/early_diag.py 12 2>&1 | grep -v amdgpu.ids | awk '{print $1, $(NF-6), $(NF-5), $(NF-1), $NF, "x", $2}'; done \
  | awk '{k=$1" "$2" "$3" "$4" "$5; c[k]+=$7} END {for (k in c) print c[k], k}' | sort -k2
for v in none unordered; do for p in 1 2 3 4; do DIAG_VAR=$v timeout 300 python tools/prefetch_diag.py occgrid 6 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$(NF-3),$(NF-2)}'; done; done | sort | uniq -c
timeout 300 python tools/chunk_stress.py 20 2>&1 | tail -2
timeout 600 python tools/aggressor_probe.py 200 2>&1 | grep -v amdgpu.ids     # every line 0 on the tree's own build
