python -m robust_e_nerf_amd.build --check
for p in 1 2 3 4; do timeout 600 python tools/early_diag.py 12 inorder,early 2>&1 | grep -v amdgpu.ids | awk '{print $1, $(NF-6), $(NF-5), $(NF-1), $NF, "x", $2}'; done | awk '{k=$1" "$2" "$3" "$4" "$5; c[k]+=$7} END {for (k in c) print c[k], k}' | sort -k2
for v in unordered; do for p in 1 2 3 4; do DIAG_VAR=$v timeout 300 python tools/prefetch_diag.py occgrid 6 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$(NF-3),$(NF-2)}'; done; done | sort | uniq -c
