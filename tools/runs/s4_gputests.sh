python -m robust_e_nerf_amd.build --check
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
