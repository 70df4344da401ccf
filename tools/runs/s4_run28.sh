python -m robust_e_nerf_amd.build --check
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -20
timeout 300 python -m pytest tests/test_gpu_vanilla.py -q -s -k "activation_alternatives_whole_step" 2>&1 | grep -E "arch mlp|gradient scale|passed|failed"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
