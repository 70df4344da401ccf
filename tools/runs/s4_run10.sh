timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "prefetched_step_front" --count 1 2>/dev/null | tail -1
for i in $(seq 1 30); do timeout 100 python -m pytest tests/test_gpu_parity.py -q -x --tb=short -k "prefetched_step_front or early_sampling" 2>&1 | grep -E "^E  |passed|failed" | head -6; done
