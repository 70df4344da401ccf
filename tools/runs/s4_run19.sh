for i in $(seq 1 40); do timeout 300 python -m pytest tests/test_gpu_parity.py -q --tb=short -k "prefetched_step_front or early_sampling or begun_sampling" 2>&1 | grep -E "^E  |^FAILED|failed|^tests.*py:[0-9]+: " | head -8; done
echo done
