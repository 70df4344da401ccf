timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -E "^FAILED|^E  |^tests.*py:[0-9]+: in|passed|failed" | head -40
