timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "Warning\|warnings.warn\|^$\|WeightNorm" | tail -40
