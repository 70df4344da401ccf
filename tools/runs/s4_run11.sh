echo "== OLD tree (9b02ecf), prefetch test alone x25"
cd _oldtree
f=0; for i in $(seq 1 25); do timeout 100 python -m pytest tests/test_gpu_parity.py -q -x -k "prefetched_step_front" 2>&1 | grep -q "1 passed" || f=$((f+1)); done; echo "old alone failures: $f / 25"
f=0; for i in $(seq 1 25); do timeout 100 python -m pytest tests/test_gpu_parity.py -q -x -k "prefetched_step_front or grad_loss_step_vs_reference" 2>&1 | grep -q "3 passed" || f=$((f+1)); done; echo "old after l_grad tests failures: $f / 25"
cd ..
echo "== NEW tree"
f=0; for i in $(seq 1 25); do timeout 100 python -m pytest tests/test_gpu_parity.py -q -x -k "prefetched_step_front" 2>&1 | grep -q "1 passed" || f=$((f+1)); done; echo "new alone failures: $f / 25"
f=0; for i in $(seq 1 25); do timeout 100 python -m pytest tests/test_gpu_parity.py -q -x -k "prefetched_step_front or grad_loss_step_vs_reference" 2>&1 | grep -q "3 passed" || f=$((f+1)); done; echo "new after l_grad tests failures: $f / 25"
f=0; for i in $(seq 1 25); do timeout 100 python -m pytest tests/test_gpu_parity.py -q -x -k "prefetched_step_front or early_sampling" 2>&1 | grep -q "2 passed" || f=$((f+1)); done; echo "new after early test failures: $f / 25"
