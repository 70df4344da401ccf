# the training leg of tools/e2e_synthetic.py alone (simulate the dataset, then scripts/train.py with the given extra arguments):
# throughput lines + exit code.   gpurun -- 'bash tools/e2e_train_only.sh --batch-size-quantum 1024'
R=${GRAFT_REPO_ROOT:-$PWD}
python - <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import e2e_synthetic as e, yaml
d = "/tmp/e2e_t"; os.makedirs(d, exist_ok=True)
e.simulate(os.path.join(d, "dataset"))
cfg = yaml.safe_load(open(os.path.join("$R", "configs", "synthetic_smoke.yaml")))
cfg["data"].update(dataset_directory=os.path.join(d, "dataset"), train_init_eff_batch_size=65536, train_eff_ray_sample_batch_size=1 << 20)
cfg["trainer"].update(max_epochs=6, limit_train_batches=500, log_every_n_steps=100)
cfg["lr_scheduler"]["multi_step_lr"]["milestones"] = [3, 4, 5]
yaml.safe_dump(cfg, open(os.path.join(d, "train.yaml"), "w"))
PY
python -X faulthandler $R/scripts/train.py --config /tmp/e2e_t/train.yaml --out /tmp/e2e_t/run "$@" > /tmp/e2e_t/log.txt 2>&1
echo "exit code $?"
grep -v amdgpu.ids /tmp/e2e_t/log.txt | tail -${TAIL:-14} | cut -c1-260
