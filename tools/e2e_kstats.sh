#!/bin/bash
# kernel stats of the end-to-end training run's steady state: simulate the dataset, then scripts/train.py under rocprofv3
# usage (GPU box, repo root): tools/e2e_kstats.sh <out csv> [extra scripts/train.py arguments, e.g. --batch-size-quantum 1024]
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/${1:-gpurun_out/e2e_kstats.csv}
shift
mkdir -p $(dirname $out)
cd /tmp; export TMPDIR=/tmp
python - <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import e2e_synthetic as e, yaml
d = "/tmp/e2e_ks"; os.makedirs(d, exist_ok=True)
if not os.path.exists(os.path.join(d, "dataset", "raw_events.npz")):
    e.simulate(os.path.join(d, "dataset"))
cfg = yaml.safe_load(open(os.path.join("$R", "configs", "synthetic_smoke.yaml")))
cfg["data"].update(dataset_directory=os.path.join(d, "dataset"), train_init_eff_batch_size=65536, train_eff_ray_sample_batch_size=1 << 20)
cfg["trainer"].update(max_epochs=3, limit_train_batches=400, log_every_n_steps=100)
yaml.safe_dump(cfg, open(os.path.join(d, "train.yaml"), "w"))
PY
rm -rf /tmp/ks_e2e
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_e2e -o x -- python $R/scripts/train.py --config /tmp/e2e_ks/train.yaml --out /tmp/e2e_ks/run --no-validation "$@" > /tmp/ks_e2e.log 2>&1
grep "M rays/s\|rror" /tmp/ks_e2e.log | tail -8 | cut -c1-250
f=$(find /tmp/ks_e2e -name '*kernel_stats.csv' | head -1)
python $R/tools/summarize_profile.py $f $out "e2e: scripts/train.py $*, 3 x 400 steps, 2^20-sample budget"
python $R/tools/gap_trace.py $(find /tmp/ks_e2e -name '*kernel_trace.csv' | head -1) 10 seq > ${out%.csv}_gaps.txt 2>&1
head -${HEADN:-30} $out
