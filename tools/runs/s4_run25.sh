run() { timeout 900 python -m pytest "tests/test_gpu_vanilla.py::test_vanilla_activation_alternatives_whole_step_vs_oracle" "tests/test_gpu_parity.py::test_activation_alternatives_whole_step_vs_oracle" -q -s 2>&1 | grep -E "whole step|control|passed|failed"; }
echo "== all six files no-slp"; run
cp robust_e_nerf_amd/build.py /tmp/build.py.orig
for keep in "pose jvp jvp2" "pose jvp jvp2 train" "pose jvp jvp2 composite" "pose jvp jvp2 sampling"; do
  cp /tmp/build.py.orig robust_e_nerf_amd/build.py
  python - "$keep" <<'PY'
import sys,re
keep=sys.argv[1].split()
p='robust_e_nerf_amd/build.py'; s=open(p).read()
for name in ("sampling","jvp2","pose","jvp","train","composite"):
    if name not in keep:
        s=s.replace('"ren_%s.hip": ["-ffp-contract=off"] + NO_SLP' % name, '"ren_%s.hip": ["-ffp-contract=off"]' % name)
        s=s.replace('"ren_%s.hip": NO_SLP' % name, '"ren_%s.hip": []' % name)
open(p,'w').write(s)
PY
  python -m robust_e_nerf_amd.build > /dev/null 2>&1
  echo "== no-slp files: $keep"; run
done
cp /tmp/build.py.orig robust_e_nerf_amd/build.py
