cp robust_e_nerf_amd/build.py /tmp/build.py.orig
python - <<'PY'
p='robust_e_nerf_amd/build.py'; s=open(p).read()
for name in ("pose","jvp","train","composite"):
    s=s.replace('"ren_%s.hip": NO_SLP' % name, '"ren_%s.hip": []' % name)
s=s.replace('"ren_sampling.hip": ["-ffp-contract=off"] + NO_SLP','"ren_sampling.hip": ["-ffp-contract=off"]')
open(p,'w').write(s)
PY
python -m robust_e_nerf_amd.build > /dev/null 2>&1
echo "== round-3 flags (SLP on everywhere but jvp2)"
timeout 900 python -m pytest "tests/test_gpu_vanilla.py::test_vanilla_activation_alternatives_whole_step_vs_oracle" -q -s --tb=short 2>&1 | grep -E "^arch mlp|control|passed|failed|^E  " | cut -c1-250
python - <<'PY'
# every gradient's error, old flags
PY
cp /tmp/build.py.orig robust_e_nerf_amd/build.py
