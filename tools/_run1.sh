mkdir -p gpurun_out/m5
for w in 2 4 8; do REN_MARCH_SEQUENTIAL=$w python tools/fuzz_march.py 2>&1 | tail -1 > gpurun_out/m5/fuzz_$w.txt; done
for w in 0 1 2 4 8 16; do
  REN_MARCH_SEQUENTIAL=$w python bench.py --sampler occgrid --loss-grad 1e-3 --events 6144 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 > gpurun_out/m5/e6k_$w.json
  REN_MARCH_SEQUENTIAL=$w python bench.py --sampler occgrid --loss-grad 1e-3 --events 16384 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 > gpurun_out/m5/e16k_$w.json
  REN_MARCH_SEQUENTIAL=$w python bench.py --sampler occgrid --events 32768 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -1 > gpurun_out/m5/e32k_$w.json
  REN_MARCH_SEQUENTIAL=$w python bench.py --sampler occgrid --events 65536 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -1 > gpurun_out/m5/e65k_$w.json
done
