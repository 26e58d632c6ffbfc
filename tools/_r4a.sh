set -o pipefail
mkdir -p gpurun_out/r4a
timeout -k 10 900 python -m pytest tests/test_gpu_saturation.py tests/test_gpu_configs.py::test_c5_2048_cubed_z_bands_against_the_oracle -x -q -m gpu > gpurun_out/r4a/pytest_new.log 2>&1; echo "pytest_new rc=$?" | tee -a gpurun_out/r4a/summary.txt
for cfg in c2 c4 c5; do for sat in 1 0; do
  st=100; [ $cfg = c5 ] && st=30
  KF_INTEGRATE_SAT=$sat timeout -k 10 300 python bench.py --config $cfg --steps $st --warmup 5 --no-cpu-baseline --no-extras --no-scaling-reference > gpurun_out/r4a/bench_${cfg}_sat$sat.json 2> gpurun_out/r4a/bench_${cfg}_sat$sat.err; echo "bench $cfg sat=$sat rc=$?" | tee -a gpurun_out/r4a/summary.txt
done; done
