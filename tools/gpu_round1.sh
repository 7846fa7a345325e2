# one GPU-box visit: parity tests, smoke, per-kernel diagnostics, bench, rocprof summary
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r1_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/r1_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r1_smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/r1_smoke.log
timeout 600 python tools/gpu_diag.py > gpurun_out/r1_diag.log 2>&1
echo "diag exit $?" >> gpurun_out/r1_diag.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r1_bench.log 2>gpurun_out/r1_bench.err
echo "bench exit $?" >> gpurun_out/r1_bench.err
tail -4 gpurun_out/r1_tests.log; tail -2 gpurun_out/r1_smoke.log; grep -E "conv3x3 64->64 relu|dcnv2|full forward|instrumented|ca_scale|conv3x3_64to64 " gpurun_out/r1_diag.log; tail -c 600 gpurun_out/r1_bench.log; tail -3 gpurun_out/r1_bench.err
