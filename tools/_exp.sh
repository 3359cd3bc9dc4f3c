timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python bench.py --workdir /tmp/fq_bench --no-front-end > gpurun_out/bench_r5f.json 2> gpurun_out/bench_r5f.err; tail -c 300 gpurun_out/bench_r5f.err
