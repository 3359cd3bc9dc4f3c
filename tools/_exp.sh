timeout 900 python -m pytest tests/test_cli_multidevice.py -x -q -m gpu 2>&1 | tail -3
( time python bench.py --workdir /tmp/fq_bench > gpurun_out/bench_r5g.json 2> gpurun_out/bench_r5g.err ) 2>&1 | tail -4; tail -c 300 gpurun_out/bench_r5g.err
