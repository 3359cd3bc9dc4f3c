timeout 600 python -m pytest tests/test_device_frontend.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python3 tools/frontend_bench.py --records 4000000
timeout 300 python3 tools/frontend_bench.py --records 5200000 --levels 1
