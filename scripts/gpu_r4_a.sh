cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4_a_tests.log
python bench.py > gpurun_out/r4_a_bench.json 2> gpurun_out/r4_a_bench.err
tail -c 1500 gpurun_out/r4_a_bench.err
