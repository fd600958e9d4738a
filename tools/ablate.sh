timeout -k 10 400 python -u -m pytest tests/test_gmm_gpu.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -3
for v in bf16x3:2x8 bf16x3:2x4 bf16x3x:2x8; do echo "$v"; timeout -k 10 200 python -u tools/perf_gmm.py swbd_tri4 35554 $v 2>&1 | grep "ms/launch" ; done
