for f in 512 1024 2048 4096 8192; do echo "FPB $f"; AMLIKE_GMM_FPB=$f timeout -k 10 200 python -u tools/perf_gmm.py swbd_tri4 35554 3x8 2>&1 | grep -A1 "^3x8" ; done
