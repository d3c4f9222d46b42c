cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "stem" 2>&1 | tail -3
for f in 2 3; do echo "FORM=$f"; W2C_STEM_FORM=$f timeout 300 python tools/bench_stem.py 2>/dev/null; done
W2C_STEM_FORM=3 python tools/bench_stem.py 128 8 1 512 2>/dev/null
W2C_STEM_FORM=3 python tools/bench_stem.py 128 2 2 1024 2>/dev/null
