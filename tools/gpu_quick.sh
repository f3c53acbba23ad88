cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "slab or strip or ildu or ldu or golden" 2>&1 | tail -2
SLAB_STRESS_MIXED=1 timeout 600 python tools/slab_stress.py 200,12,9,100 2>&1 | grep -v amdgpu | cut -c1-150
for g in 1000 -100 -128; do timeout 300 python tools/ildu_bench.py $g ildu0 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150; done
