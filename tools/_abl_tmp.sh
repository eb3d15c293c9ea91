cd $GRAFT_REPO_ROOT
export MGX_LIBRARY=$PWD/mygram-db_amd/libmygram_gpu_ablation.so MGX_BENCH_CPU_SECONDS=0 MGX_BENCH_DENSE=2 MGX_BENCH_NO_POINTS=1
for sk in 0 1 3 7; do
  MGX_DEBUG_SKIP=$sk python bench.py --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip', $sk, 'kernel_ms', round(d['roofline']['kernel_ms'],3))"
done
