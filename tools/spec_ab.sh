set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 500 python -m pytest tests/test_specialize.py -x -q 2>&1 | tail -15 > gpurun_out/r3/spec_tests.txt; cat gpurun_out/r3/spec_tests.txt
for s in 0 1; do for c in c2 c2r c3; do
 PINE_GPU_SPECIALIZE=$s timeout -k 10 300 python bench.py --config $c --steps 6 --warmup 1 --no-cpu --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('spec=$s $c', 'ms_per_step', round(j['ms_per_step'],3), 'Ms/s', round(j['value'],1), 'eq_ref', j.get('film_equals_reference'), 'kernel_ms', round(j['kernels_ms']['path_trace'],3), 'frac', round(j['roofline']['frac'],4))
" || echo "$s $c FAILED"
done; done 2>&1 | tee gpurun_out/r3/spec_ab.txt
