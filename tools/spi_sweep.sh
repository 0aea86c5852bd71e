mkdir -p gpurun_out/r3
for spi in 1 2 4 8; do
 timeout -k 10 200 python bench.py --config c2 --steps 8 --warmup 2 --headline-only --spi $spi 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('spi $spi c2', 'ms_per_step', round(j['ms_per_step'],3), 'kernels', {k: round(v,3) for k,v in j['kernels_ms'].items()}, j.get('film_equals_reference'))
"
done
for spi in 2 4; do
 timeout -k 10 200 python bench.py --config c3 --steps 3 --warmup 1 --headline-only --spi $spi 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('spi $spi c3', 'ms_per_step', round(j['ms_per_step'],3), 'kernels', {k: round(v,3) for k,v in j['kernels_ms'].items()}, j.get('film_equals_reference'))
"
done
timeout -k 10 300 python -m pytest tests/test_adapter.py -x -q 2>&1 | tail -3
