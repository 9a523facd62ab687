for cl in 512 1024 2048 4096; do SCILMM_CELL_LIMIT=$cl SCILMM_VERBOSE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/tmp/err.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('cell_limit=$cl',{k:round(c[k],1) for k in ['factorize_ms','update_ms','symbolic_s']}, c['solve_residual'])"; grep "plan" /tmp/err.txt; done
