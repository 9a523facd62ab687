for a in 0 1 2; do SCILMM_NO_LOOKAHEAD=1 SCILMM_ABLATE=$a timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('ABLATE=$a',{k:round(c[k],1) for k in ['factorize_ms','update_ms','reduce_cells_ms','potrf_ms','trsm_ms']})"; done
