for v in libscilmm_hip.so libscilmm_hip_cw16.so; do SCILMM_HIP_LIB=$PWD/scilmm_amd/csrc/$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('$v',{k:round(c[k],1) for k in ['factorize_ms','solve_ms','solve_fwd_ms','solve_bwd_ms']}, c['solve_residual'])"; done
