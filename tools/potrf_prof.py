import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import scipy.sparse as sp
import bench
from scilmm_amd.factor import Symbolic
from scilmm_amd import _lib
A,Cc,y = bench.build_problem('100k', 0)
n=A.shape[0]
sym = Symbolic([A, sp.identity(n, format='csr')])
fac = sym.factorize([0.4,0.6])
lib = _lib.lib() if callable(getattr(_lib,'lib',None)) else _lib._lib
buf=(C.c_ulonglong*16)()
lib.scilmm_debug_potrf_prof(buf, 1)
fac.refactorize([0.5,0.5])
fac.sync() if hasattr(fac,'sync') else None
lib.scilmm_debug_potrf_prof(buf, 0)
v=np.array(list(buf),dtype=float); cnt=v[15]
names=['load','diag16 (x8)','panel (x7)','trailing (x7)','inverse','log+store']
print('fronts with w==NB:', cnt)
for i,nm in enumerate(names): print('%-16s %.2f us per front'%(nm, v[i]/cnt/100.0))
print('total %.2f us'%(v[:6].sum()/cnt/100.0))
