import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from boltzmann_machines_amd.engine import RbmEngine64, RbmEngine
from boltzmann_machines_amd._ffi import DeviceArray
V,H,B=784,1024,512
rng=np.random.RandomState(0)
for cls,dt,kw in ((RbmEngine64,np.float64,{}),(RbmEngine,np.float32,dict(h_unit=2,n_samples=100))):
    eng=cls(V,H,max_batch=B,l2=1e-5,sample_v_states=True,**kw)
    eng.set('W',(rng.randn(V,H)*0.01).astype(dt))
    X=(rng.rand(B,V)<0.13).astype(dt)
    Xd=DeviceArray.from_numpy(X,dt)
    eng.seed(1)
    for _ in range(20): eng.train_step(Xd,B,0.05,0.9,1)
    eng.sync(); t0=time.perf_counter(); n=100
    for _ in range(n): eng.train_step(Xd,B,0.05,0.9,1)
    eng.sync(); dtm=(time.perf_counter()-t0)/n
    print(cls.__name__, kw, '%.1f us/update'%(dtm*1e6))
