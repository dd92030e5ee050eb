import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from ekf_slam_amd import Engine
from oracle.ekf_structured import StructuredEKF
def rel(a,b): return float(np.abs(a-b).max()/np.abs(b).max())
N=300
rng=np.random.default_rng(61); n=3+2*N
x=np.concatenate([[0.3,-0.2,40.0], rng.uniform(-20,20,2*N)])
U=rng.normal(0,0.05,(n,6)); P=np.diag(rng.uniform(0.01,0.1,n))+U@U.T; s=np.arange(1,N+1.0)
for storage in ("f32","f32_mixed"):
  for batch in (1,8,13):
    e=Engine(capacity=N+8,tile=256,storage=storage,batch=batch); ref=StructuredEKF(N+8,"known")
    e.set_state(x,P,s); ref.set_state(x,P,s)
    r=np.random.default_rng(14)
    for step in range(40):
        u=[0.1,3.0]; e.predict(u); ref.predict(u)
        i=int(r.integers(0,e.N)); z=[r.uniform(1,30), r.uniform(1,359)]; R=np.diag([z[0]*.01, z[1]*5.0])
        e.correct(z,R,i); ref.correct(z,R,i+1)
    e.flush()
    print(storage,batch,e.downdate_kernel_name(),"x %.2e P %.2e"%(rel(e.get_x(),ref.x),rel(e.get_P(),ref.P)))
