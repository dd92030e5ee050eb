import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from ekf_slam_amd import Engine
N=10000; batch=32; n=3+2*N
rng=np.random.default_rng(1)
x=np.concatenate([[0,0,0],rng.uniform(-100,100,2*N)]); d=rng.uniform(0.01,0.1,n); U=rng.normal(0,0.01,(n,8)); s=np.arange(1,N+1.0)
e=Engine(capacity=N,tile=128,batch=batch); e.load_lowrank_state(x,s,d,U)
R=np.diag([0.2,50.0]); rows=[]
for i in range(3*batch):
    e.predict([0.1,3.0]); e.correct([10.0,100.0],R,(i*37)%N)
    q=e.get_Q3().reshape(-1)[:9]
    if i>=batch: rows.append(q)
sel=np.array(rows); print("stamps:", np.round(sel.mean(axis=0))); print("diffs:", np.round(np.diff(sel,axis=1).mean(axis=0)))
