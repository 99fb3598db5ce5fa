import sys, time, numpy as np
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(__file__), '..')))
import extensisq_amd as esq
from extensisq_amd import workloads as wl
from extensisq_amd.device import DeviceContext
from scipy.integrate import solve_ivp
import traceback
N=2236
rhs=esq.Brusselator2D(N); y0=wl.bruss2d_y0(N); h=1.0/rhs.spectral_radius()
real=DeviceContext.download
calls=[]
def spy(self,*a,**k):
    calls.append(''.join(traceback.format_stack(limit=6)[:-1]))
    return real(self,*a,**k)
DeviceContext.download=spy
kw=dict(first_step=h,max_step=h,rtol=1e-6,atol=1e-9,nfev_stiff_detect=0)
for rep in range(2):
    calls.clear()
    t0=time.perf_counter()
    res=solve_ivp(rhs,(0.0,24*h),y0,method=esq.Pr8,t_eval=[24*h],**kw)
    dt=time.perf_counter()-t0
    print("t_eval run", rep, "total ms", 1e3*dt, "downloads", len(calls))
    if calls: print(calls[0][-1500:])
