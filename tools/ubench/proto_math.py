import numpy as np
LOG2E=1.4426950408889634; LN2_HI=float.fromhex('0x1.62e42fee00000p-1'); LN2_LO=float.fromhex('0x1.a39ef35793c76p-33')
assert abs(LN2_HI+LN2_LO-np.log(2))<1e-17
# exp coefficients 1/n!
import math
EC=[1.0/math.factorial(n) for n in range(0,14)]
def exp_neg(y):
    k=np.rint(y*LOG2E)
    r=y-k*LN2_HI
    r=r-k*LN2_LO
    p=np.full_like(r,EC[13])
    for n in range(12,-1,-1): p=p*r+EC[n]
    return np.ldexp(p,k.astype(int))
def rcp(b):
    r=(1.0/b).astype(np.float32).astype(np.float64)   # ~24-bit seed
    e=1.0-b*r; r=r+e*r
    e=1.0-b*r; r=r+e*r
    return r
def div(a,b):
    r=rcp(b); q=a*r; return q+(a-b*q)*r
Lg=[6.666666666666735130e-01,3.999999999940941908e-01,2.857142874366239149e-01,2.222219843214978396e-01,1.818357216161805012e-01,1.531383769920937332e-01,1.479819860511658591e-01]
def log_pos(v):
    m,e=np.frexp(v)
    lt=m<0.7071067811865476
    m=np.where(lt,m*2,m); e=np.where(lt,e-1,e).astype(float)
    f=m-1.0
    s=div(f,2.0+f)
    z=s*s
    R=np.full_like(z,Lg[6])
    for n in range(5,-1,-1): R=R*z+Lg[n]
    R=R*z
    hfsq=0.5*f*f
    return e*LN2_HI-((hfsq-(s*(hfsq+R)+e*LN2_LO))-f)
rng=np.random.default_rng(0)
y=-rng.uniform(0,745,2000000); ref=np.exp(y); got=exp_neg(y)
ok=ref>1e-300
print('exp max rel', np.max(np.abs(got[ok]-ref[ok])/ref[ok]))
v=np.exp(rng.uniform(-30,30,2000000)); print('log max abs/ulp', np.max(np.abs(log_pos(v)-np.log(v))/np.maximum(np.abs(np.log(v)),1e-300)*1), np.max(np.abs(log_pos(v)-np.log(v))))
v=rng.uniform(0.5,2,2000000); print('log near 1 abs', np.max(np.abs(log_pos(v)-np.log(v))), 'rel', np.max(np.abs(log_pos(v)-np.log(v))/np.abs(np.log(v))))
# softplus pieces
def softplus_parts(x,s):
    ax=np.abs(x); e=exp_neg(-ax)
    u=1.0+e
    # general: log1p(e)=log(u)+ (e-(u-1))/u
    l1p=log_pos(u)+div(e-(u-1.0),u)
    fast=e<9.6e-5
    l1pf=e*(1-e*(0.5-e*(1/3-e*(0.25-e*0.2))))
    l1p=np.where(fast,l1pf,l1p)
    lam=np.maximum(x,0)+l1p
    inv=rcp(u)
    sig=np.where(x>=0,inv,e*inv)
    loglam=log_pos(lam)
    return lam,sig,loglam
x=rng.uniform(-40,60,2000000)
lam,sig,ll=softplus_parts(x,None)
lam0=np.maximum(x,0)+np.log1p(np.exp(-np.abs(x))); sig0=np.where(x>=0,1/(1+np.exp(-np.abs(x))),np.exp(-np.abs(x))/(1+np.exp(-np.abs(x))))
print('lam rel',np.max(np.abs(lam-lam0)/lam0),'sig rel',np.max(np.abs(sig-sig0)/sig0),'loglam abs',np.max(np.abs(ll-np.log(lam0))), 'rel', np.max(np.abs(ll-np.log(lam0))/np.maximum(np.abs(np.log(lam0)),1e-3)))
