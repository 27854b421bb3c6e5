import sys, numpy as np, scipy.linalg as sl
sys.path.insert(0,'/root/repo')
from oracle import nngp_oracle as o
from nngp_src_amd import synth
n,d,m=3072,128,64
x,y=synth.synthetic_queries(n,d,seed=0); xt,_=synth.synthetic_queries(m,d,seed=1)
a=o.make_arch(3)
K=o.kernel_fn(x,None,"nngp",a); reg=1e-3*np.trace(K)/n; A=K+reg*np.eye(n)
ktd=o.kernel_fn(xt,x,"nngp",a); ktt=np.array([o.kernel_fn(xt[i:i+1],None,"nngp",a)[0,0] for i in range(m)])
L32=np.linalg.cholesky(A.astype(np.float32).astype(np.float64)).astype(np.float32)  # float32-rounded factor
def minv(b):  # float32 solves
    v=sl.solve_triangular(L32,b.T.astype(np.float32),lower=True,check_finite=False).astype(np.float32)
    return sl.solve_triangular(L32.T,v,lower=False,check_finite=False).astype(np.float32).T
z0=minv(ktd).astype(np.float64)
def var_from(r0):
    q=np.sum(z0*(ktd+r0),axis=1)
    v=sl.solve_triangular(L32.astype(np.float64),r0.T,lower=True,check_finite=False)
    return ktt-(q+np.sum(v*v,axis=0))
r_exact=ktd-z0@A
var_ref=ktt-np.sum(np.linalg.solve(A,ktd.T).T*ktd,axis=1)
v1=var_from(r_exact)
print("level-1 with exact residual: rel err", np.max(np.abs(v1-var_ref)/np.abs(var_ref)), "cond", np.linalg.cond(A))
def slices(mat, nsl, axis):
    # per-row (axis=1) power-of-two scale so that |mat|/scale < 1; signed 7-bit digits
    mx=np.max(np.abs(mat),axis=axis,keepdims=True); e=np.ceil(np.log2(mx)); sc=2.0**e
    rem=mat/sc; out=[]
    for s in range(nsl):
        dgt=np.trunc(rem*64.0)   # |digit| <= 64 (7 bits + sign)
        out.append(dgt); rem=rem*64.0-dgt
    return out, sc
for sa,sz,lmax in ((6,4,6),(6,4,7),(7,4,7),(7,5,8),(7,5,7),(8,5,8),(8,5,9)):
    As,asc=slices(A,sa,1)     # rows of A (symmetric: row scale)
    Zs,zsc=slices(z0,sz,1)    # rows of z0
    acc=np.zeros((m,n)); pairs=0
    for i in range(sa):
        for j in range(sz):
            if i+j+2>lmax: continue
            pairs+=1
            acc+= (Zs[j]@As[i].T) * 64.0**(-(i+j+2))
    prod=acc*zsc*asc.T   # (z0 A^T)[q, r] = sum_c z0[q,c] A[r,c]; A row scale belongs to r, z scale to q
    r0=ktd-prod
    v=var_from(r0)
    print(sa,sz,lmax,"pairs",pairs,"resid err", np.max(np.abs(r0-r_exact))/np.max(np.abs(r_exact)), "var rel err vs exact-resid", np.max(np.abs(v-v1)/np.abs(var_ref)))
