"""Emulation of the residual product r0 = k - A z0 on exactly sliced 8-bit operands (balanced base-256 digits, int8 x int8 -> int32 exact).
How many slice pairs does the level-1 variance need?  (scripts/ozaki_check.py used 6-bit truncated digits: ~40 pairs.)"""
import sys, numpy as np, scipy.linalg as sl
sys.path.insert(0,'/root/repo')
from oracle import nngp_oracle as o
from nngp_src_amd import synth
n=int(sys.argv[1]) if len(sys.argv)>1 else 3072
d,m=128,64
x,y=synth.synthetic_queries(n,d,seed=0); xt,_=synth.synthetic_queries(m,d,seed=1)
a=o.make_arch(3)
K=o.kernel_fn(x,None,"nngp",a); reg=1e-3*np.trace(K)/n; A=K+reg*np.eye(n)
ktd=o.kernel_fn(xt,x,"nngp",a); ktt=np.array([o.kernel_fn(xt[i:i+1],None,"nngp",a)[0,0] for i in range(m)])
L32=np.linalg.cholesky(A.astype(np.float32).astype(np.float64)).astype(np.float32)
def minv(b):
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
print("n",n,"level-1 exact residual: rel err", np.max(np.abs(v1-var_ref)/np.abs(var_ref)), "cond", np.linalg.cond(A))
print("row span of z0: max/median", np.median(np.max(np.abs(z0),axis=1)/np.median(np.abs(z0),axis=1)))
def slices(mat, nsl):
    mx=np.max(np.abs(mat),axis=1,keepdims=True); e=np.ceil(np.log2(mx))+1; sc=2.0**e      # |mat|/sc <= 0.5
    X=np.rint(mat/sc*2.0**(8*nsl)).astype(np.int64)   # fixed point, nsl*8 bits (nsl<=7)
    out=[]
    for s in range(nsl):
        dgt=((X+128)&255)-128; X=(X-dgt)>>8; out.append(dgt.astype(np.float64))
    assert np.all(X==0)
    return out[::-1], sc     # most significant first; digit s has weight 256^-(s+1)
for sa,sz,c in ((3,3,2),(4,4,2),(4,4,3),(5,4,3),(4,5,3),(5,5,3),(5,5,4),(6,6,5)):
    As,asc=slices(A,sa); Zs,zsc=slices(z0,sz)
    acc=np.zeros((m,n)); pairs=0
    for i in range(sa):
        for j in range(sz):
            if i+j>c: continue
            pairs+=1
            acc+=(Zs[j]@As[i].T)*256.0**(-(i+j+2))
    r0=ktd-acc*zsc*asc.T
    v=var_from(r0)
    print(sa,sz,"i+j<=",c,"pairs",pairs,"resid err %.2e"%(np.max(np.abs(r0-r_exact))/np.max(np.abs(r_exact))), "var rel err vs exact-resid %.2e"%np.max(np.abs(v-v1)/np.abs(var_ref)))
