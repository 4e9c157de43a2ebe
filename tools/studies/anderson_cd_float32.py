#!/usr/bin/env python3
"""CPU study (NumPy, no GPU): would Anderson extrapolation speed up the lasso coder's coordinate descent?
Float32 sweeps as on the GPU (exact per-sweep deltas, Gram of the deltas and the 5 x 5 solve in float64,
extrapolated point accepted only if the objective falls), against plain sweeps, on a problem of the NTF
workload's kind (d=1200, r=100, alpha=2, non-negative random dictionary).  Prints sweep counts per column
and the distance to a float64 reference solution."""
import numpy as np, time
rs=np.random.RandomState(0)
d,r,b=1200,100,64
X=rs.rand(d,b); W=rs.rand(d,r); W/=np.linalg.norm(W,axis=0)
alpha=2.0
G64=W.T@W; C64=W.T@X
G=G64.astype(np.float32); C=C64.astype(np.float32)
def obj64(H): 
    H=H.astype(np.float64); R=X-W@H; return 0.5*(R*R).sum(0)+alpha*H.sum(0)
def qobj(H,dt):   # 0.5 h'Gh - c'h + alpha sum h, in dtype dt
    H=H.astype(dt); return (H*(0.5*(G.astype(dt)@H)-C.astype(dt)+dt(alpha))).sum(0)
def sweep32(H):
    D=np.zeros_like(H)
    a=np.float32(alpha)
    for k in range(r):
        g=G[k]@H-C[k]+a
        hn=np.maximum(H[k]-g/G[k,k],np.float32(0))
        D[k]=hn-H[k]; H[k]=hn
    return H,D
H=np.zeros((r,b))
for i in range(30000):
    Hn=H.copy()
    for k in range(r):
        g=G64[k]@Hn-C64[k]+alpha; Hn[k]=np.maximum(Hn[k]-g/G64[k,k],0)
    if np.max(np.linalg.norm(Hn-H,axis=0)/np.maximum(np.linalg.norm(H,axis=0),1e-30))<1e-13: H=Hn;break
    H=Hn
Href=H; fref=obj64(Href)
def run(anderson,K=5,tol=1e-6,maxs=4000,objdt=np.float32):
    H=np.zeros((r,b),np.float32); done=np.zeros(b,bool); n=np.zeros(b,int); U=[]; nacc=0; ntry=0
    for i in range(maxs):
        Hold=H.copy(); H,D=sweep32(H)
        H[:,done]=Hold[:,done]; D[:,done]=0
        rel=np.sqrt((D.astype(np.float64)**2).sum(0))/np.maximum(np.sqrt((Hold.astype(np.float64)**2).sum(0)),1e-30)
        n[~done]+=1
        done|=(rel<=tol)
        if done.all(): break
        if anderson:
            U.append(D.copy())
            if len(U)==K:
                Ud=np.stack(U,axis=2).astype(np.float64)       # r,b,K  (exact deltas)
                M=np.einsum('rbi,rbj->bij',Ud,Ud)
                tr=np.einsum('bii->b',M)/K
                M=M+(1e-8*tr)[:,None,None]*np.eye(K)+1e-300*np.eye(K)
                z=np.linalg.solve(M,np.ones((b,K,1)))[:,:,0]
                c=z/z.sum(1,keepdims=True)
                s=np.cumsum(c,axis=1)                           # s_m = sum_{j<=m-1} c_j for m=1..K-1 -> s[:,m-1]
                He=H.astype(np.float64)
                for m in range(1,K): He-= s[:,m-1][None,:]*Ud[:,:,m]
                He=np.maximum(He,0).astype(np.float32)
                fe=qobj(He,objdt); f=qobj(H,objdt)
                acc=(fe<f)&~done
                ntry+=(~done).sum(); nacc+=acc.sum()
                H[:,acc]=He[:,acc]
                U=[]
    gap=(obj64(H)-fref)/fref; err=np.linalg.norm(H-Href,axis=0)/np.linalg.norm(Href,axis=0)
    return n,gap,err,nacc,ntry
for tol in (1e-6,1e-5):
  for a,K,dt in ((False,5,np.float32),(True,5,np.float32),(True,5,np.float64),(True,8,np.float32)):
    t=time.time(); n,gap,err,nacc,ntry=run(a,K=K,tol=tol,objdt=dt)
    print('tol',tol,'anderson',a,'K',K,dt.__name__,'sweeps mean %.0f median %.0f max %d capped %d; gap max %.1e; err max %.1e median %.1e; accepted %d/%d (%.0fs)'%(n.mean(),np.median(n),n.max(),(n>=4000).sum(),gap.max(),err.max(),np.median(err),nacc,ntry,time.time()-t))
