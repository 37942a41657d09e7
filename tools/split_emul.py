import numpy as np
rs=np.random.RandomState(0)
f16=lambda x: x.astype(np.float16).astype(np.float32)
def split(x):
    x=x.astype(np.float32); hi=f16(x); lo=f16(x-hi); return hi,lo
def mm_split(A,B,terms=3):
    Ah,Al=split(A); Bh,Bl=split(B)
    # products exact in f32; accumulate in f32 (emulate with float32 matmul)
    r=(Ah@Bh).astype(np.float32)
    r=r+(Ah@Bl).astype(np.float32)+(Al@Bh).astype(np.float32)
    if terms==4: r=r+(Al@Bl)
    return r.astype(np.float32)
def mm_f32(A,B): return (A.astype(np.float32)@B.astype(np.float32)).astype(np.float32)
def pow2scale(x,target=9):
    m=np.abs(x).max(); e=np.floor(np.log2(m)) if m>0 else 0; return np.float32(2.0**(target-e))
errs={'f32':[], 'split':[]}
for trial in range(40):
    U,k=28,40
    amp=10**rs.uniform(-2,3)
    Yw=(rs.normal(size=(U,k))*0.7*amp).astype(np.float32)
    # 16 columns (points): D weights, each point sees ~20 of 28 slots
    D=np.zeros((U,16),np.float32)
    for c in range(16):
        sel=rs.choice(U,20,replace=False); D[sel,c]=np.sqrt(rs.uniform(0,1,20))
    X=(rs.normal(size=(k,16))*3).astype(np.float32); X-=X.mean(0)
    G64=Yw.astype(np.float64)@Yw.astype(np.float64).T
    Z64=Yw.astype(np.float64)@X.astype(np.float64)
    # Gershgorin bound per column, alpha
    reg=(k-1)/1.1
    d2=(D.astype(np.float64))**2
    def run(prec):
        if prec=='f64':
            G=G64; Z=Z64
        elif prec=='f32':
            G=mm_f32(Yw,Yw.T); Z=mm_f32(Yw,X)
        else:
            sy=pow2scale(Yw); Ys=Yw*sy
            G=mm_split(Ys,Ys.T)/(sy*sy)
            sx=np.array([pow2scale(X[:,c]) for c in range(16)],np.float32)
            Z=mm_split(Ys,X*sx)/(sy*sx)
        L=np.array([ (D[:,c].astype(np.float64)*(np.abs(G64)@D[:,c].astype(np.float64))).max() for c in range(16)])*1.0001
        alpha=(2.0/L)  # A = alpha S - I has spectrum in [-1,1]
        dt=np.float64 if prec=='f64' else np.float32
        G=G.astype(dt); v0=Z.astype(dt); al=alpha.astype(dt); dd=d2.astype(dt)
        if prec=='split':
            sG=np.float32(2.0**-16)*np.float32(1.0)  # on G_acc scale; emulate: scale G so max ~2^10
            sg=pow2scale(G,10); Gs=G*sg
            sv=np.array([pow2scale(v0[:,c],8) for c in range(16)],np.float32)
            prod=lambda v: mm_split(Gs,(dd*v*sv))/(sg*sv)
        elif prec=='f32': prod=lambda v: mm_f32(G,dd*v)
        else: prod=lambda v: G@(dd*v)
        va=v0; vb=(al*prod(va)-va).astype(dt)
        acc=(0.5*va+0.3*vb).astype(dt)
        for j in range(2,15):
            vn=(2*(al*prod(vb)-vb)-va).astype(dt); va,vb=vb,vn
            acc=(acc+dt(0.3*0.6**j)*vn).astype(dt)
        return (D.astype(dt)*acc)   # t-space
    r64=run('f64')
    for p in ('f32','split'):
        r=run(p).astype(np.float64)
        errs[p].append(np.linalg.norm(r-r64)/np.linalg.norm(r64))
for p in errs: print(p,'median %.2e max %.2e'%(np.median(errs[p]),np.max(errs[p])))
