import numpy as np
rng=np.random.default_rng(1)
h=lambda a: np.asarray(a,np.float32).astype(np.float16).astype(np.float32)
def trial(n=200000, smooth=False):
    # nodes: G[ia 0..2][b 0..1][c 0..1]
    G=rng.standard_normal((n,3,2,2)).astype(np.float32)
    if smooth: G=1.0+0.1*G
    xa=rng.uniform(-1,1,n).astype(np.float32); xb=rng.uniform(-.5,.5,n).astype(np.float32); xc=rng.uniform(-.5,.5,n).astype(np.float32)
    # exact
    def bil(P,b,c):  # P[...,2,2]
        wb=b+0.5; wc=c+0.5
        return (P[:,0,0]*(1-wb)*(1-wc)+P[:,1,0]*wb*(1-wc)+P[:,0,1]*(1-wb)*wc+P[:,1,1]*wb*wc)
    v=[bil(G[:,i].astype(np.float64),xb.astype(np.float64),xc.astype(np.float64)) for i in range(3)]
    xa64=xa.astype(np.float64)
    exact=np.where(xa64<0, v[0]*(-xa64)+v[1]*(1+xa64), v[1]*(1-xa64)+v[2]*xa64)
    # corner form (today): cell = lower or upper; 8 weights fp16, 8 T fp16
    lower=xa<0
    wa=np.where(lower,1+xa,xa).astype(np.float32)
    C=np.where(lower[:,None,None,None], G[:,0:2], G[:,1:3])  # [n,2(a),2,2]
    wb=xb+0.5; wc=xc+0.5
    acc=np.zeros(n,np.float32)
    for ia in range(2):
        for ib in range(2):
            for ic in range(2):
                w=(wa if ia else 1-wa)*(wb if ib else 1-wb)*(wc if ic else 1-wc)
                acc+=h(w)*h(C[:,ia,ib,ic])
    e_corner=np.abs(acc-exact)
    # slab form: xi_a' = xa/2 ; monomials {1,b,c,bc} x {1, a', |a'|}; coefs computed in fp64 -> fp16
    # along a: val = P0 + (P1-P-1)/2 * xa + (P1+P-1-2P0)/2*|xa|  => in a' = xa/2: slope coefs x2
    G64=G.astype(np.float64)
    A0=G64[:,1]; A1=(G64[:,2]-G64[:,0]); A2=(G64[:,2]+G64[:,0]-2*G64[:,1])   # multiply a' and |a'| (a'=xa/2 -> (P1-P-1)/2*2a' )
    def bicoef(P):  # returns m, sb, sc, sbc for centered coords
        m=(P[:,0,0]+P[:,1,0]+P[:,0,1]+P[:,1,1])/4
        sb=((P[:,1,0]+P[:,1,1])-(P[:,0,0]+P[:,0,1]))/2
        sc=((P[:,0,1]+P[:,1,1])-(P[:,0,0]+P[:,1,0]))/2
        sbc=P[:,1,1]-P[:,1,0]-P[:,0,1]+P[:,0,0]
        return [m,sb,sc,sbc]
    coefs=[bicoef(A0),bicoef(A1),bicoef(A2)]
    ap=(xa*np.float32(0.5)); 
    tb=[np.ones(n,np.float32), xb, xc, xb*xc]
    ta=[np.ones(n,np.float32), ap, np.abs(ap)]
    acc=np.zeros(n,np.float32)
    for i in range(3):
        for j in range(4):
            acc+=h(ta[i]*tb[j])*h(coefs[i][j])
    e_slab=np.abs(acc-exact)
    print(("smooth" if smooth else "random"), "corner: mean %.2e max %.2e | slab: mean %.2e max %.2e"%(e_corner.mean(),e_corner.max(),e_slab.mean(),e_slab.max()))
trial(); trial(smooth=True)
