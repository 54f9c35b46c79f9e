import numpy as np, sys
sys.path.insert(0,'/root/repo')
import vulkan_rtiow_amd as V
sph, mat = V.make_cover_scene(1, 11)
S = np.stack([sph['cx'],sph['cy'],sph['cz'],sph['radius']],1).astype(np.float64)
n = len(S); print("n", n)
rng = np.random.default_rng(0)
# primary rays
W,H=1200,800
cam_from=np.array([13,2,3.]); at=np.zeros(3); up=np.array([0,1,0.])
w=(cam_from-at); w/=np.linalg.norm(w); u=np.cross(up,w); u/=np.linalg.norm(u); v=np.cross(w,u)
vfov=20; h=np.tan(np.radians(vfov)/2); vh=2*h; vw=vh*1.5
N=4000
px=rng.random(N); py=rng.random(N)
d = (-w)[None,:] + (px[:,None]-.5)*vw*u[None,:] + (py[:,None]-.5)*vh*v[None,:]
d/= np.linalg.norm(d,axis=1)[:,None]
o = np.repeat(cam_from[None,:],N,0)
def hit(o,d):
    oc = o[:,None,:]-S[None,:,:3]
    b = (oc*d[:,None,:]).sum(-1); c=(oc*oc).sum(-1)-S[None,:,3]**2
    disc=b*b-c
    t = np.where(disc>=0, -b-np.sqrt(np.maximum(disc,0)), np.inf)
    t = np.where(t>1e-3, t, np.where(disc>=0, -b+np.sqrt(np.maximum(disc,0)), np.inf))
    t = np.where(t>1e-3, t, np.inf)
    i = t.argmin(1); tt=t[np.arange(len(o)),i]
    return i,tt
i,tt = hit(o,d)
ok = np.isfinite(tt)
p = o[ok]+tt[ok,None]*d[ok]
nrm = (p - S[i[ok],:3])/S[i[ok],3:4]
r = rng.normal(size=p.shape); r/=np.linalg.norm(r,axis=1)[:,None]
d2 = nrm + r; d2/=np.linalg.norm(d2,axis=1)[:,None]
rays_o = np.concatenate([o,p]); rays_d=np.concatenate([d,d2])
# third generation
i2,t2 = hit(p,d2); ok2=np.isfinite(t2)
p3 = p[ok2]+t2[ok2,None]*d2[ok2]; n3=(p3-S[i2[ok2],:3])/S[i2[ok2],3:4]
r = rng.normal(size=p3.shape); r/=np.linalg.norm(r,axis=1)[:,None]
d3=n3+r; d3/=np.linalg.norm(d3,axis=1)[:,None]
rays_o=np.concatenate([rays_o,p3]); rays_d=np.concatenate([rays_d,d3])
print("rays", len(rays_o), "primary", N, "2nd", len(p), "3rd", len(p3))
# clusters: morton sort small
rad=np.abs(S[:,3]); med=np.median(rad); small=np.where(rad<=4*med)[0]; large=np.where(rad>4*med)[0]
lo=S[small,:3].min(0); hi=S[small,:3].max(0)
q=np.minimum(1023,((S[small,:3]-lo)/np.where(hi>lo,hi-lo,1)*1023)).astype(np.int64)
def spread(v):
    out=np.zeros_like(v)
    for b in range(10): out |= ((v>>b)&1)<<(3*b)
    return out
code=spread(q[:,0])|(spread(q[:,1])<<1)|(spread(q[:,2])<<2)
order=small[np.argsort(code,kind='stable')]
def stats(cs, margin=0.0, order=order):
    cl=[order[k:k+cs] for k in range(0,len(order),cs)]
    nc=len(cl)
    sh=np.zeros(len(rays_o)); bh=np.zeros(len(rays_o))
    for m in cl:
        c=S[m,:3].mean(0); R=(np.linalg.norm(S[m,:3]-c,axis=1)+rad[m]).max(); R2=R*R+margin
        oc=rays_o-c; b=(oc*rays_d).sum(1); cc=(oc*oc).sum(1)-R2
        disc=b*b-cc; sh += (disc>=0)&((b<0)|(cc<0))
        mr=np.sqrt(margin+rad[m]**2)
        blo=(S[m,:3]-mr[:,None]).min(0); bhi=(S[m,:3]+mr[:,None]).max(0)
        with np.errstate(divide='ignore',invalid='ignore'):
            inv=1/rays_d
            t0=(blo-rays_o)*inv; t1=(bhi-rays_o)*inv
        tn=np.minimum(t0,t1).max(1); tf=np.maximum(t0,t1).min(1)
        bh += (np.maximum(tn,0)<=tf)
    return nc, sh.mean(), bh.mean()
for cs in (4,8,16):
    for mg in (0.0,0.055,0.01):
        nc,shm,bhm=stats(cs,mg)
        print(f"cs={cs} margin={mg}: clusters={nc} sphere-bound hits/ray={shm:.2f} aabb hits/ray={bhm:.2f}  members: sph {shm*cs:.1f} aabb {bhm*cs:.1f}")

print("---- median split")
def median_split(ids, leaf):
    if len(ids) <= leaf: return [ids]
    pts = S[ids,:3]; ext = pts.max(0)-pts.min(0); ax = ext.argmax()
    o = ids[np.argsort(pts[:,ax],kind='stable')]
    # split so that left gets a multiple of leaf (keeps leaves full)
    nl = len(o)//2
    nl = max(leaf, (nl + leaf//2)//leaf*leaf) if len(o) > 2*leaf else len(o)//2
    return median_split(o[:nl], leaf) + median_split(o[nl:], leaf)
for leaf in (8,16):
    cl = median_split(small, leaf)
    order2 = np.concatenate([np.pad(c,(0,0)) for c in cl])
    sizes=[len(c) for c in cl]
    # stats() chunks by cs; emulate by custom cluster list
    def stats2(cl, margin):
        bh=np.zeros(len(rays_o)); mem=np.zeros(len(rays_o))
        for m in cl:
            mr=np.sqrt(margin+rad[m]**2)
            blo=(S[m,:3]-mr[:,None]).min(0); bhi=(S[m,:3]+mr[:,None]).max(0)
            with np.errstate(divide='ignore',invalid='ignore'):
                inv=1/rays_d; t0=(blo-rays_o)*inv; t1=(bhi-rays_o)*inv
            tn=np.minimum(t0,t1).max(1); tf=np.maximum(t0,t1).min(1)
            h=(np.maximum(tn,0)<=tf); bh+=h
        return len(cl), bh.mean()
    nc,bhm=stats2(cl,0.017)
    print(f"leaf={leaf}: clusters={nc} sizes min/max {min(sizes)}/{max(sizes)} aabb hits/ray={bhm:.2f}")
nc,shm,bhm=stats(16,0.017); print("morton16 margin .017:", nc, bhm)
nc,shm,bhm=stats(8,0.017); print("morton8 margin .017:", nc, bhm)
