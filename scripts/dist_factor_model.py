"""Host-side model of the flops per rank of the distributed factorisation (chol_inverse.h cholesky_linvt_blocked_dist + the inverse tiles a rank forms, lasso_tall.hip):
the same tile dealing and block ownership as the library, for sizes and rank counts this pool cannot run."""
# model of the flops per rank of the distributed factorisation (chol_inverse.h) + inverse tiles (lasso_tall.hip)
import math, sys
def sched(p, nparts, cus=256):
    p32=(p+31)//32*32; nrb=(p+255)//256
    def count(wb,ws,split):
        t=0
        for rb in range(nrb):
            w = wb if rb>=split else ws
            cols=min((rb+1)*256,p32); t+=(cols+w-1)//w
        return t
    slots=4*cus*max(1,nparts)
    if 4*count(32,32,0)<=3*slots: return (32,32,0)
    if count(64,64,0)<=slots: return (64,64,0)
    if 10*count(128,64,nrb//2)<=14*slots: return (128,64,nrb//2)
    return (192,64,nrb//2)
def model(p,N,grouped=True):
    p32=(p+31)//32*32; nrb=(p+255)//256
    wb,ws,split=sched(p,N)
    h=[]
    for rb in range(nrb-1,-1,-1):
        w= wb if rb>=split else ws
        cols=min((rb+1)*256,p32); ns=(cols+w-1)//w
        for sg in range(ns):
            c0=sg*w; h.append((rb,c0,min(w,cols-c0)))
    nb=(p+127)//128; pp=nb*128
    out=[]
    for part in range(N):
        mine=[]
        if grouped:
            g=-1; last=None
            for t in h:
                w= wb if t[0]>=split else ws
                gw=w//math.gcd(w,128)*128
                key=(t[0],t[1]//gw)
                if key!=last: g+=1; last=key
                if g%N==part: mine.append(t)
        else:
            mine=h[part::N]
        need=set()
        for rb,c0,wd in mine:
            for bi in (2*rb,2*rb+1):
                if bi>=nb: continue
                for bj in range(c0//128,(c0+wd-1)//128+1):
                    if bj<nb and bj<=bi: need.add((bi,bj))
        fl=0
        for k in range(nb):
            r0=k*128; nbk=min(128,p-r0); M=p-(r0+128)
            if k%N==part: fl+=2*128*nbk*(r0+nbk+max(M,0))
            for j in range(k+1,nb):
                if j%N!=part: continue
                c0=j*128; nj=min(128,p-c0)
                fl+=2*128*nj*((p-c0)+(r0+128))
        inv=sum(2*128*128*(pp-bi*128) for bi,bj in need)
        out.append(((fl)/p**3, inv/p**3, len(mine)))
    return out
for p in (2300,4096,10000,16000):
    for N in (2,4,8):
        m=model(p,N)
        print(p,N,"max total %.3f"%max(a+b for a,b,_ in m),"sum %.3f"%sum(a+b for a,b,_ in m), "ideal %.3f"%(1/N), " chol max %.3f inv max %.3f"%(max(a for a,_,_ in m),max(b for _,b,_ in m)), "tiles",[c for _,_,c in m][:4])
