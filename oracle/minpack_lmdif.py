"""Scalar restatement of MINPACK-1 `lmdif` (with fdjac2, qrfac, lmpar, qrsolv, enorm) as
SciPy's `curve_fit` / `leastsq` drive it by default (ftol = xtol = 1.49012e-8, gtol = 0,
factor = 100, epsfcn = machine epsilon, mode 1).

TEST INFRASTRUCTURE ONLY.  The device port of the carrier fit
(thrifty_amd/csrc/lmdif8.hpp) follows these steps line by line; this file exists so that
tests/test_lmdif_restatement.py can pin the *algorithm* against SciPy itself (agreement to
the last bit on random Dirichlet-fit problems) -- the reference only ever calls
`scipy.optimize.curve_fit` (carrier_sync.py:189), whose MINPACK routine is a compiled
third-party dependency (SciPy 1.15.3 here; unpinned in the reference's requirements.txt).
Restated from the published algorithm (More', Garbow, Hillstrom, "User Guide for
MINPACK-1", ANL-80-74, 1980); nothing here is shipped or imported by the product.
"""
import math
EPSMCH = 2.220446049250313e-16
DWARF = 2.2250738585072014e-308

def enorm(v):
    # MINPACK enorm for mid-range magnitudes reduces to sqrt(sum squares); keep general form simple
    rdwarf, rgiant = 3.834e-20, 1.304e19
    s1=s2=s3=0.0; x1max=x3max=0.0
    agiant = rgiant/len(v)
    for x in v:
        xabs=abs(x)
        if xabs>rdwarf and xabs<agiant:
            s2+=xabs*xabs
        elif xabs<=rdwarf:
            if xabs>x3max:
                s3=1.0+s3*(x3max/xabs)**2; x3max=xabs
            elif xabs!=0: s3+=(xabs/x3max)**2
        else:
            if xabs>x1max:
                s1=1.0+s1*(x1max/xabs)**2; x1max=xabs
            else: s1+=(xabs/x1max)**2
    if s1!=0: return x1max*math.sqrt(s1+(s2/x1max)/x1max)
    if s2!=0:
        if s2>=x3max: return math.sqrt(s2*(1.0+(x3max/s2)*(x3max*s3)))
        return math.sqrt(x3max*((s2/x3max)+(x3max*s3)))
    return x3max*math.sqrt(s3)

def qrfac(a):  # a: m x n (list of lists), pivoting
    m=len(a); n=len(a[0])
    acnorm=[enorm([a[i][j] for i in range(m)]) for j in range(n)]
    rdiag=acnorm[:]; wa=rdiag[:]; ipvt=list(range(n))
    for j in range(min(m,n)):
        kmax=j
        for k in range(j,n):
            if rdiag[k]>rdiag[kmax]: kmax=k
        if kmax!=j:
            for i in range(m): a[i][j],a[i][kmax]=a[i][kmax],a[i][j]
            rdiag[kmax]=rdiag[j]; wa[kmax]=wa[j]
            ipvt[j],ipvt[kmax]=ipvt[kmax],ipvt[j]
        ajnorm=enorm([a[i][j] for i in range(j,m)])
        if ajnorm!=0:
            if a[j][j]<0: ajnorm=-ajnorm
            for i in range(j,m): a[i][j]/=ajnorm
            a[j][j]+=1.0
            for k in range(j+1,n):
                s=0.0
                for i in range(j,m): s+=a[i][j]*a[i][k]
                temp=s/a[j][j]
                for i in range(j,m): a[i][k]-=temp*a[i][j]
                if rdiag[k]!=0:
                    temp=a[j][k]/rdiag[k]
                    rdiag[k]*=math.sqrt(max(0.0,1.0-temp*temp))
                    if 0.05*(rdiag[k]/wa[k])**2<=EPSMCH:
                        rdiag[k]=enorm([a[i][k] for i in range(j+1,m)]); wa[k]=rdiag[k]
        rdiag[j]=-ajnorm
    return ipvt,rdiag,acnorm

def qrsolv(n,r,ipvt,diag,qtb):
    # r: n x n upper triangular (full matrix list), returns x, sdiag ; r lower part overwritten with s
    r=[row[:] for row in r]
    x=[0.0]*n; sdiag=[0.0]*n; wa=[0.0]*n
    for j in range(n):
        for i in range(j,n): r[i][j]=r[j][i]
        x[j]=r[j][j]; wa[j]=qtb[j]
    for j in range(n):
        l=ipvt[j]
        if diag[l]!=0:
            for k in range(j,n): sdiag[k]=0.0
            sdiag[j]=diag[l]
            qtbpj=0.0
            for k in range(j,n):
                if sdiag[k]==0: continue
                if abs(r[k][k])<abs(sdiag[k]):
                    cotan=r[k][k]/sdiag[k]; sin_=0.5/math.sqrt(0.25+0.25*cotan*cotan); cos_=sin_*cotan
                else:
                    tan_=sdiag[k]/r[k][k]; cos_=0.5/math.sqrt(0.25+0.25*tan_*tan_); sin_=cos_*tan_
                r[k][k]=cos_*r[k][k]+sin_*sdiag[k]
                temp=cos_*wa[k]+sin_*qtbpj
                qtbpj=-sin_*wa[k]+cos_*qtbpj
                wa[k]=temp
                for i in range(k+1,n):
                    temp=cos_*r[i][k]+sin_*sdiag[i]
                    sdiag[i]=-sin_*r[i][k]+cos_*sdiag[i]
                    r[i][k]=temp
        sdiag[j]=r[j][j]; r[j][j]=x[j]
    nsing=n
    for j in range(n):
        if sdiag[j]==0 and nsing==n: nsing=j
        if nsing<n: wa[j]=0.0
    for k in range(nsing):
        j=nsing-k-1
        s=0.0
        for i in range(j+1,nsing): s+=r[i][j]*wa[i]
        wa[j]=(wa[j]-s)/sdiag[j]
    for j in range(n):
        x[ipvt[j]]=wa[j]
    return x,sdiag,r

def lmpar(n,r,ipvt,diag,qtb,delta,par):
    wa1=[0.0]*n; wa2=[0.0]*n; x=[0.0]*n
    nsing=n
    for j in range(n):
        wa1[j]=qtb[j]
        if r[j][j]==0 and nsing==n: nsing=j
        if nsing<n: wa1[j]=0.0
    for k in range(nsing):
        j=nsing-k-1
        wa1[j]/=r[j][j]; temp=wa1[j]
        for i in range(j): wa1[i]-=r[i][j]*temp
    for j in range(n): x[ipvt[j]]=wa1[j]
    it=0
    for j in range(n): wa2[j]=diag[j]*x[j]
    dxnorm=enorm(wa2); fp=dxnorm-delta
    sdiag=[0.0]*n
    if fp<=0.1*delta:
        return 0.0 if it==0 else par, x, sdiag
    parl=0.0
    if nsing>=n:
        for j in range(n):
            l=ipvt[j]; wa1[j]=diag[l]*(wa2[l]/dxnorm)
        for j in range(n):
            s=0.0
            for i in range(j): s+=r[i][j]*wa1[i]
            wa1[j]=(wa1[j]-s)/r[j][j]
        temp=enorm(wa1); parl=((fp/delta)/temp)/temp
    for j in range(n):
        s=0.0
        for i in range(j+1): s+=r[i][j]*qtb[i]
        l=ipvt[j]; wa1[j]=s/diag[l]
    gnorm=enorm(wa1); paru=gnorm/delta
    if paru==0: paru=DWARF/min(delta,0.1)
    par=max(par,parl); par=min(par,paru)
    if par==0: par=gnorm/dxnorm
    while True:
        it+=1
        if par==0: par=max(DWARF,0.001*paru)
        temp=math.sqrt(par)
        for j in range(n): wa1[j]=temp*diag[j]
        x,sdiag,rs=qrsolv(n,r,ipvt,wa1,qtb)
        for j in range(n): wa2[j]=diag[j]*x[j]
        dxnorm=enorm(wa2); temp=fp; fp=dxnorm-delta
        if abs(fp)<=0.1*delta or (parl==0 and fp<=temp and temp<0) or it==10: break
        for j in range(n):
            l=ipvt[j]; wa1[j]=diag[l]*(wa2[l]/dxnorm)
        for j in range(n):
            wa1[j]/=sdiag[j]; temp=wa1[j]
            for i in range(j+1,n): wa1[i]-=rs[i][j]*temp
        temp=enorm(wa1); parc=((fp/delta)/temp)/temp
        if fp>0: parl=max(parl,par)
        if fp<0: paru=min(paru,par)
        par=max(parl,par+parc)
    return par,x,sdiag

def lmdif(fcn,x0,m,ftol=1.49012e-8,xtol=1.49012e-8,gtol=0.0,maxfev=None,epsfcn=None,factor=100.0):
    n=len(x0); x=list(x0)
    if maxfev is None: maxfev=200*(n+1)
    if epsfcn is None: epsfcn=EPSMCH
    fvec=list(fcn(x)); nfev=1
    fnorm=enorm(fvec)
    par=0.0; it=1; info=0
    diag=[0.0]*n
    while True:
        eps=math.sqrt(max(epsfcn,EPSMCH))
        fjac=[[0.0]*n for _ in range(m)]
        for j in range(n):
            temp=x[j]; h=eps*abs(temp)
            if h==0: h=eps
            x[j]=temp+h
            wa=fcn(x); x[j]=temp
            for i in range(m): fjac[i][j]=(wa[i]-fvec[i])/h
        nfev+=n
        ipvt,wa1,wa2=qrfac(fjac)
        if it==1:
            for j in range(n):
                diag[j]=wa2[j] if wa2[j]!=0 else 1.0
            wa3=[diag[j]*x[j] for j in range(n)]
            xnorm=enorm(wa3); delta=factor*xnorm
            if delta==0: delta=factor
        wa4=fvec[:]
        qtf=[0.0]*n
        for j in range(n):
            if fjac[j][j]!=0:
                s=0.0
                for i in range(j,m): s+=fjac[i][j]*wa4[i]
                temp=-s/fjac[j][j]
                for i in range(j,m): wa4[i]+=fjac[i][j]*temp
            fjac[j][j]=wa1[j]; qtf[j]=wa4[j]
        gnorm=0.0
        if fnorm!=0:
            for j in range(n):
                l=ipvt[j]
                if wa2[l]!=0:
                    s=0.0
                    for i in range(j+1): s+=fjac[i][j]*(qtf[i]/fnorm)
                    gnorm=max(gnorm,abs(s/wa2[l]))
        if gnorm<=gtol: info=4; break
        for j in range(n): diag[j]=max(diag[j],wa2[j])
        r=[[fjac[i][j] if i<=j else 0.0 for j in range(n)] for i in range(n)]
        while True:
            par,p,sd=lmpar(n,r,ipvt,diag,qtf,delta,par)
            wa1n=[-v for v in p]
            wa2n=[x[j]+wa1n[j] for j in range(n)]
            wa3=[diag[j]*wa1n[j] for j in range(n)]
            pnorm=enorm(wa3)
            if it==1: delta=min(delta,pnorm)
            wa4=list(fcn(wa2n)); nfev+=1
            fnorm1=enorm(wa4)
            actred=-1.0
            if 0.1*fnorm1<fnorm: actred=1.0-(fnorm1/fnorm)**2
            wa3=[0.0]*n
            for j in range(n):
                l=ipvt[j]; temp=wa1n[l]
                for i in range(j+1): wa3[i]+=r[i][j]*temp
            temp1=enorm(wa3)/fnorm; temp2=(math.sqrt(par)*pnorm)/fnorm
            prered=temp1*temp1+temp2*temp2/0.5
            dirder=-(temp1*temp1+temp2*temp2)
            ratio=0.0
            if prered!=0: ratio=actred/prered
            if ratio<=0.25:
                if actred>=0: temp=0.5
                else: temp=0.5*dirder/(dirder+0.5*actred)
                if 0.1*fnorm1>=fnorm or temp<0.1: temp=0.1
                delta=temp*min(delta,pnorm/0.1); par=par/temp
            elif par==0 or ratio>=0.75:
                delta=pnorm/0.5; par=0.5*par
            if ratio>=1e-4:
                x=wa2n; wa2s=[diag[j]*x[j] for j in range(n)]
                fvec=wa4; xnorm=enorm(wa2s); fnorm=fnorm1; it+=1
            if abs(actred)<=ftol and prered<=ftol and 0.5*ratio<=1: info=1
            if delta<=xtol*xnorm: info=2
            if abs(actred)<=ftol and prered<=ftol and 0.5*ratio<=1 and info==2: info=3
            if info!=0: break
            if nfev>=maxfev: info=5
            if abs(actred)<=EPSMCH and prered<=EPSMCH and 0.5*ratio<=1: info=6
            if delta<=EPSMCH*xnorm: info=7
            if gnorm<=EPSMCH: info=8
            if info!=0: break
            if ratio>=1e-4: break
        if info!=0: break
    return x,info,nfev
