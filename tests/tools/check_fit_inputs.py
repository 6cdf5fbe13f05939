"""Dev/aux: for the blocks whose carrier offset deviates most from the oracle (weak signals,
full window), re-run SciPy on the magnitudes the GPU saw: the deviation is the fit's sensitivity
to one-ulp differences of its float32 inputs, not the solver."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import thrifty_np as onp
from thrifty_amd import _native as F, synth
from scipy.optimize import curve_fit
n,h=16384,1135
tpl=synth.gold_template(10,3,1.0)
win=onp.unique_window(n,h,len(tpl))
rng=np.random.default_rng(5)
nb=400
blocks,_=synth.synth_blocks(rng,nb,n,tpl,win,signal_frac=1.0,amp=0.02,carrier_bins=(-4000.0,4000.0))
eng=F.Engine(n,h,tpl,(0,15,0),(0,-1),(0,15,0),max_batch=64)
rec=eng.detect(blocks,np.arange(nb))[:,0]
orc=onp.OracleDetector(n,h,tpl,(0,15,0),(0,-1),(0,15,0))
worst=[]
for i in range(nb):
    try: (res,)=orc.detect_u8(i,blocks[i])
    except IndexError: continue
    if not res.carrier.detected or rec[i]["carrier_bin"]!=res.carrier.bin: continue
    worst.append((abs(rec[i]["carrier_offset"]-res.carrier.offset),i))
worst.sort(reverse=True)
print("worst 5:",worst[:5])
spec=eng.debug_fft(blocks[[w[1] for w in worst[:5]]])
for k,(d,i) in enumerate(worst[:5]):
    pk=int(rec[i]["carrier_bin"])
    z=spec[k][(pk+np.arange(-3,4))%n]
    re=z.real.astype(np.float32); im=z.imag.astype(np.float32)
    p=(re.astype(np.float64)*re.astype(np.float64)+ (im*im).astype(np.float64)).astype(np.float32)  # fma(re,re,fl(im*im))
    mags=np.sqrt(p).astype(np.float32)
    xd=np.arange(-3,4)
    def model(x,a,o): return a*np.abs(onp.dirichlet(np.array(x,dtype=np.float64)-o,n,len(tpl)))
    popt,_=curve_fit(model,xd,mags.astype(np.float64),p0=(mags[3],0))
    refm=np.abs(np.fft.fft(onp.iq_u8_to_c64(blocks[i])))[(pk+np.arange(-3,4))%n]
    print(i,"gpu off %.9f  scipy(gpu mags) %.9f  oracle %.9f   max rel mag diff %.2e"%(rec[i]["carrier_offset"],popt[1],orc.detect_u8(i,blocks[i])[0].carrier.offset, np.max(np.abs(mags-refm)/refm)))
