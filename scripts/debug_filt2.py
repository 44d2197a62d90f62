import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from decode_tonal_langauge_amd import _lib
from decode_tonal_langauge_amd._lib import ptr, check
from oracle import signal_oracle as sg
from scipy.signal import butter, lfilter_zi
x = np.random.default_rng(0).standard_normal((2, 1000))
b,a = butter(4, np.asarray([0.3,100.])/200, btype='bandpass')
zi = lfilter_zi(b,a)
dev=torch.device('cuda:0')
xd=torch.from_numpy(x).to(dev); bd=torch.from_numpy(b).to(dev); ad=torch.from_numpy(a).to(dev); zd=torch.from_numpy(zi).to(dev)
edge=27; T=1000
y=torch.empty(2,T,dtype=torch.float64,device=dev); work=torch.empty(2,T+2*edge,dtype=torch.float64,device=dev)
lib=_lib.load()
check(lib.tl_filtfilt_f64(ptr(xd),1,ptr(bd),ptr(ad),ptr(zd),ptr(y),ptr(work),2,T,9,torch.cuda.current_stream().cuda_stream),'ff')
torch.cuda.synchronize()
left = 2*x[:,:1]-x[:,edge:0:-1]; right = 2*x[:,-1:]-x[:,-2:-(edge+2):-1]
ext=np.concatenate([left,x,right],axis=1)
y1,_=sg.lfilter_df2t(b,a,ext,zi=zi[None,:]*ext[:,:1])
w=work.cpu().numpy()
d=np.abs(w-y1)
print('forward pass max abs err', d.max(), 'at', np.unravel_index(d.argmax(), d.shape), 'first', d[0,:4], 'mid', d[0,500:503])
print('b dtype', b.dtype, bd.dtype, 'zi', zi.dtype)
from fractions import Fraction as Fr
x0=ext[0,0]; z0=zi[0]*x0
print('x0',repr(x0),'zi0',repr(zi[0]),'b0',repr(b[0]))
plain = z0 + b[0]*x0
fma1 = float(Fr(z0) + Fr(b[0])*Fr(x0))            # fma(b0,x0,z0) with z0 rounded
fma2 = float(Fr(zi[0])*Fr(x0) + Fr(float(b[0]*x0)))  # fma(zi0,x0, fl(b0*x0))
exact = float((Fr(zi[0])+Fr(b[0]))*Fr(x0))
print('gpu',repr(w[0,0]),'plain',repr(plain),'fma1',repr(fma1),'fma2',repr(fma2),'exact',repr(exact),'oracle',repr(y1[0,0]))
