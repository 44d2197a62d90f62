import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from decode_tonal_langauge_amd.preprocess.signal import frequency_filter as ff
from oracle import signal_oracle as sg
from scipy.signal import butter, lfilter, lfilter_zi, filtfilt
x = np.random.default_rng(0).standard_normal((2, 1000))
for order, fr in ((4,[0.3,100]),(2,[0.3,100]),(4,[20.,100.])):
    y = ff.butter_filter(x, fr, 400, order=order)
    ref = sg.butter_filter(x, fr, 400, order=order)
    b,a = butter(order, np.asarray(fr)/200, btype='bandpass')
    ref2 = filtfilt(b,a,x,axis=-1)
    d = np.abs(y-ref)
    print(order, fr, 'max err', d.max()/np.abs(ref).max(), 'argmax', np.unravel_index(d.argmax(), d.shape), 'oracle-vs-scipy', np.abs(ref-ref2).max())
    print('   first errs', d[0,:5], 'last errs', d[0,-5:])
yc = ff.butter_filter(x, [0.3,100], 400, causal=True)
refc = sg.butter_filter(x, [0.3,100], 400, causal=True)
print('causal', np.abs(yc-refc).max()/np.abs(refc).max())
