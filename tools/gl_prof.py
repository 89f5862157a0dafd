import sys, time
sys.path.insert(0,'/root/repo')
import torch, numpy as np
import mlx_audio_primitives_amd as ap
from mlx_audio_primitives_amd.griffinlim import _project
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
g=torch.Generator(device="cuda").manual_seed(0)
y=torch.randn((64,110250),device="cuda",generator=g)*0.1
S=ap.stft(y); mag=ap.magnitude(S)
print("stft ms", t(lambda: ap.stft(y)))
print("istft ms", t(lambda: ap.istft(S,hop_length=512,length=110250)))
reb=torch.empty_like(S); tp=S.clone()
print("project ms", t(lambda: _project(1,mag,None,S,0.99,tp,reb)))
t0=time.perf_counter(); rng=np.random.default_rng(42); a=rng.uniform(-np.pi,np.pi,(64,1025,216)).astype(np.float32); t1=time.perf_counter(); ad=torch.from_numpy(a).cuda(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host rng ms", (t1-t0)*1e3, "h2d ms", (t2-t1)*1e3)
from mlx_audio_primitives_amd.griffinlim import _random_phase
def rp():
    a=_random_phase(42,(64,1025,216),torch.device("cuda",0)); torch.cuda.synchronize(); return a
t0=time.perf_counter(); rp(); print("random_phase first ms",(time.perf_counter()-t0)*1e3)
t0=time.perf_counter(); rp(); print("random_phase second ms",(time.perf_counter()-t0)*1e3)
import os; print("affinity", len(os.sched_getaffinity(0)))
print("griffinlim32 ms", t(lambda: ap.griffinlim(mag,n_iter=32,random_state=42,length=110250), n=3))
