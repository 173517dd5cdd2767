import os, sys, time, numpy as np
sys.path[:0]=['/root/repo','/root/repo/frame-interpolation_amd','/root/repo/tests']
from film_hip.engine import FilmEngine
from film_hip import weights as W
from film_hip.options import PUBLISHED
import torch
from film_hip.torch_io import DeviceInterpolator, pinned_frame
w=W.make_synthetic_weights(PUBLISHED, seed=0)
eng=FilmEngine(PUBLISHED, device=0); eng.set_weights(w)
rng=np.random.default_rng(0)
x0=rng.random((1,1080,1920,3),dtype=np.float32); x1=rng.random((1,1080,1920,3),dtype=np.float32)
d0=torch.from_numpy(x0).cuda(); d1=torch.from_numpy(x1).cuda()
di=DeviceInterpolator(eng, align=64, block_shape=[2,2])
ref=di(d0,d1).cpu().numpy()
def timeit(f, n=9, warm=3):
    for _ in range(warm): f()
    ts=[]
    for _ in range(n):
        t=time.perf_counter(); f(); ts.append((time.perf_counter()-t)*1e3)
    return np.median(ts), min(ts)
def dev():
    di(d0,d1); torch.cuda.synchronize()
p0=pinned_frame(x0.shape); p1=pinned_frame(x0.shape); po=pinned_frame(x0.shape); p0[:]=x0; p1[:]=x1
oo=np.empty_like(x0)
for rep in range(2):
    print('device resident  median %.3f min %.3f'%timeit(dev))
    for ov in (0,1):
        eng.set_option('host_overlap', ov)
        g=eng.interpolate_frames(x0,x1,align=64,block_shape=[2,2]); assert np.array_equal(g,ref), float(np.abs(g-ref).max())
        print('host_overlap',ov,'pageable, fresh out  median %.3f min %.3f'%timeit(lambda: eng.interpolate_frames(x0,x1,align=64,block_shape=[2,2])))
        print('host_overlap',ov,'pageable, reused out median %.3f min %.3f'%timeit(lambda: eng.interpolate_frames(x0,x1,align=64,block_shape=[2,2],out=oo)))
        g=eng.interpolate_frames(p0,p1,align=64,block_shape=[2,2],out=po); assert np.array_equal(g,ref)
        print('host_overlap',ov,'pinned in / out      median %.3f min %.3f'%timeit(lambda: eng.interpolate_frames(p0,p1,align=64,block_shape=[2,2],out=po)))
# other shapes through the pipeline: bits against the unpipelined call
for shape, bs in (((1,720,1280,3),[2,2]), ((1,256,256,3),None), ((2,360,640,3),[2,1]), ((1,1080,1920,3),[4,4]), ((1,540,960,3),[1,2])):
    a=rng.random(shape,dtype=np.float32); b=rng.random(shape,dtype=np.float32)
    eng.set_option('host_overlap',0); r0=eng.interpolate_frames(a,b,align=64,block_shape=bs)
    eng.set_option('host_overlap',1); r1=eng.interpolate_frames(a,b,align=64,block_shape=bs)
    print(shape, bs, 'pipeline == plain:', np.array_equal(r0,r1))
    assert np.array_equal(r0,r1)
