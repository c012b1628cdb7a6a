import sys, ctypes as C; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from hybrid_rendering_amd import api as hr, synth
W,H=1920,1080
sd=synth.sponza_like(1.0); ctx=hr.Context(0); sc=hr.Scene(ctx,sd)
cam=synth.sponza_camera(W/H); ubo=synth.make_ubo(cam,None,synth.sponza_light())
gb=sc.gbuffer(ubo,W,H); sob,sr=synth.blue_noise_tables(); sob_d,sr_d=torch.from_numpy(sob).cuda(),torch.from_numpy(sr).cuda()
p=hr.RayTracedShadows(ctx,W,H); fi=hr.frame_inputs(gb,gb,ubo,0,0,sob_d,sr_d)
r,n,t=p.trace_stats(sc,fi); m=C.c_uint64(0); hr.lib().hr_shadows_trace_divergence(p.h,C.byref(m))
print('rays',r,'nodes/ray',n/r,'tris/ray',t/r,'wave_max_steps',m.value,'lane utilisation',(n+t)/(64*m.value), 'waves', (W//8)*(H//8), 'avg max steps/wave', m.value/((W//8)*(H//8)))
