# (needs the EXPERIMENT build: make -C webgpu-pathtracer_amd/csrc experiments; MI3PT_LIBRARY=webgpu-pathtracer_amd/libmi3pt_exp.so
#  MI3PT_FORCE_SLOW_SLAB=1 python profiles/check_slow_slab.py -- the release library has no such switch)
import sys, os, time
sys.path.insert(0,'webgpu-pathtracer_amd/py'); sys.path.insert(0,'tests')
import ptcommon as pc
from mi3pt_host import capi, scenes
sc = scenes.demo_scene(); sc.build_bvh(); env = scenes.synthetic_env()
ctx = capi.Context(0); pc.upload_scene(ctx, sc, env); ctx.resize(1920,1080); ctx.set_kernel_variant(4)
for f in range(2,6):
    pc.gpu_frame(ctx, pc.rt_uniforms(sc,1920,1080,frame=f,bounces=8), pc.acc_uniforms(1920,1080,f), 3)
ctx.sync(); ctx.reset_counters()
t=time.time()
for f in range(6,26):
    pc.gpu_frame(ctx, pc.rt_uniforms(sc,1920,1080,frame=f,bounces=8), pc.acc_uniforms(1920,1080,f), 3)
ctx.sync(); dt=(time.time()-t)/20
c = ctx.counters(); print(os.environ.get('MI3PT_FORCE_SLOW_SLAB'), 'ms/frame', dt*1e3, 'rays', c['rays'], 'slow segments', c['reserved'])
