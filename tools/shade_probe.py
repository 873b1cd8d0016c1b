import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import raytracer_amd as ra
from raytracer_amd import scenes
w,h=1920,1080
scene,camera=scenes.sponza_class(w/h)
vp=ra.Viewport(w,h,seed=1,max_ray_depth=0)   # depth 0 only: generate, trace, shade(0), trace(shadow), accumulate
vp.set_renderer(scene)
lib=ra.rtgpu_lib(); ctx=vp.device_context()
lib.rtgpu_set_intersection_counters(ctx,0)
vp.render(camera,8); lib.rtgpu_synchronize(ctx)
lib.rtgpu_enable_timing(ctx,1)
vp.render(camera,16); lib.rtgpu_synchronize(ctx)
ms=(C.c_double*8)(); n=(C.c_uint64*8)(); names=(C.c_char_p*8)()
lib.rtgpu_get_kernel_times(ctx,ms,n,names)
print({names[i].decode():(round(ms[i],3),int(n[i])) for i in range(4)})
