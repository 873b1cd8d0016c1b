"""Config 5 (rough-glass BDPT) is bimodal from run to run (round 4: 2510-2600 or 2700-2780 Msamples/s for ONE library).  Where does the mode live?
  A. several timed regions on ONE viewport (same allocations, same process): does the rate move between regions?
  B. several fresh viewports in one process (new allocations each): does it move between viewports, and with what -- the arena addresses,
     the clocks (rocm-smi before / after), the per-class kernel times of the SAME viewport measured right behind its timed regions?
python tools/vcm_bimodal.py [--viewports 6 --regions 4 --steps 20 --warmup 5]"""
import argparse, ctypes as C, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--viewports", type=int, default=6); ap.add_argument("--regions", type=int, default=4)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
args = ap.parse_args()
import raytracer_amd as ra
from raytracer_amd import scenes
lib, host = ra.rtgpu_lib(), ra.host_lib()
w, h = args.width, args.height
scene, camera = scenes.rough_glass_slab(w / h)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out); card = d[sorted(d)[0]]
        return {k: v for k, v in card.items() if any(s in k.lower() for s in ("sclk", "mclk", "fclk", "power", "junction"))}
    except Exception as e:
        return {"error": repr(e)}


def kernel_times(ctx):
    ms = (C.c_double * 8)(); launches = (C.c_uint64 * 8)(); names = (C.c_char_p * 8)()
    lib.rtgpu_get_kernel_times(ctx, ms, launches, names)
    return {names[i].decode(): round(ms[i], 2) for i in range(8) if names[i]}


for v in range(args.viewports):
    vp = ra.Viewport(w, h, seed=515, max_ray_depth=8)
    vp.set_renderer(scene, name="VCM", intersection_counters=False)
    vp.set_vcm(**scenes.ROUGH_GLASS_SLAB_VCM)
    ctx = vp.device_context()
    vp.render(camera, args.warmup); lib.rtgpu_synchronize(ctx)
    sum_ptr, sec_ptr, nfl = C.c_void_p(), C.c_void_p(), C.c_size_t()
    lib.rtgpu_get_device_sum(ctx, C.byref(sum_ptr), C.byref(sec_ptr), C.byref(nfl))
    before = smi()
    rates = []
    for r in range(args.regions):
        c0 = vp.counters(); lib.rtgpu_synchronize(ctx)
        t0 = time.perf_counter()
        vp.render(camera, args.steps)
        host.rth_viewport_fetch_sum(vp._h)
        dt = time.perf_counter() - t0
        c1 = vp.counters()
        rates.append(round((c1["numRays"] - c0["numRays"]) / dt / 1e6, 1))
    after = smi()
    # the same viewport, serial kernels with HIP-event timing: which class carries the difference
    lib.rtgpu_set_concurrency(ctx, 1); lib.rtgpu_enable_timing(ctx, 1)
    vp.render(camera, args.steps); lib.rtgpu_synchronize(ctx)
    kt = kernel_times(ctx)
    print(json.dumps({"viewport": v, "Msamples_per_s_per_region": rates, "sum_buffer_address": hex(sum_ptr.value or 0), "serial_kernel_ms": kt, "smi_before": before, "smi_after": after}), flush=True)
    del vp
