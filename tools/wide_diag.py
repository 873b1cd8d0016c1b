"""Walk statistics: the binary tree (reference counters) or the 4-wide tree (RTGPU_WIDE=1 RTGPU_WIDE_DIAG=1 [RTGPU_WIDE_SORT=1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import scenes
w,h=1920,1080
scene,camera=scenes.sponza_class(w/h)
vp=ra.Viewport(w,h,seed=515,max_ray_depth=8); vp.set_renderer(scene)
ra.rtgpu_lib().rtgpu_set_intersection_counters(vp.device_context(), 1 if os.environ.get("COUNT") else 0)
vp.render(camera,4); c=vp.counters()
rays=c["numRays"]+c["numShadowRays"]
print({k:c[k] for k in ("numPassedRayBoxTests","numPassedRayTriangleTests","numMeshHits","numAnalyticHits","numPrimaryRays","numShadowRaysHit","numRays","numShadowRays","numRetracedRays","numUntrustedRays","numStackOverflowRays","diag2","numRayBoxTests","numShadowRayBoxTests","numRayTriangleTests","numShadowRayTriangleTests")})
if os.environ.get("RTGPU_WIDE_DIAG") == "2":
    print("stack depth: rays whose stack held more than 9 / 13 / 17 entries (a 12 / 16 / 20-entry stack hands them over): %d / %d / %d of %d rays = %.4f %% / %.4f %% / %.4f %%" % (
        c["numUntrustedRays"], c["numStackOverflowRays"], c["diag2"], rays, 100.0*c["numUntrustedRays"]/rays, 100.0*c["numStackOverflowRays"]/rays, 100.0*c["diag2"]/rays))
elif os.environ.get("RTGPU_WIDE_DIAG"):
    print("interior visits/ray %.2f  interior-loop lane utilisation %.3f  leaf visits/ray %.2f" % (c["numUntrustedRays"]/rays, c["numUntrustedRays"]/max(1,c["numStackOverflowRays"]), c["diag2"]/rays))
    tot = c["numPassedRayTriangleTests"]
    print("wave clocks: refill %.1f %%  interior %.1f %%  leaf/finish %.1f %%  (sum %.1f %% of the waves' lifetime)" % (100*c["numRayBoxTests"]/tot, 100*c["numPassedRayBoxTests"]/tot, 100*c["numRayTriangleTests"]/tot, 100*(c["numRayBoxTests"]+c["numPassedRayBoxTests"]+c["numRayTriangleTests"])/tot))
    print("phase runs per wave-lifetime: refill %d interior %d leaf %d; clocks per run: refill %.0f interior %.0f leaf %.0f; interior wave-steps per run %.2f, clocks per interior wave-step %.0f" % (
        c["numMeshHits"], c["numShadowRayBoxTests"], c["numShadowRayTriangleTests"], c["numRayBoxTests"]/max(1,c["numMeshHits"]), c["numPassedRayBoxTests"]/max(1,c["numShadowRayBoxTests"]), c["numRayTriangleTests"]/max(1,c["numShadowRayTriangleTests"]),
        c["numStackOverflowRays"]/64/max(1,c["numShadowRayBoxTests"]), c["numPassedRayBoxTests"]/max(1,c["numStackOverflowRays"]/64)))
    print("drain phase (queue empty -> the wave's last ray done): %.1f %% of the wave time" % (100.0 * c["numShadowRayTriangleTests"] / tot))
    base_primary, base_shadow_hit = w * h * 4, int(os.environ.get("BASE_SHADOW_HIT", "0"))
    print("refill runs %d (%.0f clocks each); cursor claims %d, %.0f clocks each = %.1f %% of the wave time (needs BASE_SHADOW_HIT = numShadowRaysHit of a plain run; raw %d)" % (
        c["numAnalyticHits"], c["numRayBoxTests"] / max(1, c["numAnalyticHits"]), c["numShadowRaysHit"] - base_shadow_hit,
        (c["numPrimaryRays"] - base_primary) / max(1, c["numShadowRaysHit"] - base_shadow_hit), 100.0 * (c["numPrimaryRays"] - base_primary) / tot, c["numShadowRaysHit"]))
if os.environ.get("COUNT"):
    print("binary: visits/ray %.2f  triangle tests/ray %.2f" % ((c["numRayBoxTests"]+c["numShadowRayBoxTests"])/2/rays, (c["numRayTriangleTests"]+c["numShadowRayTriangleTests"])/rays))
