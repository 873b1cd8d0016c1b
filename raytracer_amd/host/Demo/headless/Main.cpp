// Headless counterpart of the reference's Demo (Demo/Main.cpp:6-40 takes -w/--width, -h/--height, -s/--scene, --renderer,
// --data and opens a window): loads a JSON scene with helpers::LoadScene, renders N passes with the device "Path Tracer MIS"
// through the same rt::Viewport API the window loop uses (Demo.cpp: Resize -> SetRenderer -> Render per frame ->
// GetFrontBuffer) and writes the tone-mapped front buffer as a BMP.  The extra options are --passes, --depth, --output, --seed;
// the environment variable RTGPU_DEVICES ("0,1,2,3" / "all") spreads the frame over several GPUs (Core/Rendering/Renderer.h).
#include "../Demo.h"
#include "../SceneLoader.h"
#include "../../Core/Rendering/Viewport.h"
#include "../../Core/Rendering/Renderer.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>

using namespace rt;

static bool SaveBMP(const char* path, const Bitmap& front)   // 24-bit, rows bottom-up like the BMP files the loader reads
{
    const uint32 w = front.GetWidth(), h = front.GetHeight(), row = (w * 3u + 3u) & ~3u;
    FILE* f = fopen(path, "wb");
    if (!f) return false;
#pragma pack(push, 2)
    struct { uint16 type; uint32 size; uint16 r1, r2; uint32 offBits; uint32 infoSize; int32 width, height; uint16 planes, bitCount; uint32 compression, sizeImage; int32 xppm, yppm; uint32 clrUsed, clrImportant; }
        hdr = { 0x4D42, 54u + row * h, 0, 0, 54, 40, (int32)w, (int32)h, 1, 24, 0, row * h, 2835, 2835, 0, 0 };
#pragma pack(pop)
    bool ok = fwrite(&hdr, sizeof(hdr), 1, f) == 1;
    std::string line(row, '\0');
    for (uint32 y = 0; ok && y < h; ++y)
    {
        const uint32* src = reinterpret_cast<const uint32*>(front.GetBytes() + (size_t)front.GetStride() * (h - 1u - y));
        for (uint32 x = 0; x < w; ++x) { line[3 * x + 0] = (char)(src[x] & 255u); line[3 * x + 1] = (char)((src[x] >> 8) & 255u); line[3 * x + 2] = (char)((src[x] >> 16) & 255u); }
        ok = fwrite(line.data(), row, 1, f) == 1;
    }
    fclose(f);
    return ok;
}

int main(int argc, char* argv[])
{
    uint32 width = 1280, height = 720, passes = 64, depth = 20;
    std::string scenePath, rendererName = "Path Tracer MIS", output = "out.bmp";
    unsigned long long seed = 0; bool haveSeed = false;
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        auto value = [&](const char* name) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", name); exit(2); } return argv[++i]; };
        if (a == "-w" || a == "--width") width = (uint32)atoi(value("--width"));
        else if (a == "-h" || a == "--height") height = (uint32)atoi(value("--height"));
        else if (a == "-s" || a == "--scene") scenePath = value("--scene");
        else if (a == "--renderer") rendererName = value("--renderer");
        else if (a == "--data") gOptions.dataPath = value("--data");
        else if (a == "--passes") passes = (uint32)atoi(value("--passes"));
        else if (a == "--depth") depth = (uint32)atoi(value("--depth"));
        else if (a == "--output") output = value("--output");
        else if (a == "--seed") { seed = strtoull(value("--seed"), nullptr, 10); haveSeed = true; }
        else { fprintf(stderr, "usage: rt_demo -s scene.json [--data dir/] [-w W] [-h H] [--passes N] [--depth D] [--renderer name] [--output out.bmp] [--seed N]\n"); return 2; }
    }
    if (scenePath.empty()) { fprintf(stderr, "no scene given (-s scene.json)\n"); return 2; }

    Scene scene;
    Camera camera;
    if (!helpers::LoadScene(scenePath, scene, camera)) return 1;
    if (!scene.BuildBVH()) return 1;
    camera.SetPerspective((float)width / (float)height, camera.mFieldOfView);

    Viewport viewport;
    RenderingParams params;
    params.maxRayDepth = depth;
    if (haveSeed) viewport.SetSeed(seed);   // reproducible frames (the reference seeds its generators from the clock)
    if (!viewport.SetRenderingParams(params) || !viewport.Resize(width, height)) return 1;
    RendererPtr renderer = CreateRenderer(rendererName, scene);
    if (!renderer) { fprintf(stderr, "renderer '%s' is not available (no GPU?)\n", rendererName.c_str()); return 1; }
    if (!viewport.SetRenderer(renderer)) return 1;
    viewport.Reset();

    const auto t0 = std::chrono::steady_clock::now();
    for (uint32 i = 0; i < passes; ++i) if (!viewport.Render(camera)) return 1;
    const RayTracingCounters counters = viewport.GetTotalCounters();   // synchronises
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%u passes of %ux%u in %.3f s: %.1f Msamples/s (%llu paths x bounces, %llu shadow rays), average error %g\n", passes, width, height, seconds,
           (double)counters.numRays / seconds / 1.0e6, (unsigned long long)counters.numRays, (unsigned long long)counters.numShadowRays,
           (double)viewport.GetProgress().averageError);
    if (!SaveBMP(output.c_str(), viewport.GetFrontBuffer())) { fprintf(stderr, "cannot write %s\n", output.c_str()); return 1; }
    printf("wrote %s\n", output.c_str());
    return 0;
}
