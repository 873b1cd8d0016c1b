// What the reference's own Tests/RaytracingTests.cpp (compiled unchanged by tests/test_cpp_api.py) does not touch of the C++ mirror:
//   * CreateRenderer with a name the factory does not know returns nullptr (Renderer.cpp:45-69);
//   * Viewport::SetPostprocessParams + GetFrontBuffer: the displayable B8G8R8A8 image of a uniformly lit frame
//     (Viewport.cpp:359-431) for each of the three renderer names the reference's tests loop over.
// Exit code 0: all checks passed; 2: no GPU renderer; 1: a check failed (printed).
#include "Core/Scene/Scene.h"
#include "Core/Scene/Camera.h"
#include "Core/Rendering/Viewport.h"
#include "Core/Scene/Light/BackgroundLight.h"
#include "Core/Scene/Object/SceneObject_Light.h"

#include <stdio.h>

using namespace rt;
using namespace math;

static int CheckFrontBuffer(const char* rendererName)
{
    constexpr uint32 size = 48;
    Scene scene;
    scene.AddObject(std::make_unique<LightSceneObject>(std::make_unique<BackgroundLight>(Vector4(0.25f, 0.5f, 0.75f))));
    scene.BuildBVH();

    Viewport viewport;
    viewport.Resize(size, size);
    viewport.SetRenderer(CreateRenderer(rendererName, scene));
    viewport.Reset();
    Camera camera;
    camera.SetPerspective(1.0f, DegToRad(60.0f));
    for (int pass = 0; pass < 3; ++pass) viewport.Render(camera);

    PostprocessParams post;
    post.ditheringStrength = 0.0f;
    post.tonemapper = Tonemapper::Clamped;
    if (!viewport.SetPostprocessParams(post)) { printf("%s: SetPostprocessParams refused valid parameters\n", rendererName); return 1; }
    const Bitmap& front = viewport.GetFrontBuffer();
    if (front.GetWidth() != size || front.GetHeight() != size || front.GetFormat() != Bitmap::Format::B8G8R8A8_UNorm)
    {
        printf("%s: front buffer is not a %ux%u B8G8R8A8 bitmap\n", rendererName, size, size);
        return 1;
    }
    // every pixel sees the same background: one colour, blue > green > red > 0 (BGRA byte order: byte 0 = blue = 0.75)
    const uint32* px = reinterpret_cast<const uint32*>(front.GetBytes());
    int failures = 0;
    for (uint32 i = 1; i < size * size; ++i) if (px[i] != px[0]) failures++;
    const uint32 b = px[0] & 255u, g = (px[0] >> 8) & 255u, r = (px[0] >> 16) & 255u;
    if (failures != 0 || !(b > g && g > r && r > 0u))
    {
        printf("%s: front buffer not uniform (%d pixels differ) or channels out of order (b %u g %u r %u)\n", rendererName, failures, b, g, r);
        return 1;
    }
    printf("%-16s front buffer ok (b %u g %u r %u)\n", rendererName, b, g, r);
    return 0;
}

int main()
{
    Scene probe;
    if (!CreateRenderer("Path Tracer MIS", probe)) { printf("no GPU renderer available\n"); return 2; }
    int failures = 0;
    if (CreateRenderer("no such renderer", probe) != nullptr) { printf("unknown renderer names must give nullptr\n"); failures++; }
    for (const char* name : { "Path Tracer", "Path Tracer MIS", "VCM" }) failures += CheckFrontBuffer(name);
    printf("%d check(s) failed\n", failures);
    return failures ? 1 : 0;
}
