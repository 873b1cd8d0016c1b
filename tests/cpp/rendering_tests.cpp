// The reference's RenderingTest.* suite (Tests/RaytracingTests.cpp:263-523) written against this repository's
// C++ mirror of the reference API: same scene set-up calls, same 32x32 viewport, same pass counts, same
// per-pixel tolerances on the (un-tone-mapped) sum buffer.  The reference loops over {"Path Tracer", "Path Tracer MIS", "VCM"}
// (Tests/RaytracingTests.cpp:17-22) and so does this.  No gtest in the image: a tiny harness.
//
// Build (tests/test_cpp_api.py does this):
//   g++ -std=c++17 -O1 -I raytracer_amd/host tests/cpp/rendering_tests.cpp -L raytracer_amd/lib -lraytracer_amd_host -lrtgpu
#include "Core/Scene/Scene.h"
#include "Core/Scene/Camera.h"
#include "Core/Rendering/Context.h"
#include "Core/Material/Material.h"
#include "Core/Rendering/Viewport.h"
#include "Core/Scene/Light/BackgroundLight.h"
#include "Core/Scene/Object/SceneObject_Shape.h"
#include "Core/Scene/Object/SceneObject_Light.h"
#include "Core/Shapes/SphereShape.h"

#include <stdio.h>
#include <string>

using namespace rt;
using namespace math;

static const char* gRendererNames[] = { "Path Tracer", "Path Tracer MIS", "VCM" };   // Tests/RaytracingTests.cpp:17-22
static const char* gRendererName = gRendererNames[1];
static constexpr uint32 ViewportSize = 32;
static int gFailures = 0;

static void ValidateBitmap(const Bitmap& bitmap, const Vector4& expectedColor, float maxError, const char* test)
{
    int bad = 0;
    float worst = 0.0f;
    for (uint32 i = 0; i < bitmap.GetWidth(); ++i)
    {
        for (uint32 j = 0; j < bitmap.GetHeight(); ++j)
        {
            const Vector4 c = bitmap.GetPixel(i, j);
            const float e = Max(fabsf(c.x - expectedColor.x), Max(fabsf(c.y - expectedColor.y), fabsf(c.z - expectedColor.z)));
            if (!(e <= maxError)) bad++;
            if (e > worst) worst = e;
        }
    }
    printf("%-28s %s (worst per-channel error %.5f, tolerance %.3f)\n", test, bad ? "FAILED" : "ok", worst, maxError);
    if (bad) gFailures++;
}

struct Fixture
{
    std::unique_ptr<Scene> mScene = std::make_unique<Scene>();
    std::unique_ptr<Viewport> mViewport = std::make_unique<Viewport>();
};

static void EmptyScene()
{
    Fixture f;
    f.mViewport->Resize(ViewportSize, ViewportSize);
    Camera camera;
    camera.SetPerspective(1.0f, DegToRad(90.0f));
    f.mScene->BuildBVH();
    RendererPtr renderer = CreateRenderer(gRendererName, *f.mScene);
    f.mViewport->SetRenderer(renderer);
    f.mViewport->Reset();
    f.mViewport->Render(camera);
    ValidateBitmap(f.mViewport->GetSumBuffer(), Vector4::Zero(), 0.0f, "EmptyScene");
}

static void BackgroundLightOnly()
{
    Fixture f;
    const Vector4 lightColor(1.0f, 2.0f, 3.0f);
    auto backgroundLight = std::make_unique<BackgroundLight>(lightColor);
    auto lightObject = std::make_unique<LightSceneObject>(std::move(backgroundLight));
    f.mScene->AddObject(std::move(lightObject));
    f.mScene->BuildBVH();
    f.mViewport->Resize(ViewportSize, ViewportSize);
    Camera camera;
    camera.SetPerspective(1.0f, DegToRad(90.0f));
    RendererPtr renderer = CreateRenderer(gRendererName, *f.mScene);
    f.mViewport->SetRenderer(renderer);
    f.mViewport->Reset();
    f.mViewport->Render(camera);
    ValidateBitmap(f.mViewport->GetSumBuffer(), lightColor, 0.01f, "BackgroundLightOnly");
}

static void Furnace(const char* test, const MaterialPtr& material, uint32 numPasses, const Vector4& expected, float tolerance)
{
    Fixture f;
    const Vector4 lightColor(1.0f, 2.0f, 3.0f);
    auto backgroundLight = std::make_unique<BackgroundLight>(lightColor);
    auto lightObject = std::make_unique<LightSceneObject>(std::move(backgroundLight));
    f.mScene->AddObject(std::move(lightObject));

    ShapePtr shape = std::make_unique<SphereShape>(1.0f);
    ShapeSceneObjectPtr sceneObject = std::make_unique<ShapeSceneObject>(std::move(shape));
    sceneObject->SetDefaultMaterial(material);
    f.mScene->AddObject(std::move(sceneObject));
    f.mScene->BuildBVH();

    f.mViewport->Resize(ViewportSize, ViewportSize);
    Camera camera;
    camera.SetPerspective(1.0f, DegToRad(10.0f));
    camera.SetTransform(Transform(Vector4(0.0f, 0.0f, -3.0f)));

    RendererPtr renderer = CreateRenderer(gRendererName, *f.mScene);
    f.mViewport->SetRenderer(renderer);
    f.mViewport->Reset();
    for (uint32 i = 0; i < numPasses; ++i) f.mViewport->Render(camera);

    Bitmap bitmap = f.mViewport->GetSumBuffer();
    bitmap.Scale(Vector4(1.0f / numPasses));
    ValidateBitmap(bitmap, expected, tolerance, test);

    // the displayable image (Viewport::GetFrontBuffer): B8G8R8A8, uniform for a furnace scene up to dithering
    PostprocessParams post;
    post.ditheringStrength = 0.0f;
    post.tonemapper = Tonemapper::Clamped;
    if (!f.mViewport->SetPostprocessParams(post)) { printf("%s: SetPostprocessParams failed\n", test); gFailures++; }
    const Bitmap& front = f.mViewport->GetFrontBuffer();
    if (front.GetWidth() != ViewportSize || front.GetFormat() != Bitmap::Format::B8G8R8A8_UNorm) { printf("%s: bad front buffer\n", test); gFailures++; return; }
    const uint32* px = reinterpret_cast<const uint32*>(front.GetBytes());
    uint32 lo[3] = { 255, 255, 255 }, hi[3] = { 0, 0, 0 };
    for (uint32 i = 0; i < ViewportSize * ViewportSize; ++i)
        for (int c = 0; c < 3; ++c) { const uint32 v = (px[i] >> (8 * c)) & 255u; lo[c] = v < lo[c] ? v : lo[c]; hi[c] = v > hi[c] ? v : hi[c]; }
    for (int c = 0; c < 3; ++c)
        if (hi[c] - lo[c] > 40) { printf("%s: front buffer channel %d spreads %u..%u\n", test, c, lo[c], hi[c]); gFailures++; }
}

int main()
{
    if (!CreateRenderer(gRendererName, *std::make_unique<Scene>()))
    {
        printf("no GPU renderer available\n");
        return 2;
    }
    if (CreateRenderer("no such renderer", *std::make_unique<Scene>()) != nullptr) { printf("unknown names must give nullptr\n"); return 1; }

    for (const char* rendererName : gRendererNames)
    {
    gRendererName = rendererName;
    printf("---- %s\n", rendererName);
    EmptyScene();
    BackgroundLightOnly();

    const Vector4 lightColor(1.0f, 2.0f, 3.0f);
    {
        const Vector4 materialColor(0.4f, 0.6f, 0.8f);
        MaterialPtr material = std::make_unique<Material>();
        material->SetBsdf("diffuse");
        material->baseColor = materialColor;
        material->Compile();
        Furnace("FurnaceTest_Diffuse", material, 100, lightColor * materialColor, 0.05f);
    }
    {
        const Vector4 emissionColor(3.0f, 2.0f, 1.0f);
        MaterialPtr material = std::make_unique<Material>();
        material->SetBsdf("null");
        material->baseColor = Vector4::Zero();
        material->emission = emissionColor;
        material->Compile();
        Furnace("FurnaceTest_Emissive", material, 1, emissionColor, 0.0f);
    }
    {
        const Vector4 materialColor(0.4f, 0.6f, 0.8f);
        MaterialPtr material = std::make_unique<Material>();
        material->SetBsdf("metal");
        material->baseColor = materialColor;
        material->IoR = 0.0f;
        material->K = 100.0;
        material->Compile();
        Furnace("FurnaceTest_Metal", material, 20, lightColor * materialColor, 0.05f);
    }
    {
        MaterialPtr material = std::make_unique<Material>();
        material->SetBsdf("dielectric");
        material->baseColor = Vector4(1.0f);
        material->Compile();
        Furnace("FurnaceTest_Dielectric", material, 1000, lightColor, 0.075f);
    }
    }
    printf("%d test(s) failed\n", gFailures);
    return gFailures ? 1 : 0;
}
