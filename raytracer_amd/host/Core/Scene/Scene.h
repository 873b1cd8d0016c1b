// Scene: owns the objects, builds the top-level BVH and flattens everything into the RtSceneDesc the
// device library uploads (reference: Core/Scene/Scene.h, Scene::BuildBVH Core/Scene/Scene.cpp:36-126).
#pragma once

#include "Object/SceneObject.h"
#include "../BVH/BVH.h"
#include "../../../../include/rtgpu.h"

namespace rt {

class RAYLIB_API Scene
{
public:
    Scene();
    ~Scene();
    Scene(Scene&&);
    Scene& operator=(Scene&&);

    void AddObject(SceneObjectPtr object);
    bool BuildBVH();

    uint32 GetNumObjects() const { return (uint32)mAllObjects.size(); }
    const std::vector<const LightSceneObject*>& GetLights() const { return mLights; }
    const std::vector<const LightSceneObject*>& GetGlobalLights() const { return mGlobalLights; }
    const BVH& GetBVH() const { return mTraceableObjectsBVH; }

    // Flat, pointer-stable description of the scene as of the last BuildBVH().  Valid until the next
    // BuildBVH() or destruction.  blueNoise is filled in by the renderer (it is sampler data, not scene data).
    const RtSceneDesc& GetDesc() const { return mDesc; }
    uint64 GetBuildId() const { return mBuildId; }

private:
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    bool Flatten();

    std::vector<SceneObjectPtr> mAllObjects;
    std::vector<const ISceneObject*> mTraceableObjects;   // BVH leaf order after BuildBVH
    std::vector<const LightSceneObject*> mLights;
    std::vector<const LightSceneObject*> mGlobalLights;
    BVH mTraceableObjectsBVH;

    // flattened storage backing mDesc
    std::vector<RtNode> mFlatTopNodes, mFlatMeshNodes;
    std::vector<RtObject> mFlatObjects;
    std::vector<RtLight> mFlatLights;
    std::vector<uint32> mFlatGlobalLights;
    std::vector<RtMaterial> mFlatMaterials;
    std::vector<RtMesh> mFlatMeshes;
    std::vector<RtTriangle> mFlatTriangles;
    std::vector<RtVertexIndices> mFlatVertexIndices;
    std::vector<RtVertexShading> mFlatVertexShading;
    std::vector<RtTexture> mFlatTextures;
    std::vector<uint8> mFlatTexels;
    RtSceneDesc mDesc;
    uint64 mBuildId = 0;
};

} // namespace rt
