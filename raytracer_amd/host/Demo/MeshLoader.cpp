// Mesh ingestion: OBJ -> de-duplicated vertex streams with Lengyel tangents -> MeshShape (Demo/MeshLoader.cpp:120-425).
// The arithmetic follows the reference's Vector4 semantics where it decides stored bits: dpps-ordered dot products,
// the fused cross product (Vector4ImplSSE.h:476-485), Normalize3 as a division of all four lanes by sqrt(dot3).
#include "MeshLoader.h"
#include "ObjReader.h"
#include "../Core/Textures/BitmapTexture.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <unordered_map>

namespace helpers {

using namespace rt;
using namespace rt::math;

namespace {

inline float Dot3(const Vector4& a, const Vector4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + 0.0f); }   // dpps 0x7F
inline Vector4 Cross3(const Vector4& a, const Vector4& b)   // r = a.yzx * b.zxy; fnmadd(a.zxy, b.yzx, r)
{
    const float rx = a.y * b.z, ry = a.z * b.x, rz = a.x * b.y, rw = a.w * b.w;
    return Vector4(fmaf(-a.z, b.y, rx), fmaf(-a.x, b.z, ry), fmaf(-a.y, b.x, rz), fmaf(-a.w, b.w, rw));
}
inline Vector4 Normalized3(const Vector4& a) { const float l = sqrtf(Dot3(a, a)); return Vector4(a.x / l, a.y / l, a.z / l, a.w / l); }
inline Vector4 Orthogonalize(const Vector4& v, const Vector4& ref)   // NegMulAndAdd(Dot3V(v, ref), ref, v), Vector4ImplSSE.h:587-591
{
    const float d = Dot3(v, ref);
    return Vector4(fmaf(-d, ref.x, v.x), fmaf(-d, ref.y, v.y), fmaf(-d, ref.z, v.z), fmaf(-d, ref.w, v.w));
}
inline float CopySign(float x, float y)
{
    uint32 xi, yi; memcpy(&xi, &x, 4); memcpy(&yi, &y, 4);
    xi = (0x7fffffffu & xi) | (0x80000000u & yi); memcpy(&x, &xi, 4); return x;
}
inline void BuildOrthonormalBasis(const Vector4& n, Vector4& u, Vector4& v)   // Core/Math/Geometry.cpp:15-32
{
    const float sign = CopySign(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    u = Vector4(1.0f + sign * n.x * n.x * a, sign * n.x * n.y * a, -sign * n.x);
    v = Vector4(n.x * n.y * a, sign + n.y * n.y * a, -n.y);
}
inline float TriangleSurfaceArea(const Vector4& edge0, const Vector4& edge1) { const Vector4 c = Cross3(edge1, edge0); return sqrtf(Dot3(c, c)) * 0.5f; }

struct IndexHash { size_t operator()(const obj::Index& k) const { return (size_t)(k.vertex_index ^ k.normal_index ^ k.texcoord_index); } };
struct IndexEq { bool operator()(const obj::Index& a, const obj::Index& b) const { return a.vertex_index == b.vertex_index && a.normal_index == b.normal_index && a.texcoord_index == b.texcoord_index; } };

// MeshLoader::ComputeTangentVectors, MeshLoader.cpp:274-369 (Lengyel's method)
void ComputeTangentVectors(MeshStreams& m)
{
    m.tangents.assign(m.normals.size(), Float3());
    std::vector<Vector4> bitangents(m.normals.size(), Vector4::Zero());
    const uint32 numTriangles = (uint32)(m.vertexIndices.size() / 3);
    for (uint32 i = 0; i < numTriangles; ++i)
    {
        const uint32 i0 = m.vertexIndices[3 * i + 0], i1 = m.vertexIndices[3 * i + 1], i2 = m.vertexIndices[3 * i + 2];
        const Vector4 p0(m.positions[i0]), p1(m.positions[i1]), p2(m.positions[i2]);
        const Vector4 e1 = p1 - p0, e2 = p2 - p0;
        const Float2& w0 = m.texCoords[i0]; const Float2& w1 = m.texCoords[i1]; const Float2& w2 = m.texCoords[i2];
        const float s1 = w1.x - w0.x, t1 = w1.y - w0.y, s2 = w2.x - w0.x, t2 = w2.y - w0.y;
        const float det = s1 * t2 - s2 * t1;
        if (fabsf(det) < 1.0e-10f) continue;
        const float r = 1.0f / det;
        const Vector4 sdir = (t2 * e1 - t1 * e2) * r;
        const Vector4 tdir = (s1 * e2 - s2 * e1) * r;
        for (uint32 k : { i0, i1, i2 })
        {
            m.tangents[k].x += sdir.x; m.tangents[k].y += sdir.y; m.tangents[k].z += sdir.z;
            bitangents[k] += tdir;
        }
    }
    for (size_t i = 0; i < m.positions.size(); ++i)
    {
        Vector4 tangent(m.tangents[i]);
        const Vector4 normal(m.normals[i]);
        Vector4 bitangent = bitangents[i];
        bool tangentIsValid = false;
        if (Dot3(tangent, tangent) > 0.1f)
        {
            tangent = Normalized3(tangent);
            const Vector4 c = Cross3(tangent, normal);
            if (Dot3(c, c) > 0.01f) { tangent = Orthogonalize(tangent, normal); tangentIsValid = true; }
        }
        if (!tangentIsValid) BuildOrthonormalBasis(normal, tangent, bitangent);
        tangent = Normalized3(tangent);
        m.tangents[i] = tangent.ToFloat3();
    }
}

// LoadMaterial, MeshLoader.cpp:78-94
MaterialPtr LoadMaterial(const std::string& baseDir, const obj::Material& source)
{
    MaterialPtr material = Material::Create();
    material->SetBsdf("diffuse");
    material->debugName = source.name;
    material->baseColor = Vector4(source.diffuse[0], source.diffuse[1], source.diffuse[2], 0.0f);
    material->emission.baseValue = Vector4(source.emission[0], source.emission[1], source.emission[2], 0.0f);
    material->baseColor.texture = LoadTexture(baseDir, source.diffuse_texname);
    material->normalMap = LoadTexture(baseDir, source.normal_texname);
    // maskMap (alpha_texname): Material::GetMaskValue has no caller in the reference
    material->roughness = 0.075f;
    material->Compile();
    return material;
}

} // namespace

BitmapPtr LoadBitmapObject(const std::string& baseDir, const std::string& path)   // MeshLoader.cpp:32-58
{
    if (path.empty()) return nullptr;
    std::string fullPath = baseDir + path;
    if (fullPath.length() >= 4 && ((fullPath.rfind(".png") == fullPath.length() - 4) || (fullPath.rfind(".jpg") == fullPath.length() - 4)))
        fullPath.replace(fullPath.length() - 4, 4, ".bmp");
    static std::map<std::string, BitmapPtr> bitmapsList;   // bitmaps are loaded only once
    BitmapPtr& bitmapPtr = bitmapsList[fullPath];
    if (!bitmapPtr)
    {
        bitmapPtr = BitmapPtr(new Bitmap(path.c_str()));
        if (!bitmapPtr->Load(fullPath.c_str())) return nullptr;
    }
    return bitmapPtr;
}

TexturePtr LoadTexture(const std::string& baseDir, const std::string& path)   // MeshLoader.cpp:60-76
{
    BitmapPtr bitmap = LoadBitmapObject(baseDir, path);
    if (!bitmap) return nullptr;
    if (bitmap->GetWidth() > 0 && bitmap->GetHeight() > 0) return std::make_shared<BitmapTexture>(bitmap);
    return nullptr;
}

MaterialPtr CreateDefaultMaterial(MaterialsMap& outMaterials)   // MeshLoader.cpp:96-109
{
    MaterialPtr material = Material::Create();
    material->debugName = "default";
    material->baseColor = Vector4(0.8f, 0.8f, 0.8f, 0.0f);
    material->emission.baseValue = Vector4(0.0f, 0.0f, 0.0f, 0.0f);
    material->roughness = 0.75f;
    material->SetBsdf(Material::DefaultBsdfName);
    material->Compile();
    outMaterials[material->debugName] = material;
    return material;
}

bool LoadMeshStreams(const std::string& filePath, MaterialsMap& outMaterials, const float scale, MeshStreams& m)   // MeshLoader::LoadMesh, :120-271
{
    m = MeshStreams();
    const float MinEdgeLength = 0.001f, MinEdgeLengthSqr = MinEdgeLength * MinEdgeLength;
    const std::string meshBaseDir = filePath.substr(0, filePath.find_last_of("\\/")) + "/";
    obj::Model model;
    std::string warning, err;
    const bool ret = obj::LoadObj(filePath, meshBaseDir, model, warning, err);
    if (!warning.empty()) fprintf(stderr, "[rt] WARNING: Mesh '%s' loading message:\n%s", filePath.c_str(), warning.c_str());
    if (!err.empty()) fprintf(stderr, "[rt] ERROR: Mesh '%s' loading message:\n%s", filePath.c_str(), err.c_str());
    if (!ret) { fprintf(stderr, "[rt] ERROR: Failed to load mesh '%s'\n", filePath.c_str()); return false; }

    std::unordered_map<obj::Index, uint32, IndexHash, IndexEq> uniqueIndices;
    const size_t numFaces = model.indices.size() / 3;
    for (size_t faceIndex = 0; faceIndex < numFaces; faceIndex++)
    {
        const obj::Index idx[3] = { model.indices[3 * faceIndex + 0], model.indices[3 * faceIndex + 1], model.indices[3 * faceIndex + 2] };
        // a face that names a vertex the file never defined is a load error; a normal / uv out of range counts as missing
        const int numVertices = (int)(model.vertices.size() / 3), numNormals = (int)(model.normals.size() / 3), numTexCoords = (int)(model.texcoords.size() / 2);
        bool hasNormals = true, hasTexCoords = true;
        for (size_t i = 0; i < 3; i++)
        {
            if (idx[i].vertex_index < 0 || idx[i].vertex_index >= numVertices)
            {
                fprintf(stderr, "[rt] ERROR: Mesh '%s': face %zu references vertex %d of %d\n", filePath.c_str(), faceIndex, idx[i].vertex_index + 1, numVertices);
                return false;
            }
            hasNormals = hasNormals && idx[i].normal_index >= 0 && idx[i].normal_index < numNormals;
            hasTexCoords = hasTexCoords && idx[i].texcoord_index >= 0 && idx[i].texcoord_index < numTexCoords;
        }
        Vector4 verts[3];
        for (size_t i = 0; i < 3; i++)
            verts[i] = scale * Vector4(model.vertices[3 * idx[i].vertex_index + 0], model.vertices[3 * idx[i].vertex_index + 1], model.vertices[3 * idx[i].vertex_index + 2]);
        // discard degenerate triangles
        const Vector4 edge1 = verts[1] - verts[0], edge2 = verts[2] - verts[0], edge3 = verts[2] - verts[1];
        if (Dot3(edge1, edge1) < MinEdgeLengthSqr || Dot3(edge2, edge2) < MinEdgeLengthSqr || Dot3(edge3, edge3) < MinEdgeLengthSqr ||
            TriangleSurfaceArea(edge1, edge2) < MinEdgeLengthSqr)
            continue;
        const Vector4 faceNormal = Normalized3(Cross3(verts[1] - verts[0], verts[2] - verts[0]));
        for (size_t i = 0; i < 3; i++)
        {
            const obj::Index indices = idx[i];
            const auto iter = uniqueIndices.find(indices);
            uint32 uniqueIndex = 0;
            if (iter != uniqueIndices.end()) uniqueIndex = iter->second;
            else
            {
                uniqueIndex = (uint32)uniqueIndices.size();
                uniqueIndices[indices] = uniqueIndex;
                m.positions.push_back(verts[i].ToFloat3());
                if (hasNormals)
                {
                    const Vector4 normal(model.normals[3 * indices.normal_index + 0], model.normals[3 * indices.normal_index + 1], model.normals[3 * indices.normal_index + 2]);
                    m.normals.push_back(Normalized3(normal).ToFloat3());
                }
                else m.normals.push_back(faceNormal.ToFloat3());   // fallback to the face normal
                if (hasTexCoords) m.texCoords.push_back(Float2(model.texcoords[2 * idx[i].texcoord_index], model.texcoords[2 * idx[i].texcoord_index + 1]));
                else m.texCoords.push_back(Float2());
            }
            m.vertexIndices.push_back(uniqueIndex);
        }
        m.materialIndices.push_back((uint32)model.material_ids[faceIndex]);
    }
    ComputeTangentVectors(m);

    m.materials.reserve(model.materials.size());
    for (const obj::Material& source : model.materials)
    {
        MaterialPtr material = LoadMaterial(meshBaseDir, source);
        m.materials.push_back(material);
        outMaterials[material->debugName] = material;
    }
    if (model.materials.empty())   // fallback to the default material
    {
        fprintf(stderr, "[rt] WARNING: No materials found in mesh '%s'. Falling back to the default material.\n", filePath.c_str());
        m.materials.push_back(CreateDefaultMaterial(outMaterials));
        for (uint32& index : m.materialIndices) index = 0;
    }
    return true;
}

MeshShapePtr LoadMesh(const std::string& filePath, MaterialsMap& outMaterials, const float scale)
{
    MeshStreams m;
    if (!LoadMeshStreams(filePath, outMaterials, scale, m)) return nullptr;
    MeshDesc meshDesc;   // MeshLoader::BuildMesh, MeshLoader.cpp:371-395
    meshDesc.path = filePath;
    meshDesc.vertexBufferDesc.numTriangles = (uint32)(m.vertexIndices.size() / 3);
    meshDesc.vertexBufferDesc.numVertices = (uint32)m.positions.size();
    meshDesc.vertexBufferDesc.numMaterials = (uint32)m.materials.size();
    meshDesc.vertexBufferDesc.materials = m.materials.data();
    meshDesc.vertexBufferDesc.materialIndexBuffer = m.materialIndices.data();
    meshDesc.vertexBufferDesc.vertexIndexBuffer = m.vertexIndices.data();
    meshDesc.vertexBufferDesc.positions = m.positions.data();
    meshDesc.vertexBufferDesc.normals = m.normals.data();
    meshDesc.vertexBufferDesc.tangents = m.tangents.data();
    meshDesc.vertexBufferDesc.texCoords = m.texCoords.data();
    MeshShapePtr mesh = std::make_shared<MeshShape>();
    if (!mesh->Initialize(meshDesc)) return nullptr;
    return mesh;
}

} // namespace helpers
