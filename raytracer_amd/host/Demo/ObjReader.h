// Wavefront OBJ / MTL reader for helpers::LoadMesh.
//
// The reference's Demo/MeshLoader.cpp:126-142 calls tinyobj::LoadObj (External/tiny_obj_loader.h, tinyobjloader
// v1.4.0, vendored in the reference tree) with triangulate = true.  That library is a third-party dependency; what
// follows is a restatement of the parts of its published algorithm that decide the numbers LoadMesh sees:
//   * number parsing (tryParseDouble, tiny_obj_loader.h:567-690: decimal digits accumulated in a double, exponent
//     applied as ldexp(mantissa * 5^e, e), then narrowed to float) -- NOT strtod, the last bit can differ;
//   * index parsing (parseTriple / fixIndex, :501-522, :818-870: 1-based, negative = relative to the count so far);
//   * face order = file order (exportGroupsToShape appends faces to the current shape, shapes are emitted in
//     order, faces with fewer than 3 vertices are dropped);
//   * polygon triangulation by ear clipping in the dominant-axis projection (:1107-1300);
//   * usemtl name -> material id through the map filled by LoadMtl (:1353-), -1 when unknown;
//   * MTL: newmtl, Kd, Ke, map_Kd, norm, map_d and the texture-option skipping of ParseTextureNameAndOption.
// Vertex colours, lines, tags, smoothing groups and the other MTL fields do not reach LoadMesh and are skipped.
#pragma once

#include <string>
#include <vector>

namespace helpers {
namespace obj {

struct Index { int vertex_index = -1, normal_index = -1, texcoord_index = -1; };   // tinyobj::index_t

struct Material   // the fields of tinyobj::material_t that LoadMaterial reads (Demo/MeshLoader.cpp:78-94)
{
    std::string name;
    float diffuse[3] = { 0.0f, 0.0f, 0.0f };
    float emission[3] = { 0.0f, 0.0f, 0.0f };
    std::string diffuse_texname, normal_texname, alpha_texname;
};

struct Model
{
    std::vector<float> vertices, normals, texcoords;   // attrib_t: xyz, xyz, uv
    std::vector<Index> indices;                        // 3 per triangle, all shapes concatenated in file order
    std::vector<int> material_ids;                     // per triangle
    std::vector<Material> materials;
};

// returns false (with a message in err) when the file cannot be read or a face line is malformed
bool LoadObj(const std::string& path, const std::string& mtlBaseDir, Model& out, std::string& warn, std::string& err);

// exposed for the tests
bool TryParseDouble(const char* s, const char* s_end, double* result);

} // namespace obj
} // namespace helpers
