#include "ObjReader.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits>
#include <map>
#include <sstream>

namespace helpers {
namespace obj {

namespace {

inline bool IsSpace(char x) { return x == ' ' || x == '\t'; }
inline bool IsDigit(char x) { return (unsigned int)(x - '0') < 10u; }
inline bool IsNewLine(char x) { return x == '\r' || x == '\n' || x == '\0'; }

// safeGetline, tiny_obj_loader.h:461-493: splits at \n, \r or \r\n
bool GetLine(const std::string& text, size_t& pos, std::string& line)
{
    line.clear();
    if (pos >= text.size()) return false;
    while (pos < text.size())
    {
        const char c = text[pos++];
        if (c == '\n') return true;
        if (c == '\r') { if (pos < text.size() && text[pos] == '\n') ++pos; return true; }
        line += c;
    }
    return true;
}

bool ReadFile(const std::string& path, std::string& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

float ParseReal(const char** token, double defaultValue = 0.0)   // parseReal, :692-700
{
    (*token) += strspn(*token, " \t");
    const char* end = (*token) + strcspn(*token, " \t\r");
    double val = defaultValue;
    TryParseDouble(*token, end, &val);
    (*token) = end;
    return (float)val;
}

std::string ParseString(const char** token)   // parseString, :524-531
{
    (*token) += strspn(*token, " \t");
    const size_t e = strcspn(*token, " \t\r");
    std::string s(*token, (*token) + e);
    (*token) += e;
    return s;
}

bool FixIndex(int idx, int n, int* ret)   // :501-522
{
    if (idx > 0) { *ret = idx - 1; return true; }
    if (idx == 0) return false;
    *ret = n + idx;
    return true;
}

struct VertexIndex { int v = -1, vt = -1, vn = -1; };

bool ParseTriple(const char** token, int vsize, int vnsize, int vtsize, VertexIndex* ret)   // :818-870
{
    VertexIndex vi;
    if (!FixIndex(atoi(*token), vsize, &vi.v)) return false;
    (*token) += strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') { *ret = vi; return true; }
    (*token)++;
    if ((*token)[0] == '/')   // i//k
    {
        (*token)++;
        if (!FixIndex(atoi(*token), vnsize, &vi.vn)) return false;
        (*token) += strcspn(*token, "/ \t\r");
        *ret = vi; return true;
    }
    if (!FixIndex(atoi(*token), vtsize, &vi.vt)) return false;   // i/j/k or i/j
    (*token) += strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') { *ret = vi; return true; }
    (*token)++;
    if (!FixIndex(atoi(*token), vnsize, &vi.vn)) return false;
    (*token) += strcspn(*token, "/ \t\r");
    *ret = vi;
    return true;
}

// pnpoly, tiny_obj_loader.h:1065-1077
int PnPoly(int nvert, const float* vertx, const float* verty, float testx, float testy)
{
    int i, j, c = 0;
    for (i = 0, j = nvert - 1; i < nvert; j = i++)
    {
        if (((verty[i] > testy) != (verty[j] > testy)) && (testx < (vertx[j] - vertx[i]) * (testy - verty[i]) / (verty[j] - verty[i]) + vertx[i])) c = !c;
    }
    return c;
}

void EmitTriangle(Model& out, const VertexIndex& a, const VertexIndex& b, const VertexIndex& c, int materialId)
{
    const VertexIndex* t[3] = { &a, &b, &c };
    for (int k = 0; k < 3; ++k)
    {
        Index idx; idx.vertex_index = t[k]->v; idx.normal_index = t[k]->vn; idx.texcoord_index = t[k]->vt;
        out.indices.push_back(idx);
    }
    out.material_ids.push_back(materialId);
}

// the triangulating branch of exportGroupsToShape, tiny_obj_loader.h:1107-1300
void TriangulateFace(Model& out, const std::vector<VertexIndex>& face, int materialId, const std::vector<float>& v)
{
    size_t npolys = face.size();
    if (npolys < 3) return;
    VertexIndex i0, i1, i2;
    size_t axes[2] = { 1, 2 };
    for (size_t k = 0; k < npolys; ++k)   // find the two axes to work in
    {
        i0 = face[(k + 0) % npolys]; i1 = face[(k + 1) % npolys]; i2 = face[(k + 2) % npolys];
        const size_t vi0 = (size_t)i0.v, vi1 = (size_t)i1.v, vi2 = (size_t)i2.v;
        if ((3 * vi0 + 2) >= v.size() || (3 * vi1 + 2) >= v.size() || (3 * vi2 + 2) >= v.size()) continue;
        const float v0x = v[vi0 * 3 + 0], v0y = v[vi0 * 3 + 1], v0z = v[vi0 * 3 + 2];
        const float v1x = v[vi1 * 3 + 0], v1y = v[vi1 * 3 + 1], v1z = v[vi1 * 3 + 2];
        const float v2x = v[vi2 * 3 + 0], v2y = v[vi2 * 3 + 1], v2z = v[vi2 * 3 + 2];
        const float e0x = v1x - v0x, e0y = v1y - v0y, e0z = v1z - v0z;
        const float e1x = v2x - v1x, e1y = v2y - v1y, e1z = v2z - v1z;
        const float cx = fabsf(e0y * e1z - e0z * e1y), cy = fabsf(e0z * e1x - e0x * e1z), cz = fabsf(e0x * e1y - e0y * e1x);
        const float epsilon = std::numeric_limits<float>::epsilon();
        if (cx > epsilon || cy > epsilon || cz > epsilon)
        {
            if (cx > cy && cx > cz) {}
            else
            {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    float area = 0;
    for (size_t k = 0; k < npolys; ++k)
    {
        i0 = face[(k + 0) % npolys]; i1 = face[(k + 1) % npolys];
        const size_t vi0 = (size_t)i0.v, vi1 = (size_t)i1.v;
        if ((vi0 * 3 + axes[0]) >= v.size() || (vi0 * 3 + axes[1]) >= v.size() || (vi1 * 3 + axes[0]) >= v.size() || (vi1 * 3 + axes[1]) >= v.size()) continue;
        const float v0x = v[vi0 * 3 + axes[0]], v0y = v[vi0 * 3 + axes[1]], v1x = v[vi1 * 3 + axes[0]], v1y = v[vi1 * 3 + axes[1]];
        area += (v0x * v1y - v0y * v1x) * 0.5f;
    }

    std::vector<VertexIndex> remaining = face;
    size_t guessVert = 0;
    VertexIndex ind[3];
    float vx[3], vy[3];
    size_t remainingIterations = face.size();
    size_t previousRemainingVertices = remaining.size();
    while (remaining.size() > 3 && remainingIterations > 0)
    {
        npolys = remaining.size();
        if (guessVert >= npolys) guessVert -= npolys;
        if (previousRemainingVertices != npolys) { previousRemainingVertices = npolys; remainingIterations = npolys; }
        else remainingIterations--;
        for (size_t k = 0; k < 3; k++)
        {
            ind[k] = remaining[(guessVert + k) % npolys];
            const size_t vi = (size_t)ind[k].v;
            if ((vi * 3 + axes[0]) >= v.size() || (vi * 3 + axes[1]) >= v.size()) { vx[k] = 0.0f; vy[k] = 0.0f; }
            else { vx[k] = v[vi * 3 + axes[0]]; vy[k] = v[vi * 3 + axes[1]]; }
        }
        const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
        const float cross = e0x * e1y - e0y * e1x;
        if (cross * area < 0.0f) { guessVert += 1; continue; }   // an internal angle
        bool overlap = false;
        for (size_t otherVert = 3; otherVert < npolys; ++otherVert)
        {
            const size_t idx = (guessVert + otherVert) % npolys;
            if (idx >= remaining.size()) continue;
            const size_t ovi = (size_t)remaining[idx].v;
            if ((ovi * 3 + axes[0]) >= v.size() || (ovi * 3 + axes[1]) >= v.size()) continue;
            const float tx = v[ovi * 3 + axes[0]], ty = v[ovi * 3 + axes[1]];
            if (PnPoly(3, vx, vy, tx, ty)) { overlap = true; break; }
        }
        if (overlap) { guessVert += 1; continue; }
        EmitTriangle(out, ind[0], ind[1], ind[2], materialId);   // this triangle is an ear
        size_t removed = (guessVert + 1) % npolys;               // remove v1 from the list
        while (removed + 1 < npolys) { remaining[removed] = remaining[removed + 1]; removed += 1; }
        remaining.pop_back();
    }
    if (remaining.size() == 3) EmitTriangle(out, remaining[0], remaining[1], remaining[2], materialId);
}

// ParseTextureNameAndOption, tiny_obj_loader.h:905-995 (the option values are parsed and dropped)
bool ParseTextureName(std::string* texname, const char* linebuf)
{
    bool found = false;
    std::string name;
    const char* token = linebuf;
    auto skipWord = [&]() { token += strspn(token, " \t"); token += strcspn(token, " \t\r"); };
    while (!IsNewLine(*token))
    {
        token += strspn(token, " \t");
        if ((0 == strncmp(token, "-blendu", 7)) && IsSpace(token[7])) { token += 8; skipWord(); }
        else if ((0 == strncmp(token, "-blendv", 7)) && IsSpace(token[7])) { token += 8; skipWord(); }
        else if ((0 == strncmp(token, "-clamp", 6)) && IsSpace(token[6])) { token += 7; skipWord(); }
        else if ((0 == strncmp(token, "-boost", 6)) && IsSpace(token[6])) { token += 7; ParseReal(&token, 1.0); }
        else if ((0 == strncmp(token, "-bm", 3)) && IsSpace(token[3])) { token += 4; ParseReal(&token, 1.0); }
        else if ((0 == strncmp(token, "-o", 2)) && IsSpace(token[2])) { token += 3; ParseReal(&token); ParseReal(&token); ParseReal(&token); }
        else if ((0 == strncmp(token, "-s", 2)) && IsSpace(token[2])) { token += 3; ParseReal(&token, 1.0); ParseReal(&token, 1.0); ParseReal(&token, 1.0); }
        else if ((0 == strncmp(token, "-t", 2)) && IsSpace(token[2])) { token += 3; ParseReal(&token); ParseReal(&token); ParseReal(&token); }
        else if ((0 == strncmp(token, "-type", 5)) && IsSpace(token[5])) { token += 5; skipWord(); }
        else if ((0 == strncmp(token, "-imfchan", 8)) && IsSpace(token[8])) { token += 9; skipWord(); }
        else if ((0 == strncmp(token, "-mm", 3)) && IsSpace(token[3])) { token += 4; ParseReal(&token, 0.0); ParseReal(&token, 1.0); }
        else if ((0 == strncmp(token, "-colorspace", 11)) && IsSpace(token[11])) { token += 12; ParseString(&token); }
        else { name = std::string(token); token += name.length(); found = true; }   // the rest of the line is the file name
    }
    if (found) *texname = name;
    return found;
}

// LoadMtl, tiny_obj_loader.h:1353-1790 (fields that reach LoadMaterial)
void LoadMtl(std::map<std::string, int>& materialMap, std::vector<Material>& materials, const std::string& text)
{
    Material material;
    size_t pos = 0;
    std::string linebuf;
    while (GetLine(text, pos, linebuf))
    {
        if (!linebuf.empty()) linebuf = linebuf.substr(0, linebuf.find_last_not_of(" \t") + 1);   // trim trailing whitespace
        if (linebuf.empty()) continue;
        const char* token = linebuf.c_str();
        token += strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#') continue;
        if ((0 == strncmp(token, "newmtl", 6)) && IsSpace(token[6]))
        {
            if (!material.name.empty())
            {
                materialMap.insert(std::pair<std::string, int>(material.name, (int)materials.size()));
                materials.push_back(material);
            }
            material = Material();
            token += 7;
            material.name = token;
            continue;
        }
        if (token[0] == 'K' && token[1] == 'd' && IsSpace(token[2]))
        {
            token += 2;
            material.diffuse[0] = ParseReal(&token); material.diffuse[1] = ParseReal(&token); material.diffuse[2] = ParseReal(&token);
            continue;
        }
        if (token[0] == 'K' && token[1] == 'e' && IsSpace(token[2]))
        {
            token += 2;
            material.emission[0] = ParseReal(&token); material.emission[1] = ParseReal(&token); material.emission[2] = ParseReal(&token);
            continue;
        }
        if ((0 == strncmp(token, "map_Kd", 6)) && IsSpace(token[6])) { token += 7; ParseTextureName(&material.diffuse_texname, token); continue; }
        if ((0 == strncmp(token, "map_d", 5)) && IsSpace(token[5])) { token += 6; material.alpha_texname = token; continue; }
        if ((0 == strncmp(token, "norm", 4)) && IsSpace(token[4])) { token += 5; ParseTextureName(&material.normal_texname, token); continue; }
    }
    materialMap.insert(std::pair<std::string, int>(material.name, (int)materials.size()));   // flush the last material
    materials.push_back(material);
}

} // namespace

// tryParseDouble, tiny_obj_loader.h:567-690
bool TryParseDouble(const char* s, const char* s_end, double* result)
{
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', expSign = '+';
    const char* curr = s;
    int read = 0;
    bool endNotReached = false;

    if (*curr == '+' || *curr == '-') { sign = *curr; curr++; }
    else if (IsDigit(*curr)) {}
    else return false;

    endNotReached = (curr != s_end);
    while (endNotReached && IsDigit(*curr))
    {
        mantissa *= 10;
        mantissa += (int)(*curr - 0x30);
        curr++; read++;
        endNotReached = (curr != s_end);
    }
    if (read == 0) return false;
    bool assemble = !endNotReached;
    if (!assemble)
    {
        if (*curr == '.')
        {
            curr++;
            read = 1;
            endNotReached = (curr != s_end);
            while (endNotReached && IsDigit(*curr))
            {
                static const double powLut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
                const int lutEntries = sizeof powLut / sizeof powLut[0];
                mantissa += (int)(*curr - 0x30) * (read < lutEntries ? powLut[read] : pow(10.0, -read));
                read++; curr++;
                endNotReached = (curr != s_end);
            }
        }
        else if (*curr == 'e' || *curr == 'E') {}
        else assemble = true;
    }
    if (!assemble && endNotReached && (*curr == 'e' || *curr == 'E'))
    {
        curr++;
        endNotReached = (curr != s_end);
        if (endNotReached && (*curr == '+' || *curr == '-')) { expSign = *curr; curr++; }
        else if (IsDigit(*curr)) {}
        else return false;   // empty E is not allowed
        read = 0;
        endNotReached = (curr != s_end);
        while (endNotReached && IsDigit(*curr))
        {
            exponent *= 10;
            exponent += (int)(*curr - 0x30);
            curr++; read++;
            endNotReached = (curr != s_end);
        }
        exponent *= (expSign == '+' ? 1 : -1);
        if (read == 0) return false;
    }
    *result = (sign == '+' ? 1 : -1) * (exponent ? ldexp(mantissa * pow(5.0, exponent), exponent) : mantissa);
    return true;
}

bool LoadObj(const std::string& path, const std::string& mtlBaseDir, Model& out, std::string& warn, std::string& err)
{
    out = Model();
    std::string text;
    if (!ReadFile(path, text)) { err += "Cannot open file [" + path + "]\n"; return false; }

    std::vector<float>& v = out.vertices; std::vector<float>& vn = out.normals; std::vector<float>& vt = out.texcoords;
    std::map<std::string, int> materialMap;
    int material = -1;
    std::vector<std::vector<VertexIndex>> faceGroup;
    auto exportGroup = [&]() { for (const auto& face : faceGroup) TriangulateFace(out, face, material, v); faceGroup.clear(); };

    size_t pos = 0, lineNum = 0;
    std::string linebuf;
    while (GetLine(text, pos, linebuf))
    {
        lineNum++;
        if (linebuf.empty()) continue;
        const char* token = linebuf.c_str();
        token += strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#') continue;

        if (token[0] == 'v' && IsSpace(token[1]))
        {
            token += 2;
            const float x = ParseReal(&token), y = ParseReal(&token), z = ParseReal(&token);
            v.push_back(x); v.push_back(y); v.push_back(z);
            continue;
        }
        if (token[0] == 'v' && token[1] == 'n' && IsSpace(token[2]))
        {
            token += 3;
            const float x = ParseReal(&token), y = ParseReal(&token), z = ParseReal(&token);
            vn.push_back(x); vn.push_back(y); vn.push_back(z);
            continue;
        }
        if (token[0] == 'v' && token[1] == 't' && IsSpace(token[2]))
        {
            token += 3;
            const float x = ParseReal(&token), y = ParseReal(&token);
            vt.push_back(x); vt.push_back(y);
            continue;
        }
        if (token[0] == 'f' && IsSpace(token[1]))
        {
            token += 2;
            token += strspn(token, " \t");
            std::vector<VertexIndex> face;
            while (!IsNewLine(token[0]))
            {
                VertexIndex vi;
                if (!ParseTriple(&token, (int)(v.size() / 3), (int)(vn.size() / 3), (int)(vt.size() / 2), &vi))
                {
                    std::stringstream ss;
                    ss << "Failed parse `f' line(e.g. zero value for face index. line " << lineNum << ".)\n";
                    err += ss.str();
                    return false;
                }
                face.push_back(vi);
                token += strspn(token, " \t\r");
            }
            faceGroup.push_back(face);
            continue;
        }
        if ((0 == strncmp(token, "usemtl", 6)) && IsSpace(token[6]))
        {
            token += 7;
            const std::string namebuf = token;
            int newMaterialId = -1;
            const auto it = materialMap.find(namebuf);
            if (it != materialMap.end()) newMaterialId = it->second;
            if (newMaterialId != material) { exportGroup(); material = newMaterialId; }
            continue;
        }
        if ((0 == strncmp(token, "mtllib", 6)) && IsSpace(token[6]))
        {
            token += 7;
            std::vector<std::string> filenames;   // SplitString(token, ' ')
            { std::stringstream ss(token); std::string item; while (std::getline(ss, item, ' ')) if (!item.empty()) filenames.push_back(item); }
            if (filenames.empty()) warn += "Looks like empty filename for mtllib. Use default material.\n";
            bool found = false;
            for (const std::string& name : filenames)
            {
                std::string mtlText;
                if (ReadFile(mtlBaseDir + name, mtlText)) { LoadMtl(materialMap, out.materials, mtlText); found = true; break; }
                warn += "Material file [ " + mtlBaseDir + name + " ] not found.\n";
            }
            if (!found && !filenames.empty()) warn += "Failed to load material file(s). Use default material.\n";
            continue;
        }
        if ((token[0] == 'g' || token[0] == 'o') && IsSpace(token[1])) { exportGroup(); continue; }   // a new shape: faces keep file order
        // l, t, s and anything else: not used by LoadMesh
    }
    exportGroup();
    return true;
}

} // namespace obj
} // namespace helpers
