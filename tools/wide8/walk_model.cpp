// walk_model.cpp -- CPU step model of the traversal formats studied in round 6 (review item 1: "octant-ordered 8-wide 64-byte node, step model FIRST").
//
// Reads what tools/wide8/dump_walk_inputs.py wrote (the benchmark mesh's binary tree = the reference's BVH::Node array, its triangles in leaf order, a sample of
// the rays a depth-8 PathTracerMIS pass traces) and walks every ray through
//   W4   today's tree: 4-wide collapse (rt_trace_wide.inl buildWideBvh), boxes on the 16-bit grid, children visited in order of their entry distance (the
//        kernel's five-exchange sort);
//   W8   the 8-wide collapse of the same binary tree (same SAH splits, same leaves), in four flavours: planes on the 16-bit grid or as 8-bit offsets in the
//        node's own frame (origin on the grid + one power-of-two scale per axis: the 64-byte node of DESIGN 8), children visited by entry distance (a full
//        sort: the cost the format cannot afford) or in the order of the ray's octant (slot ^ octant ascending, children assigned to slots at build time by
//        their position in the node: no keys, no sort);
//   W4o  the 4-wide tree with octant order (what dropping the sort alone would cost).
// and prints, per ray kind, interior visits (= node fetches = dependent round trips), leaf visits, triangle tests, box tests, children entered per visit and the
// depth the deferred-work stack reaches (one entry per deferred CHILD for W4, one per deferred NODE GROUP for W8).
// The arithmetic is plain float slab / Moeller-Trumbore tests: this is a model of visit COUNTS, not of the kernel's exactness rules (leaf gates, runner-up
// hand-over), which do not change what is visited.
//
//   g++ -O2 -std=c++17 tools/wide8/walk_model.cpp -o /tmp/walk_model/walk_model && /tmp/walk_model/walk_model /tmp/walk_model
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct Node { float mn[3]; uint32_t child; float mx[3]; uint32_t leaves; };
struct Tri { float v0[3], e1[3], e2[3]; };
struct RayIn { float o[3], d[3], dist, bounce, p[3], n[3]; };
struct Ray { float o[3], d[3], inv[3], tmax; bool anyHit; uint32_t oct; };

static int gCullBits = 0;   // > 0: the deferred distance is kept as the top gCullBits bits of the float below its sign (a lower bound)
static bool gCullOnPop = false;   // the deferred entry carries its entry distance and is dropped at the pop if a hit found meanwhile lies in front of it
static std::vector<Node> gNodes;
static std::vector<Tri> gTris;

static inline bool isLeaf(uint32_t n) { return (gNodes[n].leaves & 0x3FFFFFFFu) != 0u; }
static inline double area(uint32_t n)
{
    const double ex = (double)gNodes[n].mx[0] - gNodes[n].mn[0], ey = (double)gNodes[n].mx[1] - gNodes[n].mn[1], ez = (double)gNodes[n].mx[2] - gNodes[n].mn[2];
    return ex * ey + ey * ez + ez * ex;
}

// ---- topology optimisation of the binary tree ABOVE its leaves (the leaves -- exact boxes, triangle runs -- stay the reference's): tree rotations that lower the
// surface-area cost, bottom-up, a few sweeps (Kensler 2008).  Any tree over the same leaves is a legal walk for the 4-wide kernel (its boxes are conservative and
// the leaf gates are the reference's), so this is purely a visit-count question.
struct BN { float mn[3], mx[3]; int l, r; uint32_t child, leaves; };
static inline double areaOf(const float mn[3], const float mx[3]) { const double ex = (double)mx[0] - mn[0], ey = (double)mx[1] - mn[1], ez = (double)mx[2] - mn[2]; return ex * ey + ey * ez + ez * ex; }
static void refit(std::vector<BN>& t, int n) { for (int a = 0; a < 3; ++a) { t[n].mn[a] = fminf(t[t[n].l].mn[a], t[t[n].r].mn[a]); t[n].mx[a] = fmaxf(t[t[n].l].mx[a], t[t[n].r].mx[a]); } }
static double unionArea(const BN& a, const BN& b) { float mn[3], mx[3]; for (int k = 0; k < 3; ++k) { mn[k] = fminf(a.mn[k], b.mn[k]); mx[k] = fmaxf(a.mx[k], b.mx[k]); } return areaOf(mn, mx); }
static void optimiseTopology(int sweeps)
{
    std::vector<BN> t(gNodes.size());
    for (size_t n = 0; n < gNodes.size(); ++n)
    {
        if (n == 1) continue;
        BN& b = t[n]; memcpy(b.mn, gNodes[n].mn, 12); memcpy(b.mx, gNodes[n].mx, 12); b.child = gNodes[n].child; b.leaves = gNodes[n].leaves;
        if (isLeaf((uint32_t)n)) b.l = b.r = -1; else { b.l = (int)gNodes[n].child; b.r = (int)gNodes[n].child + 1; }
    }
    auto cost = [&]() { double c = 0; std::vector<int> st; st.push_back(0); while (!st.empty()) { int n = st.back(); st.pop_back(); if (t[n].l < 0) continue; c += areaOf(t[n].mn, t[n].mx); st.push_back(t[n].l); st.push_back(t[n].r); } return c; };
    const double before = cost();
    long rotations = 0;
    for (int sweep = 0; sweep < sweeps; ++sweep)
    {
        // post-order
        std::vector<int> order, st; st.push_back(0);
        while (!st.empty()) { int n = st.back(); st.pop_back(); if (t[n].l < 0) continue; order.push_back(n); st.push_back(t[n].l); st.push_back(t[n].r); }
        for (size_t i = order.size(); i-- > 0;)
        {
            const int n = order[i];
            refit(t, n);
            // best of the four rotations: a child swapped with a grandchild of the other side
            double bestGain = 1e-12; int bestSide = -1, bestGrand = -1;
            for (int side = 0; side < 2; ++side)
            {
                const int c = side ? t[n].r : t[n].l, o = side ? t[n].l : t[n].r;   // c: the child whose box changes (must be interior), o: the other child
                if (t[c].l < 0) continue;
                const double old = areaOf(t[c].mn, t[c].mx);
                const double g0 = old - unionArea(t[o], t[t[c].r]);   // swap o with c.l: c = {o, c.r}
                const double g1 = old - unionArea(t[t[c].l], t[o]);   // swap o with c.r: c = {c.l, o}
                if (g0 > bestGain) { bestGain = g0; bestSide = side; bestGrand = 0; }
                if (g1 > bestGain) { bestGain = g1; bestSide = side; bestGrand = 1; }
            }
            if (bestSide >= 0)
            {
                const int c = bestSide ? t[n].r : t[n].l; int& oRef = bestSide ? t[n].l : t[n].r; int& gRef = bestGrand ? t[c].r : t[c].l;
                std::swap(oRef, gRef); refit(t, c); refit(t, n); rotations++;
            }
        }
    }
    const double after = cost();
    // back into the array form (child pairs adjacent, breadth first)
    std::vector<Node> out; out.resize(2); std::vector<std::pair<int, uint32_t>> todo; todo.push_back({ 0, 0u });
    for (size_t i = 0; i < todo.size(); ++i)
    {
        const int n = todo[i].first; const uint32_t at = todo[i].second;
        Node nd; memcpy(nd.mn, t[n].mn, 12); memcpy(nd.mx, t[n].mx, 12);
        if (t[n].l < 0) { nd.child = t[n].child; nd.leaves = t[n].leaves; }
        else { nd.child = (uint32_t)out.size(); nd.leaves = 0u; out.resize(out.size() + 2); todo.push_back({ t[n].l, nd.child }); todo.push_back({ t[n].r, nd.child + 1u }); }
        out[at] = nd;
    }
    printf("topology optimisation: %d sweeps, %ld rotations, surface-area cost of the interior nodes %.4g -> %.4g (%.3f)\n", sweeps, rotations, before, after, after / before);
    gNodes = out;
}

// ---- the 16-bit grid of rt_wide_grid.inl (conservative, a step to spare) ----
static float gBase[3], gStep[3];
static void buildGrid()
{
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t n = 0; n < gNodes.size(); ++n) { if (n == 1) continue; for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], gNodes[n].mn[a]); hi[a] = fmaxf(hi[a], gNodes[n].mx[a]); } }
    const float largest = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    for (int a = 0; a < 3; ++a) { gStep[a] = fmaxf(hi[a] - lo[a], 1e-6f * largest) / (65535.0f - 8.0f); gBase[a] = lo[a] - 4.0f * gStep[a]; }
}
static inline int qmin(int a, float x) { long v = (long)floor(((double)x - gBase[a]) / gStep[a]) - 1; return (int)(v < 0 ? 0 : v); }
static inline int qmax(int a, float x) { long v = (long)ceil(((double)x - gBase[a]) / gStep[a]) + 1; return (int)(v > 65535 ? 65535 : v); }

// ---- a wide tree over the binary one ----
struct WChild { uint32_t ref; bool leaf; bool valid; int q[6]; float lo[3], hi[3]; };   // ref: wide node index or binary leaf node; q: planes on the 16-bit grid; lo / hi: the planes the walk tests
struct WNode { WChild c[8]; int num; };
struct WTree { std::vector<WNode> nodes; int width; };

static void setPlanes16(WChild& c) { for (int a = 0; a < 3; ++a) { c.lo[a] = gBase[a] + c.q[a] * gStep[a]; c.hi[a] = gBase[a] + c.q[3 + a] * gStep[a]; } }

static WTree collapse(int width, bool octantSlots, bool planes8)
{
    WTree t; t.width = width;
    std::vector<uint32_t> order; order.push_back(0u);
    std::vector<uint32_t> wideOf(gNodes.size(), 0xFFFFFFFFu); wideOf[0] = 0;
    for (size_t i = 0; i < order.size(); ++i)
    {
        const uint32_t n = order[i];
        uint32_t list[8]; int num = 2;
        list[0] = gNodes[n].child; list[1] = gNodes[n].child + 1u;
        while (num < width)
        {
            int pick = -1; double best = -1.0;
            for (int k = 0; k < num; ++k) if (!isLeaf(list[k]) && area(list[k]) > best) { best = area(list[k]); pick = k; }
            if (pick < 0) break;
            const uint32_t c = gNodes[list[pick]].child;
            list[pick] = c; list[num++] = c + 1u;
        }
        WNode w; w.num = num;
        for (int k = 0; k < 8; ++k) w.c[k].valid = false;
        // slot assignment
        int slotOf[8];
        if (!octantSlots) for (int k = 0; k < num; ++k) slotOf[k] = k;
        else
        {
            // a child's slot says where it lies in the node: bit a of the slot set = towards +axis a.  Greedy assignment by the projection of the child's
            // centre (relative to the node's) on the slot's diagonal, as the compressed-wide-BVH literature does
            float cen[3];
            for (int a = 0; a < 3; ++a) cen[a] = 0.5f * (gNodes[n].mn[a] + gNodes[n].mx[a]);
            if (n == 0) for (int a = 0; a < 3; ++a) { float lo = INFINITY, hi = -INFINITY; for (int k = 0; k < num; ++k) { lo = fminf(lo, gNodes[list[k]].mn[a]); hi = fmaxf(hi, gNodes[list[k]].mx[a]); } cen[a] = 0.5f * (lo + hi); }
            bool usedC[8] = { false }, usedS[8] = { false };
            for (int round = 0; round < num; ++round)
            {
                int bc = -1, bs = -1; float bestCost = -INFINITY;
                for (int k = 0; k < num; ++k) if (!usedC[k])
                    for (int s = 0; s < 8; ++s) if (!usedS[s])
                    {
                        float cost = 0.0f;
                        for (int a = 0; a < 3; ++a) { const float d = 0.5f * (gNodes[list[k]].mn[a] + gNodes[list[k]].mx[a]) - cen[a]; cost += ((s >> a) & 1) ? d : -d; }
                        if (cost > bestCost) { bestCost = cost; bc = k; bs = s; }
                    }
                usedC[bc] = true; usedS[bs] = true; slotOf[bc] = bs;
            }
        }
        for (int k = 0; k < num; ++k)
        {
            WChild& c = w.c[slotOf[k]];
            c.valid = true; c.leaf = isLeaf(list[k]);
            for (int a = 0; a < 3; ++a) { c.q[a] = qmin(a, gNodes[list[k]].mn[a]); c.q[3 + a] = qmax(a, gNodes[list[k]].mx[a]); }
            if (c.leaf) c.ref = list[k];
            else { wideOf[list[k]] = (uint32_t)order.size(); c.ref = (uint32_t)order.size(); order.push_back(list[k]); }
            setPlanes16(c);
        }
        if (planes8)
        {
            // the node's frame: origin = the children's common minimum on the grid, one power-of-two scale per axis so that the extent fits 255 steps
            for (int a = 0; a < 3; ++a)
            {
                int lo = 65535, hi = 0;
                for (int k = 0; k < 8; ++k) if (w.c[k].valid) { lo = std::min(lo, w.c[k].q[a]); hi = std::max(hi, w.c[k].q[3 + a]); }
                int e = 0; while (((hi - lo) + (1 << e) - 1) >> e > 255) ++e;
                for (int k = 0; k < 8; ++k) if (w.c[k].valid)
                {
                    const int ql = (w.c[k].q[a] - lo) >> e, qh = ((w.c[k].q[3 + a] - lo) + (1 << e) - 1) >> e;
                    w.c[k].lo[a] = gBase[a] + (lo + (ql << e)) * gStep[a]; w.c[k].hi[a] = gBase[a] + (lo + (qh << e)) * gStep[a];
                }
            }
        }
        t.nodes.push_back(w);
    }
    return t;
}

// ---- tests ----
static inline bool slab(const Ray& r, const float lo[3], const float hi[3], float& nearOut, float& farOut)
{
    float tn = 0.0f, tf = INFINITY;
    for (int a = 0; a < 3; ++a)
    {
        const float t0 = (lo[a] - r.o[a]) * r.inv[a], t1 = (hi[a] - r.o[a]) * r.inv[a];
        tn = fmaxf(tn, fminf(t0, t1)); tf = fminf(tf, fmaxf(t0, t1));
    }
    nearOut = tn; farOut = tf;
    return tf >= tn;
}
static inline bool triHit(const Ray& r, const Tri& t, float& dist)
{
    const float* d = r.d;
    const float px = d[1] * t.e2[2] - d[2] * t.e2[1], py = d[2] * t.e2[0] - d[0] * t.e2[2], pz = d[0] * t.e2[1] - d[1] * t.e2[0];
    const float det = t.e1[0] * px + t.e1[1] * py + t.e1[2] * pz;
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    const float tx = r.o[0] - t.v0[0], ty = r.o[1] - t.v0[1], tz = r.o[2] - t.v0[2];
    const float u = (tx * px + ty * py + tz * pz) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    const float qx = ty * t.e1[2] - tz * t.e1[1], qy = tz * t.e1[0] - tx * t.e1[2], qz = tx * t.e1[1] - ty * t.e1[0];
    const float v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    dist = (t.e2[0] * qx + t.e2[1] * qy + t.e2[2] * qz) * inv;
    return dist > 0.0f;
}

static inline float cullKey(float nearD) { if (gCullBits <= 0) return nearD; uint32_t u; memcpy(&u, &nearD, 4); u &= ~((1u << (31 - gCullBits)) - 1u); float f; memcpy(&f, &u, 4); return f; }
struct Stats
{
    double rays = 0, interior = 0, leaves = 0, triTests = 0, boxTests = 0, entered = 0, maxStack = 0, stackOver[4] = { 0, 0, 0, 0 }, hits = 0, distMismatch = 0;
    void add(const Stats& s) { rays += s.rays; interior += s.interior; leaves += s.leaves; triTests += s.triTests; boxTests += s.boxTests; entered += s.entered; maxStack = std::max(maxStack, s.maxStack); for (int i = 0; i < 4; ++i) stackOver[i] += s.stackOver[i]; hits += s.hits; distMismatch += s.distMismatch; }
};

// One ray through a wide tree.  order: 0 = by entry distance (nearest first, the rest deferred farthest first), 1 = octant order (slot ^ octant ascending).
// groupStack: deferred work is counted in node groups (a node with children still to visit = one entry) instead of single children.
static float walk(const WTree& t, const Ray& r, int order, bool groupStack, Stats& st)
{
    struct Entry { uint32_t ref; bool leaf; float nearD; float farD; };
    std::vector<Entry> stack; stack.reserve(64);
    std::vector<int> groupSizes;   // groupStack: children still deferred per group (for the depth statistic)
    float best = r.tmax;
    Entry cur = { 0u, false, 0.0f, 0.0f };
    bool have = true;
    size_t deepest = 0;
    while (have)
    {
        if (!cur.leaf)
        {
            st.interior++;
            const WNode& n = t.nodes[cur.ref];
            Entry hit[8]; int numHit = 0; int slotKey[8];
            for (int k = 0; k < 8; ++k)
            {
                if (!n.c[k].valid) continue;
                st.boxTests++;
                float nd, fd;
                if (slab(r, n.c[k].lo, n.c[k].hi, nd, fd) && nd < best) { hit[numHit] = { n.c[k].ref, n.c[k].leaf, nd, fminf(fd, best) }; slotKey[numHit] = k ^ (int)r.oct; numHit++; }
            }
            st.entered += numHit;
            // order: the LAST element is visited next
            if (order == 0) { for (int i = 1; i < numHit; ++i) for (int j = i; j > 0 && hit[j - 1].nearD < hit[j].nearD; --j) { std::swap(hit[j - 1], hit[j]); } }
            else if (order == 2) { std::reverse(hit, hit + numHit); }   // storage order (slot 0 first)
            else if (order == 6) { for (int i = 1; i < numHit; ++i) for (int j = i; j > 0 && hit[j - 1].nearD > hit[j].nearD; --j) { std::swap(hit[j - 1], hit[j]); } }   // FARTHEST first
            else if (order == 7) { for (int i = 1; i < numHit; ++i) for (int j = i; j > 0 && (hit[j - 1].farD - hit[j - 1].nearD) > (hit[j].farD - hit[j].nearD); --j) { std::swap(hit[j - 1], hit[j]); } }   // LONGEST chord first
            else if (order == 8) { for (int i = 1; i < numHit; ++i) for (int j = i; j > 0 && hit[j - 1].farD > hit[j].farD; --j) { std::swap(hit[j - 1], hit[j]); } }   // largest EXIT distance first
            else if (order == 9) { for (int i = 1; i < numHit; ++i) for (int j = i; j > 0 && slotKey[j - 1] > slotKey[j]; --j) { std::swap(hit[j - 1], hit[j]); std::swap(slotKey[j - 1], slotKey[j]); } }   // REVERSE octant order (the slot farthest along the ray first)
            else if (order == 4) { }                                    // reverse storage order (the highest slot first)
            else if (order == 5) { static uint32_t lc = 12345u; for (int i = numHit - 1; i > 0; --i) { lc = lc * 1664525u + 1013904223u; std::swap(hit[i], hit[(lc >> 16) % (uint32_t)(i + 1)]); } }   // random order
            else if (order == 3)
            {
                // largest box first (any-hit rays: the child most likely to hold an occluder) -- keys ride in nearD
                for (int i = 0; i < numHit; ++i) for (int k = 0; k < 8; ++k) if (n.c[k].valid && n.c[k].ref == hit[i].ref && n.c[k].leaf == hit[i].leaf)
                { const float ex = n.c[k].hi[0] - n.c[k].lo[0], ey = n.c[k].hi[1] - n.c[k].lo[1], ez = n.c[k].hi[2] - n.c[k].lo[2]; hit[i].nearD = ex * ey + ey * ez + ez * ex; }
                for (int i = 1; i < numHit; ++i) for (int j = i; j > 0 && hit[j - 1].nearD > hit[j].nearD; --j) { std::swap(hit[j - 1], hit[j]); }
            }
            else { for (int i = 1; i < numHit; ++i) for (int j = i; j > 0 && slotKey[j - 1] < slotKey[j]; --j) { std::swap(hit[j - 1], hit[j]); std::swap(slotKey[j - 1], slotKey[j]); } }
            if (numHit == 0)
            {
                if (stack.empty()) have = false;
                else { cur = stack.back(); stack.pop_back(); if (groupStack) { if (--groupSizes.back() == 0) groupSizes.pop_back(); } while (gCullOnPop && !r.anyHit && cullKey(cur.nearD) >= best) { if (stack.empty()) { have = false; break; } cur = stack.back(); stack.pop_back(); } }
            }
            else
            {
                for (int i = 0; i + 1 < numHit; ++i) stack.push_back(hit[i]);
                if (groupStack && numHit > 1) groupSizes.push_back(numHit - 1);
                cur = hit[numHit - 1];
            }
            deepest = std::max(deepest, groupStack ? groupSizes.size() : stack.size());
        }
        else
        {
            // a deferred child may have been passed by a hit found meanwhile: the kernel tests the leaf's triangles regardless (it keeps no entry distance),
            // so does the model
            st.leaves++;
            const Node& leaf = gNodes[cur.ref];
            const uint32_t count = leaf.leaves & 0x3FFFFFFFu;
            for (uint32_t i = 0; i < count; ++i)
            {
                st.triTests++;
                float d;
                if (triHit(r, gTris[leaf.child + i], d) && d < best) { best = d; if (r.anyHit) { have = false; } }
            }
            if (have)
            {
                if (stack.empty()) have = false;
                else { cur = stack.back(); stack.pop_back(); if (groupStack) { if (--groupSizes.back() == 0) groupSizes.pop_back(); } while (gCullOnPop && !r.anyHit && cullKey(cur.nearD) >= best) { if (stack.empty()) { have = false; break; } cur = stack.back(); stack.pop_back(); } }
            }
        }
    }
    st.rays++;
    st.maxStack = std::max(st.maxStack, (double)deepest);
    const size_t limits[4] = { 6, 8, 10, 13 };
    for (int i = 0; i < 4; ++i) if (deepest > limits[i]) st.stackOver[i]++;
    if (best < r.tmax) st.hits++;
    return best;
}

static Ray makeRay(const float o[3], const float d[3], float offset, float tmax, bool anyHit)
{
    Ray r;
    const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    r.oct = 0;
    for (int a = 0; a < 3; ++a) { r.d[a] = d[a] / len; r.o[a] = o[a] + r.d[a] * offset; r.inv[a] = 1.0f / r.d[a]; if (r.d[a] < 0.0f) r.oct |= 1u << a; }
    r.tmax = tmax; r.anyHit = anyHit;
    return r;
}

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/tmp/walk_model";
    gCullOnPop = argc > 2 && atoi(argv[2]) != 0; gCullBits = argc > 2 && atoi(argv[2]) > 1 ? atoi(argv[2]) : 0;
    const int topologySweeps = argc > 3 ? atoi(argv[3]) : 0;
    size_t numNodes = 0, numTris = 0, numRays = 0; float sun[3];
    { FILE* f = fopen((dir + "/meta.txt").c_str(), "r"); if (!f || fscanf(f, "%zu %zu %zu %f %f %f", &numNodes, &numTris, &numRays, &sun[0], &sun[1], &sun[2]) != 6) { fprintf(stderr, "no meta.txt in %s\n", dir.c_str()); return 1; } fclose(f); }
    gNodes.resize(numNodes); gTris.resize(numTris);
    std::vector<RayIn> in(numRays);
    auto slurp = [&](const char* name, void* dst, size_t bytes) { FILE* f = fopen((dir + "/" + name).c_str(), "rb"); if (!f || fread(dst, 1, bytes, f) != bytes) { fprintf(stderr, "bad %s\n", name); exit(1); } fclose(f); };
    slurp("nodes.bin", gNodes.data(), numNodes * sizeof(Node)); slurp("tris.bin", gTris.data(), numTris * sizeof(Tri)); slurp("rays.bin", in.data(), numRays * sizeof(RayIn));
    if (topologySweeps > 0) optimiseTopology(topologySweeps);
    buildGrid();

    // the rays: every recorded path segment as a closest-hit ray; from every vertex that hit something a next-event ray -- towards the sun (a 1-degree cone: its axis)
    // or over the hemisphere of the shading normal (the background light), the light picked like LightSamplingStrategy::Single does, skipped when it points into the surface
    std::vector<Ray> closest, shadow;
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (float)((lcg >> 40) & 0xFFFFFF) / 16777216.0f; };
    for (const RayIn& r : in)
    {
        closest.push_back(makeRay(r.o, r.d, r.bounce > 0.0f ? 0.001f : 0.0f, INFINITY, false));
        if (!(r.dist < 1e30f)) continue;
        float l[3];
        if (rnd() < 0.5f) { l[0] = sun[0]; l[1] = sun[1]; l[2] = sun[2]; }
        else { do { l[0] = 2 * rnd() - 1; l[1] = 2 * rnd() - 1; l[2] = 2 * rnd() - 1; } while (l[0] * l[0] + l[1] * l[1] + l[2] * l[2] > 1.0f || l[0] * l[0] + l[1] * l[1] + l[2] * l[2] < 1e-4f); }
        if (l[0] * r.n[0] + l[1] * r.n[1] + l[2] * r.n[2] <= 0.0f) continue;
        shadow.push_back(makeRay(r.p, l, 0.0001f, INFINITY, true));
    }
    printf("binary tree: %zu nodes, %zu triangles; rays: %zu closest-hit, %zu any-hit (%.2f per closest)\n", numNodes, numTris, closest.size(), shadow.size(), (double)shadow.size() / closest.size());

    struct Variant { const char* name; int width; bool octSlots, planes8; int order; bool group; };
    const Variant variants[] = {
        { "W4  16-bit planes, distance order (round 5)", 4, false, false, 0, false },
        { "W4o 16-bit planes, octant order", 4, true, false, 1, false },
        { "W8  16-bit planes, distance order", 8, false, false, 0, true },
        { "W8  16-bit planes, octant order", 8, true, false, 1, true },
        { "W8  8-bit planes, distance order", 8, false, true, 0, true },
        { "W8  8-bit planes, octant order (the 64-byte node)", 8, true, true, 1, true },
        { "W6  8-bit planes, octant order (header + 6 x 8 B)", 6, true, true, 1, true },
        { "W6  8-bit planes, distance order", 6, false, true, 0, true },
        { "W4  16-bit planes, STORAGE order (no sort at all; any-hit rays only are meaningful)", 4, false, false, 2, false },
        { "W4  16-bit planes, LARGEST child first (any-hit rays only are meaningful)", 4, false, false, 3, false },
        { "W4  16-bit planes, REVERSE storage order", 4, false, false, 4, false },
        { "W4  16-bit planes, RANDOM order", 4, false, false, 5, false },
        { "W4  16-bit planes, FARTHEST child first", 4, false, false, 6, false },
        { "W4  16-bit planes, LONGEST chord first", 4, false, false, 7, false },
        { "W4  16-bit planes, largest EXIT distance first", 4, false, false, 8, false },
        { "W8  8-bit planes, FARTHEST child first", 8, false, true, 6, true },
        { "W8  8-bit planes, REVERSE storage order", 8, false, true, 4, true },
        { "W8  8-bit planes, REVERSE octant order (any-hit rays only are meaningful)", 8, true, true, 9, true },
        { "W4  16-bit planes, REVERSE octant order (any-hit rays only are meaningful)", 4, true, false, 9, false },
    };
    double base[2] = { 0, 0 };
    for (const Variant& v : variants)
    {
        const WTree t = collapse(v.width, v.octSlots, v.planes8);
        double children = 0; for (const WNode& n : t.nodes) children += n.num;
        printf("\n%s: %zu nodes (%.2f children per node, %.1f MB at 64 B)\n", v.name, t.nodes.size(), children / t.nodes.size(), t.nodes.size() * 64.0 / 1e6);
        Stats both;
        for (int kind = 0; kind < 2; ++kind)
        {
            Stats s;
            double mismatch = 0;
            const std::vector<Ray>& rays = kind ? shadow : closest;
            for (size_t i = 0; i < rays.size(); ++i)
            {
                const float d = walk(t, rays[i], v.order, v.group, s);
                if (kind == 0 && in[i].dist < 1e30f && fabsf(d - in[i].dist) > 2e-3f * fmaxf(1.0f, in[i].dist)) mismatch++;
            }
            if (&v == &variants[0]) base[kind] = s.interior / s.rays;
            printf("  %-11s interior %6.2f per ray (%.3f of round 5's), leaf visits %5.2f, triangle tests %5.2f, box tests %6.1f, children entered per visit %.2f, hit %.3f, deepest stack %2.0f (rays over 6 / 8 / 10 / 13 entries: %.0f / %.0f / %.0f / %.0f)%s\n",
                   kind ? "any-hit:" : "closest-hit:", s.interior / s.rays, s.interior / s.rays / base[kind], s.leaves / s.rays, s.triTests / s.rays, s.boxTests / s.rays, s.entered / s.interior, s.hits / s.rays,
                   s.maxStack, s.stackOver[0], s.stackOver[1], s.stackOver[2], s.stackOver[3], kind == 0 ? (std::string("; hit distance differs from the oracle's for ") + std::to_string((long)mismatch) + " rays").c_str() : "");
            both.add(s);
        }
        printf("  %-11s interior %6.2f per ray (%.3f of round 5's), leaf visits %5.2f, triangle tests %5.2f\n", "all:", both.interior / both.rays,
               both.interior / both.rays / ((base[0] * closest.size() + base[1] * shadow.size()) / (closest.size() + shadow.size())), both.leaves / both.rays, both.triTests / both.rays);
    }
    return 0;
}
