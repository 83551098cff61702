/*
 * bvh_build.cpp — CPU BVH builder, host side of the tracer (no GPU calls).
 *
 * Keeps the role of BVH::build in the reference (src/bvh.cpp:30-238): a
 * top-down binary BVH over triangle bounds, emitted in depth-first pre-order
 * with 40-byte nodes (src/bvh.h:55-65) and a leaf-order triangle index array.
 * Split policy, unchanged so that the tree (and therefore the traversal order
 * that decides exact-tie hits) is the one the reference builds:
 *   - more than 32 triangles: 11 sampled planes per axis over the centroid
 *     bounds, cost 1 + (A_l/(A_l+A_r))*n_l + (A_r/(A_l+A_r))*n_r, std::partition;
 *   - otherwise median of the node's longest axis via std::nth_element;
 *   - leaf at <= 4 triangles, at the depth cap, or when a split leaves a side empty.
 * The device re-flattens this output into its own layout (see bvh_flatten in
 * the device sources); this file only produces the topology.
 */
#include "../../../include/mi_pathtrace.h"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

void pt_set_error(const char* fmt, ...);

namespace {

inline float fminG(float x, float y) { return x < y ? x : y; }   /* glm::min */
inline float fmaxG(float x, float y) { return x > y ? x : y; }   /* glm::max */

struct Bounds {
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    float hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    void addPoint(const float* p)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = fminG(lo[k], p[k]); hi[k] = fmaxG(hi[k], p[k]); }
    }
    void addBounds(const Bounds& b)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = fminG(lo[k], b.lo[k]); hi[k] = fmaxG(hi[k], b.hi[k]); }
    }
    float halfAreaTimes2() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
    int widestAxis() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx > dy && dx > dz) return 0;
        if (dy > dz) return 1;
        return 2;
    }
};

struct Prim {
    int index;
    Bounds box;
    float mid[3];
};

constexpr int kLeafSize = 4;       /* BVH_MAX_LEAF_TRIANGLES, src/bvh.h:11 */
constexpr int kSahThreshold = 32;  /* src/bvh.cpp:97 */
constexpr int kSahPlanes = 12;     /* BVH_SAH_SAMPLES, src/bvh.h:13 */

} // namespace

struct PtBvh {
    std::vector<PtBvhNode> nodes;
    std::vector<int32_t> order;
    std::vector<Prim> prims;
    int depthCap = 24;
    int deepest = 0;

    int emitLeaf(int self, int first, int last)
    {
        PtBvhNode& n = nodes[self];
        n.leftChild = -1;
        n.rightChild = -1;
        n.triangleStart = (int)order.size();
        n.triangleCount = last - first;
        for (int i = first; i < last; ++i) order.push_back(prims[i].index);
        return self;
    }

    int medianSplit(int first, int last, int axis)
    {
        int mid = (first + last) / 2;
        std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + last,
                         [axis](const Prim& a, const Prim& b) { return a.mid[axis] < b.mid[axis]; });
        return mid;
    }

    float planeCost(int first, int last, int axis, float plane, int& splitAt) const
    {
        Bounds L, R;
        int nl = 0, nr = 0;
        for (int i = first; i < last; ++i) {
            if (prims[i].mid[axis] < plane) { L.addBounds(prims[i].box); ++nl; }
            else { R.addBounds(prims[i].box); ++nr; }
        }
        splitAt = first + nl;
        if (nl == 0 || nr == 0) return FLT_MAX;
        float al = L.halfAreaTimes2(), ar = R.halfAreaTimes2();
        float sum = al + ar;
        return 1.0f + 1.0f * (al / sum * nl + ar / sum * nr);
    }

    int sahSplit(int first, int last)
    {
        if (last - first <= 2) return (first + last) / 2;
        Bounds cb;
        for (int i = first; i < last; ++i) cb.addPoint(prims[i].mid);
        int bestAxis = -1, bestAt = -1;
        float bestPlane = 0.0f, bestCost = FLT_MAX;
        for (int axis = 0; axis < 3; ++axis) {
            float lo = cb.lo[axis], hi = cb.hi[axis];
            if (lo == hi) continue;
            for (int k = 1; k < kSahPlanes; ++k) {
                float f = k / (float)kSahPlanes;
                float plane = lo + f * (hi - lo);
                int at;
                float cost = planeCost(first, last, axis, plane, at);
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestPlane = plane; bestAt = at; }
            }
        }
        if (bestAxis == -1 || bestAt <= first || bestAt >= last) return medianSplit(first, last, cb.widestAxis());
        auto it = std::partition(prims.begin() + first, prims.begin() + last,
                                 [bestAxis, bestPlane](const Prim& p) { return p.mid[bestAxis] < bestPlane; });
        return (int)(it - prims.begin());
    }

    int subdivide(int first, int last, int depth)
    {
        const int self = (int)nodes.size();
        PtBvhNode fresh;
        memset(&fresh, 0, sizeof(fresh));
        nodes.push_back(fresh);
        if (depth > deepest) deepest = depth;

        Bounds b;
        for (int i = first; i < last; ++i) b.addBounds(prims[i].box);
        memcpy(nodes[self].bmin, b.lo, 12);
        memcpy(nodes[self].bmax, b.hi, 12);

        const int count = last - first;
        if (depth >= depthCap || count <= kLeafSize) return emitLeaf(self, first, last);
        int mid = count > kSahThreshold ? sahSplit(first, last) : medianSplit(first, last, b.widestAxis());
        if (mid == first || mid == last) return emitLeaf(self, first, last);

        int l = subdivide(first, mid, depth + 1);
        int r = subdivide(mid, last, depth + 1);
        nodes[self].leftChild = l;
        nodes[self].rightChild = r;
        nodes[self].triangleStart = -1;
        nodes[self].triangleCount = 0;
        return self;
    }
};

extern "C" {

int pt_bvh_build(const PtTriangle* tris, int n, int maxDepth, PtBvh** out)
{
    if (!out || n < 0 || (n > 0 && !tris)) { pt_set_error("pt_bvh_build: bad argument"); return PT_EINVAL; }
    PtBvh* B = new PtBvh();
    B->depthCap = maxDepth;
    B->prims.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        Prim& p = B->prims[i];
        p.index = i;
        p.box.addPoint(tris[i].v0);
        p.box.addPoint(tris[i].v1);
        p.box.addPoint(tris[i].v2);
        for (int k = 0; k < 3; ++k) p.mid[k] = (p.box.lo[k] + p.box.hi[k]) * 0.5f;
    }
    B->nodes.reserve(2 * (size_t)n + 1);
    B->order.reserve((size_t)n);
    B->subdivide(0, n, 0);
    B->prims.clear();
    B->prims.shrink_to_fit();
    *out = B;
    return PT_OK;
}

const PtBvhNode* pt_bvh_nodes(const PtBvh* b, int* count)
{
    if (count) *count = b ? (int)b->nodes.size() : 0;
    return b ? b->nodes.data() : nullptr;
}
const int32_t* pt_bvh_indices(const PtBvh* b, int* count)
{
    if (count) *count = b ? (int)b->order.size() : 0;
    return b ? b->order.data() : nullptr;
}
int pt_bvh_stats(const PtBvh* b, int* nodeCount, int* leafCount, int* maxDepth)
{
    if (!b) return PT_EINVAL;
    int leaves = 0;
    for (const auto& n : b->nodes) if (n.leftChild == -1) ++leaves;
    if (nodeCount) *nodeCount = (int)b->nodes.size();
    if (leafCount) *leafCount = leaves;
    if (maxDepth) *maxDepth = b->deepest;
    return PT_OK;
}
void pt_bvh_free(PtBvh* b) { delete b; }

} // extern "C"
