// bvh_build.cpp -- see bvh_build.h. Host C++ only (no device code).
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace rptr {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = INFINITY;
            hi[k] = -INFINITY;
        }
    }
    void grow(const float *l, const float *h) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], l[k]);
            hi[k] = std::fmax(hi[k], h[k]);
        }
    }
    void grow(const Box &b) { grow(b.lo, b.hi); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f && dy >= 0.f && dz >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Tmp {
    Box box;
    int32_t left, right; // Tmp indices, -1 for leaves
    uint32_t first, count;
    int32_t depth;
};

struct Builder {
    const BuildPrim *prims;
    uint32_t *order;
    std::vector<Tmp> pool;
    std::atomic<uint32_t> next{0};
    std::atomic<int> live_threads{1};
    int max_threads;
    uint32_t max_leaf;
    int max_depth;
    static constexpr int NB = 32;

    uint32_t alloc() { return next.fetch_add(1); }

    static inline float centroid(const BuildPrim &p, int ax) { return 0.5f * (p.lo[ax] + p.hi[ax]); }

    void build(uint32_t self, uint32_t begin, uint32_t end, int depth) {
        Tmp &node = pool[self];
        node.box.reset();
        Box cb;
        cb.reset();
        for (uint32_t i = begin; i < end; ++i) {
            const BuildPrim &p = prims[order[i]];
            node.box.grow(p.lo, p.hi);
            float c[3] = {centroid(p, 0), centroid(p, 1), centroid(p, 2)};
            cb.grow(c, c);
        }
        node.left = node.right = -1;
        node.first = begin;
        node.count = end - begin;
        node.depth = depth;
        const uint32_t n = end - begin;
        if (n <= 1) return;

        uint32_t mid = 0;
        bool split_found = false;
        if (depth < max_depth - 8) { // SAH region; the last levels fall back to median splits to bound depth
            int best_axis = -1, best_bin = -1;
            float best_cost = INFINITY;
            for (int ax = 0; ax < 3; ++ax) {
                const float ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.f)) continue;
                Box bb[NB];
                uint32_t bc[NB];
                for (int b = 0; b < NB; ++b) {
                    bb[b].reset();
                    bc[b] = 0;
                }
                const float scale = NB / ext;
                for (uint32_t i = begin; i < end; ++i) {
                    const BuildPrim &p = prims[order[i]];
                    int b = (int)((centroid(p, ax) - cb.lo[ax]) * scale);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    bb[b].grow(p.lo, p.hi);
                    bc[b]++;
                }
                float ra[NB];
                uint32_t rc[NB];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = NB - 1; b >= 0; --b) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    ra[b] = acc.half_area();
                    rc[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    if (cnt == 0 || rc[b + 1] == 0) continue;
                    const float cost = acc.half_area() * cnt + ra[b + 1] * rc[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = ax;
                        best_bin = b;
                    }
                }
            }
            if (best_axis >= 0) {
                // SAH termination: leaf if cheaper than the best split
                const float pa = node.box.half_area();
                const float split_cost = 1.0f + (pa > 0.f ? best_cost / pa : (float)n);
                if (n <= max_leaf && (float)n <= split_cost) return;
                const float ext = cb.hi[best_axis] - cb.lo[best_axis];
                const float scale = NB / ext;
                const float clo = cb.lo[best_axis];
                uint32_t *m = std::partition(order + begin, order + end, [&](uint32_t id) {
                    int b = (int)((centroid(prims[id], best_axis) - clo) * scale);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    return b <= best_bin;
                });
                mid = (uint32_t)(m - order);
                split_found = mid > begin && mid < end;
            }
        }
        if (!split_found) {
            if (n <= max_leaf) return;
            // median split along the widest centroid axis (or by index when all centroids coincide)
            int ax = 0;
            float w = -1.f;
            for (int k = 0; k < 3; ++k) {
                float e = cb.hi[k] - cb.lo[k];
                if (e > w) {
                    w = e;
                    ax = k;
                }
            }
            mid = begin + n / 2;
            if (w > 0.f)
                std::nth_element(order + begin, order + mid, order + end,
                                 [&](uint32_t a, uint32_t b) { return centroid(prims[a], ax) < centroid(prims[b], ax); });
        }
        const uint32_t l = alloc(), r = alloc();
        pool[self].left = (int32_t)l;
        pool[self].right = (int32_t)r;
        const uint32_t nl = mid - begin, nr = end - mid;
        if (std::min(nl, nr) >= 32768 && live_threads.load() < max_threads) {
            live_threads.fetch_add(1);
            std::thread t([=]() {
                build(l, begin, mid, depth + 1);
                live_threads.fetch_sub(1);
            });
            build(r, mid, end, depth + 1);
            t.join();
        } else {
            build(l, begin, mid, depth + 1);
            build(r, mid, end, depth + 1);
        }
    }
};

} // namespace

void build_bvh2(const BuildPrim *prims, uint32_t n, uint32_t max_leaf, int max_depth, int threads, BuiltTree &out) {
    out.nodes.clear();
    out.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) out.order[i] = i;
    Builder b;
    b.prims = prims;
    b.order = out.order.data();
    b.pool.resize(std::max<size_t>(1, 2 * (size_t)n));
    b.max_leaf = std::max<uint32_t>(1, max_leaf);
    b.max_depth = max_depth;
    b.max_threads = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency());
    const uint32_t root = b.alloc();
    if (n == 0) {
        b.pool[root].box.reset();
        b.pool[root].left = b.pool[root].right = -1;
        b.pool[root].first = b.pool[root].count = 0;
        b.pool[root].depth = 0;
    } else
        b.build(root, 0, n, 0);
    const std::vector<Tmp> &tmp = b.pool;
    memcpy(out.lo, tmp[root].box.lo, sizeof(out.lo));
    memcpy(out.hi, tmp[root].box.hi, sizeof(out.hi));

    // flatten depth-first, left subtree directly after its parent
    auto leaf = [&](uint32_t i) { return tmp[i].left < 0; };
    auto set_child = [&](RptrBvhNode &nd, int which, uint32_t t, int32_t out_idx) {
        float *lo = which ? nd.lo1 : nd.lo0, *hi = which ? nd.hi1 : nd.hi0;
        memcpy(lo, tmp[t].box.lo, 12);
        memcpy(hi, tmp[t].box.hi, 12);
        if (leaf(t)) {
            (which ? nd.child1 : nd.child0) = RPTR_BVH_LEAF(tmp[t].first, tmp[t].count);
            (which ? nd.cnt1 : nd.cnt0) = (int32_t)tmp[t].count;
        } else {
            (which ? nd.child1 : nd.child0) = out_idx;
            (which ? nd.cnt1 : nd.cnt0) = 0;
        }
    };
    out.depth = 0;
    out.nodes.reserve(n / 2 + 4);
    out.nodes.push_back(RptrBvhNode());
    if (leaf(root)) {
        RptrBvhNode nd;
        memset(&nd, 0, sizeof(nd));
        set_child(nd, 0, root, -1);
        for (int k = 0; k < 3; ++k) {
            nd.lo1[k] = INFINITY;
            nd.hi1[k] = -INFINITY;
        }
        nd.child1 = RPTR_BVH_LEAF(0, 0);
        nd.cnt1 = 0;
        out.nodes[0] = nd;
        out.depth = 1;
        return;
    }
    struct Item {
        uint32_t t;
        int32_t o;
    };
    std::vector<Item> stack;
    stack.push_back({root, 0});
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        const uint32_t l = (uint32_t)tmp[it.t].left, r = (uint32_t)tmp[it.t].right;
        out.depth = std::max(out.depth, tmp[it.t].depth + 1);
        int32_t lo_idx = -1, ro_idx = -1;
        if (!leaf(l)) {
            lo_idx = (int32_t)out.nodes.size();
            out.nodes.push_back(RptrBvhNode());
        }
        if (!leaf(r)) {
            ro_idx = (int32_t)out.nodes.size();
            out.nodes.push_back(RptrBvhNode());
        }
        RptrBvhNode nd;
        memset(&nd, 0, sizeof(nd));
        set_child(nd, 0, l, lo_idx);
        set_child(nd, 1, r, ro_idx);
        out.nodes[it.o] = nd;
        if (!leaf(r)) stack.push_back({r, ro_idx});
        if (!leaf(l)) stack.push_back({l, lo_idx});
    }
}

void refit_bvh2(BuiltTree &tree, const BuildPrim *p) {
    // nodes are stored parents-before-children (DFS), so a reverse sweep sees children first
    std::vector<Box> nb(tree.nodes.size());
    for (int64_t i = (int64_t)tree.nodes.size() - 1; i >= 0; --i) {
        RptrBvhNode &nd = tree.nodes[i];
        for (int w = 0; w < 2; ++w) {
            const int32_t child = w ? nd.child1 : nd.child0;
            const int32_t cnt = w ? nd.cnt1 : nd.cnt0;
            Box b;
            b.reset();
            if (child >= 0)
                b = nb[child];
            else
                for (int32_t k = 0; k < cnt; ++k) b.grow(p[RPTR_BVH_LEAF_FIRST(child) + k].lo, p[RPTR_BVH_LEAF_FIRST(child) + k].hi);
            memcpy(w ? nd.lo1 : nd.lo0, b.lo, 12);
            memcpy(w ? nd.hi1 : nd.hi0, b.hi, 12);
            if (w == 0)
                nb[i] = b;
            else
                nb[i].grow(b);
        }
    }
    if (!tree.nodes.empty()) {
        memcpy(tree.lo, nb[0].lo, 12);
        memcpy(tree.hi, nb[0].hi, 12);
    }
}

// ------------------------------------------------------------------ BVH2 -> BVH4
void collapse_bvh4(const BuiltTree &t, Wide4Tree &out) {
    out.nodes.clear();
    memcpy(out.lo, t.lo, 12);
    memcpy(out.hi, t.hi, 12);
    if (t.nodes.empty()) {
        Wide4 w;
        memset(&w, 0, sizeof(w));
        for (int k = 0; k < 4; ++k) w.child[k] = RPTR_BVH4_EMPTY;
        out.nodes.push_back(w);
        return;
    }
    struct Slot {
        int32_t ref;
        float lo[3], hi[3];
    };
    auto area = [](const Slot &s) {
        const float dx = s.hi[0] - s.lo[0], dy = s.hi[1] - s.lo[1], dz = s.hi[2] - s.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto children_of = [&](int32_t n, std::vector<Slot> &dst, size_t at) {
        const RptrBvhNode &nd = t.nodes[n];
        size_t pos = at;
        for (int w = 0; w < 2; ++w) {
            const int32_t c = w ? nd.child1 : nd.child0;
            if (c < 0 && RPTR_BVH_LEAF_COUNT(c) == 0) continue; // empty half of a degenerate node
            Slot s;
            s.ref = c;
            memcpy(s.lo, w ? nd.lo1 : nd.lo0, 12);
            memcpy(s.hi, w ? nd.hi1 : nd.hi0, 12);
            dst.insert(dst.begin() + pos, s);
            ++pos;
        }
    };
    std::vector<int32_t> queue{0}; // binary node behind every wide node, breadth first
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        std::vector<Slot> slots;
        children_of(queue[qi], slots, 0);
        while (slots.size() < 4) {
            int pick = -1;
            float best = -1.0f;
            for (size_t i = 0; i < slots.size(); ++i)
                if (slots[i].ref >= 0) {
                    const float a = area(slots[i]);
                    if (a > best) {
                        best = a;
                        pick = (int)i;
                    }
                }
            if (pick < 0) break;
            const int32_t n = slots[pick].ref;
            slots.erase(slots.begin() + pick);
            children_of(n, slots, (size_t)pick);
        }
        Wide4 w;
        memset(&w, 0, sizeof(w));
        for (int k = 0; k < 4; ++k) {
            w.child[k] = RPTR_BVH4_EMPTY;
            for (int a = 0; a < 3; ++a) {
                w.box.lo[k][a] = INFINITY;
                w.box.hi[k][a] = -INFINITY;
            }
        }
        for (size_t k = 0; k < slots.size(); ++k) {
            memcpy(w.box.lo[k], slots[k].lo, 12);
            memcpy(w.box.hi[k], slots[k].hi, 12);
            if (slots[k].ref >= 0) {
                w.child[k] = (int32_t)queue.size();
                queue.push_back(slots[k].ref);
            } else
                w.child[k] = slots[k].ref;
        }
        out.nodes.push_back(w);
    }
}

} // namespace rptr
