// bvh_build.cpp -- see bvh_build.h. Host C++ only (no device code).
#include "bvh_build.h"

#include <algorithm>
#include <functional>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace rptr {

BuildTuning &build_tuning() {
    // per THREAD: a caller sets it and builds on the same thread (collapse_bvh4 / build_bvh2_ploc read it before they start their own
    // workers), so two handles with different options that build side by side -- one host thread per GPU -- never see each other's values
    static thread_local BuildTuning t;
    return t;
}
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
void collapse_bvh4(const BuiltTree &t, Wide4Tree &out, int rule, int fallback) {
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
    if (rule < 0) rule = build_tuning().collapse_rule >= 0 ? build_tuning().collapse_rule : fallback;
    const bool even_rule = rule == COLLAPSE_EVEN; // the device's rebuild rule (lbvh.h rp_k_lbvh_emit)
    const bool dp_rule = rule == COLLAPSE_OPTIMAL; // bvh_build.h
    const size_t nb = t.nodes.size();
    std::vector<float> cost1, F; // cost1[n]: subtree n as ONE wide node; F[4 n + i - 1]: the children of n spread over at most i slots
    std::vector<float> node_area;
    if (dp_rule) {
        cost1.assign(nb, 0.0f);
        F.assign(4 * nb, 0.0f);
        node_area.assign(nb, 0.0f);
        auto half_area = [](const float *lo, const float *hi) {
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        // children are allocated behind their parents (Builder::alloc, build_bvh2_ploc's relabelling): last to first is bottom-up
        auto G = [&](int32_t c, int i) -> float { // subtree c through at most i slots
            if (c < 0) return 0.0f;
            return i == 1 ? cost1[c] : std::min(cost1[c], F[4 * (size_t)c + i - 1]);
        };
        for (size_t k = nb; k-- > 0;) {
            const RptrBvhNode &nd = t.nodes[k];
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::fmin(nd.lo0[a], nd.lo1[a]);
                hi[a] = std::fmax(nd.hi0[a], nd.hi1[a]);
            }
            node_area[k] = half_area(lo, hi);
            if ((nd.child0 >= 0 && (size_t)nd.child0 <= k) || (nd.child1 >= 0 && (size_t)nd.child1 <= k)) { // not in that order: fall back
                cost1.clear();
                break;
            }
            F[4 * k + 0] = INFINITY; // two children never fit one slot
            for (int i = 2; i <= 4; ++i) {
                float best = INFINITY;
                for (int l = 1; l < i; ++l) best = std::min(best, G(nd.child0, l) + G(nd.child1, i - l));
                F[4 * k + i - 1] = best;
            }
            cost1[k] = node_area[k] + F[4 * k + 3];
        }
    }
    std::function<void(int32_t, const float *, const float *, int, std::vector<Slot> &)> emit_dp = [&](int32_t c, const float *lo, const float *hi, int i,
                                                                                                         std::vector<Slot> &dst) {
        if (c < 0 && RPTR_BVH_LEAF_COUNT(c) == 0) return;
        if (c < 0 || i == 1 || cost1[c] <= F[4 * (size_t)c + i - 1]) {
            Slot sl;
            sl.ref = c;
            memcpy(sl.lo, lo, 12);
            memcpy(sl.hi, hi, 12);
            dst.push_back(sl);
            return;
        }
        const RptrBvhNode &nd = t.nodes[c];
        int best_l = 1;
        float best = INFINITY;
        auto G = [&](int32_t x, int j) -> float { return x < 0 ? 0.0f : (j == 1 ? cost1[x] : std::min(cost1[x], F[4 * (size_t)x + j - 1])); };
        for (int l = 1; l < i; ++l) {
            const float v = G(nd.child0, l) + G(nd.child1, i - l);
            if (v < best) {
                best = v;
                best_l = l;
            }
        }
        emit_dp(nd.child0, nd.lo0, nd.hi0, best_l, dst);
        emit_dp(nd.child1, nd.lo1, nd.hi1, i - best_l, dst);
    };
    std::vector<int32_t> queue{0}; // binary node behind every wide node, breadth first
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        std::vector<Slot> slots;
        if (dp_rule && !cost1.empty()) {
            const RptrBvhNode &nd = t.nodes[queue[qi]];
            int best_l = 1;
            float best = INFINITY;
            auto G = [&](int32_t x, int j) -> float { return x < 0 ? 0.0f : (j == 1 ? cost1[x] : std::min(cost1[x], F[4 * (size_t)x + j - 1])); };
            for (int l = 1; l < 4; ++l) {
                const float v = G(nd.child0, l) + G(nd.child1, 4 - l);
                if (v < best) {
                    best = v;
                    best_l = l;
                }
            }
            emit_dp(nd.child0, nd.lo0, nd.hi0, best_l, slots);
            emit_dp(nd.child1, nd.lo1, nd.hi1, 4 - best_l, slots);
        } else
        children_of(queue[qi], slots, 0);
        if (even_rule) { // every inner child hands its two children up, whatever their size (csrc/lbvh.h rp_k_lbvh_emit)
            std::vector<Slot> up;
            for (const Slot &sl : slots) {
                if (sl.ref >= 0)
                    children_of(sl.ref, up, up.size());
                else
                    up.push_back(sl);
            }
            slots.swap(up);
        }
        while (!even_rule && !(dp_rule && !cost1.empty()) && slots.size() < 4) {
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

// ------------------------------------------------------------------ PLOC (bvh_build.h): host statement of the device builder's clustering
namespace {
inline uint64_t expand21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
inline float union_half_area(const Box &a, const Box &b) {
    const float dx = std::fmax(a.hi[0], b.hi[0]) - std::fmin(a.lo[0], b.lo[0]), dy = std::fmax(a.hi[1], b.hi[1]) - std::fmin(a.lo[1], b.lo[1]),
                dz = std::fmax(a.hi[2], b.hi[2]) - std::fmin(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}
} // namespace

void build_bvh2_ploc(const BuildPrim *prims, uint32_t n, int radius, uint32_t max_leaf, int threads, BuiltTree &out) {
    out.nodes.clear();
    out.order.resize(n);
    if (n < 2) {
        build_bvh2(prims, n, max_leaf, 48, threads, out);
        return;
    }
    const int nt = std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 64));
    auto parallel = [&](size_t count, auto &&fn) {
        std::vector<std::thread> pool;
        const size_t chunk = (count + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            const size_t b = std::min(count, (size_t)t * chunk), e = std::min(count, (size_t)(t + 1) * chunk);
            if (b < e) pool.emplace_back([=, &fn] { fn(b, e); });
        }
        for (auto &th : pool) th.join();
    };
    // Morton order of the centroids on a cubic grid (21 bits per axis)
    Box cb;
    cb.reset();
    for (uint32_t i = 0; i < n; ++i) {
        const float c[3] = {prims[i].lo[0] + prims[i].hi[0], prims[i].lo[1] + prims[i].hi[1], prims[i].lo[2] + prims[i].hi[2]};
        cb.grow(c, c);
    }
    float ext = 0;
    for (int k = 0; k < 3; ++k) ext = std::fmax(ext, cb.hi[k] - cb.lo[k]);
    // (the device's keys, lbvh.h rp_k_lbvh_keys: the code gets the bits the triangle index leaves in a 64-bit key; ties by index)
    int index_bits = 1;
    while ((1ull << index_bits) < (unsigned long long)n) ++index_bits;
    const int axis_bits = std::min(21, (64 - index_bits) / 3);
    const float cells = (float)((1u << axis_bits) - 1u), inv = ext > 0 ? cells / ext : 0.f;
    std::vector<std::pair<uint64_t, uint32_t>> keys(n);
    parallel(n, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            uint32_t q[3];
            for (int k = 0; k < 3; ++k) q[k] = (uint32_t)std::fmin(std::fmax(((prims[i].lo[k] + prims[i].hi[k]) - cb.lo[k]) * inv, 0.f), cells);
            keys[i] = {(expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]), (uint32_t)i};
        }
    });
    std::sort(keys.begin(), keys.end());
    // binary tree: nodes 0..n-1 are the primitives in Morton order, n.. are merges in creation order
    struct PNode {
        Box box;
        int32_t left, right;
        uint32_t count;
    };
    std::vector<PNode> nodes(2 * (size_t)n - 1);
    std::vector<uint32_t> cur(n), next;
    parallel(n, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            PNode &p = nodes[i];
            memcpy(p.box.lo, prims[keys[i].second].lo, 12);
            memcpy(p.box.hi, prims[keys[i].second].hi, 12);
            p.left = p.right = -1;
            p.count = 1;
            cur[i] = (uint32_t)i;
        }
    });
    size_t made = n;
    std::vector<uint32_t> nn;
    std::vector<uint32_t> slot;
    size_t top_k = RP_PLOC_TOP_DEFAULT; // stop clustering at this many clusters and put a binned-SAH tree over them: the device builder's default (csrc/ploc.h)
    if (build_tuning().ploc_top > 0) top_k = build_tuning().ploc_top;
    while (cur.size() > top_k) {
        const size_t m = cur.size();
        nn.resize(m);
        parallel(m, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) {
                const size_t lo = i > (size_t)radius ? i - radius : 0, hi = std::min(m - 1, i + radius);
                float best = INFINITY;
                uint32_t arg = (uint32_t)(i == lo ? hi : lo);
                for (size_t j = lo; j <= hi; ++j) {
                    if (j == i) continue;
                    const float a = union_half_area(nodes[cur[i]].box, nodes[cur[j]].box);
                    if (a < best) {
                        best = a;
                        arg = (uint32_t)j;
                    }
                }
                nn[i] = arg;
            }
        });
        // mutual nearest neighbours merge (the lower position makes the node), everything else is carried over; order is kept
        slot.assign(m + 1, 0);
        size_t merges = 0;
        for (size_t i = 0; i < m; ++i) {
            const bool mutual = nn[nn[i]] == i;
            if (mutual && i < nn[i]) ++merges;
            slot[i + 1] = slot[i] + ((mutual && i > nn[i]) ? 0u : 1u); // the upper partner disappears
        }
        next.resize(slot[m]);
        size_t id = made;
        for (size_t i = 0; i < m; ++i) {
            const bool mutual = nn[nn[i]] == i;
            if (mutual && i > nn[i]) continue;
            if (mutual) {
                PNode &p = nodes[id];
                const uint32_t a = cur[i], b = cur[nn[i]];
                p.box = nodes[a].box;
                p.box.grow(nodes[b].box);
                p.left = (int32_t)a;
                p.right = (int32_t)b;
                p.count = nodes[a].count + nodes[b].count;
                next[slot[i]] = (uint32_t)id++;
            } else
                next[slot[i]] = cur[i];
        }
        made = id;
        cur.swap(next);
        (void)merges;
    }
    if (cur.size() > 1) { // top-down binned SAH over the remaining clusters, stitched on
        std::vector<BuildPrim> cp(cur.size());
        for (size_t i = 0; i < cur.size(); ++i) {
            memcpy(cp[i].lo, nodes[cur[i]].box.lo, 12);
            memcpy(cp[i].hi, nodes[cur[i]].box.hi, 12);
        }
        BuiltTree top;
        build_bvh2(cp.data(), (uint32_t)cp.size(), 1, 56, threads, top);
        std::vector<uint32_t> id_of(top.nodes.size());
        for (int64_t i = (int64_t)top.nodes.size() - 1; i >= 0; --i) { // children lie behind their parents: backwards = bottom-up
            const RptrBvhNode &t = top.nodes[(size_t)i];
            auto ref = [&](int32_t c) -> uint32_t { return c >= 0 ? id_of[(size_t)c] : cur[top.order[(size_t)RPTR_BVH_LEAF_FIRST(c)]]; };
            PNode &p = nodes[made];
            const uint32_t a = ref(t.child0), b2 = ref(t.child1);
            p.box = nodes[a].box;
            p.box.grow(nodes[b2].box);
            p.left = (int32_t)a;
            p.right = (int32_t)b2;
            p.count = nodes[a].count + nodes[b2].count;
            id_of[(size_t)i] = (uint32_t)made++;
        }
        cur.assign(1, id_of[0]);
    }
    const uint32_t root = cur[0];
    // leaf collapse by SAH: cost of a subtree = min(triangles as one leaf, 1 + area-weighted cost of the children)
    const int force_leaf = build_tuning().ploc_leaf; // the device's rule instead: a range of <= k triangles is a leaf
    std::vector<float> cost(nodes.size());
    std::vector<uint8_t> is_leaf(nodes.size(), 0);
    for (size_t i = 0; i < nodes.size(); ++i) { // children are created before their parents: ascending order is bottom-up
        const PNode &p = nodes[i];
        if (p.left < 0) {
            cost[i] = 1.0f;
            is_leaf[i] = 1;
            continue;
        }
        const float a = p.box.half_area();
        const float split = 1.0f + (a > 0 ? (nodes[p.left].box.half_area() * cost[p.left] + nodes[p.right].box.half_area() * cost[p.right]) / a
                                          : cost[p.left] + cost[p.right]);
        if (force_leaf > 0 ? p.count <= (uint32_t)force_leaf : (p.count <= max_leaf && (float)p.count <= split)) {
            cost[i] = (float)p.count;
            is_leaf[i] = 1;
        } else
            cost[i] = split;
    }
    // flatten: depth-first, primitives of a subtree contiguous in `order`
    memcpy(out.lo, nodes[root].box.lo, 12);
    memcpy(out.hi, nodes[root].box.hi, 12);
    uint32_t at = 0;
    auto emit_leaf = [&](uint32_t t) -> std::pair<uint32_t, uint32_t> { // (first, count): the primitives below t, in tree order
        const uint32_t first = at;
        std::vector<uint32_t> st{t};
        while (!st.empty()) {
            const uint32_t x = st.back();
            st.pop_back();
            if (nodes[x].left < 0)
                out.order[at++] = keys[x].second;
            else {
                st.push_back((uint32_t)nodes[x].right);
                st.push_back((uint32_t)nodes[x].left);
            }
        }
        return {first, at - first};
    };
    auto set_child = [&](RptrBvhNode &nd, int which, uint32_t t, int32_t out_idx) {
        memcpy(which ? nd.lo1 : nd.lo0, nodes[t].box.lo, 12);
        memcpy(which ? nd.hi1 : nd.hi0, nodes[t].box.hi, 12);
        if (is_leaf[t]) {
            const auto fc = emit_leaf(t);
            (which ? nd.child1 : nd.child0) = RPTR_BVH_LEAF(fc.first, fc.second);
            (which ? nd.cnt1 : nd.cnt0) = (int32_t)fc.second;
        } else {
            (which ? nd.child1 : nd.child0) = out_idx;
            (which ? nd.cnt1 : nd.cnt0) = 0;
        }
    };
    out.depth = 0;
    out.nodes.push_back(RptrBvhNode());
    if (is_leaf[root]) {
        RptrBvhNode nd;
        memset(&nd, 0, sizeof(nd));
        set_child(nd, 0, root, -1);
        for (int k = 0; k < 3; ++k) {
            nd.lo1[k] = INFINITY;
            nd.hi1[k] = -INFINITY;
        }
        nd.child1 = RPTR_BVH_LEAF(0, 0);
        out.nodes[0] = nd;
        out.depth = 1;
        return;
    }
    struct Item {
        uint32_t t;
        int32_t o, depth;
    };
    std::vector<Item> stack{{root, 0, 0}};
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        const uint32_t l = (uint32_t)nodes[it.t].left, r = (uint32_t)nodes[it.t].right;
        out.depth = std::max(out.depth, it.depth + 1);
        int32_t lo_idx = -1, ro_idx = -1;
        if (!is_leaf[l]) {
            lo_idx = (int32_t)out.nodes.size();
            out.nodes.push_back(RptrBvhNode());
        }
        if (!is_leaf[r]) {
            ro_idx = (int32_t)out.nodes.size();
            out.nodes.push_back(RptrBvhNode());
        }
        RptrBvhNode nd;
        memset(&nd, 0, sizeof(nd));
        set_child(nd, 0, l, lo_idx);
        set_child(nd, 1, r, ro_idx);
        out.nodes[it.o] = nd;
        if (!is_leaf[r]) stack.push_back({r, ro_idx, it.depth + 1});
        if (!is_leaf[l]) stack.push_back({l, lo_idx, it.depth + 1});
    }
}

// ------------------------------------------------------------------ triangle pre-splitting (bvh_build.h)
namespace {

struct Poly {
    int n;
    double p[16][3];
};

// Sutherland-Hodgman against the plane x[ax] = s: `l` gets the part with x <= s, `r` the part with x >= s
void clip_poly(const Poly &in, int ax, double s, Poly &l, Poly &r) {
    l.n = r.n = 0;
    for (int i = 0; i < in.n; ++i) {
        const double *a = in.p[i], *b = in.p[(i + 1) % in.n];
        if (a[ax] <= s && l.n < 16) memcpy(l.p[l.n++], a, 24);
        if (a[ax] >= s && r.n < 16) memcpy(r.p[r.n++], a, 24);
        if ((a[ax] < s && b[ax] > s) || (a[ax] > s && b[ax] < s)) {
            const double t = (s - a[ax]) / (b[ax] - a[ax]);
            double q[3];
            for (int k = 0; k < 3; ++k) q[k] = a[k] + t * (b[k] - a[k]);
            q[ax] = s;
            if (l.n < 16) memcpy(l.p[l.n++], q, 24);
            if (r.n < 16) memcpy(r.p[r.n++], q, 24);
        }
    }
}

inline float round_down(double x) {
    float f = (float)x;
    return (double)f > x ? std::nextafterf(f, -INFINITY) : f;
}
inline float round_up(double x) {
    float f = (float)x;
    return (double)f < x ? std::nextafterf(f, INFINITY) : f;
}

// box of a polygon, rounded outwards, inside `limit` (the triangle's own box: no point of the triangle lies outside it)
void poly_box(const Poly &p, const BuildPrim &limit, BuildPrim &out) {
    for (int k = 0; k < 3; ++k) {
        double lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < p.n; ++i) {
            lo = std::fmin(lo, p.p[i][k]);
            hi = std::fmax(hi, p.p[i][k]);
        }
        out.lo[k] = std::fmax(round_down(lo), limit.lo[k]);
        out.hi[k] = std::fmin(round_up(hi), limit.hi[k]);
    }
}

struct SplitGrid {
    double lo[3], cell; // cubic grid: 2^40 cells of size `cell` along every axis from lo
    static constexpr int BITS = 40;
    // the most important grid plane strictly inside (a, b) along axis ax: its level (0 = the scene's median plane ... BITS - 1) and position
    bool plane(int ax, double a, double b, int &level, double &pos) const {
        if (!(b > a)) return false;
        const double scale = 1.0 / cell;
        const double fa = (a - lo[ax]) * scale, fb = (b - lo[ax]) * scale;
        const uint64_t top = (1ull << BITS) - 1;
        const uint64_t qa = fa <= 0 ? 0 : (fa >= (double)top ? top : (uint64_t)fa), qb = fb <= 0 ? 0 : (fb >= (double)top ? top : (uint64_t)fb);
        if (qa == qb) return false;
        const int bit = 63 - __builtin_clzll(qa ^ qb);
        const uint64_t q = (qb >> bit) << bit;
        pos = lo[ax] + (double)q * cell;
        level = BITS - 1 - bit;
        return pos > a && pos < b;
    }
};

struct Splitter {
    const SplitGrid &grid;
    const BuildPrim &limit;
    std::vector<BuildPrim> &out;
    void run(const Poly &poly, const BuildPrim &box, int splits) {
        if (splits > 0 && poly.n >= 3) {
            int best_ax = -1, best_level = 1 << 30;
            double best_pos = 0, best_ext = -1;
            for (int ax = 0; ax < 3; ++ax) {
                int level;
                double pos;
                if (!grid.plane(ax, box.lo[ax], box.hi[ax], level, pos)) continue;
                const double ext = (double)box.hi[ax] - box.lo[ax];
                if (level < best_level || (level == best_level && ext > best_ext)) {
                    best_ax = ax;
                    best_level = level;
                    best_pos = pos;
                    best_ext = ext;
                }
            }
            if (best_ax >= 0) {
                Poly l, r;
                clip_poly(poly, best_ax, best_pos, l, r);
                if (l.n >= 3 && r.n >= 3) {
                    BuildPrim bl, br;
                    poly_box(l, limit, bl);
                    poly_box(r, limit, br);
                    bl.hi[best_ax] = std::fmin(bl.hi[best_ax], round_up(best_pos));
                    br.lo[best_ax] = std::fmax(br.lo[best_ax], round_down(best_pos));
                    auto longest = [](const BuildPrim &b) {
                        return std::fmax((double)b.hi[0] - b.lo[0], std::fmax((double)b.hi[1] - b.lo[1], (double)b.hi[2] - b.lo[2]));
                    };
                    const double wl = longest(bl), wr = longest(br);
                    int sl = (wl + wr) > 0 ? (int)((splits - 1) * (wl / (wl + wr)) + 0.5) : (splits - 1) / 2;
                    sl = std::max(0, std::min(splits - 1, sl));
                    run(l, bl, sl);
                    run(r, br, splits - 1 - sl);
                    return;
                }
            }
        }
        out.push_back(box);
    }
};

} // namespace

void presplit_triangles(const TriVerts *tris, uint32_t n, float density, float budget, int max_refs_per_tri, int threads, std::vector<BuildPrim> &out_box,
                        std::vector<uint32_t> &out_tri) {
    out_box.clear();
    out_tri.clear();
    if (n == 0) return;
    const int nt = std::max(1, std::min(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), 64));
    auto parallel = [&](auto &&fn) { // fn(thread, begin, end) over contiguous chunks of the triangles
        std::vector<std::thread> pool;
        const uint32_t chunk = (n + nt - 1) / nt;
        for (int t = 0; t < nt; ++t) {
            const uint32_t b = std::min<uint64_t>((uint64_t)t * chunk, n), e = std::min<uint64_t>((uint64_t)(t + 1) * chunk, n);
            if (b < e) pool.emplace_back([=, &fn] { fn(t, b, e); });
        }
        for (auto &th : pool) th.join();
    };
    // scene bounds -> cubic grid
    std::vector<Box> tb((size_t)nt);
    for (Box &b : tb) b.reset();
    std::vector<BuildPrim> full(n);
    parallel([&](int t, uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; ++i) {
            BuildPrim &bp = full[i];
            for (int k = 0; k < 3; ++k) {
                bp.lo[k] = std::fmin(tris[i].v[0][k], std::fmin(tris[i].v[1][k], tris[i].v[2][k]));
                bp.hi[k] = std::fmax(tris[i].v[0][k], std::fmax(tris[i].v[1][k], tris[i].v[2][k]));
            }
            tb[(size_t)t].grow(bp.lo, bp.hi);
        }
    });
    Box sb;
    sb.reset();
    for (const Box &b : tb) sb.grow(b);
    SplitGrid grid;
    double ext = 0;
    for (int k = 0; k < 3; ++k) ext = std::fmax(ext, (double)sb.hi[k] - sb.lo[k]);
    if (!(ext > 0) || !(budget > 0.f) || !(density > 0.f)) { // nothing to split along / switched off
        out_box = std::move(full);
        out_tri.resize(n);
        for (uint32_t i = 0; i < n; ++i) out_tri[i] = i;
        return;
    }
    for (int k = 0; k < 3; ++k) grid.lo[k] = sb.lo[k];
    grid.cell = ext / (double)(1ull << SplitGrid::BITS);
    // priority of every triangle
    std::vector<float> prio(n);
    parallel([&](int, uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; ++i) {
            const BuildPrim &bp = full[i];
            int level = 1 << 30;
            for (int ax = 0; ax < 3; ++ax) {
                int lv;
                double pos;
                if (grid.plane(ax, bp.lo[ax], bp.hi[ax], lv, pos)) level = std::min(level, lv);
            }
            if (level == (1 << 30)) {
                prio[i] = 0.f;
                continue;
            }
            const double dx = (double)bp.hi[0] - bp.lo[0], dy = (double)bp.hi[1] - bp.lo[1], dz = (double)bp.hi[2] - bp.lo[2];
            const double a_box = dx * dy + dy * dz + dz * dx;
            double e1[3], e2[3];
            for (int k = 0; k < 3; ++k) {
                e1[k] = (double)tris[i].v[1][k] - tris[i].v[0][k];
                e2[k] = (double)tris[i].v[2][k] - tris[i].v[0][k];
            }
            const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const double a_ideal = 0.5 * (std::fabs(cx) + std::fabs(cy) + std::fabs(cz)); // the box of an axis-aligned right triangle of this area
            const double excess = std::fmax(0.0, a_box - a_ideal) / (ext * ext);
            prio[i] = (float)std::cbrt(std::ldexp(excess, -level));
        }
    });
    // the scale D with sum floor(D * prio) <= budget * n (bisection; counts are capped per triangle)
    const int cap = std::max(1, max_refs_per_tri) - 1;
    const double want = (double)budget * n;
    auto total = [&](double D) {
        std::vector<double> part((size_t)nt, 0.0);
        parallel([&](int t, uint32_t b, uint32_t e) {
            double s = 0;
            for (uint32_t i = b; i < e; ++i) s += std::min<double>(cap, std::floor(D * prio[i]));
            part[(size_t)t] = s;
        });
        double s = 0;
        for (double v : part) s += v;
        return s;
    };
    double D = (double)density;
    if (total(D) > want) { // over the budget: the largest scale that fits
        double d_lo = 0, d_hi = D;
        for (int it = 0; it < 40; ++it) {
            const double mid = 0.5 * (d_lo + d_hi);
            (total(mid) <= want ? d_lo : d_hi) = mid;
        }
        D = d_lo;
    }
    // split (per thread into its own list, then concatenated in triangle order)
    std::vector<std::vector<BuildPrim>> boxes((size_t)nt);
    std::vector<std::vector<uint32_t>> ids((size_t)nt);
    parallel([&](int t, uint32_t b, uint32_t e) {
        std::vector<BuildPrim> &ob = boxes[(size_t)t];
        std::vector<uint32_t> &oi = ids[(size_t)t];
        ob.reserve((size_t)((e - b) * (1.0 + budget) * 1.1) + 16);
        oi.reserve(ob.capacity());
        for (uint32_t i = b; i < e; ++i) {
            const int splits = (int)std::min<double>(cap, std::floor(D * prio[i]));
            const size_t before = ob.size();
            if (splits <= 0)
                ob.push_back(full[i]);
            else {
                Poly p;
                p.n = 3;
                for (int v = 0; v < 3; ++v)
                    for (int k = 0; k < 3; ++k) p.p[v][k] = tris[i].v[v][k];
                Splitter sp{grid, full[i], ob};
                sp.run(p, full[i], splits);
            }
            oi.insert(oi.end(), ob.size() - before, i);
        }
    });
    size_t total_refs = 0;
    for (auto &v : boxes) total_refs += v.size();
    out_box.reserve(total_refs);
    out_tri.reserve(total_refs);
    for (int t = 0; t < nt; ++t) {
        out_box.insert(out_box.end(), boxes[(size_t)t].begin(), boxes[(size_t)t].end());
        out_tri.insert(out_tri.end(), ids[(size_t)t].begin(), ids[(size_t)t].end());
    }
}

} // namespace rptr
