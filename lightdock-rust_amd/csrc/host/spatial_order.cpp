#include "spatial_order.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <utility>

namespace ld {

namespace {

void split(const double *xyz, std::vector<uint32_t> &ids, size_t begin, size_t end) {
    const size_t n = end - begin;
    if (n <= 8) return;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t k = begin; k < end; k++)
        for (int c = 0; c < 3; c++) {
            const double v = xyz[3 * (size_t)ids[k] + c];
            lo[c] = std::min(lo[c], v);
            hi[c] = std::max(hi[c], v);
        }
    int axis = 0;
    for (int c = 1; c < 3; c++)
        if (hi[c] - lo[c] > hi[axis] - lo[axis]) axis = c;
    std::stable_sort(ids.begin() + (long)begin, ids.begin() + (long)end, [&](uint32_t a, uint32_t b) {
        return xyz[3 * (size_t)a + axis] < xyz[3 * (size_t)b + axis];
    });
    const size_t unit = n > 64 ? 64 : 8;
    size_t half = ((n / 2 + unit / 2) / unit) * unit;
    if (half < unit) half = unit;
    if (half >= n) half = (n - 1) / unit * unit;
    split(xyz, ids, begin, begin + half);
    split(xyz, ids, begin + half, end);
}

// Cost of one 8-atom subtile for the 8x8 box test: the volume of its bounding box grown by the
// cutoff plus a typical partner box (15 A + 15 A + ~6 A), i.e. proportional to the number of
// partner subtiles the box test lets through.
constexpr double kGrow = 36.0;

double subtile_cost(const double *xyz, const uint32_t *ids) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int k = 0; k < 8; k++)
        for (int c = 0; c < 3; c++) {
            const double v = xyz[3 * (size_t)ids[k] + c];
            lo[c] = std::min(lo[c], v);
            hi[c] = std::max(hi[c], v);
        }
    return (hi[0] - lo[0] + kGrow) * (hi[1] - lo[1] + kGrow) * (hi[2] - lo[2] + kGrow);
}

// The median splits leave subtiles whose boxes overlap along the split planes.  Swap atoms
// between the subtiles of a window while that lowers the summed cost (deterministic sweep order,
// a few passes): first inside every full 64-atom tile, then across every two consecutive tiles
// (consecutive tiles are siblings or cousins of the split tree, i.e. neighbours in space).  On
// the 1k4c example this removes 9 % of the 8x8 blocks the kernel has to evaluate; the number of
// 64x64 tile pairs stays the same.
void refine_window(const double *xyz, uint32_t *t, int atoms) {
    const int groups = atoms / 8;
    std::vector<double> cost((size_t)groups);
    for (int g = 0; g < groups; g++) cost[(size_t)g] = subtile_cost(xyz, t + 8 * g);
    for (int pass = 0; pass < 4; pass++) {
        bool improved = false;
        for (int i = 0; i < atoms; i++)
            for (int j = (i / 8 + 1) * 8; j < atoms; j++) {  // j in a later subtile than i
                const int a = i / 8, b = j / 8;
                std::swap(t[i], t[j]);
                const double ca = subtile_cost(xyz, t + 8 * a), cb = subtile_cost(xyz, t + 8 * b);
                if (ca + cb < cost[(size_t)a] + cost[(size_t)b] - 1e-9) {
                    cost[(size_t)a] = ca;
                    cost[(size_t)b] = cb;
                    improved = true;
                } else {
                    std::swap(t[i], t[j]);
                }
            }
        if (!improved) break;
    }
}

void refine_subtiles(const double *xyz, std::vector<uint32_t> &ids, size_t n) {
    for (size_t base = 0; base + 64 <= n; base += 64) refine_window(xyz, ids.data() + base, 64);
    for (size_t base = 0; base + 128 <= n; base += 64) refine_window(xyz, ids.data() + base, 128);
}

// ---- refinement beyond the windows (round 6) ----------------------------------------------------------------------------
// The window sweeps above stop in local minima of a cost that is flat wherever a swap leaves a box's extreme atoms alone, and
// they never move an atom further than the next tile.  Two more sweeps over ALL full subtiles, each against its kNeighbours
// nearest subtiles (by centroid) whatever tile they lie in: first with a smooth cost -- the sum of the 28 distances inside a
// subtile, which pulls stragglers in although no box shrinks yet --, then with the box cost itself.  After that the subtiles
// are regrouped into tiles (whole subtiles swapped between neighbouring tiles while the summed tile-box cost falls), because the
// atom swaps have let the tiles' boxes grow.  Replaying the example poses through the culling's two box tests
// (tools/cluster_sim.py): 8x8 blocks per pose 1k4c 5872 -> 5311 (-9.6 %), 1ppe 1288 -> 1209, 2uuy 1061 -> 984; 64x64 tile pairs
// per pose 369 -> 383, 63 -> 67, 62 -> 64.  The pair kernel's time is proportional to the blocks.
constexpr int kNeighbours = 10, kTileNeighbours = 8, kGlobalPasses = 4;

double dist(const double *xyz, uint32_t a, uint32_t b) {
    double d2 = 0.0;
    for (int c = 0; c < 3; c++) {
        const double d = xyz[3 * (size_t)a + c] - xyz[3 * (size_t)b + c];
        d2 += d * d;
    }
    return std::sqrt(d2);
}

// sum of the distances from atom `a` to the atoms of the subtile `t` other than slot `skip`
double dist_to_group(const double *xyz, uint32_t a, const uint32_t *t, int skip) {
    double s = 0.0;
    for (int k = 0; k < 8; k++)
        if (k != skip) s += dist(xyz, a, t[k]);
    return s;
}

// groups [g * size, (g + 1) * size) of `ids`, g < groups: for every group its `knn` nearest groups by centroid, nearest first
std::vector<uint32_t> nearest_groups(const double *xyz, const uint32_t *ids, size_t groups, int size, int knn) {
    std::vector<double> cen(3 * groups, 0.0);
    for (size_t g = 0; g < groups; g++)
        for (int k = 0; k < size; k++)
            for (int c = 0; c < 3; c++) cen[3 * g + c] += xyz[3 * (size_t)ids[g * size + k] + c] / size;
    const size_t take = std::min<size_t>((size_t)knn, groups - 1);
    std::vector<uint32_t> out(groups * (size_t)knn, std::numeric_limits<uint32_t>::max());
    std::vector<std::pair<double, uint32_t>> d(groups);
    for (size_t g = 0; g < groups; g++) {
        for (size_t h = 0; h < groups; h++) {
            double d2 = 0.0;
            for (int c = 0; c < 3; c++) d2 += (cen[3 * g + c] - cen[3 * h + c]) * (cen[3 * g + c] - cen[3 * h + c]);
            d[h] = {h == g ? 1e300 : d2, (uint32_t)h};
        }
        std::partial_sort(d.begin(), d.begin() + (long)take, d.end());
        for (size_t k = 0; k < take; k++) out[g * (size_t)knn + k] = d[k].second;
    }
    return out;
}

// One family of sweeps over the full subtiles [0, groups): atom i of subtile a against the eight atoms of a neighbouring
// subtile b > a, the best of the eight swaps taken when it lowers the two subtiles' summed cost.
void refine_all_subtiles(const double *xyz, uint32_t *ids, size_t groups, bool smooth) {
    if (groups < 2) return;
    std::vector<double> cost(groups);
    auto group_cost = [&](size_t g) {
        if (!smooth) return subtile_cost(xyz, ids + 8 * g);
        double s = 0.0;
        for (int i = 0; i < 8; i++)
            for (int j = i + 1; j < 8; j++) s += dist(xyz, ids[8 * g + i], ids[8 * g + j]);
        return s;
    };
    for (int pass = 0; pass < kGlobalPasses; pass++) {
        const std::vector<uint32_t> near = nearest_groups(xyz, ids, groups, 8, kNeighbours);
        for (size_t g = 0; g < groups; g++) cost[g] = group_cost(g);
        bool improved = false;
        for (size_t a = 0; a < groups; a++)
            for (int nb = 0; nb < kNeighbours; nb++) {
                const size_t b = near[a * kNeighbours + nb];
                if (b == std::numeric_limits<uint32_t>::max() || b < a) continue;
                uint32_t *ta = ids + 8 * a, *tb = ids + 8 * b;
                for (int i = 0; i < 8; i++) {
                    int best = -1;
                    double best_sum = cost[a] + cost[b] - 1e-9, best_a = 0.0, best_b = 0.0;
                    for (int j = 0; j < 8; j++) {
                        double ca, cb;
                        if (smooth) {
                            ca = cost[a] - dist_to_group(xyz, ta[i], ta, i) + dist_to_group(xyz, tb[j], ta, i);
                            cb = cost[b] - dist_to_group(xyz, tb[j], tb, j) + dist_to_group(xyz, ta[i], tb, j);
                        } else {
                            std::swap(ta[i], tb[j]);
                            ca = subtile_cost(xyz, ta);
                            cb = subtile_cost(xyz, tb);
                            std::swap(ta[i], tb[j]);
                        }
                        if (ca + cb < best_sum) {
                            best = j;
                            best_sum = ca + cb;
                            best_a = ca;
                            best_b = cb;
                        }
                    }
                    if (best >= 0) {
                        std::swap(ta[i], tb[best]);
                        cost[a] = best_a;
                        cost[b] = best_b;
                        improved = true;
                    }
                }
            }
        if (!improved) break;
    }
}

// Whole subtiles swapped between a tile and its nearest tiles while the two tiles' summed cost falls; a tile's cost is the sum
// of the 28 distances between its subtiles' centroids (independent of the frame: a ligand's tiles are boxed after posing).
void regroup_tiles(const double *xyz, uint32_t *ids, size_t tiles) {
    if (tiles < 2) return;
    std::vector<double> cen(tiles * 8 * 3), cost(tiles);
    std::vector<uint32_t> sub(tiles * 8);   // tile slot -> subtile of the incoming order
    for (size_t s = 0; s < tiles * 8; s++) {
        sub[s] = (uint32_t)s;
        for (int c = 0; c < 3; c++) {
            double m = 0.0;
            for (int k = 0; k < 8; k++) m += xyz[3 * (size_t)ids[8 * s + k] + c];
            cen[3 * s + c] = m / 8;
        }
    }
    auto cdist = [&](uint32_t p, uint32_t q) {
        double d2 = 0.0;
        for (int c = 0; c < 3; c++) d2 += (cen[3 * (size_t)p + c] - cen[3 * (size_t)q + c]) * (cen[3 * (size_t)p + c] - cen[3 * (size_t)q + c]);
        return std::sqrt(d2);
    };
    auto to_tile = [&](uint32_t p, const uint32_t *t, int skip) {
        double s = 0.0;
        for (int k = 0; k < 8; k++)
            if (k != skip) s += cdist(p, t[k]);
        return s;
    };
    for (int pass = 0; pass < kGlobalPasses; pass++) {
        // nearest tiles by the mean of their subtiles' centroids
        std::vector<double> tc(tiles * 3, 0.0);
        for (size_t t = 0; t < tiles; t++)
            for (int k = 0; k < 8; k++)
                for (int c = 0; c < 3; c++) tc[3 * t + c] += cen[3 * (size_t)sub[8 * t + k] + c] / 8;
        for (size_t t = 0; t < tiles; t++) {
            cost[t] = 0.0;
            for (int i = 0; i < 8; i++)
                for (int j = i + 1; j < 8; j++) cost[t] += cdist(sub[8 * t + i], sub[8 * t + j]);
        }
        bool improved = false;
        std::vector<std::pair<double, uint32_t>> d(tiles);
        const size_t take = std::min<size_t>((size_t)kTileNeighbours, tiles - 1);
        for (size_t a = 0; a < tiles; a++) {
            for (size_t h = 0; h < tiles; h++) {
                double d2 = 0.0;
                for (int c = 0; c < 3; c++) d2 += (tc[3 * a + c] - tc[3 * h + c]) * (tc[3 * a + c] - tc[3 * h + c]);
                d[h] = {h == a ? 1e300 : d2, (uint32_t)h};
            }
            std::partial_sort(d.begin(), d.begin() + (long)take, d.end());
            for (size_t nb = 0; nb < take; nb++) {
                const size_t b = d[nb].second;
                uint32_t *ta = sub.data() + 8 * a, *tb = sub.data() + 8 * b;
                for (int i = 0; i < 8; i++) {
                    int best = -1;
                    double best_sum = cost[a] + cost[b] - 1e-9, best_a = 0.0, best_b = 0.0;
                    for (int j = 0; j < 8; j++) {
                        const double ca = cost[a] - to_tile(ta[i], ta, i) + to_tile(tb[j], ta, i);
                        const double cb = cost[b] - to_tile(tb[j], tb, j) + to_tile(ta[i], tb, j);
                        if (ca + cb < best_sum) {
                            best = j;
                            best_sum = ca + cb;
                            best_a = ca;
                            best_b = cb;
                        }
                    }
                    if (best >= 0) {
                        std::swap(ta[i], tb[best]);
                        cost[a] = best_a;
                        cost[b] = best_b;
                        improved = true;
                    }
                }
            }
        }
        if (!improved) break;
    }
    std::vector<uint32_t> out(tiles * 64);
    for (size_t s = 0; s < tiles * 8; s++)
        for (int k = 0; k < 8; k++) out[8 * s + k] = ids[8 * (size_t)sub[s] + k];
    std::copy(out.begin(), out.end(), ids);
}

void refine_globally(const double *xyz, std::vector<uint32_t> &ids, size_t n) {
    const size_t tiles = n / 64;   // the full tiles; a short tail keeps the order the splits gave it
    refine_all_subtiles(xyz, ids.data(), tiles * 8, true);
    refine_all_subtiles(xyz, ids.data(), tiles * 8, false);
    regroup_tiles(xyz, ids.data(), tiles);
}

}  // namespace

std::vector<uint32_t> spatial_tile_order(const double *xyz, size_t n) {
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; i++) ids[i] = (uint32_t)i;
    split(xyz, ids, 0, n);
    refine_subtiles(xyz, ids, n);
    refine_globally(xyz, ids, n);
    const size_t padded = (n + 63) / 64 * 64;
    ids.resize(padded, std::numeric_limits<uint32_t>::max());
    return ids;
}

std::vector<uint32_t> pair_types_for_patches(const double *xyz, const uint32_t *types, const std::vector<uint32_t> &order,
                                             uint32_t n_types) {
    const uint32_t kPad = std::numeric_limits<uint32_t>::max();
    std::vector<double> w((size_t)n_types * n_types, 0.0);
    for (size_t base = 0; base + 8 <= order.size(); base += 8)
        for (int i = 0; i < 8; i++)
            for (int j = i + 1; j < 8; j++) {
                const uint32_t a = order[base + i], b = order[base + j];
                if (a == kPad || b == kPad) continue;
                const uint32_t ta = types[a], tb = types[b];
                if (ta == tb || ta >= n_types || tb >= n_types) continue;
                double d2 = 0.0;
                for (int c = 0; c < 3; c++) {
                    const double d = xyz[3 * (size_t)a + c] - xyz[3 * (size_t)b + c];
                    d2 += d * d;
                }
                const double v = 1.0 / (1.0 + d2);
                w[(size_t)ta * n_types + tb] += v;
                w[(size_t)tb * n_types + ta] += v;
            }
    struct Edge {
        double weight;
        uint32_t a, b;
    };
    std::vector<Edge> edges;
    for (uint32_t a = 0; a < n_types; a++)
        for (uint32_t b = a + 1; b < n_types; b++)
            if (w[(size_t)a * n_types + b] > 0.0) edges.push_back({w[(size_t)a * n_types + b], a, b});
    std::stable_sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) { return x.weight > y.weight; });
    std::vector<uint32_t> perm(n_types, kPad);
    uint32_t next = 0;
    for (const Edge &e : edges)
        if (perm[e.a] == kPad && perm[e.b] == kPad) {
            perm[e.a] = next;
            perm[e.b] = next + 1;
            next += 2;
        }
    for (uint32_t a = 0; a < n_types; a++)
        if (perm[a] == kPad) perm[a] = next++;
    return perm;
}

void refine_order_for_pairs(const double *xyz, const uint32_t *types, const std::vector<uint32_t> &perm,
                            std::vector<uint32_t> &order, size_t n, double mu) {
    auto cost_of = [&](const uint32_t *ids) {
        uint32_t pt[8];
        for (int k = 0; k < 8; k++) pt[k] = perm[types[ids[k]]];
        int lonely = 0;
        for (int k = 0; k < 8; k++) {
            bool found = false;
            for (int m = 0; m < 8; m++) found = found || pt[m] == (pt[k] ^ 1u);
            lonely += found ? 0 : 1;
        }
        return subtile_cost(xyz, ids) * (1.0 + mu * lonely / 8.0);
    };
    for (size_t base = 0; base + 64 <= n; base += 64) {
        uint32_t *t = order.data() + base;
        double cost[8];
        for (int g = 0; g < 8; g++) cost[g] = cost_of(t + 8 * g);
        for (int pass = 0; pass < 4; pass++) {
            bool improved = false;
            for (int i = 0; i < 64; i++)
                for (int j = (i / 8 + 1) * 8; j < 64; j++) {
                    const int a = i / 8, b = j / 8;
                    std::swap(t[i], t[j]);
                    const double ca = cost_of(t + 8 * a), cb = cost_of(t + 8 * b);
                    if (ca + cb < cost[a] + cost[b] - 1e-9) {
                        cost[a] = ca;
                        cost[b] = cb;
                        improved = true;
                    } else {
                        std::swap(t[i], t[j]);
                    }
                }
            if (!improved) break;
        }
    }
}

DfireTileLayout dfire_tile_layout(const double *xyz, const uint32_t *types, size_t n) {
    DfireTileLayout out;
    out.order = spatial_tile_order(xyz, n);
    out.type_perm = pair_types_for_patches(xyz, types, out.order, 169);
    refine_order_for_pairs(xyz, types, out.type_perm, out.order, n, 0.15);
    out.type_perm = pair_types_for_patches(xyz, types, out.order, 169);
    return out;
}

}  // namespace ld
