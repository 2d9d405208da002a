#include "spatial_order.hpp"

#include <algorithm>
#include <limits>

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

}  // namespace

std::vector<uint32_t> spatial_tile_order(const double *xyz, size_t n) {
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; i++) ids[i] = (uint32_t)i;
    split(xyz, ids, 0, n);
    const size_t padded = (n + 63) / 64 * 64;
    ids.resize(padded, std::numeric_limits<uint32_t>::max());
    return ids;
}

}  // namespace ld
