// test harness: the Gibbs plan's host-side colouring (sailfish_amd/csrc/colour.h) as a plain C function
#include "../sailfish_amd/csrc/colour.h"
extern "C" uint32_t colour_classes(const uint32_t* list, uint64_t n, const uint32_t* rowptr, uint64_t C, const uint32_t* ids, uint64_t L,
                                   uint64_t M, uint32_t* colour_out) {
    std::vector<uint32_t> wl(list, list + n), rp(rowptr, rowptr + C + 1), id(ids, ids + L), col;
    const uint32_t k = sfgpu::colour_wide_classes(wl, rp, id, M, col);
    for (uint64_t i = 0; i < n; ++i) colour_out[i] = col[i];
    return k;
}
extern "C" uint32_t class_components(const uint32_t* list, uint64_t n, const uint32_t* rowptr, uint64_t C, const uint32_t* ids, uint64_t L,
                                     uint64_t M, uint32_t* comp_out) {
    std::vector<uint32_t> wl(list, list + n), rp(rowptr, rowptr + C + 1), id(ids, ids + L), comp;
    const uint32_t k = sfgpu::components_of_wide_classes(wl, rp, id, M, comp);
    for (uint64_t i = 0; i < n; ++i) comp_out[i] = comp[i];
    return k;
}
