// rtiow_rng.h — the counter-keyed RNG of the PATH mode (BUILD-SPEC, SURVEY.md
// section 7 hard part 2: the reference has no RNG at all, and the book's
// sequential rand() cannot be reproduced on a GPU).  One PCG-RXS-M-XS-32
// stream per (seed, pixel, sample); a path consumes its stream sequentially,
// so the image does not depend on which lane or GPU traced the path.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RTIOW_HD __host__ __device__ __forceinline__
#else
#define RTIOW_HD inline
#endif

namespace rtiow {

RTIOW_HD uint32_t pcg_advance(uint32_t s) { return s * 747796405u + 2891336453u; }

RTIOW_HD uint32_t pcg_permute(uint32_t s) {
    const uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

RTIOW_HD uint32_t pcg_seed(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t s = pcg_permute(pcg_advance(seed));
    s = pcg_permute(pcg_advance(s ^ pixel));
    return pcg_permute(pcg_advance(s ^ sample));
}

struct Pcg {
    uint32_t state;
    RTIOW_HD Pcg(uint32_t seed, uint32_t pixel, uint32_t sample)
        : state(pcg_seed(seed, pixel, sample)) {}
    RTIOW_HD explicit Pcg(uint32_t raw) : state(raw) {}
    RTIOW_HD uint32_t next() {
        state = pcg_advance(state);
        return pcg_permute(state);
    }
    // [0,1) with 24 random bits: exact in binary32
    RTIOW_HD float uniform() { return static_cast<float>(next() >> 8) * 0x1p-24f; }
    // [-1,1): 2u-1 is exact
    RTIOW_HD float symmetric() { return 2.0f * uniform() - 1.0f; }
};

}  // namespace rtiow
