// Work layout of the kernels that carry 32 horizontally adjacent pixels in the bits of a register and walk a
// strip of image rows: which lane of which wave has which 32-pixel word of which strip.
//
// A lane needs the words left and right of its own (window columns are funnel shifts), and gets them from the
// neighbouring lanes of its wave, so a wave covers consecutive words of one row band and its first and last lane
// only serve as neighbours.  A row is cut into segments of 62 words (64 lanes); what is left of it (fewer than 62
// words, + 2 lanes) is packed several strips to a wave.  Everything is on the flat view of the image: word j of
// row y is the 32 pixels from flat offset y * W + 32 j, whether or not they all lie in that row.
#pragma once
#include "iak_common.hpp"

namespace iak {

struct BitRows {
    int rows;        // output rows per strip
    int strips;      // strips per image
    int words;       // 32-pixel words per image row: ceil(W / 32)
    int nfull;       // 62-word segments per row
    int rem_lanes;   // lanes of the row's remaining segment (its words + 2), 0 if there is none
    int pack;        // remaining segments (of different strips) per wave
    int waves_full;  // strips * nfull
    int waves;       // waves per image
    int images;

    static int waves_per_image(int W, int H, int rows_) {
        const int words_ = (W + 31) / 32, nfull_ = words_ / 62, rem = words_ - 62 * nfull_;
        const int strips_ = (H + rows_ - 1) / rows_;
        return strips_ * nfull_ + (rem ? (strips_ + 64 / (rem + 2) - 1) / (64 / (rem + 2)) : 0);
    }
    static BitRows make(int W, int H, int B, int rows_) {
        BitRows g;
        g.rows = rows_;
        g.strips = (H + rows_ - 1) / rows_;
        g.words = (W + 31) / 32;
        g.nfull = g.words / 62;
        const int rem = g.words - 62 * g.nfull;
        g.rem_lanes = rem ? rem + 2 : 0;
        g.pack = rem ? 64 / g.rem_lanes : 0;
        g.waves_full = g.strips * g.nfull;
        g.waves = waves_per_image(W, H, rows_);
        g.images = B;
        return g;
    }
    unsigned grid_blocks() const { return (unsigned)(((long long)waves * images + 3) / 4); }  // 256-lane workgroups

    // gwave: blockIdx.x * 4 + wave of the workgroup, as a wave-uniform value (read through readfirstlane: the
    // image's buffer descriptor depends on it and must live in scalar registers).  False for a wave without work (uniform over the wave).
    // strip: may be >= strips for lanes of a packed wave that have none; word: -1 .. words (the row's neighbours
    // included); owns: the lane stores its word's results.
    __device__ __forceinline__ bool locate(int gwave, int lane, int *image, int *strip, int *word, bool *owns) const {
        const int im = gwave / waves, wave = gwave - im * waves;
        if (im >= images) return false;
        *image = im;
        bool o;
        if (wave < waves_full) {
            const int st = wave / nfull, seg = wave - st * nfull;
            *strip = st;
            *word = seg * 62 + lane - 1;
            o = lane >= 1 && lane <= 62;
        } else {
            const int sub = lane / rem_lanes, li = lane - sub * rem_lanes;
            *strip = (wave - waves_full) * pack + sub;
            *word = nfull * 62 + li - 1;
            o = sub < pack && *strip < strips && li >= 1 && li <= rem_lanes - 2;
        }
        *owns = o && *word < words;
        return true;
    }
};

// the word of the lane below / above in the wave (zero for the wave's first / last lane)
__device__ __forceinline__ uint32_t word_from_left(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true); }   // wave_shr:1
__device__ __forceinline__ uint32_t word_from_right(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true); }  // wave_shl:1

}  // namespace iak
