// HBM layout of the profile arena.  Everything a sweep streams is stored "tile-transposed":
// 64 consecutive node ids form a tile, and inside a tile the 64 nodes' values for one alignment column are
// contiguous, so that a wavefront whose lane l owns node 64*t+l issues one fully coalesced 16-byte-per-lane
// load per column group while every lane walks its own profile in the reference's column order.
// Dense ML rows.  The tile streams are built for the one-vs-all sweeps of the NJ phase; rewriting ONE node means
// re-packing its tile (k_tile_commit: ~30 us at 200 columns, ~250 us at 1000).  The ML phase rewrites single nodes all
// the time (recomputeProfile after every split, up-profiles) and never sweeps, so its writes go to a plain row per
// internal node instead: mlW[(node - nSeqs) * nPos + p], mlC[...] (code), mlF[...][nCodes] (valid when the code is
// NOCODE and the weight positive), with one byte mlIs[node - nSeqs] saying that the node's current profile is that
// row.  ML kernels read through vft_load_col_ml (row if flagged, tile streams otherwise: leaves and the NJ-phase
// averages the first ML round starts from); any write through the tile path (k_tile_commit) clears the flag.  NJ-phase
// kernels never look at the rows.  Allocated on the first ML-phase write ((maxNodes - nSeqs) * nPos * (nCodes + 1)
// numeric_t + 1 byte per column).
#pragma once
#include <stdint.h>

#define VFT_TILE 64        // nodes per tile = wavefront width on gfx950
#define VFT_CHUNK 16       // alignment columns per 16-byte code chunk
#define VFT_NOCODE_ 127

#if defined(__HIPCC__)
#define VFT_HD __host__ __device__ __forceinline__
#else
#define VFT_HD inline
#endif

struct VftDims {
    int64_t nSeqs, nPos, maxNodes;
    int32_t nCodes;
    int32_t nChunk;        // ceil(nPos / 16)
    int64_t nPosPad;       // nChunk * 16: column stride of the internal-profile arrays; padding columns keep
                           // weight 0 / mask 0, so kernels may run whole chunks without bounds checks
    int64_t firstProfTile; // nSeqs / 64: first tile that can hold an internal node
    int64_t nTiles;        // ceil(maxNodes / 64)
};

// ---- leaf codes: uint4 leafT[tile][chunk][lane]; byte b of the uint4 is column chunk*16+b (encoded, see vft_encode)
VFT_HD int64_t vft_leaf_idx(const VftDims &d, int64_t tile, int32_t chunk, int32_t lane) {
    return (tile * d.nChunk + chunk) * VFT_TILE + lane;
}

// nt leaves are stored as 0x10 | (1 << code) and gaps as 0, so that for two encoded bytes a,b:
//   (a & b & 0x10) != 0  <=> both present,   (a & b & 0x0F) != 0 <=> same base
// aa leaves keep the reference code (0..19, 127).
VFT_HD uint8_t vft_encode(uint8_t code, int nCodes) {
    if (nCodes != 4) return code;
    return code == VFT_NOCODE_ ? 0 : (uint8_t) (0x10 | (1u << code));
}

// ---- internal profiles (tile index is relative to firstProfTile)
//   uint4 profC[ptile][chunk][lane]      raw reference codes, 16 columns per uint4, dense
//   ColMask colMask[ptile][pos]          .vec: bit l set <=> node 64*tile+l holds a frequency vector at this column
//                                        .w:   bit l set <=> node 64*tile+l stores an EXPLICIT weight at this column.
//                                              Without the bit the weight is implied: 1 if the column holds a code
//                                              or a vector, 0 if it is an empty gap (what leaves and almost every
//                                              column of a low-gap alignment have, NJ.tcc:2078-2112)
//   ColOff  colOff[ptile][pos]           .vec / .w: how many vectors / explicit weights the tile holds in the columns
//                                              before this one (exclusive prefix sums of the mask popcounts)
//   REAL  profF[ptile][slot][nCodes]     the tile's vectors as ONE contiguous stream in (column, lane) order: the
//                                        vector of lane l at column p sits in slot
//                                              colOff[p].vec + popcount(colMask[p].vec & ((1<<l)-1))
//                                        (20-state alphabets: inside a column's block of slots the vectors are
//                                        transposed in 16-byte pieces, vft_fidx in vft_device.h)
//   REAL  profW[ptile][slot]             the tile's explicit weights, same scheme with the .w members
// A wavefront walking the columns of its tile therefore reads exactly the vectors that exist (the reference's
// sparse profiles, NJ.h:126-141) as one ascending, hole-free address stream.  Streams are rebuilt per tile by
// k_tile_commit whenever nodes of the tile are written (vft_kernels_profile.h); capacity is the dense worst case.
struct ColMask { unsigned long long vec, w; };
struct ColOff { uint32_t vec, w; };
VFT_HD int64_t vft_meta_idx(const VftDims &d, int64_t ptile, int64_t pos) { return ptile * d.nPosPad + pos; }
VFT_HD int64_t vft_wstream_base(const VftDims &d, int64_t ptile) { return ptile * d.nPosPad * VFT_TILE; }
VFT_HD int64_t vft_fstream_base(const VftDims &d, int64_t ptile) { return ptile * d.nPosPad * VFT_TILE * d.nCodes; }
VFT_HD int64_t vft_c_idx(const VftDims &d, int64_t ptile, int32_t chunk, int32_t lane) {
    return (ptile * d.nChunk + chunk) * VFT_TILE + lane;
}
