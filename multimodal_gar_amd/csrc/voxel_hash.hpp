// voxel_hash.hpp -- device-side lookup of the voxel hash table built by mgar_voxel_hash_build (csrc/sparse_conv.hip):
// open addressing, linear probing, 64-bit keys ((b*Z + z)*Y + y)*X + x -> row id.  Shared by the rulebook kernels and the
// hash-table flavour of the voxel query (csrc/voxel_query.hip).
#pragma once
#include "common.hpp"

namespace mgar {

constexpr long long SPH_EMPTY = -1LL;

__device__ __forceinline__ unsigned long long sph_mix(unsigned long long k) {   // murmur3 finaliser
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

__device__ __forceinline__ int sph_find(const long long *__restrict__ tkeys, const int *__restrict__ tvals, int mask, long long key) {
    unsigned slot = (unsigned)sph_mix((unsigned long long)key) & (unsigned)mask;
    for (int probe = 0; probe <= mask; ++probe) {
        const long long k = tkeys[slot];
        if (k == key) return tvals[slot];
        if (k == SPH_EMPTY) return -1;
        slot = (slot + 1) & (unsigned)mask;
    }
    return -1;
}

// coords (N, 4) int32 [b, z, y, x] -> key, or -1 if outside [0, Z) x [0, Y) x [0, X)
__device__ __forceinline__ long long sph_key(int b, int z, int y, int x, int Z, int Y, int X) {
    if (z < 0 || y < 0 || x < 0 || z >= Z || y >= Y || x >= X || b < 0) return SPH_EMPTY;
    return (((long long)b * Z + z) * Y + y) * X + x;
}


}  // namespace mgar
