// nsf.h -- argument block and small helpers shared by the normalise -> weight -> sum kernels (fuse.hip, tables.hip).
#pragma once
#include "common.h"

namespace fz {

// cache-policy operand of global_load_lds on gfx940+: 2 = nt (the planes are streamed exactly once)
#define FZ_CPOL_NT 2

struct NsfArgs {
    const float* planes[FZ_MAX_SYSTEMS];
    const int32_t* ranks[FZ_MAX_SYSTEMS];
    const uint32_t* vbits[FZ_MAX_SYSTEMS];   // validity of system s as a bitmap [Q][ldb] (bit j & 31 of word j >> 5), built once per system
    int ldb;                                 //   (fz_rank_to_bitmap): 1/32 of the bytes of the rank plane it replaces in the fusion passes
    const float* distr[FZ_MAX_SYSTEMS];
    int P[FZ_MAX_SYSTEMS];
    float w[FZ_MAX_SYSTEMS];
    const float* sa[FZ_MAX_SYSTEMS];         // per-system row statistics [Q] (min | mean) and (max | unbiased std) for the flat passes: each system
    const float* sb[FZ_MAX_SYSTEMS];         //   brings its own (a by-product of the sort that ranked it), nothing is concatenated per fusion call
    int S, N, ld, Q;
};

// Barrier that does NOT drain pending LDS-DMA (a __syncthreads() would: hipcc emits vmcnt(0) in front of it while a
// global_load_lds is in flight, which would serialise the prefetch below).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// validity of the 4 columns j0 .. j0+3 (j0 % 4 == 0) of row q as a nibble: from the bitmap when the system has one, else from its rank plane
__device__ __forceinline__ uint32_t valid_nibble(const NsfArgs& a, int s, int q, size_t rowoff, int j0) {
    if (a.vbits[s]) return (a.vbits[s][(size_t)q * a.ldb + (j0 >> 5)] >> (j0 & 31)) & 0xfu;
    if (!a.ranks[s]) return 0xfu;
    const int4 r = *reinterpret_cast<const int4*>(a.ranks[s] + rowoff + j0);
    return (r.x >= 0 ? 1u : 0u) | (r.y >= 0 ? 2u : 0u) | (r.z >= 0 ? 4u : 0u) | (r.w >= 0 ? 8u : 0u);
}

// fills the argument block from the C-ABI arrays; FZ_OK or FZ_ERR_ARG (fuse.hip)
int nsf_fill_args(NsfArgs& a, const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N, int ld,
                  bool needs_distr, const float* const* distr_h, const int32_t* P_h, const uint32_t* const* valid_bits_h, int ldb);
// the LDS-resident table kernel (all S tables at once): returns 1 when it cannot take the call (fuse.hip)
int launch_nsf_tables(const NsfArgs& a, bool nce, int Q, float* fused, hipStream_t st);
// true when launch_nsf_tables would take the call
bool nsf_tables_fit_lds(const NsfArgs& a, const float* fused);

}  // namespace fz
