// Row-panel solve building blocks on v_mfma_f64_16x16x4_f64 (gfx950), shared by trsm_panel128_kernel / ep_link_kernel
// (kernels_diag.hip) and the fused EP block kernel (gpcore_ep.hip): X (rows x 128, an LDS strip with column stride XS) <- X L^-T
// with L a 128 x 128 lower-triangular block whose fragments come from L2 and whose 16 x 16 diagonal tiles are applied as inverses.
#pragma once
#include <hip/hip_runtime.h>

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

constexpr int TRSM_XS = 80;      // LDS column stride of the 64-row X strip: 640 B = 128 (mod 256)

// tile index q of a lower-triangular enumeration -> (bi, bj), bi >= bj
__device__ __forceinline__ void tri_coords(int q, int &bi, int &bj) {
    bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= q) ++bi;
    bj = q - bi * (bi + 1) / 2;
}

// L fragments of chunk CB (rows 16*CB.., all previous columns): 4*CB values per lane, A-operand layout.
template <int CB>
__device__ __forceinline__ void trsm_load_frags(const double *__restrict__ L, int ldl, int fr, int fg, double (&lf)[28]) {
#pragma unroll
    for (int q = 0; q < 4 * CB; ++q) lf[q] = L[(16 * CB + fr) + (size_t)(4 * q + fg) * ldl];
}

// one 16-column chunk: R = B_cb - X_prev L(cb,prev)^T on the matrix cores, then X_cb = invD_cb R with the
// accumulator registers fed straight back as the B operand (register r carries k = fg + 4r).
template <int CB, int XS = TRSM_XS>
__device__ __forceinline__ void trsm_chunk(double *xs, int sp, int fr, int fg, const double (&lf)[28],
                                           const double *__restrict__ dinv, double &ss) {
    constexpr int c0 = 16 * CB;
    double4_t acc0, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = xs[(c0 + fg + 4 * r) * XS + sp];
    double dq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dq[r] = dinv[CB * 256 + fr + 16 * (fg + 4 * r)];
#pragma unroll
    for (int q = 0; q < 4 * CB; q += 2) {
        acc0 = MFMA(-lf[q], xs[(4 * q + fg) * XS + sp], acc0);
        acc1 = MFMA(-lf[q + 1], xs[(4 * q + 4 + fg) * XS + sp], acc1);
    }
    acc0 += acc1;
    double4_t nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) nw = MFMA(dq[r], acc0[r], nw);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        xs[(c0 + fg + 4 * r) * XS + sp] = nw[r];
        ss = fma(nw[r], nw[r], ss);
    }
}


// the same chunk with the tile inverse already in registers (dq[r] = dinv[CB * 256 + fr + 16 * (fg + 4 r)], fetched up front by
// the caller: on a latency-bound single-workgroup path the four global loads at the head of every chunk are four L2 round trips)
template <int CB, int XS = TRSM_XS>
__device__ __forceinline__ void trsm_chunk_pre(double *xs, int sp, int fr, int fg, const double (&lf)[28], const double (&dq)[4]) {
    constexpr int c0 = 16 * CB;
    double4_t acc0, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = xs[(c0 + fg + 4 * r) * XS + sp];
#pragma unroll
    for (int q = 0; q < 4 * CB; q += 2) {
        acc0 = MFMA(-lf[q], xs[(4 * q + fg) * XS + sp], acc0);
        acc1 = MFMA(-lf[q + 1], xs[(4 * q + 4 + fg) * XS + sp], acc1);
    }
    acc0 += acc1;
    double4_t nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) nw = MFMA(dq[r], acc0[r], nw);
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[(c0 + fg + 4 * r) * XS + sp] = nw[r];
}


// two 16-row sub-strips (rows spa, spb) through chunk CB at once: four independent accumulator chains, and the LDS operands of
// step q + 2 requested BEFORE the MFMAs of step q are issued (explicit double buffering, fenced with sched_barrier: left to itself
// the compiler reads, waits out the full LDS latency and only then issues the MFMAs -- ~110 cycles per 64-cycle MFMA for a wave
// that has its SIMD to itself, as every wave of the fused EP block kernel has)
template <int CB, int XS = TRSM_XS>
__device__ __forceinline__ void trsm_chunk_pre2(double *xs, int spa, int spb, int fr, int fg, const double (&lf)[28], const double (&dq)[4]) {
    constexpr int c0 = 16 * CB;
    double4_t a0, a1 = {0.0, 0.0, 0.0, 0.0}, b0, b1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = xs[(c0 + fg + 4 * r) * XS + spa], b0[r] = xs[(c0 + fg + 4 * r) * XS + spb];
    if constexpr (CB > 0) {
        double cur[4], nxt[4];
        cur[0] = xs[fg * XS + spa], cur[1] = xs[fg * XS + spb], cur[2] = xs[(4 + fg) * XS + spa], cur[3] = xs[(4 + fg) * XS + spb];
#pragma unroll
        for (int q = 0; q < 4 * CB; q += 2) {
            if (q + 2 < 4 * CB) {
                nxt[0] = xs[(4 * (q + 2) + fg) * XS + spa], nxt[1] = xs[(4 * (q + 2) + fg) * XS + spb];
                nxt[2] = xs[(4 * (q + 2) + 4 + fg) * XS + spa], nxt[3] = xs[(4 * (q + 2) + 4 + fg) * XS + spb];
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = MFMA(-lf[q], cur[0], a0);
            b0 = MFMA(-lf[q], cur[1], b0);
            a1 = MFMA(-lf[q + 1], cur[2], a1);
            b1 = MFMA(-lf[q + 1], cur[3], b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
        }
    }
    a0 += a1, b0 += b1;
    double4_t na = {0.0, 0.0, 0.0, 0.0}, nb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) na = MFMA(dq[r], a0[r], na), nb = MFMA(dq[r], b0[r], nb);
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[(c0 + fg + 4 * r) * XS + spa] = na[r], xs[(c0 + fg + 4 * r) * XS + spb] = nb[r];
}

}  // namespace
