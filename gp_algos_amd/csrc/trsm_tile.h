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

}  // namespace
