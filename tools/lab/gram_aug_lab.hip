// Tuning harness (not part of the product): wave-private Gram tiles with the whole exponent on the matrix cores.
//   z~_i = (z_i, -|z_i|^2/2, 1),  z~_j = (z_j, 1, -|z_j|^2/2)   ->   z~_i . z~_j = -r_ij^2 / 2
// One wave = one 64 x 64 tile, operands straight from global memory in MFMA layout, no LDS tile, no barrier.
// build: hipcc --offload-arch=gfx950 -O3 -o gram_aug_lab gram_aug_lab.hip ; run: ./gram_aug_lab [m] [n] [spread]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct GP { double sf2, sn2, extra; double inv_ls[16]; };

__device__ __forceinline__ void tile_lower(int t, int &bi, int &bj) {
    int b = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (b * (b + 1) / 2 > t) --b;
    while ((b + 1) * (b + 2) / 2 <= t) ++b;
    bi = b;
    bj = t - b * (b + 1) / 2;
}

// one operand block: 16 points from p0 (point of this lane: p0 + perm(fr)), KS k-steps; ROW role puts (-|z|^2/2, 1) behind the features,
// COL role (1, -|z|^2/2)
template <int KS, bool ROW, int VAR = 0>
__device__ __forceinline__ void load_block(const double *__restrict__ X, int n, int ldx, int p, int d, const GP &prm, const double *__restrict__ cen, int ldcen,
                                           int fk, double (&z)[KS]) {
    const double *xp = X + min(p, n - 1);
    double part = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + fk;
        double v = 0.0;
        if (VAR & 1) v = 1e-3 * (double)((p * 7 + k * 13) & 1023);      // no vector loads at all
        else if (k < d) v = (xp[(size_t)k * ldx] - cen[(size_t)k * ldcen]) * prm.inv_ls[k];
        z[ks] = v;
        part = fma(v, v, part);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    const double hn = -0.5 * part;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + fk;
        if (k == d) z[ks] = ROW ? hn : 1.0;
        if (k == d + 1) z[ks] = ROW ? 1.0 : hn;
    }
}

// MODE 0: cross (all tiles, no symmetry); 1: symmetric lower only; 2: symmetric, mirrored
// VAR bits: 1 operands computed from indices (no vector loads); 2 one store per wave (compute only); 4 nontemporal stores; 8 no exp; 16 no MFMA
template <int KS, int MODE, int WPB, int VAR = 0>
__global__ __launch_bounds__(64 * WPB) void ga(const double *__restrict__ Xr, int nr, int ldxr, const double *__restrict__ Xc, int nc, int ldxc, int d, GP prm,
                                               const double *__restrict__ cen, int ldcen, double *__restrict__ K, int ldk, int nbr, int ntiles) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    const int t = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    int bi, bj;
    if (MODE) tile_lower(t, bi, bj);
    else { bi = t % nbr; bj = t / nbr; }
    const int i0 = bi * 64, j0 = bj * 64;
    // column points are fed in the order that leaves each lane with 4 CONSECUTIVE columns (j = 4 fk + r) when the mirror is wanted
    const int cperm = (MODE == 2) ? 4 * (fr & 3) + (fr >> 2) : fr;
    double zr[4][KS], zc[4][KS];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        load_block<KS, true, VAR>(Xr, nr, ldxr, i0 + 16 * b + fr, d, prm, cen, ldcen, fk, zr[b]);
        load_block<KS, false, VAR>(Xc, nc, ldxc, j0 + 16 * b + cperm, d, prm, cen, ldcen, fk, zc[b]);
    }
    const bool diag = MODE && bi == bj;
    const bool plain = !diag && i0 + 64 <= nr && j0 + 64 <= nc;
    const double dval = (prm.sf2 + prm.sn2) + prm.extra;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        d4 acc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[it] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (VAR & 16) acc[it][ks & 3] -= zc[cb][ks] * zr[it][ks];
                else acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[cb][ks], zr[it][ks], acc[it], 0, 0, 0);
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (MODE == 1 && diag && it < cb) continue;    // block above the diagonal
            const int gi = i0 + 16 * it + fr;
            d4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (VAR & 8) ? acc[it][r] : prm.sf2 * exp(fmin(acc[it][r], 0.0));
            const int jb = j0 + 16 * cb + ((MODE == 2) ? 4 * fk : fk);
            constexpr int JS = (MODE == 2) ? 1 : 4;     // column step between a lane's registers
            double *Kp = K + gi + (size_t)jb * ldk;
            if (VAR & 2) {
                if (v[0] + v[1] + v[2] + v[3] == 123.456) Kp[0] = v[0];
            } else if (plain && (VAR & 4)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(v[r], Kp + (size_t)(JS * r) * ldk);
            } else if (plain) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Kp[(size_t)(JS * r) * ldk] = v[r];
                if (MODE == 2) *reinterpret_cast<d4 *>(K + jb + (size_t)gi * ldk) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gj = jb + JS * r;
                    if (MODE && gi == gj) v[r] = dval;
                    if (gi < nr && gj < nc && !(diag && gi < gj)) Kp[(size_t)(JS * r) * ldk] = v[r];
                    if (MODE == 2 && gi < nr && gj < nc && (!diag || gi > gj)) K[gj + (size_t)gi * ldk] = v[r];
                }
            }
        }
    }
}

// ---- row strips: one wave keeps its 64 row points and walks CH column tiles, the next tile's raw features are requested before the
// current tile's 64 stores are issued (a load issued behind the CU's queued stores takes microseconds to come back)
template <int KS>
__device__ __forceinline__ void raw_load(const double *__restrict__ X, int n, int ldx, int p, int d, int fk, double (&raw)[KS]) {
    const double *xp = X + min(p, n - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) raw[ks] = xp[(size_t)min(4 * ks + fk, d - 1) * ldx];
}
template <int KS, bool ROW>
__device__ __forceinline__ void finish_block(const double (&raw)[KS], int d, const double (&cen)[KS], const double (&il)[KS], int fk, double (&z)[KS]) {
    double part = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const double v = (4 * ks + fk < d) ? (raw[ks] - cen[ks]) * il[ks] : 0.0;
        z[ks] = v;
        part = fma(v, v, part);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    const double hn = -0.5 * part;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + fk;
        if (k == d) z[ks] = ROW ? hn : 1.0;
        if (k == d + 1) z[ks] = ROW ? 1.0 : hn;
    }
}

template <int KS, int MODE>
__global__ __launch_bounds__(64) void gs(const double *__restrict__ Xr, int nr, int ldxr, const double *__restrict__ Xc, int nc, int ldxc, int d, GP prm,
                                         const double *__restrict__ cenp, int ldcen, double *__restrict__ K, int ldk, int nbr, int nbc, int CH) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    int bi, c0, c1;
    if (MODE == 0) { bi = blockIdx.x % nbr; c0 = (blockIdx.x / nbr) * CH; c1 = min(c0 + CH, nbc); }
    else {
        // strips g*CH .. g*CH+CH-1 have g+1 chunks each: jobs before group g = CH g (g+1) / 2
        const int t = blockIdx.x;
        int g = (int)((sqrt(8.0 * (double)t / CH + 1.0) - 1.0) * 0.5);
        while (CH * g * (g + 1) / 2 > t) --g;
        while (CH * (g + 1) * (g + 2) / 2 <= t) ++g;
        const int rem = t - CH * g * (g + 1) / 2;
        bi = g * CH + rem / (g + 1);
        c0 = (rem % (g + 1)) * CH;
        c1 = min(c0 + CH, bi + 1);
        if (bi >= nbr) return;
    }
    const int i0 = bi * 64;
    const int cperm = (MODE == 2) ? 4 * (fr & 3) + (fr >> 2) : fr;
    double cen[KS], il[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = min(4 * ks + fk, d - 1);
        cen[ks] = cenp[(size_t)k * ldcen];
        il[ks] = prm.inv_ls[k];
    }
    double zr[4][KS], zc[4][KS], raw[4][KS];
#pragma unroll
    for (int b = 0; b < 4; ++b) raw_load<KS>(Xr, nr, ldxr, i0 + 16 * b + fr, d, fk, raw[b]);
#pragma unroll
    for (int b = 0; b < 4; ++b) finish_block<KS, true>(raw[b], d, cen, il, fk, zr[b]);
#pragma unroll
    for (int b = 0; b < 4; ++b) raw_load<KS>(Xc, nc, ldxc, c0 * 64 + 16 * b + cperm, d, fk, raw[b]);
    const double dval = (prm.sf2 + prm.sn2) + prm.extra;
    for (int c = c0; c < c1; ++c) {
#pragma unroll
        for (int b = 0; b < 4; ++b) finish_block<KS, false>(raw[b], d, cen, il, fk, zc[b]);
        if (c + 1 < c1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) raw_load<KS>(Xc, nc, ldxc, (c + 1) * 64 + 16 * b + cperm, d, fk, raw[b]);
        }
        const int j0 = c * 64;
        const bool diag = MODE && bi == c;
        const bool plain = !diag && i0 + 64 <= nr && j0 + 64 <= nc;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            d4 acc[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) acc[it] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[cb][ks], zr[it][ks], acc[it], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (MODE == 1 && diag && it < cb) continue;
                const int gi = i0 + 16 * it + fr;
                d4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = prm.sf2 * exp(fmin(acc[it][r], 0.0));
                const int jb = j0 + 16 * cb + ((MODE == 2) ? 4 * fk : fk);
                constexpr int JS = (MODE == 2) ? 1 : 4;
                double *Kp = K + gi + (size_t)jb * ldk;
                if (plain) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Kp[(size_t)(JS * r) * ldk] = v[r];
                    if (MODE == 2) *reinterpret_cast<d4 *>(K + jb + (size_t)gi * ldk) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gj = jb + JS * r;
                        if (MODE && gi == gj) v[r] = dval;
                        if (gi < nr && gj < nc && !(diag && gi < gj)) Kp[(size_t)(JS * r) * ldk] = v[r];
                        if (MODE == 2 && gi < nr && gj < nc && (!diag || gi > gj)) K[gj + (size_t)gi * ldk] = v[r];
                    }
                }
            }
        }
    }
}

// ---- unit ranges: a unit = 64 rows x 16 columns of one row strip; every wave takes `upw` consecutive units of the strip-major order
// (lower forms: strip b has 4 (b + 1) units), keeps the strip's row operands, asks for the next tile's column features one tile ahead,
// stores through a uniform base + 32-bit lane offset, and has ln(sf^2) folded into the column operand's norm slot
template <int KS, int KR, bool ROW>
__device__ __forceinline__ void finish2(const double (&raw)[KR], int d, const double (&cen)[KR], const double (&il)[KR], int fk, double lnsf2, double (&z)[KS]) {
    double part = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) z[ks] = 0.0;
#pragma unroll
    for (int ks = 0; ks < KR; ++ks) {
        const double v = (4 * ks + fk < d) ? (raw[ks] - cen[ks]) * il[ks] : 0.0;
        z[ks] = v;
        part = fma(v, v, part);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    const double hn = ROW ? -0.5 * part : fma(-0.5, part, lnsf2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + fk;
        if (k == d) z[ks] = ROW ? hn : 1.0;
        if (k == d + 1) z[ks] = ROW ? 1.0 : hn;
    }
}

template <int KS, int KR, int MODE>
__global__ __launch_bounds__(64) void gu(const double *__restrict__ Xr, int nr, int ldxr, const double *__restrict__ Xc, int nc, int ldxc, int d, GP prm,
                                         const double *__restrict__ cenp, int ldcen, double *__restrict__ K, int ldk, int nbr, int nbc, int upw, long total) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    long u0 = (long)blockIdx.x * upw, u1 = u0 + upw < total ? u0 + upw : total;
    if (u0 >= u1) return;
    const int cperm = (MODE == 2) ? 4 * (fr & 3) + (fr >> 2) : fr;
    double cen[KR], il[KR];
#pragma unroll
    for (int ks = 0; ks < KR; ++ks) {
        const int k = min(4 * ks + fk, d - 1);
        cen[ks] = cenp[(size_t)k * ldcen];
        il[ks] = prm.inv_ls[k];
    }
    const double lnsf2 = log(prm.sf2), dval = (prm.sf2 + prm.sn2) + prm.extra;
    const unsigned voff = (unsigned)(fr + (size_t)((MODE == 2) ? 4 * fk : fk) * ldk) * 8u;     // lane part of the primary store
    const unsigned moff = (unsigned)(4 * fk + (size_t)fr * ldk) * 8u;                            // lane part of the mirrored store
    constexpr int JS = (MODE == 2) ? 1 : 4;
    double zr[4][KS], zc[4][KS], raw[4][KR];
    while (u0 < u1) {
        // the strip of u0 and this wave's column-block range in it
        int bi; long ub;
        if (MODE == 0) { bi = (int)(u0 / (4 * nbc)); ub = (long)bi * 4 * nbc; }
        else {
            bi = (int)((sqrt(2.0 * (double)u0 + 1.0) - 1.0) * 0.5);
            while (2L * bi * (bi + 1) > u0) --bi;
            while (2L * (bi + 1) * (bi + 2) <= u0) ++bi;
            bi = __builtin_amdgcn_readfirstlane(bi);     // uniform by construction; the sqrt left it in a vector register
            ub = 2L * bi * (bi + 1);
        }
        const int slen = (MODE == 0) ? 4 * nbc : 4 * (bi + 1);
        const int q0 = (int)(u0 - ub), q1 = (int)((u1 - ub < slen) ? u1 - ub : slen);
        const int i0 = bi * 64;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double *xp = Xr + min(i0 + 16 * b + fr, nr - 1);
#pragma unroll
            for (int ks = 0; ks < KR; ++ks) raw[b][ks] = xp[(size_t)min(4 * ks + fk, d - 1) * ldxr];
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) finish2<KS, KR, true>(raw[b], d, cen, il, fk, 0.0, zr[b]);
        const int c0 = q0 >> 2, c1 = (q1 + 3) >> 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double *xp = Xc + min(c0 * 64 + 16 * b + cperm, nc - 1);
#pragma unroll
            for (int ks = 0; ks < KR; ++ks) raw[b][ks] = xp[(size_t)min(4 * ks + fk, d - 1) * ldxc];
        }
        for (int c = c0; c < c1; ++c) {
#pragma unroll
            for (int b = 0; b < 4; ++b) finish2<KS, KR, false>(raw[b], d, cen, il, fk, lnsf2, zc[b]);
            if (c + 1 < c1) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double *xp = Xc + min((c + 1) * 64 + 16 * b + cperm, nc - 1);
#pragma unroll
                    for (int ks = 0; ks < KR; ++ks) raw[b][ks] = xp[(size_t)min(4 * ks + fk, d - 1) * ldxc];
                }
            }
            const int j0 = c * 64;
            const bool diag = MODE && bi == c;
            const bool plain = !diag && i0 + 64 <= nr && j0 + 64 <= nc && (MODE != 2 || (ldk & 3) == 0);
            const int cbs = max(q0 - 4 * c, 0), cbe = min(q1 - 4 * c, 4);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if (cb < cbs || cb >= cbe) continue;
                d4 acc[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[cb][ks], zr[it][ks], acc[it], 0, 0, 0);
                char *cbase = reinterpret_cast<char *>(K + i0 + (size_t)(j0 + 16 * cb) * ldk);
                char *mbase = reinterpret_cast<char *>(K + (j0 + 16 * cb) + (size_t)i0 * ldk);
                if (plain) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        d4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = exp(fmin(acc[it][r], lnsf2));
#pragma unroll
                        for (int r = 0; r < 4; ++r) *reinterpret_cast<double *>(cbase + ((size_t)(16 * it) + (size_t)(JS * r) * ldk) * 8 + voff) = v[r];
                        if (MODE == 2) *reinterpret_cast<d4 *>(mbase + (size_t)(16 * it) * ldk * 8 + moff) = v;
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        if (MODE == 1 && diag && it < cb) continue;
                        const int gi = i0 + 16 * it + fr;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int gj = j0 + 16 * cb + ((MODE == 2) ? 4 * fk : fk) + JS * r;
                            double v = exp(fmin(acc[it][r], lnsf2));
                            if (MODE && gi == gj) v = dval;
                            if (gi < nr && gj < nc && !(diag && gi < gj)) K[gi + (size_t)gj * ldk] = v;
                            if (MODE == 2 && gi < nr && gj < nc && (!diag || gi > gj)) K[gj + (size_t)gi * ldk] = v;
                        }
                    }
                }
            }
        }
        u0 = ub + q1;
    }
}

int main(int argc, char **argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 65536, n = argc > 2 ? atoi(argv[2]) : 8192;
    const double spread = argc > 3 ? atof(argv[3]) : 4.0;
    const int d = 8;
    double *K, *X, *Xs;
    CK(hipMalloc(&K, sizeof(double) * (size_t)m * n));
    CK(hipMalloc(&X, sizeof(double) * (size_t)n * d)); CK(hipMalloc(&Xs, sizeof(double) * (size_t)m * d));
    std::vector<double> h((size_t)m * d);
    for (size_t i = 0; i < h.size(); ++i) h[i] = spread * (-0.5 + ((i * 2654435761u) % 100003) / 100003.0);
    std::vector<double> hx((size_t)n * d);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = spread * (-0.5 + (((i + 77) * 2246822519u) % 100019) / 100019.0);
    CK(hipMemcpy(Xs, h.data(), sizeof(double) * (size_t)m * d, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, hx.data(), sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice));
    GP prm; prm.sf2 = 2.25; prm.sn2 = 0.01; prm.extra = 0; for (int k = 0; k < 16; ++k) prm.inv_ls[k] = 1.0 / (1.0 + 0.25 * (k % 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double gb, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-52s %8.1f us  %.2f TB/s\n", name, ms * 1e3, gb / ms);
    };
    auto kref = [&](const double *a, int lda, int i, const double *b, int ldb, int j) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) { double df = a[i + (size_t)k * lda] - b[j + (size_t)k * ldb]; s += (df * (prm.inv_ls[k] * prm.inv_ls[k])) * df; }
        return prm.sf2 * exp(-0.5 * s);
    };
    std::vector<double> hk;
    auto check = [&](const char *name, int rows, int cols, const double *a, int lda, bool sym, bool lower_only) {
        hk.resize((size_t)rows * cols);
        CK(hipMemcpy(hk.data(), K, sizeof(double) * hk.size(), hipMemcpyDeviceToHost));
        double worst = 0.0; long bad = 0;
        for (long s = 0; s < 400000; ++s) {
            int i = (int)((s * 7919L + 13) % rows), j = (int)((s * 104729L + 5) % cols);
            if (s < 4096) { i = (int)(s % rows); j = (int)(s % cols); }          // the diagonal
            if (lower_only && i < j) { int tmp = i; i = j; j = tmp; }
            double ref = (sym && i == j) ? (prm.sf2 + prm.sn2) + prm.extra : kref(a, lda, i, hx.data(), n, j);
            double got = hk[i + (size_t)j * rows];
            double rel = fabs(got - ref) / fmax(fabs(ref), 1e-300);
            if (ref < 1e-290) continue;
            if (rel > worst) worst = rel;
            if (rel > 1e-13) ++bad;
        }
        printf("    check %-40s worst rel %.2e, > 1e-13: %ld\n", name, worst, bad);
    };
    const int nbr = m / 64, nbc = n / 64;
    const double gb_cross = 8.0 * m * n / 1e9, gb_low = 8.0 * n * (n + 1.0) / 2 / 1e9, gb_full = 8.0 * n * (double)n / 1e9;
    const int nt_cross = nbr * nbc, nb = (n + 63) / 64, nt_sym = nb * (nb + 1) / 2;
    CK(hipMemset(K, 0, sizeof(double) * (size_t)m * n));
    timeit("ga<3,0,1> cross, 1 wave/WG", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    check("cross", m, n, h.data(), m, false, false);
    timeit("ga<3,0,4> cross, 4 waves/WG", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 4>), dim3((nt_cross + 3) / 4), dim3(256), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("ga<3,0,1,VAR1> cross, no vector loads", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 1>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("ga<3,0,1,VAR2> cross, no stores", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 2>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("ga<3,0,1,VAR3> cross, nontemporal stores", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 4>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("cross, no stores, no exp", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 2 + 8>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("cross, no stores, no MFMA", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 2 + 16>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("cross, no stores, no loads", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 2 + 1>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("cross, no stores, no loads, no MFMA", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 2 + 1 + 16>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("cross, no stores, no loads, no exp", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 2 + 1 + 8>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("cross, no loads, no exp (MFMA + stores)", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 1 + 8>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    timeit("cross, no loads, no MFMA (exp + stores)", gb_cross, [&] { hipLaunchKernelGGL((ga<3, 0, 1, 1 + 16>), dim3(nt_cross), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nt_cross); });
    CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
    timeit("ga<3,1,4,VAR1> lower, no vector loads", gb_low, [&] { hipLaunchKernelGGL((ga<3, 1, 4, 1>), dim3((nt_sym + 3) / 4), dim3(256), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nt_sym); });
    timeit("ga<3,1,4,VAR2> lower, no stores", gb_low, [&] { hipLaunchKernelGGL((ga<3, 1, 4, 2>), dim3((nt_sym + 3) / 4), dim3(256), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nt_sym); });
    timeit("ga<3,1,4,VAR3> lower, nontemporal stores", gb_low, [&] { hipLaunchKernelGGL((ga<3, 1, 4, 4>), dim3((nt_sym + 3) / 4), dim3(256), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nt_sym); });
    CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
    timeit("ga<3,1,1> lower, 1 wave/WG", gb_low, [&] { hipLaunchKernelGGL((ga<3, 1, 1>), dim3(nt_sym), dim3(64), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nt_sym); });
    check("lower", n, n, hx.data(), n, true, true);
    timeit("ga<3,1,4> lower, 4 waves/WG", gb_low, [&] { hipLaunchKernelGGL((ga<3, 1, 4>), dim3((nt_sym + 3) / 4), dim3(256), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nt_sym); });
    CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
    timeit("ga<3,2,1> mirrored, 1 wave/WG", gb_full, [&] { hipLaunchKernelGGL((ga<3, 2, 1>), dim3(nt_sym), dim3(64), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nt_sym); });
    check("mirrored", n, n, hx.data(), n, true, false);
    timeit("ga<3,2,4> mirrored, 4 waves/WG", gb_full, [&] { hipLaunchKernelGGL((ga<3, 2, 4>), dim3((nt_sym + 3) / 4), dim3(256), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nt_sym); });
    for (int CH : {2, 4, 8, 16}) {
        char nm[96];
        const int jobs = nbr * ((nbc + CH - 1) / CH);
        snprintf(nm, sizeof nm, "gs<3,0> cross strips, CH=%d", CH);
        CK(hipMemset(K, 0, sizeof(double) * (size_t)m * n));
        timeit(nm, gb_cross, [&] { hipLaunchKernelGGL((gs<3, 0>), dim3(jobs), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nbc, CH); });
        if (CH == 8) check("cross strips", m, n, h.data(), m, false, false);
    }
    for (int CH : {2, 4, 8, 16}) {
        char nm[96];
        int jobs = 0;
        for (int b = 0; b < nb; ++b) jobs += b / CH + 1;
        const int ngrp = (nb + CH - 1) / CH;
        const int grid = CH * ngrp * (ngrp + 1) / 2;      // whole groups; strips past nb return at once
        snprintf(nm, sizeof nm, "gs<3,1> lower strips, CH=%d (%d jobs, grid %d)", CH, jobs, grid);
        CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
        timeit(nm, gb_low, [&] { hipLaunchKernelGGL((gs<3, 1>), dim3(grid), dim3(64), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nb, CH); });
        if (CH == 4) check("lower strips", n, n, hx.data(), n, true, true);
        snprintf(nm, sizeof nm, "gs<3,2> mirrored strips, CH=%d", CH);
        CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
        timeit(nm, gb_full, [&] { hipLaunchKernelGGL((gs<3, 2>), dim3(grid), dim3(64), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nb, CH); });
        if (CH == 4) check("mirrored strips", n, n, hx.data(), n, true, false);
    }
    for (int upw : {8, 16, 32, 64, 128, 171, 256}) {
        char nm[96];
        const long total = (long)nbr * nbc * 4;
        const int grid = (int)((total + upw - 1) / upw);
        snprintf(nm, sizeof nm, "gu<3,2,0> cross units, %d per wave (%d waves)", upw, grid);
        CK(hipMemset(K, 0, sizeof(double) * (size_t)m * n));
        timeit(nm, gb_cross, [&] { hipLaunchKernelGGL((gu<3, 2, 0>), dim3(grid), dim3(64), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr, nbc, upw, total); });
        if (upw == 32 || upw == 171) check("cross units", m, n, h.data(), m, false, false);
    }
    for (int upw : {2, 4, 6, 8, 11, 16}) {
        char nm[96];
        const long total = 2L * nb * (nb + 1);
        const int grid = (int)((total + upw - 1) / upw);
        snprintf(nm, sizeof nm, "gu<3,2,1> lower units, %d per wave (%d waves)", upw, grid);
        CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
        timeit(nm, gb_low, [&] { hipLaunchKernelGGL((gu<3, 2, 1>), dim3(grid), dim3(64), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nb, upw, total); });
        if (upw == 6 || upw == 11) check("lower units", n, n, hx.data(), n, true, true);
        snprintf(nm, sizeof nm, "gu<3,2,2> mirrored units, %d per wave", upw);
        CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
        timeit(nm, gb_full, [&] { hipLaunchKernelGGL((gu<3, 2, 2>), dim3(grid), dim3(64), 0, 0, X, n, n, X, n, n, d, prm, X, n, K, n, nb, nb, upw, total); });
        if (upw == 6 || upw == 11) check("mirrored units", n, n, hx.data(), n, true, false);
    }
    // ragged sizes
    {
        const int n2 = n - 37, m2 = 1000;
        CK(hipMemset(K, 0, sizeof(double) * (size_t)n * n));
        const int nb2 = (n2 + 63) / 64, nt2 = nb2 * (nb2 + 1) / 2;
        const long tot2 = 2L * nb2 * (nb2 + 1);
        (void)nt2;
        hipLaunchKernelGGL((gu<3, 2, 2>), dim3((int)((tot2 + 6) / 7)), dim3(64), 0, 0, X, n2, n, X, n2, n, d, prm, X, n, K, n2, nb2, nb2, 7, tot2);
        // reuse check with rows = n2 (hx has ld n)
        hk.resize((size_t)n2 * n2);
        CK(hipMemcpy(hk.data(), K, sizeof(double) * hk.size(), hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (long s = 0; s < 200000; ++s) {
            int i = (int)((s * 7919L + 13) % n2), j = (int)((s * 104729L + 5) % n2);
            if (s < 2000) { i = n2 - 1 - (int)(s % 70); j = n2 - 1 - (int)((s / 70) % 70); }
            double ref = (i == j) ? (prm.sf2 + prm.sn2) : kref(hx.data(), n, i, hx.data(), n, j);
            double rel = fabs(hk[i + (size_t)j * n2] - ref) / fmax(ref, 1e-300);
            if (ref > 1e-290 && rel > worst) worst = rel;
        }
        printf("    check ragged mirrored n=%d worst rel %.2e\n", n2, worst);
        (void)m2;
    }
    return 0;
}
