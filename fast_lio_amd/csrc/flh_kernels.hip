// flh_kernels.hip -- the HIP kernels of the hot path, written for gfx950 (CDNA4, wave64) only.
//
//   K0  map index build   keys -> (radix sort) -> gather -> brick/cell tables        [setup]
//   A   k_search<LPQ>     body->world transform + exact 5-NN over the cell grid       [search passes]
//   B   k_fit             plane fit + residual gate + 12-col Jacobian row + 16x16 Gram
//                         contraction on v_mfma_f64_16x16x4_f64                       [every pass]
//   R   k_reduce1/2       deterministic cross-block sum of the Gram partials          [every pass]
//
// Reference lines replaced: src/laserMapping.cpp:650-693 (A,B), :695-752 + esekfom.hpp:1784,1804 (B,R).
// Built with -ffp-contract=off (see flh_device.hpp).
#include "flh_kernels.hpp"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "flh_device.hpp"

namespace flh {

typedef unsigned long long u64;
constexpr u64 kInfKey = ~0ull;

// ------------------------------------------------------------------------------------------------
// K0: map index
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_map_keys(GridParams g, const float4* __restrict__ pts, uint32_t M,
                                                  u64* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float4 p = pts[i];
    int cx, cy, cz;
    float fx, fy, fz;
    cell_of(g, p.x, p.y, p.z, cx, cy, cz, fx, fy, fz);
    // the host sized the grid from the exact AABB with padding, so these clamps never fire for finite input
    cx = min(max(cx, 0), g.nx - 1);
    cy = min(max(cy, 0), g.ny - 1);
    cz = min(max(cz, 0), g.nz - 1);
    keys[i] = ((u64)brick_key(cx, cy, cz) << 6) | (u64)cell_local(cx, cy, cz);
    vals[i] = i;
}

// sorted float4 (xyz, original index bits) + head flags of bricks
__global__ void __launch_bounds__(256) k_map_gather(const float4* __restrict__ pts, const u64* __restrict__ keys_sorted,
                                                    const uint32_t* __restrict__ vals_sorted, uint32_t M,
                                                    float4* __restrict__ out, uint32_t* __restrict__ brick_head) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const uint32_t src = vals_sorted[i];
    float4 p = pts[src];
    p.w = __uint_as_float(src);
    out[i] = p;
    const u64 k = keys_sorted[i];
    brick_head[i] = (i == 0 || (keys_sorted[i - 1] >> 6) != (k >> 6)) ? 1u : 0u;
}

// brick_rank_incl[i] = inclusive prefix sum of brick_head (rank+1).  Cell heads write (start,count);
// brick heads insert (key -> rank) into the open-addressing directory.
__global__ void __launch_bounds__(256) k_map_cells(const u64* __restrict__ keys_sorted,
                                                   const uint32_t* __restrict__ brick_rank_incl, uint32_t M,
                                                   uint2* __restrict__ cells, uint2* __restrict__ hash,
                                                   uint32_t hash_mask, int hash_shift) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const u64 k = keys_sorted[i];
    const bool cell_head = (i == 0) || keys_sorted[i - 1] != k;
    if (!cell_head) return;
    const uint32_t rank = brick_rank_incl[i] - 1;
    uint32_t e = i + 1;
    while (e < M && keys_sorted[e] == k) ++e;
    cells[(size_t)rank * 64 + (uint32_t)(k & 63)] = make_uint2(i, e - i);
    const bool bhead = (i == 0) || (keys_sorted[i - 1] >> 6) != (k >> 6);
    if (bhead) {
        const uint32_t bkey = (uint32_t)(k >> 6);
        uint32_t slot = hash_slot(bkey, hash_shift);
        for (;;) {
            const uint32_t prev = atomicCAS(&hash[slot].x, kEmptyKey, bkey);
            if (prev == kEmptyKey) {
                hash[slot].y = rank;
                break;
            }
            slot = (slot + 1) & hash_mask;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A: exact 5-NN.  LPQ lanes cooperate on one query: lane t owns the cells t, t+LPQ, ... of the
// (2r+1)^3 cube around the query's cell, keeps a private sorted top-5 of what it sees, then the group
// merges the private lists with 5 rounds of {64-bit min butterfly, ballot, pop}.  Keys are
// (d2 bits << 32 | original map index): one unsigned compare orders by (d2, index), the tie-break the
// oracle uses.  Ring r is exact once the 5th distance is within the cube's guaranteed radius
// (r + distance to the nearest face of the centre cell) * c; otherwise the cube grows, up to the gate
// radius sqrt(max_sqdist) beyond which a result can never be selected (src/laserMapping.cpp:671).
// ------------------------------------------------------------------------------------------------
template <int LPQ>
__global__ void __launch_bounds__(256)
k_search(GridParams g, StateDev s, const float4* __restrict__ body, int N, float max_sqdist, int rmax,
         float4* __restrict__ world, float4* __restrict__ nn_pts, float* __restrict__ nn_d2,
         uint8_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected, u64* __restrict__ cand_counter) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int q = tid / LPQ;
    const int lane = threadIdx.x & (LPQ - 1);
    if (q >= N) return;  // group-uniform
    const float4 b = body[q];
    float qx, qy, qz;
    body_to_world(s, b.x, b.y, b.z, qx, qy, qz);
    int cx, cy, cz;
    float fx, fy, fz;
    cell_of(g, qx, qy, qz, cx, cy, cz, fx, fy, fz);
    const float minfrac = fminf(fminf(fminf(fx, 1.f - fx), fminf(fy, 1.f - fy)), fminf(fz, 1.f - fz));

    const int wl0 = (threadIdx.x & 63) & ~(LPQ - 1);  // first wave-lane of this group
    const u64 gmask = (LPQ == 64) ? ~0ull : (((1ull << (LPQ & 63)) - 1ull) << wl0);

    u64 rk[5];
    uint32_t rp[5];
    int cnt = 0;
    float d5 = INFINITY;
    float ub = max_sqdist;
    uint32_t ncand = 0;

    for (int r = 1; r <= rmax; ++r) {
        u64 k[5];
        uint32_t p[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { k[j] = kInfKey; p[j] = 0; }
        const int side = 2 * r + 1;
        const int side2 = side * side;
        const int ncell = side2 * side;
        for (int t = lane; t < ncell; t += LPQ) {
            const int iz = t / side2;
            const int rem = t - iz * side2;
            const int iy = rem / side;
            const int dx = rem - iy * side - r, dy = iy - r, dz = iz - r;
            if (r > 1) {
                // lower bound of the distance from the query to this cell's box; skip if beyond the bound
                const float gx = dx > 0 ? (float)dx - fx : (dx < 0 ? fx - (float)(dx + 1) : 0.f);
                const float gy = dy > 0 ? (float)dy - fy : (dy < 0 ? fy - (float)(dy + 1) : 0.f);
                const float gz = dz > 0 ? (float)dz - fz : (dz < 0 ? fz - (float)(dz + 1) : 0.f);
                const float lb = ((gx * gx + gy * gy) + gz * gz) * (g.c * g.c) * 0.995f - 1e-5f;
                if (lb > ub) continue;
            }
            const uint2 e = lookup_cell(g, cx + dx, cy + dy, cz + dz);
            const uint32_t end = e.x + e.y;
            ncand += e.y;
            for (uint32_t i0 = e.x; i0 < end; i0 += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = g.pts[min(i0 + u, end - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (i0 + u < end) {
                        const float d = dist2(qx, qy, qz, v[u].x, v[u].y, v[u].z);
                        const u64 key = ((u64)__float_as_uint(d) << 32) | (u64)__float_as_uint(v[u].w);
                        if (key < k[4]) {
                            k[4] = key;
                            p[4] = i0 + u;
#pragma unroll
                            for (int j = 4; j > 0; --j) {
                                if (k[j] < k[j - 1]) {
                                    const u64 tk = k[j]; k[j] = k[j - 1]; k[j - 1] = tk;
                                    const uint32_t tp = p[j]; p[j] = p[j - 1]; p[j - 1] = tp;
                                }
                            }
                        }
                    }
                }
            }
        }
        // ---- group merge: 5 x (min butterfly, ballot, pop)
        cnt = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            u64 m = k[0];
#pragma unroll
            for (int off = LPQ / 2; off >= 1; off >>= 1) {
                const u64 o = __shfl_xor(m, off, LPQ);
                m = o < m ? o : m;
            }
            const bool win = (k[0] == m) && (m != kInfKey);
            const u64 bal = __ballot(win) & gmask;
            const int wl = bal ? (__ffsll((long long)bal) - 1) : wl0;
            const uint32_t wp = __shfl(p[0], wl, 64);
            rk[j] = m;
            rp[j] = wp;
            if (m != kInfKey) ++cnt;
            if (win) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { k[t] = k[t + 1]; p[t] = p[t + 1]; }
                k[4] = kInfKey;
            }
        }
        d5 = (cnt == 5) ? __uint_as_float((uint32_t)(rk[4] >> 32)) : INFINITY;
        const float gr = ((float)r + minfrac) * g.c - 2e-3f * g.c;  // guaranteed-complete radius (with fp margin)
        const float gr2 = gr * gr;
        if ((cnt == 5 && d5 <= gr2) || gr2 >= max_sqdist) break;
        ub = fminf(d5, max_sqdist);
    }

    if (lane < 5) {
        u64 kk = rk[0];
        uint32_t pp = rp[0];
#pragma unroll
        for (int j = 1; j < 5; ++j)
            if (lane == j) { kk = rk[j]; pp = rp[j]; }
        const bool has = lane < cnt;
        float4 v = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (has) v = g.pts[pp];
        nn_pts[(size_t)lane * N + q] = v;
        nn_d2[(size_t)lane * N + q] = has ? __uint_as_float((uint32_t)(kk >> 32)) : INFINITY;
    }
    if (lane == 0) {
        nn_cnt[q] = (uint8_t)cnt;
        selected[q] = (cnt == 5 && !(d5 > max_sqdist)) ? 1 : 0;  // laserMapping.cpp:671
        world[q] = make_float4(qx, qy, qz, 0.f);
    }
    if (cand_counter) atomicAdd(cand_counter, (u64)ncand);
}

// ------------------------------------------------------------------------------------------------
// B: one thread per scan point: plane fit, residual gate, Jacobian row; then the wave's 64 rows are
// contracted into a 16x16 Gram block on the f64 matrix core.  v = [row(12) | h=-pd2 | 1 | |pd2| | 0]:
//   G[i][j] (i,j<12) = HTH,  G[i][12] = HTh,  G[13][13] = n_eff,  G[14][13] = total_residual.
// v_mfma_f64_16x16x4_f64 takes A[i][k] in lane (i + 16k) and B[k][j] in lane (j + 16k): with A = B^T
// = the same register, one LDS transpose ([point][16] -> lane (col, point%4)) feeds both operands.
// ------------------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kTileStride = 17;  // doubles per row: 16 + 1 pad (conflict-free ds_write_b64 / ds_read_b64)

__global__ void __launch_bounds__(256)
k_fit(StateDev s, const float4* __restrict__ body, const float4* __restrict__ nn_pts, int N, int ext, float thr,
      uint8_t* __restrict__ selected, float4* __restrict__ normvec, double* __restrict__ partials) {
    __shared__ double lds[4 * 64 * kTileStride];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;

    if (i < N && selected[i]) {  // laserMapping.cpp:674
        const float4 b = body[i];
        float P[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float4 n = nn_pts[(size_t)j * N + i];
            P[j][0] = n.x; P[j][1] = n.y; P[j][2] = n.z;
        }
        float pabcd[4];
        const bool ok = esti_plane(P, thr, pabcd);  // :678
        bool sel = false;
        float pd2 = 0.f;
        if (ok) {
            float wx, wy, wz;
            body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
            pd2 = ((pabcd[0] * wx + pabcd[1] * wy) + pabcd[2] * wz) + pabcd[3];  // :680
            const double bx = (double)b.x, by = (double)b.y, bz = (double)b.z;
            const double nb = sqrt((bx * bx + by * by) + bz * bz);
            const float sg = (float)(1 - 0.9 * (double)fabsf(pd2) / sqrt(nb));  // :681
            sel = (double)sg > 0.9;                                             // :683
        }
        selected[i] = sel ? 1 : 0;
        if (sel) {
            normvec[i] = make_float4(pabcd[0], pabcd[1], pabcd[2], pd2);  // :686-689
            // Jacobian row, fp64 (:723-752)
            const double bx = (double)b.x, by = (double)b.y, bz = (double)b.z;
            double px, py, pz;
            quat_rot(s.offR, bx, by, bz, px, py, pz);
            px = px + s.offT[0]; py = py + s.offT[1]; pz = pz + s.offT[2];
            const double rotc[4] = {-s.rot[0], -s.rot[1], -s.rot[2], s.rot[3]};
            const double nx = (double)pabcd[0], ny = (double)pabcd[1], nz = (double)pabcd[2];
            double Cx, Cy, Cz;
            quat_rot(rotc, nx, ny, nz, Cx, Cy, Cz);  // C = R^T n
            v[0] = nx; v[1] = ny; v[2] = nz;
            v[3] = (-pz) * Cy + py * Cz;             // A = hat(p_I) C
            v[4] = pz * Cx + (-px) * Cz;
            v[5] = (-py) * Cx + px * Cy;
            if (ext) {
                const double offRc[4] = {-s.offR[0], -s.offR[1], -s.offR[2], s.offR[3]};
                double Dx, Dy, Dz;
                quat_rot(offRc, Cx, Cy, Cz, Dx, Dy, Dz);
                v[6] = (-bz) * Dy + by * Dz;         // B = hat(p_b) R_LI^T C
                v[7] = bz * Dx + (-bx) * Dz;
                v[8] = (-by) * Dx + bx * Dy;
                v[9] = Cx; v[10] = Cy; v[11] = Cz;
            }
            v[12] = -(double)pd2;          // h(i) = -norm_p.intensity (:750)
            v[13] = 1.0;                   // effct_feat_num
            v[14] = (double)fabsf(pd2);    // res_last -> total_residual (:702)
        }
    }
    double* T = lds + wave * 64 * kTileStride;
#pragma unroll
    for (int c = 0; c < 16; ++c) T[lane * kTileStride + c] = v[c];
    __syncthreads();
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const int col = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const double a = T[(4 * m + kq) * kTileStride + col];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    __syncthreads();
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* Rb = lds;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rb[wave * 256 + (kq + 4 * r) * 16 + col] = acc[r];
    __syncthreads();
    const int t = threadIdx.x;
    partials[(size_t)blockIdx.x * 256 + t] = (Rb[t] + Rb[256 + t]) + (Rb[512 + t] + Rb[768 + t]);
}

// R: deterministic two-level sum of the per-block Gram partials (fixed order -> run-to-run identical)
__global__ void __launch_bounds__(256)
k_reduce1(const double* __restrict__ partials, int nblk, int per, double* __restrict__ part2) {
    const int t = threadIdx.x;
    const int b0 = blockIdx.x * per;
    const int b1 = min(b0 + per, nblk);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += partials[(size_t)(b + 0) * 256 + t];
        s1 += partials[(size_t)(b + 1) * 256 + t];
        s2 += partials[(size_t)(b + 2) * 256 + t];
        s3 += partials[(size_t)(b + 3) * 256 + t];
    }
    for (; b < b1; ++b) s0 += partials[(size_t)b * 256 + t];
    part2[(size_t)blockIdx.x * 256 + t] = (s0 + s1) + (s2 + s3);
}
__global__ void __launch_bounds__(256) k_reduce2(const double* __restrict__ part2, int n2, double* __restrict__ out) {
    const int t = threadIdx.x;
    double s = 0.0;
    for (int b = 0; b < n2; ++b) s += part2[(size_t)b * 256 + t];
    out[t] = s;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

hipError_t launch_map_keys(const GridParams& g, const float4* pts, uint32_t M, u64* keys, uint32_t* vals, hipStream_t st) {
    hipLaunchKernelGGL(k_map_keys, dim3(cdiv(M, 256)), dim3(256), 0, st, g, pts, M, keys, vals);
    return hipGetLastError();
}
hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, const uint32_t* vin, uint32_t* vout,
                      uint32_t M, hipStream_t st) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)M, 0, 36, st);
}
hipError_t inclusive_sum(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint32_t* out, uint32_t M, hipStream_t st) {
    return hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, in, out, (int)M, st);
}
hipError_t launch_map_gather(const float4* pts, const u64* ks, const uint32_t* vs, uint32_t M, float4* out,
                             uint32_t* brick_head, hipStream_t st) {
    hipLaunchKernelGGL(k_map_gather, dim3(cdiv(M, 256)), dim3(256), 0, st, pts, ks, vs, M, out, brick_head);
    return hipGetLastError();
}
hipError_t launch_map_cells(const u64* ks, const uint32_t* rank_incl, uint32_t M, uint2* cells, uint2* hash,
                            uint32_t hash_mask, int hash_shift, hipStream_t st) {
    hipLaunchKernelGGL(k_map_cells, dim3(cdiv(M, 256)), dim3(256), 0, st, ks, rank_incl, M, cells, hash, hash_mask,
                       hash_shift);
    return hipGetLastError();
}

hipError_t launch_search(int lpq, const GridParams& g, const StateDev& s, const float4* body, int N, float max_sqdist,
                         int rmax, float4* world, float4* nn_pts, float* nn_d2, uint8_t* nn_cnt, uint8_t* selected,
                         u64* cand_counter, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    const dim3 blk(256);
#define FLH_LAUNCH_SEARCH(L)                                                                                       \
    hipLaunchKernelGGL((k_search<L>), dim3(cdiv((long long)N * L, 256)), blk, 0, st, g, s, body, N, max_sqdist, rmax, \
                       world, nn_pts, nn_d2, nn_cnt, selected, cand_counter)
    switch (lpq) {
        case 8: FLH_LAUNCH_SEARCH(8); break;
        case 16: FLH_LAUNCH_SEARCH(16); break;
        case 64: FLH_LAUNCH_SEARCH(64); break;
        default: FLH_LAUNCH_SEARCH(32); break;
    }
#undef FLH_LAUNCH_SEARCH
    return hipGetLastError();
}

int fit_blocks(int N) { return cdiv(N > 0 ? N : 1, 256); }
int reduce1_blocks(int nblk, int* per_out) {
    int per = 16;
    int n2 = cdiv(nblk, per);
    if (per_out) *per_out = per;
    return n2;
}

hipError_t launch_fit(const StateDev& s, const float4* body, const float4* nn_pts, int N, int ext, float thr,
                      uint8_t* selected, float4* normvec, double* partials, double* part2, double* out256,
                      hipStream_t st) {
    const int nblk = fit_blocks(N);
    hipLaunchKernelGGL(k_fit, dim3(nblk), dim3(256), 0, st, s, body, nn_pts, N, ext, thr, selected, normvec, partials);
    int per;
    const int n2 = reduce1_blocks(nblk, &per);
    hipLaunchKernelGGL(k_reduce1, dim3(n2), dim3(256), 0, st, partials, nblk, per, part2);
    hipLaunchKernelGGL(k_reduce2, dim3(1), dim3(256), 0, st, part2, n2, out256);
    return hipGetLastError();
}

}  // namespace flh
