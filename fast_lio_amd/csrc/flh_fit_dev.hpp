// flh_fit_dev.hpp -- device code of the per-point plane fit / residual gate / Jacobian row and of the 16x16 Gram contraction
// on the f64 matrix core, shared by k_fit (flh_kernels.hip) and k_pass (flh_pass.hip).  gfx950 / wave64 only.
// Reference lines replaced: src/laserMapping.cpp:674-691 (fit + gate), :720-752 (rows), esekfom.hpp:1784,1804 (H^T H, H^T h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flh_device.hpp"

namespace flh {

// ONE scan point that passed the kNN gate: plane through its five neighbours P (laserMapping.cpp:678), point-to-plane residual
// (:680), the mixed-precision gate (:681-683), and -- if it passes -- its row of the 16-column Gram operand
//   v = [row(12) | h = -pd2 | 1 | |pd2| | 0]      (v stays as the caller initialised it -- zeros -- otherwise).
// PM = 2: the plane is taken from pc (the plane cache) instead of being fitted; PM = 0 / 1: fitted (ok = esti_plane's verdict).
// Returns the point's final point_selected_surf flag.  The expressions are the reference's, operation for operation.
template <int ORD, bool HALF, int PM>
__device__ __forceinline__ bool fit_point(const StateDev& s, float bxf, float byf, float bzf, float wx, float wy, float wz,
                                          const float (&P)[5][3], const float4& pc, int ext, float thr, float (&pabcd)[4], bool& ok,
                                          float& pd2, double (&v)[16]) {
    if (PM == 2) {  // compile-time: this instantiation has no fit in it
        pabcd[0] = pc.x; pabcd[1] = pc.y; pabcd[2] = pc.z; pabcd[3] = pc.w;
        ok = true;
    } else {
        ok = HALF ? esti_plane_half<ORD>(P, thr, pabcd) : esti_plane<ORD>(P, thr, pabcd);  // :678
    }
    bool sel = false;
    pd2 = 0.f;
    if (ok) {
        pd2 = ((pabcd[0] * wx + pabcd[1] * wy) + pabcd[2] * wz) + pabcd[3];  // :680
        const double bx = (double)bxf, by = (double)byf, bz = (double)bzf;
        const double nb = sqrt((bx * bx + by * by) + bz * bz);
        const float sg = (float)(1 - 0.9 * (double)fabsf(pd2) / sqrt(nb));  // :681
        sel = (double)sg > 0.9;                                             // :683
    }
    if (sel) {
        // Jacobian row, fp64 (:723-752)
        const double bx = (double)bxf, by = (double)byf, bz = (double)bzf;
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
    return sel;
}

// The wave's 64 rows contracted into a 16x16 Gram block G = sum v v^T:
//   G[i][j] (i,j<12) = HTH,  G[i][12] = HTh,  G[13][13] = n_eff,  G[14][13] = total_residual.
// v_mfma_f64_16x16x4_f64 takes A[i][k] in lane (i + 16k) and B[k][j] in lane (j + 16k): with A = B^T = the same register, one
// LDS transpose ([point][16] -> lane (col, point%4)) feeds both operands.  T: 64 x kTileStride doubles of LDS owned by the wave;
// the caller orders the wave's earlier LDS traffic before and after.  Result layout (C/D of the f64 MFMA): col = lane & 15,
// row = (lane >> 4) + 4 * reg.
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kTileStride = 17;  // doubles per row: 16 + 1 pad (conflict-free ds_write_b64 / ds_read_b64)
__device__ __forceinline__ void tile_store(double* __restrict__ T, int lane, const double (&v)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) T[lane * kTileStride + c] = v[c];
}
__device__ __forceinline__ v4f64 tile_gram(const double* __restrict__ T, int lane) {
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const int col = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const double a = T[(4 * m + kq) * kTileStride + col];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    return acc;
}

// Compact layout of the entries of the 16x16 Gram block the filter reads: the upper triangle of the leading ncol x ncol block
// (ncol = 12 with extrinsic estimation, else 6: the last six columns are structurally zero, laserMapping.cpp:745), then the
// ncol entries of column 12 (H^T h), then n_eff (G[13][13]) and total_residual (G[14][13]).  -1 = not transmitted.
__host__ __device__ inline int gram_nslots(int ncol) { return ncol * (ncol + 1) / 2 + ncol + 2; }
__host__ __device__ inline int gram_slot(int r, int c, int ncol) {
    const int tri = ncol * (ncol + 1) / 2;
    if (c < ncol && r <= c) return r * ncol - r * (r - 1) / 2 + (c - r);
    if (c == 12 && r < ncol) return tri + r;
    if (r == 13 && c == 13) return tri + ncol;
    if (r == 14 && c == 13) return tri + ncol + 1;
    return -1;
}

// one 16-byte system-scope store of a {value, sequence} granule (the host polls the sequence word of every granule it needs;
// a 16-byte store is a single PCIe write, so value and tag arrive together)
__device__ __forceinline__ void store_granule(double* dst, double value, double seq) {
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    const v2f64 g2 = {value, seq};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(g2) : "memory");
}


// Where a pass's group sums go: granule buffers in pinned host memory -- this rank's own and, when a scan is sharded over GPUs,
// every peer's (one shared-memory segment registered with every device: no collective, no extra launch).  A rank's SECTION of a
// buffer: granule 0 = {number of value granules that follow, sequence}, then [group][slot] with
// slot < gran_section_slots(ncol) = the Gram entries the filter reads + one statistic (queries that needed the second search).
constexpr int kPeersMax = 8;
struct GranOut {
    double* dst[kPeersMax];
    int n_dst;     // 0: no granule output
    int sect_off;  // granules before this rank's section in every buffer
};
__host__ __device__ inline int gran_section_slots(int ncol) { return gram_nslots(ncol) + 1; }
__device__ __forceinline__ void publish_granule(const GranOut& o, size_t idx, double value, double seq) {
#pragma unroll
    for (int d = 0; d < kPeersMax; ++d)  // static indices: the pointers stay in SGPRs
        if (d < o.n_dst) store_granule(o.dst[d] + ((size_t)o.sect_off + idx) * 2, value, seq);
}


// ---- the cross-workgroup sum both pass kernels share.  A UNIT is 64 consecutive scan points (a workgroup of k_pass, a wave of
// k_fit: the same points in the same lanes), a QUAD four consecutive units, a GROUP `red` consecutive units (a multiple of four).
// The one summation tree of a pass, whichever kernel runs it (so a no-search pass produces the bits a searching pass would at the
// same state, and the one-launch pass the bits of the three-launch pass):
//     quad  = ((u0 + u1) + u2) + u3                     missing units of the scan's last quad count as +0.0
//     half  = q_a + q_(a+1) + ...  in quad order        lower half = the first ceil(quads / 2) quads of the group
//     group = lower half + upper half                   -> one {value, sequence} granule per entry; the host adds the groups in order
// k_pass stores one partial per unit and its group's last arriver forms the quads; a block of k_fit IS a quad: it adds its four
// waves in LDS, stores one partial, and its group's last arriver reads a quarter of the partials.
// Hand-off: the entries of the 16x16 block that the filter reads (+ one statistic) are stored write-through at agent scope, the
// stores drained (vmcnt(0)), then a ticket of the group is taken; the last arriver reads with agent-scope (L1-bypassing) loads.
// No release/acquire fences, hence no L2 write-back sweep.
typedef __attribute__((address_space(1))) double gdouble;
__device__ __forceinline__ void unit_partial_store(double* partials, int unit, int nsl, int ncol, const v4f64& acc, int wl, double stat) {
    gdouble* gp = (gdouble*)partials + (size_t)unit * nsl;
    const int col = wl & 15, kq = wl >> 4;  // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int slot = gram_slot(kq + 4 * r, col, ncol);
        if (slot >= 0) __hip_atomic_store(gp + FLH_IDX(601, slot, nsl - 1), acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wl == 0) __hip_atomic_store(gp + (nsl - 1), stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One wave; the group's units are [group * red, group * red + gsize).  QUADS: partials holds one record per QUAD (k_fit), indexed by
// quad number = unit / 4; else one per unit (k_pass).
template <bool QUADS>
__device__ __forceinline__ void group_sum_publish(const double* partials, int group, int gsize, int red, int nsl, int ngroups,
                                                  const GranOut& gout, double seq, int wl, double* part2 = nullptr) {
    const gdouble* gpart = (const gdouble*)partials;
    const int u0 = group * red;
    const int nq = (gsize + 3) >> 2;   // quads of the group
    const int halfq = (nq + 1) >> 1;   // quads of its lower half
    const int hi = wl >> 5, sl = wl & 31;
    const int qlo = hi ? halfq : 0, qhi = hi ? nq : halfq;
    // k_fit's reducer reads the eight quad records of a half group in ONE load round trip (the same tree, the same order of
    // additions as with two trips of four; same box, two alternating pairs, profiles/r05_call1/: no-search pass 18.3 / 18.4 ->
    // 17.8 / 18.0 us, 7 431 / 7 399 -> 7 512 / 7 484 scans/s).  k_pass's reducer keeps sixteen loads per trip: thirty-two do
    // not fit its 72 registers
    constexpr int QB = QUADS ? 8 : 4;
    if (QUADS && nsl > 64 && nsl <= 96) {
        // the extrinsic columns (94 slots = three trips of 32): k_fit's reducer takes its three slots at once -- the 24 loads of a
        // round in flight together, one round trip where round 5 made three (extrinsic_est_en = 1 is the reference's default,
        // src/laserMapping.cpp:789).  Per slot the same additions in the same order: the same bits.
        constexpr int NT = 3;
        double s3[NT] = {0.0, 0.0, 0.0};
        for (int q0 = qlo; q0 < qhi; q0 += QB) {
            double qv[NT][QB];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int slot = sl + 32 * tt;
                const int sc = slot < nsl ? slot : nsl - 1;
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    qv[tt][j] = (q0 + j < qhi) ? __hip_atomic_load(gpart + (size_t)((u0 >> 2) + q0 + j) * nsl + sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : 0.0;
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    if (q0 + j < qhi) s3[tt] += qv[tt][j];
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int slot = sl + 32 * tt;
            const double other = __shfl_xor(s3[tt], 32, 64);
            const double total = hi ? other + s3[tt] : s3[tt] + other;  // lower half + upper half on both sides
            if (hi == 0 && slot < nsl) {
                if (gout.n_dst > 0) publish_granule(gout, 1 + (size_t)group * nsl + slot, total, seq);
                else part2[(size_t)group * nsl + slot] = total;
            }
        }
        if (gout.n_dst > 0 && wl == 0 && group == 0) publish_granule(gout, 0, (double)(ngroups * nsl), seq);  // the section's header
        return;
    }
    // (k_pass with the extrinsic columns does not come here: three of its waves take a trip each, group_sum_trip below)
    for (int slot = sl; slot < ((nsl + 31) & ~31); slot += 32) {
        const int sc = slot < nsl ? slot : nsl - 1;
        double s0 = 0.0;
        for (int q0 = qlo; q0 < qhi; q0 += QB) {  // sixteen loads in flight (four quads of units, or four quad records)
            double qv[QB];
            if (QUADS) {
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    qv[j] = (q0 + j < qhi) ? __hip_atomic_load(gpart + (size_t)((u0 >> 2) + q0 + j) * nsl + sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : 0.0;
            } else {
                double pv[4 * QB];
#pragma unroll
                for (int j = 0; j < 4 * QB; ++j) {
                    const int u = 4 * q0 + j;  // unit inside the group
                    pv[j] = (q0 + (j >> 2) < qhi && u < gsize)
                                ? __hip_atomic_load(gpart + (size_t)(u0 + u) * nsl + sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : 0.0;
                }
#pragma unroll
                for (int j = 0; j < QB; ++j) qv[j] = ((pv[4 * j] + pv[4 * j + 1]) + pv[4 * j + 2]) + pv[4 * j + 3];
            }
#pragma unroll
            for (int j = 0; j < QB; ++j)
                if (q0 + j < qhi) s0 += qv[j];  // (uniform over the wave's half: qhi is)
        }
        const double other = __shfl_xor(s0, 32, 64);
        const double total = hi ? other + s0 : s0 + other;  // lower half + upper half on both sides
        if (hi == 0 && slot < nsl) {
            if (gout.n_dst > 0) publish_granule(gout, 1 + (size_t)group * nsl + slot, total, seq);
            else part2[(size_t)group * nsl + slot] = total;  // group totals in device memory: the ranks' are all-reduced, k_publish_groups adds the groups
        }
    }
    if (gout.n_dst > 0 && wl == 0 && group == 0) publish_granule(gout, 0, (double)(ngroups * nsl), seq);  // the section's header
}

// ONE 32-slot trip of k_pass's group sum (slot = 32 * trip + (lane & 31)) by one wave.  With the extrinsic columns a group's record
// has 93 slots = three trips; the last arriver's workgroup takes them with three of its waves side by side instead of one wave
// taking them in turn (extrinsic_est_en = 1 is the reference's default, src/laserMapping.cpp:789; profiles/r06_call33/: the columns
// cost k_pass 2.5 us, most of it here).  Per slot the additions of group_sum_publish<false> in its order: the same bits.  The
// section's header stays with the caller.
__device__ __forceinline__ void group_sum_trip(const double* partials, int group, int gsize, int red, int nsl, const GranOut& gout,
                                               double seq, int wl, int trip, double* part2) {
    const gdouble* gpart = (const gdouble*)partials;
    const int u0 = group * red;
    const int nq = (gsize + 3) >> 2;
    const int halfq = (nq + 1) >> 1;
    const int hi = wl >> 5, sl = wl & 31;
    const int qlo = hi ? halfq : 0, qhi = hi ? nq : halfq;
    constexpr int QB = 4;
    const int slot = sl + 32 * trip;
    const int sc = slot < nsl ? slot : nsl - 1;
    double s0 = 0.0;
    for (int q0 = qlo; q0 < qhi; q0 += QB) {
        double pv[4 * QB];
#pragma unroll
        for (int j = 0; j < 4 * QB; ++j) {
            const int u = 4 * q0 + j;  // unit inside the group
            pv[j] = (q0 + (j >> 2) < qhi && u < gsize) ? __hip_atomic_load(gpart + (size_t)(u0 + u) * nsl + sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                       : 0.0;
        }
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const double qv = ((pv[4 * j] + pv[4 * j + 1]) + pv[4 * j + 2]) + pv[4 * j + 3];
            if (q0 + j < qhi) s0 += qv;
        }
    }
    const double other = __shfl_xor(s0, 32, 64);
    const double total = hi ? other + s0 : s0 + other;  // lower half + upper half on both sides
    if (hi == 0 && slot < nsl) {
        if (gout.n_dst > 0) publish_granule(gout, 1 + (size_t)group * nsl + slot, total, seq);
        else part2[(size_t)group * nsl + slot] = total;
    }
}

}  // namespace flh
