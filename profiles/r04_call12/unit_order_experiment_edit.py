def rep(s,a,b):
    assert s.count(a)==1, (s.count(a), a[:80])
    return s.replace(a,b)
# ---- kernel on the copy stream (flh_kernels.hip, next to k_scan_gather's launcher)
p='/root/repo/fast_lio_amd/csrc/flh_kernels.hip'; s=open(p).read()
s=rep(s,'''hipError_t launch_scan_gather(const float4* raw, const uint32_t* perm, uint32_t N, float4* body, hipStream_t st) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_gather, dim3(cdiv(N, 256)), dim3(256), 0, st, raw, perm, N, body);
    return hipGetLastError();
}''','''hipError_t launch_scan_gather(const float4* raw, const uint32_t* perm, uint32_t N, float4* body, hipStream_t st) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_gather, dim3(cdiv(N, 256)), dim3(256), 0, st, raw, perm, N, body);
    return hipGetLastError();
}

// The order in which k_pass's workgroups take the scan's units (64 consecutive points of the staged order): the SPARSEST first.
// A unit whose 64 Morton-neighbours spread over tens of metres is far field: every dependent load of its search misses, its
// waves are the slowest of the launch and decide when the kernel ends -- so they should be dispatched first.  Class of a unit =
// log2 of the extent (largest coordinate span) of eight sampled points; classes in descending order, inside a class from the end
// of the Morton order backwards (roughly: the placement uses atomics).  Any permutation gives the same results -- the summation
// tree is indexed by unit -- so this only moves time.  One workgroup, on the copy stream beside the previous scan's update.
constexpr int kOrderMaxUnits = 8192;
__global__ void __launch_bounds__(1024) k_unit_order(const float4* __restrict__ body, uint32_t N, uint32_t nunits,
                                                     uint32_t* __restrict__ order) {
    __shared__ uint32_t cnt[8], cursor[8];
    __shared__ uint8_t cls[kOrderMaxUnits];
    const uint32_t tid = threadIdx.x;
    if (tid < 8) { cnt[tid] = 0u; cursor[tid] = 0u; }
    __syncthreads();
    for (uint32_t k = tid; k < nunits; k += 1024) {
        const uint32_t u = nunits - 1u - k;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t q = min(u * 64u + (uint32_t)j * 9u, N - 1u);  // points 0, 9, ..., 63 of the unit
            const float4 p = body[q];
            lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
            lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
            lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
        }
        const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        int c = 0;  // < 1 m: 0; [1, 2): 1; [2, 4): 2; ... ; >= 64 m (or not finite): 7
        if (!(ext < 1.f)) c = min(7, 1 + (int)floorf(log2f(fminf(ext, 1.0e6f))));
        cls[k] = (uint8_t)c;
        atomicAdd(&cnt[c], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int c = 7; c >= 0; --c) { const uint32_t n = cnt[c]; cnt[c] = acc; acc += n; }  // cnt[c] = first position of class c
    }
    __syncthreads();
    for (uint32_t k = tid; k < nunits; k += 1024) {
        const uint32_t c = cls[k];
        order[cnt[c] + atomicAdd(&cursor[c], 1u)] = nunits - 1u - k;
    }
}
bool unit_order_supported(uint32_t N) { return N > 0 && (N + 63u) / 64u <= (uint32_t)kOrderMaxUnits; }
hipError_t launch_unit_order(const float4* body, uint32_t N, uint32_t* order, hipStream_t st) {
    if (!unit_order_supported(N)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_unit_order, dim3(1), dim3(1024), 0, st, body, N, (N + 63u) / 64u, order);
    return hipGetLastError();
}''')
open(p,'w').write(s)
p='/root/repo/fast_lio_amd/csrc/flh_kernels.hpp'; s=open(p).read()
s=rep(s,'''hipError_t launch_scan_gather(const float4* raw, const uint32_t* perm, uint32_t N, float4* body, hipStream_t st);''','''hipError_t launch_scan_gather(const float4* raw, const uint32_t* perm, uint32_t N, float4* body, hipStream_t st);
// the dispatch order of k_pass's units for a staged scan (sparsest units first); (N + 63) / 64 words
bool unit_order_supported(uint32_t N);
hipError_t launch_unit_order(const float4* body, uint32_t N, uint32_t* order, hipStream_t st);''')
s=rep(s,'''                       int own_axis, float own_lo, float own_hi, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t launch_publish256''','''                       int own_axis, float own_lo, float own_hi, const uint32_t* unit_order /* may be null */, hipStream_t st,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t launch_publish256''')
open(p,'w').write(s)
# ---- k_pass
p='/root/repo/fast_lio_amd/csrc/flh_pass.hip'; s=open(p).read()
s=rep(s,'''       int red, u64* __restrict__ cand_counter, int own_axis, float own_lo, float own_hi) {''','''       int red, u64* __restrict__ cand_counter, int own_axis, float own_lo, float own_hi, const uint32_t* __restrict__ unit_order) {''')
a=s.index('    // which 64 scan points (a unit of the summation tree) this workgroup takes: the LAST ones first.')
b=s.index('    const int q0 = unit * kPassQueries;')
s=s[:a]+'''    // Which 64 scan points (a unit of the summation tree) this workgroup takes.  Workgroups are dispatched in blockIdx order over
    // ~2 us, and the slowest waves of the launch -- the scan's far field: sparse queries, every dependent load of the search
    // misses -- decide when the kernel ends, so they start first: in the order the staging left for this scan (k_unit_order:
    // sparsest units first), else from the END of the Morton order backwards (measured on BASELINE configs[1], same box, two pairs:
    // 41.9 -> 38.8 us per launch against the forward order, profiles/r04_call8/).  Any order gives the same bits.
#if defined(FLH_PASS_FORWARD)  // (developer A/B builds)
    const int unit = (int)blockIdx.x;
#elif defined(FLH_PASS_REVERSED)
    const int unit = (int)gridDim.x - 1 - (int)blockIdx.x;
#else
    const int unit = unit_order ? (int)unit_order[blockIdx.x] : (int)gridDim.x - 1 - (int)blockIdx.x;
#endif
'''+s[b:]
s=rep(s,'''                       int own_axis, float own_lo, float own_hi, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {''','''                       int own_axis, float own_lo, float own_hi, const uint32_t* unit_order, hipStream_t st, hipEvent_t ev_start,
                       hipEvent_t ev_stop) {''')
s=s.replace('''ncol, nn_pts, nn_cnt, selected, plane_cache, partials, tickets, out, seq, red, cand_counter, own_axis, \\
                                  own_lo, own_hi);''','''ncol, nn_pts, nn_cnt, selected, plane_cache, partials, tickets, out, seq, red, cand_counter, own_axis, \\
                                  own_lo, own_hi, unit_order);''')
s=s.replace('''selected, plane_cache, partials, tickets, out, seq, red, cand_counter, own_axis, own_lo, own_hi);       \\''','''selected, plane_cache, partials, tickets, out, seq, red, cand_counter, own_axis, own_lo, own_hi, unit_order); \\''')
open(p,'w').write(s)
# ---- host
p='/root/repo/fast_lio_amd/csrc/flh_api.cpp'; s=open(p).read()
s=rep(s,'''        DevBuf<float4> body;            // Morton-ordered (internal order); .w = original index''','''        DevBuf<float4> body;            // Morton-ordered (internal order); .w = original index
        DevBuf<uint32_t> order;         // the order in which k_pass's workgroups take this scan's units (k_unit_order); has_order says whether it is set
        bool has_order = false;''')
s=rep(s,'''    HIPC(hipEventRecord(sl.ready, cs));
    sl.N = N;''','''    sl.has_order = false;
    if (flh::unit_order_supported((uint32_t)N)) {
        HIPC(sl.order.reserve((N + 63) / 64));
        HIPC(flh::launch_unit_order(sl.body.p, (uint32_t)N, sl.order.p, cs));
        sl.has_order = true;
    }
    HIPC(hipEventRecord(sl.ready, cs));
    sl.N = N;''')
s=rep(s,'''                              h->stats ? h->counter.p : nullptr, h->own_axis, h->own_lo, h->own_hi, st, timed ? ev3[0] : nullptr,
                              timed ? ev3[3] : nullptr));''','''                              h->stats ? h->counter.p : nullptr, h->own_axis, h->own_lo, h->own_hi,
                              (h->cur && h->cur->has_order && h->cur->N == h->N) ? h->cur->order.p : nullptr, st, timed ? ev3[0] : nullptr,
                              timed ? ev3[3] : nullptr));''')
open(p,'w').write(s)
print('ok')
