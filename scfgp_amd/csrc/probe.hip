// Box probe: what THIS device delivers on the two resources the hot path lives on, measured in ~100 ms so that two bench
// lines taken on two leases of the pool can be normalised instead of argued about (boxes differ by several percent in
// sustained MFMA clock and in memory bandwidth).  No reference counterpart; bench.py prints it as `secondary.box`.
//   out[0] fp32 MFMA rate of a register-only loop (v_mfma_f32_16x16x4_f32, 8 independent accumulators per wave; the best of
//          1, 2 and 8 waves per SIMD, whose individual rates are out[3], out[4], out[5] when n >= 6), TFLOP/s
//   out[1] shader clock held during that loop, GHz (s_memtime ticks per s_memrealtime tick of the 100 MHz constant clock,
//          median over workgroups)
//   out[2] streaming copy of 2 GiB (16 bytes per lane, read + write counted), GB/s
//   out[6] (n >= 7) read-only stream of the same 2 GiB (four independent 16-byte loads per lane in flight, summed into a
//          register), GB/s: the ceiling of the sweeps that only read (X~^T Zbar reads Phi and Phibar once)
#include "../../include/scfgp_hip.h"
#include "common.h"

#include <algorithm>
#include <vector>

__global__ __launch_bounds__(256) void probe_mfma_kernel(float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters) {
    const int lane = threadIdx.x & 63;
    v4f acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = v4f{0.f, 0.f, 0.f, 0.f};
    // full-range pseudo-random operands, different for every lane and every MFMA of the 8: constant or trivial operands
    // let the chip hold a higher clock than real data does (MI355X_MICROARCH.md, DVFS give-back)
    float a[8], b[8];
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u; a[j] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        h = h * 1664525u + 1013904223u; b[j] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    (void)lane;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {                            // inline asm: the builtin form made hipcc shuffle accumulators
#pragma unroll                                                   // between misaligned AGPR tuples inside the loop
        for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a[j]), "v"(b[j]));
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 123.456f) sink[0] = s;                               // never true: keeps the loop alive
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void probe_read_kernel(const v4f* __restrict__ src, float* __restrict__ sink, int64_t n) {
    v4f a0 = v4f{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const v4f x0 = src[i], x1 = src[i + stride], x2 = src[i + 2 * stride], x3 = src[i + 3 * stride];
        a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    }
    for (; i < n; i += stride) a0 += src[i];
    const v4f a = a0 + a1 + a2 + a3;
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) sink[1] = a[0];     // never true: keeps the loads alive
}

extern "C" int scfgp_box_probe(int device, double* out, int n) {
    if (!out || n < 3) return SCFGP_EARG;
    if (hipSetDevice(device) != hipSuccess) return SCFGP_EHIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SCFGP_EHIP;
    const int ncu = prop.multiProcessorCount, nwg = ncu * 8;
    float* sink = nullptr; unsigned long long* stamps = nullptr; char* buf = nullptr;
    const size_t half = (size_t)1 << 30;                          // 1 GiB in, 1 GiB out
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t ps = nullptr;                                     // the probe's own non-blocking stream: no implicit ordering against anybody's work
    int rc = SCFGP_OK;
    if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&sink, 64) != hipSuccess || hipMalloc((void**)&stamps, sizeof(unsigned long long) * 2 * nwg) != hipSuccess ||
        hipMalloc((void**)&buf, 2 * half) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        rc = SCFGP_EHIP;
    } else {
        float ms = 0;
        (void)hipMemsetAsync(buf, 1, 2 * half, ps);
        hipLaunchKernelGGL(probe_mfma_kernel, dim3(nwg), dim3(256), 0, ps, sink, stamps, 2000);      // warm-up
        out[0] = 0; out[1] = 0;
        const int wps[3] = {1, 2, 8};                                 // waves per SIMD = 4-wave workgroups per CU
        for (int v = 0; v < 3; ++v) {
            const int g = ncu * wps[v], iters = 240000 / wps[v];       // ~25-30 ms each
            (void)hipEventRecord(e0, ps);
            hipLaunchKernelGGL(probe_mfma_kernel, dim3(g), dim3(256), 0, ps, sink, stamps, iters);
            (void)hipEventRecord(e1, ps);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double tf = (double)g * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;
            if (n >= 6) out[3 + v] = tf;
            if (tf > out[0]) {
                out[0] = tf;
                std::vector<unsigned long long> h(2 * g);
                (void)hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * g, hipMemcpyDeviceToHost);
                std::vector<double> ghz(g);
                for (int i = 0; i < g; ++i) ghz[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.0;
                std::nth_element(ghz.begin(), ghz.begin() + g / 2, ghz.end());
                out[1] = ghz[g / 2];
            }
        }
        double best = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0, ps);
            hipLaunchKernelGGL(probe_copy_kernel, dim3(ncu * 16), dim3(256), 0, ps, (const v4f*)buf, (v4f*)(buf + half), (int64_t)(half / 16));
            (void)hipEventRecord(e1, ps);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0) best = std::max(best, 2.0 * half / (ms * 1e-3) / 1e9);
        }
        out[2] = best;
        if (n >= 7) {
            best = 0;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, ps);
                hipLaunchKernelGGL(probe_read_kernel, dim3(ncu * 16), dim3(256), 0, ps, (const v4f*)buf, sink, (int64_t)(2 * half / 16));
                (void)hipEventRecord(e1, ps);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0) best = std::max(best, 2.0 * half / (ms * 1e-3) / 1e9);
            }
            out[6] = best;
        }
        if (hipGetLastError() != hipSuccess) rc = SCFGP_EHIP;
    }
    if (ps) (void)hipStreamSynchronize(ps);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (ps) (void)hipStreamDestroy(ps);
    if (sink) (void)hipFree(sink);
    if (stamps) (void)hipFree(stamps);
    if (buf) (void)hipFree(buf);
    return rc;
}
