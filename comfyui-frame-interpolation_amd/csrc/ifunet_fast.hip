// CBAM (IFUNet_arch.py:411-503) on cooperating lanes.  Each kernel replaces a per-element body of ifunet_bodies.h whose
// parallelism was the OUTPUT count — N x C = 256 threads for the channel gate's two-layer MLP, N x strips x C threads
// walking thousands of pixels each for the pooling — which leaves an MI355X's 1024 SIMDs idle and every access strided.
//   chan_pool_partial_wg_kernel   global mean / max per channel, pass 1: a workgroup per strip, 256 / C pixel lanes x C
//                                 channels (a wave reads whole pixel rows); double sums, as the body.
//   chan_pool_final_wave_kernel   pass 2: a wave per (image, channel) over the strips.
//   cbam_gate_wg_kernel           sigmoid(mlp(avg) + mlp(max)): a workgroup per image; hidden units by waves (lanes over the C
//                                 inputs, shuffle sum), then a thread per output channel over the R hidden units.
//   cbam_scale_compress_wave_kernel  xs = x * scale, comp = (max_c, mean_c): lanes over channels, segmented shuffle max / sum.
#include "ifunet_fast.h"

#include "vfi_common.h"

namespace vfi {

using namespace vfi_ifunet;

namespace {

__global__ __launch_bounds__(256) void chan_pool_partial_wg_kernel(const PoolPartArgs a) {
    __shared__ double ssum[256];
    __shared__ float smax[256];
    const int s = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int P = 256 / a.C;
    const int p = tid / a.C, c = tid - p * a.C;
    const long per = (a.HW + a.strips - 1) / a.strips, lo = s * per, hi = lo + per < a.HW ? lo + per : a.HW;
    double sum = 0.0;
    float mx = -INFINITY;
    if (p < P) {
        const float* b = a.x + (size_t)n * a.HW * a.cs + c;
#pragma unroll 8
        for (long i = lo + p; i < hi; i += P) {
            const float v = b[i * a.cs];
            sum += v;
            mx = v > mx ? v : mx;
        }
    }
    ssum[tid] = sum, smax[tid] = mx;
    __syncthreads();
    if (tid < a.C) {
        for (int k = 1; k < P; ++k) {
            sum += ssum[tid + k * a.C];
            mx = smax[tid + k * a.C] > mx ? smax[tid + k * a.C] : mx;
        }
        const size_t q = ((size_t)n * a.strips + s) * a.C + tid;
        a.psum[q] = sum;
        a.pmax[q] = mx;
    }
}

__global__ __launch_bounds__(256) void chan_pool_final_wave_kernel(const PoolFinalArgs a) {
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= (long)a.N * a.C) return;
    const int c = (int)(idx % a.C), n = (int)(idx / a.C);
    double sum = 0.0;
    float mx = -INFINITY;
    for (int s = lane; s < a.strips; s += 64) {
        const size_t q = ((size_t)n * a.strips + s) * a.C + c;
        sum += a.psum[q];
        mx = a.pmax[q] > mx ? a.pmax[q] : mx;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sum += __shfl_xor(sum, o);
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) {
        a.stats[idx * 2] = (float)(sum / (double)a.HW);
        a.stats[idx * 2 + 1] = mx;
    }
}

constexpr int GATE_MAXR = 64;

__global__ __launch_bounds__(256) void cbam_gate_wg_kernel(const GateArgs a) {
    __shared__ float hid[2][GATE_MAXR];
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int j = wave; j < 2 * a.R; j += 4) {
        const int which = j / a.R, r = j - which * a.R;
        float s = 0.f;
        for (int k = lane; k < a.C; k += 64) s += a.w1[(size_t)r * a.C + k] * a.stats[((size_t)n * a.C + k) * 2 + which];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const float h = a.b1[r] + s;
            hid[which][r] = h > 0.f ? h : 0.f;
        }
    }
    __syncthreads();
    for (int c = tid; c < a.C; c += 256) {
        float att = 0.f;
        for (int which = 0; which < 2; ++which) {
            float o = a.b2[c];
            for (int r = 0; r < a.R; ++r) o += a.w2[(size_t)c * a.R + r] * hid[which][r];
            att += o;
        }
        a.scale[(size_t)n * a.C + c] = 1.0f / (1.0f + expf(-att));
    }
}

// G lanes per pixel (G = the power of two >= min(C, 64)), 64 / G pixels per wave; channels c = g, g + G, ...
__global__ __launch_bounds__(256) void cbam_scale_compress_wave_kernel(const ScaleCompArgs a, int G) {
    const int lane = threadIdx.x & 63;
    const int ppw = 64 / G;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long pix = wave * ppw + lane / G;
    const int g = lane & (G - 1);
    const bool ok = pix < (long)a.N * a.HW;
    float mx = -INFINITY, sum = 0.f;
    if (ok) {
        const int n = (int)(pix / a.HW);
        const float* b = a.x + pix * a.cs;
        const float* sc = a.scale + (size_t)n * a.C;
        float* o = a.xs + pix * a.xs_cs;
        for (int c = g; c < a.C; c += G) {
            const float v = b[c] * sc[c];
            o[c] = v;
            mx = v > mx ? v : mx;
            sum += v;
        }
    }
    for (int o = G >> 1; o >= 1; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        sum += __shfl_xor(sum, o);
    }
    if (ok && g == 0) {
        a.comp[pix * 2] = mx;
        a.comp[pix * 2 + 1] = sum / (float)a.C;
    }
}

// SpatialGate (:469-482) with G lanes per pixel: the 49 taps of the 7x7 conv over the 2-channel compressed map are split over the
// lanes and summed by shuffles, then every lane scales its channels (coalesced; the body's per-thread loop over C strided floats
// was the cost)
__global__ __launch_bounds__(256) void cbam_spatial_wave_kernel(const SpatialArgs a, int G) {
    const int lane = threadIdx.x & 63;
    const int ppw = 64 / G;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long pix = wave * ppw + lane / G;
    const int g = lane & (G - 1);
    const long total = (long)a.N * a.H * a.W;
    const bool ok = pix < total;
    const long pid = ok ? pix : 0;
    const int X = (int)(pid % a.W), Y = (int)((pid / a.W) % a.H);
    const int n = (int)(pid / ((long)a.W * a.H));
    const float* cp = a.comp + (size_t)n * a.H * a.W * 2;
    float s = 0.f;
    for (int t = g; t < 49; t += G) {
        const int ky = t / 7, kx = t - ky * 7;
        const int y = Y + ky - 3, x = X + kx - 3;
        if (y < 0 || y >= a.H || x < 0 || x >= a.W) continue;
        const float* q = cp + ((size_t)y * a.W + x) * 2;
        s += q[0] * a.w[t * 2] + q[1] * a.w[t * 2 + 1];
    }
    for (int o = G >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float gate = 1.0f / (1.0f + expf(-(s * a.bn_a + a.bn_b)));
    if (ok) {
        float* o = a.xs + pix * a.cs;
        for (int c = g; c < a.C; c += G) o[c] *= gate;
    }
}

}  // namespace

int cbam_spatial_wave_launch(const SpatialArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int G = 1;
    while (G < a.C && G < 64) G <<= 1;
    const long pixels = (long)a.N * a.H * a.W, waves = (pixels + 64 / G - 1) / (64 / G);
    TraceScope ts("cbam_spatial", s);
    hipLaunchKernelGGL(cbam_spatial_wave_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a, G);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

bool chan_pool_partial_wg_fits(const PoolPartArgs& a) { return a.C <= 256; }

int chan_pool_partial_wg_launch(const PoolPartArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("channel_pool_partial", s);
    hipLaunchKernelGGL(chan_pool_partial_wg_kernel, dim3(a.strips, a.N), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int chan_pool_final_wave_launch(const PoolFinalArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("channel_pool_final", s);
    hipLaunchKernelGGL(chan_pool_final_wave_kernel, dim3((unsigned)(((long)a.N * a.C + 3) / 4)), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

bool cbam_gate_wg_fits(const GateArgs& a) { return a.R <= GATE_MAXR; }

int cbam_gate_wg_launch(const GateArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("cbam_gate", s);
    hipLaunchKernelGGL(cbam_gate_wg_kernel, dim3(a.N), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int cbam_scale_compress_wave_launch(const ScaleCompArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int G = 1;
    while (G < a.C && G < 64) G <<= 1;
    const long pixels = (long)a.N * a.HW, waves = (pixels + 64 / G - 1) / (64 / G);
    TraceScope ts("cbam_scale_compress", s);
    hipLaunchKernelGGL(cbam_scale_compress_wave_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a, G);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vfi
