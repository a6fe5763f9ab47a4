// Generic NHWC fp32 building blocks behind the C ABI, used by the FILM path (film.py): every tensor argument is
// a pointer to the first channel of a channel WINDOW inside a possibly wider NHWC tensor (`*_cs` = floats per
// pixel of the underlying tensor), so torch.cat along channels never materialises — producers write their slice.
//
// Reference semantics restated (vfi_models/film/film_arch.py): conv helper 'same' padding :784-798 (2x2 kernels
// pad bottom/right), avg_pool2d pyramids :655-674, warp (align_corners=False, border) :677-724, bilinear flow
// up-sampling F.interpolate(2*v, size=...) :597,610,752, nearest up-sampling :286.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/vfi_hip.h"
#include "vfi_common.h"

namespace vfi {

struct Bil2 {
    int i0, i1;
    float w0, w1;
};
// torch area_pixel_compute_source_index(scale, dst, align_corners=False) + guard_index_and_lambda
__device__ static inline Bil2 bil2(int d, float rscale, int in_size) {
    float src = __fsub_rn(__fmul_rn(rscale, __fadd_rn((float)d, 0.5f)), 0.5f);
    if (src < 0.f) src = 0.f;
    int i0 = (int)floorf(src);
    if (i0 > in_size - 1) i0 = in_size - 1;
    float l = fminf(fmaxf(__fsub_rn(src, (float)i0), 0.f), 1.f);
    Bil2 b;
    b.i0 = i0;
    b.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    b.w1 = l;
    b.w0 = __fsub_rn(1.0f, l);
    return b;
}

// ---- avg_pool2d(2, 2): out[y][x] = (((a + b) + c) + d) / 4, odd sizes floor -------------------------------
__global__ void avgpool2_kernel(const float* __restrict__ in, int in_cs, float* __restrict__ out, int out_cs, int N,
                                int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * Ho * Wo * C4) return;
    const int q = idx % C4;
    long p = idx / C4;
    const int x = p % Wo;
    p /= Wo;
    const int y = p % Ho;
    const int n = p / Ho;
    const float* b = in + ((size_t)(n * H + 2 * y) * W + 2 * x) * in_cs + q * 4;
    const float4 a0 = *(const float4*)b, a1 = *(const float4*)(b + in_cs);
    const float4 b0 = *(const float4*)(b + (size_t)W * in_cs), b1 = *(const float4*)(b + (size_t)W * in_cs + in_cs);
    float4 r;
    r.x = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(a0.x, a1.x), b0.x), b1.x), 0.25f);
    r.y = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(a0.y, a1.y), b0.y), b1.y), 0.25f);
    r.z = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(a0.z, a1.z), b0.z), b1.z), 0.25f);
    r.w = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(a0.w, a1.w), b0.w), b1.w), 0.25f);
    *(float4*)(out + ((size_t)(n * Ho + y) * Wo + x) * out_cs + q * 4) = r;
}

// ---- nearest up-sampling to a given size: src = min(floor(dst * in/out), in-1) -----------------------------
__global__ void upsample_nearest_kernel(const float* __restrict__ in, int in_cs, float* __restrict__ out, int out_cs,
                                        int N, int Hi, int Wi, int Ho, int Wo, int C4, float sy, float sx) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * Ho * Wo * C4) return;
    const int q = idx % C4;
    long p = idx / C4;
    const int x = p % Wo;
    p /= Wo;
    const int y = p % Ho;
    const int n = p / Ho;
    const int iy = min((int)floorf(__fmul_rn((float)y, sy)), Hi - 1);
    const int ix = min((int)floorf(__fmul_rn((float)x, sx)), Wi - 1);
    *(float4*)(out + ((size_t)(n * Ho + y) * Wo + x) * out_cs + q * 4) =
        *(const float4*)(in + ((size_t)(n * Hi + iy) * Wi + ix) * in_cs + q * 4);
}

// ---- bilinear resize to a given size, align_corners=False, input pre-multiplied by `mul` -------------------
__global__ void resize_bilinear_kernel(const float* __restrict__ in, int in_cs, float* __restrict__ out, int out_cs,
                                       int N, int Hi, int Wi, int Ho, int Wo, int C, float sy, float sx, float mul) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * Ho * Wo) return;
    const int x = idx % Wo, y = (idx / Wo) % Ho;
    const int n = idx / ((long)Wo * Ho);
    const Bil2 by = bil2(y, sy, Hi), bx = bil2(x, sx, Wi);
    const float* b = in + (size_t)n * Hi * Wi * in_cs;
    float* o = out + (size_t)idx * out_cs;
    for (int c = 0; c < C; ++c) {
        const float a = __fmul_rn(b[((size_t)by.i0 * Wi + bx.i0) * in_cs + c], mul);
        const float bb = __fmul_rn(b[((size_t)by.i0 * Wi + bx.i1) * in_cs + c], mul);
        const float cc = __fmul_rn(b[((size_t)by.i1 * Wi + bx.i0) * in_cs + c], mul);
        const float d = __fmul_rn(b[((size_t)by.i1 * Wi + bx.i1) * in_cs + c], mul);
        o[c] = __fadd_rn(__fmul_rn(by.w0, __fadd_rn(__fmul_rn(bx.w0, a), __fmul_rn(bx.w1, bb))),
                         __fmul_rn(by.w1, __fadd_rn(__fmul_rn(bx.w0, cc), __fmul_rn(bx.w1, d))));
    }
}

// ---- FILM warp: out(x,y) = bilinear(image, x + fx*mul, y + fy*mul), border clamp, align_corners=False ------
// fp32 expression order of film_arch.warp + grid_sample: grid = linspace(-(1-1/W), 1-1/W, W)[x] + f/(W*0.5);
// ix = ((grid + 1) * W - 1) / 2, clipped to [0, W-1].
__device__ static inline float lin_ls(int i, int n, float startf, float endf, float step) {
    return i < n / 2 ? __fadd_rn(startf, __fmul_rn(step, (float)i)) : __fsub_rn(endf, __fmul_rn(step, (float)(n - 1 - i)));
}
__global__ void warp_film_kernel(const float* __restrict__ in, int in_cs, const float* __restrict__ flow, int flow_cs,
                                 float fmul, float* __restrict__ out, int out_cs, int N, int H, int W, int C4,
                                 float sx0, float sx1, float stepx, float sy0, float sy1, float stepy) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W * C4) return;
    const int q = idx % C4;
    const long p = idx / C4;
    const int x = p % W, y = (p / W) % H;
    const int n = p / ((long)W * H);
    const float fx = __fmul_rn(flow[p * flow_cs], fmul), fy = __fmul_rn(flow[p * flow_cs + 1], fmul);
    // grid = lin - ((-f) / (size*0.5))
    const float gx = __fsub_rn(lin_ls(x, W, sx0, sx1, stepx), __fdiv_rn(-fx, __fmul_rn((float)W, 0.5f)));
    const float gy = __fsub_rn(lin_ls(y, H, sy0, sy1, stepy), __fdiv_rn(-fy, __fmul_rn((float)H, 0.5f)));
    float px = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), (float)W), 1.0f), 2.0f);
    float py = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), (float)H), 1.0f), 2.0f);
    px = fminf((float)(W - 1), fmaxf(px, 0.f));
    py = fminf((float)(H - 1), fmaxf(py, 0.f));
    const float x0f = floorf(px), y0f = floorf(py);
    const float w = __fsub_rn(px, x0f), e = __fsub_rn(1.0f, w), nn = __fsub_rn(py, y0f), s = __fsub_rn(1.0f, nn);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = x0 + (x0 < W - 1 ? 1 : 0), y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float nw = __fmul_rn(s, e), ne = __fmul_rn(s, w), sw = __fmul_rn(nn, e), se = __fmul_rn(nn, w);
    const float* b = in + (size_t)n * H * W * in_cs + q * 4;
    const float4 a = *(const float4*)(b + ((size_t)y0 * W + x0) * in_cs), bb = *(const float4*)(b + ((size_t)y0 * W + x1) * in_cs);
    const float4 c = *(const float4*)(b + ((size_t)y1 * W + x0) * in_cs), d = *(const float4*)(b + ((size_t)y1 * W + x1) * in_cs);
    float4 r;
    r.x = a.x * nw + bb.x * ne + c.x * sw + d.x * se;
    r.y = a.y * nw + bb.y * ne + c.y * sw + d.y * se;
    r.z = a.z * nw + bb.z * ne + c.z * sw + d.z * se;
    r.w = a.w * nw + bb.w * ne + c.w * sw + d.w * se;
    *(float4*)(out + p * out_cs + q * 4) = r;
}

// ---- out = alpha * a + beta * b (b may be null), arbitrary channel windows -------------------------------
__global__ void axpby_kernel(const float* __restrict__ a, int a_cs, const float* __restrict__ b, int b_cs,
                             float* __restrict__ out, int out_cs, long px, int C, float alpha, float beta) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= px * C) return;
    const int c = idx % C;
    const long p = idx / C;
    float v = __fmul_rn(a[p * a_cs + c], alpha);
    if (b) v = __fadd_rn(v, __fmul_rn(b[p * b_cs + c], beta));
    out[p * out_cs + c] = v;
}

// the same for C % 4 == 0 with 16-byte aligned windows: one float4 per thread, 32-bit index arithmetic (the scalar form spends a 64-bit
// division per ELEMENT: IFUNet's 73 and FILM's 40 calls per frame ran at 2.5 TB/s)
__global__ void axpby4_kernel(const float* __restrict__ a, int a_cs, const float* __restrict__ b, int b_cs, float* __restrict__ out, int out_cs,
                              unsigned total, unsigned q, float alpha, float beta) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const unsigned p = idx / q, g = idx - p * q;
    const float4 x = *(const float4*)(a + (size_t)p * a_cs + 4 * g);
    float4 v = make_float4(__fmul_rn(x.x, alpha), __fmul_rn(x.y, alpha), __fmul_rn(x.z, alpha), __fmul_rn(x.w, alpha));
    if (b) {
        const float4 y = *(const float4*)(b + (size_t)p * b_cs + 4 * g);
        v.x = __fadd_rn(v.x, __fmul_rn(y.x, beta));
        v.y = __fadd_rn(v.y, __fmul_rn(y.y, beta));
        v.z = __fadd_rn(v.z, __fmul_rn(y.z, beta));
        v.w = __fadd_rn(v.w, __fmul_rn(y.w, beta));
    }
    *(float4*)(out + (size_t)p * out_cs + 4 * g) = v;
}

}  // namespace vfi

using namespace vfi;

struct vfi_conv {
    float* w = nullptr;
    float* ww = nullptr;     // Winograd F(2x2,3x3) pack of a 3x3 stride-1 layer (conv_wino.hip), next to the direct kernel's
    float* bias = nullptr;
    float* prelu = nullptr;  // per-channel PReLU slopes [Cout_p] (optional)
    // ConvTranspose2d(4, 2, 1) as ONE 3x3 layer with 4 * Cout channels on the Winograd kernel (conv_wino.hip: pack_deconv_as_conv3x3):
    // ww holds that pack, bias3 / prelu3 the bias / per-channel PReLU slopes repeated per parity group (none / LeakyReLU / PReLU layers
    // take this form)
    float* bias3 = nullptr;
    float* prelu3 = nullptr;
    int Cout3_p = 0;
    int Cout = 0, Cout_p = 0, Cin = 0, Cin_p = 0, kh = 0, kw = 0, taps = 0;
    int stride = 1, pad_mode = 0, kind = 0;  // kind 0: Conv2d, 1: ConvTranspose2d(4, 2, 1), 2: nearest x2 up-sampling + Conv2d(2, 'same') (vfi_conv_create_up2x2)
    unsigned char* tapmask = nullptr;         // kind 2: taps of every 64-channel N block's parity
};

static unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

extern "C" {

vfi_conv_t* vfi_conv_create(const float* w_oihw_host, const float* bias_host, int Cout, int Cin, int kh, int kw,
                            const int* chan_map, int Cin_phys) {
    if (!w_oihw_host || Cout <= 0 || Cin <= 0 || kh != kw || (kh != 1 && kh != 2 && kh != 3) || Cin_phys % 8 || Cin_phys < Cin) {
        set_error("vfi_conv_create: bad arguments (Cout=%d Cin=%d k=%dx%d Cin_phys=%d; k in {1,2,3}, Cin_phys a multiple of 8)",
                  Cout, Cin, kh, kw, Cin_phys);
        return nullptr;
    }
    vfi_conv* c = new vfi_conv();
    c->Cout = Cout;
    c->Cout_p = round_up(Cout, 32);
    c->Cin = Cin;
    c->Cin_p = Cin_phys;
    c->kh = kh;
    c->kw = kw;
    c->taps = kh * kw;
    const int cin8 = Cin_phys / 8;
    std::vector<float> wp((size_t)c->taps * cin8 * c->Cout_p * 8, 0.f), bp(c->Cout_p, 0.f);
    for (int co = 0; co < Cout; ++co) {
        if (bias_host) bp[co] = bias_host[co];
        for (int ci = 0; ci < Cin; ++ci) {
            const int pc = chan_map ? chan_map[ci] : ci;
            if (pc < 0 || pc >= Cin_phys) {
                set_error("vfi_conv_create: chan_map[%d]=%d outside 0..%d", ci, pc, Cin_phys - 1);
                delete c;
                return nullptr;
            }
            for (int t = 0; t < c->taps; ++t)
                wp[(((size_t)t * cin8 + pc / 8) * c->Cout_p + co) * 8 + (pc & 7)] = w_oihw_host[((size_t)co * Cin + ci) * c->taps + t];
        }
    }
    if (hipMalloc((void**)&c->w, wp.size() * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->bias, bp.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(c->w, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("vfi_conv_create: device allocation/upload failed");
        vfi_conv_destroy(c);
        return nullptr;
    }
    // Winograd F(2x2,3x3) pack: every 3x3 layer — and the 2x2 'same' layers with a long reduction (FILM's first fusion conv,
    // 1936 -> 512): embedded in a 3x3 kernel (w3[1 + dy][1 + dx] = w[dy][dx]: 'same' for k = 2 pads bottom / right) they cost the
    // Winograd kernel 2.25x the multiplications of the direct 2x2 form, which it wins back exactly — at its 0.78 matrix-pipe
    // utilisation for Cin >= 512 against the direct kernel's 0.66 (measured: 2.46 -> ~2.05 ms); shorter reductions do not gain.
    std::vector<float> w3;
    if (kh == 2 && Cin_phys >= 1024) {
        w3.assign((size_t)Cout * Cin * 9, 0.f);
        for (size_t i = 0; i < (size_t)Cout * Cin; ++i)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) w3[i * 9 + (1 + dy) * 3 + 1 + dx] = w_oihw_host[i * 4 + dy * 2 + dx];
    }
    if (kh == 3 || !w3.empty()) {
        std::vector<float> ww;
        pack_wino3x3(w3.empty() ? w_oihw_host : w3.data(), Cout, Cin, chan_map, Cin_phys, c->Cout_p, ww);
        if (hipMalloc((void**)&c->ww, ww.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(c->ww, ww.data(), ww.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("vfi_conv_create: device allocation/upload failed (Winograd pack)");
            vfi_conv_destroy(c);
            return nullptr;
        }
    }
    return c;
}

// F.interpolate(x, scale 2, 'nearest') followed by Conv2d(Cin, Cout, 2, padding 'same') — FILM's Fusion (film_arch.py:282-292) — as ONE
// layer on the LOW-resolution input: 4 * Cout output channels (parity g = 2 py + px of the up-sampled image), 2x2 taps at (y + a, x + b)
// with the weights of the original taps (a', b') that read that input pixel summed: a = (py + a') >> 1, b = (px + b') >> 1.  Parity (0,0)
// keeps one tap, (0,1) / (1,0) two, (1,1) four — the kernel walks 9 tap blocks instead of 16 (conv_mfma2.hip, MASKED).
vfi_conv_t* vfi_conv_create_up2x2(const float* w_oihw_host, const float* bias_host, int Cout, int Cin, const int* chan_map, int Cin_phys) {
    if (!w_oihw_host || Cout <= 0 || Cout % 64 || Cin <= 0 || Cin_phys % 8 || Cin_phys < Cin) {
        set_error("vfi_conv_create_up2x2: bad arguments (Cout=%d must be a multiple of 64, Cin=%d, Cin_phys=%d a multiple of 8)", Cout, Cin, Cin_phys);
        return nullptr;
    }
    vfi_conv* c = new vfi_conv();
    c->kind = 2;
    c->Cout = Cout;
    c->Cout_p = 4 * Cout;
    c->Cin = Cin;
    c->Cin_p = Cin_phys;
    c->kh = c->kw = 2;
    c->taps = 4;
    const int cin8 = Cin_phys / 8;
    std::vector<float> wp((size_t)4 * cin8 * c->Cout_p * 8, 0.f), bp(c->Cout_p, 0.f);
    for (int g = 0; g < 4; ++g) {
        const int py = g >> 1, px = g & 1;
        for (int co = 0; co < Cout; ++co) {
            bp[g * Cout + co] = bias_host ? bias_host[co] : 0.f;
            for (int ci = 0; ci < Cin; ++ci) {
                const int pc = chan_map ? chan_map[ci] : ci;
                if (pc < 0 || pc >= Cin_phys) {
                    set_error("vfi_conv_create_up2x2: chan_map[%d]=%d outside 0..%d", ci, pc, Cin_phys - 1);
                    delete c;
                    return nullptr;
                }
                for (int ta = 0; ta < 2; ++ta)
                    for (int tb = 0; tb < 2; ++tb) {      // original tap (ta, tb) of the up-sampled image -> low-resolution tap (a, b)
                        const int a = (py + ta) >> 1, b = (px + tb) >> 1;
                        wp[(((size_t)(a * 2 + b) * cin8 + pc / 8) * c->Cout_p + g * Cout + co) * 8 + (pc & 7)] += w_oihw_host[((size_t)co * Cin + ci) * 4 + ta * 2 + tb];
                    }
            }
        }
    }
    std::vector<unsigned char> mask(c->Cout_p / 64);
    const unsigned char gmask[4] = {0x1, 0x3, 0x5, 0xf};      // taps (bit a * 2 + b) parity g can reach: a <= py, b <= px
    for (size_t y = 0; y < mask.size(); ++y) mask[y] = gmask[(y * 64) / Cout];
    if (hipMalloc((void**)&c->w, wp.size() * sizeof(float)) != hipSuccess || hipMalloc((void**)&c->bias, bp.size() * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->tapmask, mask.size()) != hipSuccess ||
        hipMemcpy(c->w, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->tapmask, mask.data(), mask.size(), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("vfi_conv_create_up2x2: device allocation/upload failed");
        vfi_conv_destroy(c);
        return nullptr;
    }
    return c;
}

void vfi_conv_destroy(vfi_conv_t* c) {
    if (!c) return;
    if (c->tapmask) (void)hipFree(c->tapmask);
    if (c->w) (void)hipFree(c->w);
    if (c->ww) (void)hipFree(c->ww);
    if (c->bias) (void)hipFree(c->bias);
    if (c->prelu) (void)hipFree(c->prelu);
    if (c->bias3) (void)hipFree(c->bias3);
    if (c->prelu3) (void)hipFree(c->prelu3);
    delete c;
}

static bool upload(float** dst, const std::vector<float>& v) {
    return hipMalloc((void**)dst, v.size() * sizeof(float)) == hipSuccess &&
           hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
}

vfi_conv_t* vfi_conv_create_ex(int kind, const float* w_host, const float* bias_host, int Cout, int Cin, int k, int stride,
                               int pad_mode, const int* chan_map, int Cin_phys, const float* prelu_host) {
    const bool conv_ok = kind == 0 && ((k == 3 && (stride == 1 || stride == 2)) || (k == 2 && stride == 2) || (k == 1 && stride == 1));
    const bool deconv_ok = kind == 1 && k == 4 && stride == 2;
    if (!w_host || Cout <= 0 || Cin <= 0 || !(conv_ok || deconv_ok) || Cin_phys % 8 || Cin_phys < Cin || pad_mode < 0 || pad_mode > 1) {
        set_error("vfi_conv_create_ex: unsupported layer (kind=%d Cout=%d Cin=%d k=%d stride=%d pad_mode=%d Cin_phys=%d)", kind, Cout,
                  Cin, k, stride, pad_mode, Cin_phys);
        return nullptr;
    }
    vfi_conv* c = new vfi_conv();
    c->kind = kind;
    c->Cout = Cout;
    c->Cout_p = round_up(Cout, 32);
    c->Cin = Cin;
    c->Cin_p = Cin_phys;
    c->kh = c->kw = k;
    c->taps = kind == 1 ? 4 : k * k;
    c->stride = stride;
    c->pad_mode = pad_mode;
    const int cin8 = Cin_phys / 8, ngrp = kind == 1 ? 4 : 1;
    std::vector<float> wp((size_t)ngrp * c->taps * cin8 * c->Cout_p * 8, 0.f), bp((size_t)ngrp * c->Cout_p, 0.f);
    for (int ci = 0; ci < Cin; ++ci) {
        const int pc = chan_map ? chan_map[ci] : ci;
        if (pc < 0 || pc >= Cin_phys) {
            set_error("vfi_conv_create_ex: chan_map[%d]=%d outside 0..%d", ci, pc, Cin_phys - 1);
            delete c;
            return nullptr;
        }
        for (int co = 0; co < Cout; ++co) {
            if (kind == 0) {
                for (int t = 0; t < c->taps; ++t)
                    wp[(((size_t)t * cin8 + pc / 8) * c->Cout_p + co) * 8 + (pc & 7)] = w_host[((size_t)co * Cin + ci) * c->taps + t];
            } else {
                // out[co, 2y+py, 2x+px] = b[co] + sum in[ci, y+py-1+a, x+px-1+b] * w[ci, co, 3-py-2a, 3-px-2b]
                for (int g = 0; g < 4; ++g)
                    for (int t = 0; t < 4; ++t) {
                        const int ky = 3 - (g >> 1) - 2 * (t >> 1), kx = 3 - (g & 1) - 2 * (t & 1);
                        wp[((((size_t)g * 4 + t) * cin8 + pc / 8) * c->Cout_p + co) * 8 + (pc & 7)] =
                            w_host[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
                    }
            }
        }
    }
    if (bias_host)
        for (int g = 0; g < ngrp; ++g)
            for (int co = 0; co < Cout; ++co) bp[(size_t)g * c->Cout_p + co] = bias_host[co];
    bool ok = upload(&c->w, wp) && upload(&c->bias, bp);
    if (ok && kind == 0 && k == 3 && stride == 1) {
        std::vector<float> ww;
        pack_wino3x3(w_host, Cout, Cin, chan_map, Cin_phys, c->Cout_p, ww);
        ok = upload(&c->ww, ww);
    }
    if (ok && prelu_host) {
        std::vector<float> pp(c->Cout_p, 0.f);
        for (int co = 0; co < Cout; ++co) pp[co] = prelu_host[co];
        ok = upload(&c->prelu, pp);
    }
    // the transposed convolution's 3x3 / Winograd form (4x the U-transform weight memory): packed only for layers that can ever take it —
    // vfi_conv_forward_ex sends replicate-padded ones to the direct kernel whatever the image.  Per-channel PReLU layers take it
    // since r5 (conv_wino.hip MODE 1): their slopes are repeated per parity group like the bias.
    if (ok && kind == 1 && 4 * Cout <= 1024 && !pad_mode) {
        std::vector<float> w3, b3, ww;
        pack_deconv_as_conv3x3(w_host, bias_host, Cin, Cout, w3, b3);
        c->Cout3_p = round_up(4 * Cout, 32);
        pack_wino3x3(w3.data(), 4 * Cout, Cin, chan_map, Cin_phys, c->Cout3_p, ww);
        b3.resize(c->Cout3_p, 0.f);
        ok = upload(&c->ww, ww) && upload(&c->bias3, b3);
        if (ok && prelu_host) {
            std::vector<float> p3(c->Cout3_p, 0.f);
            for (int g = 0; g < 4; ++g)
                for (int co = 0; co < Cout; ++co) p3[(size_t)g * Cout + co] = prelu_host[co];
            ok = upload(&c->prelu3, p3);
        }
    }
    if (!ok) {
        set_error("vfi_conv_create_ex: device allocation/upload failed");
        vfi_conv_destroy(c);
        return nullptr;
    }
    return c;
}

int vfi_conv_forward_ex(const vfi_conv_t* c, const float* in_dev, int in_cs, int Hin, int Win, float* out_dev, int out_cs, int N,
                        int act, float slope, float post_scale, float post_shift, const float* res_dev, int res_cs, void* stream) {
    VFI_REQUIRE(c && in_dev && out_dev && N > 0 && Hin > 0 && Win > 0, "vfi_conv_forward_ex: bad arguments");
    VFI_REQUIRE(in_cs >= c->Cin_p && in_cs % 4 == 0 && ((uintptr_t)in_dev & 15) == 0,
                "vfi_conv_forward_ex: input window must hold %d channels, 16-byte aligned (in_cs=%d)", c->Cin_p, in_cs);
    VFI_REQUIRE(act != 3 || c->prelu, "vfi_conv_forward_ex: act 3 (per-channel PReLU) needs slopes given at create time");
    VFI_REQUIRE(c->stride == 1 || (Hin % 2 == 0 && Win % 2 == 0) || c->kind == 1,
                "vfi_conv_forward_ex: stride-2 layers need even input sizes (%dx%d)", Hin, Win);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_dev;
    a.w = c->w;
    a.bias = c->bias;
    a.prelu = c->prelu;
    a.res = res_dev;
    a.res_cs = res_cs;
    a.out = out_dev;
    a.N = N;
    a.Hin = Hin;
    a.Win = Win;
    a.in_cs = in_cs;
    a.out_cs = out_cs;
    a.Cin_p = c->Cin_p;
    a.Cout_p = c->Cout_p;
    a.Cout = c->Cout;
    a.ntaps = c->taps;
    a.act = act;
    a.slope = slope;
    a.split_ok = 1;
    a.post_scale = post_scale;
    a.post_shift = post_shift;
    a.pad_replicate = c->pad_mode;
    char name[80];
    if (c->kind == 1) {
        VFI_REQUIRE(!res_dev, "vfi_conv_forward_ex: residual not supported for transposed convs");
        a.Hout = Hin;  // per parity group; the kernel interleaves the 4 groups into [2*Hin, 2*Win]
        a.Wout = Win;
        a.out_mode = 2;
        snprintf(name, sizeof(name), "deconv4x4s2_%dto%d", c->Cin_p, c->Cout);
    } else {
        a.Hout = Hin / c->stride;
        a.Wout = Win / c->stride;
        a.tap_y0 = a.tap_x0 = c->kh == 3 ? -1 : 0;
        snprintf(name, sizeof(name), "conv%dx%ds%d_%dto%d", c->kh, c->kw, c->stride, c->Cin_p, c->Cout);
    }
    static const bool by_shape = getenv("VFI_TRACE_SHAPES") != nullptr;      // per-shape trace rows: append the OUTPUT size and batch
    if (by_shape) snprintf(name + strlen(name), sizeof(name) - strlen(name), "@%dx%dx%d", N, c->kind == 1 ? 2 * Hin : a.Hout, c->kind == 1 ? 2 * Win : a.Wout);
    static std::map<std::string, const char*> names;
    auto it = names.find(name);
    if (it == names.end()) it = names.emplace(name, strdup(name)).first;
    if (c->ww && c->kind == 0 && c->stride == 1 && conv_wino_eligible(a)) {
        a.w = c->ww;
        return conv_wino_launch(a, 0, (hipStream_t)stream, it->second);
    }
    if (c->ww && c->kind == 1 && option(kOptDeconvWino) && conv_wino_mode(-1) != 1 && !c->pad_mode && post_scale == 0.f && !res_dev &&
        (act == 0 || (act == 1 && slope >= 0.f && slope <= 1.f) || (act == 3 && c->prelu3))) {
        // the transposed convolution as one 3x3 layer with 4 * Cout channels (no padding of Cout to a 32-wide N tile per parity group);
        // chosen from the image, never from the batch
        ConvArgs b = a;
        conv3x3_taps(b);
        b.w = c->ww;
        b.bias = c->bias3;
        b.prelu = c->prelu3;
        b.Cout = 4 * c->Cout;
        b.Cout_p = c->Cout3_p;
        const long regions = 2L * cdiv(Hin, 8) * cdiv(Win, 16);
        // ... and from 128 x 128 input pixels up: below that (the 34x60 / 68x120 pyramid levels at 1080p, K = 256 ... 768) the direct kernel's
        // split-K wins by 10-75 % whatever the item count (profiles/r05_deconv_ab.txt: 512->256 @34x60 153 vs 267 us, 256->128 @68x120 114 vs 143 us;
        // from 136x240 up the Winograd form wins: 64->16 @544x960 339 vs 696 us)
        if (regions / 4 * (b.Cout_p / 32) >= 192 && (long)Hin * Win >= 16384 && (long)4 * Hin * Win * out_cs * 4 < 0x7fffffffL && (long)Hin * Win * in_cs * 4 < 0x7fffffffL)
            return conv_wino_launch(b, 8, (hipStream_t)stream, it->second);
    }
    return conv_launch(a, c->kind == 1 ? 1 : c->stride, c->kind == 1, -1, (hipStream_t)stream, it->second);
}

int vfi_conv_forward(const vfi_conv_t* c, const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int H, int W,
                     int act, float slope, void* stream) {
    VFI_REQUIRE(c && in_dev && out_dev && N > 0 && H > 0 && W > 0, "vfi_conv_forward: bad arguments");
    VFI_REQUIRE(in_cs >= c->Cin_p && in_cs % 4 == 0 && ((uintptr_t)in_dev & 15) == 0,
                "vfi_conv_forward: input window must hold %d channels, 16-byte aligned (in_cs=%d)", c->Cin_p, in_cs);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_dev;
    a.w = c->w;
    a.bias = c->bias;
    a.out = out_dev;
    a.N = N;
    a.Hin = a.Hout = H;
    a.Win = a.Wout = W;
    a.in_cs = in_cs;
    a.out_cs = out_cs;
    a.Cin_p = c->Cin_p;
    a.Cout_p = c->Cout_p;
    a.Cout = c->Cout;
    a.ntaps = c->taps;
    a.tap_y0 = a.tap_x0 = c->kh == 3 ? -1 : 0;  // 'same': 3x3 centred, 2x2 pads bottom/right, 1x1
    a.act = act;
    a.slope = slope;
    a.split_ok = 1;
    char name[64];
    static const bool by_shape = getenv("VFI_TRACE_SHAPES") != nullptr;      // per-shape trace rows (tools/film_bench.py --shapes)
    if (c->kind == 2) {      // H x W is the LOW-resolution input; the output is [2H, 2W, out_cs] (vfi_conv_create_up2x2)
        VFI_REQUIRE(act == 0 || act == 1, "vfi_conv_forward: the up-sample x2 + 2x2 layer takes act 0 / 1");
        a.tapmask = c->tapmask;
        a.par_cout = c->Cout;
        a.Cout = c->Cout_p;
        a.out_mode = 3;
        if (by_shape) snprintf(name, sizeof(name), "up2conv2x2_%dto%d@%dx%d", c->Cin_p, c->Cout, 2 * H, 2 * W);
        else snprintf(name, sizeof(name), "up2conv2x2");
    } else if (by_shape) snprintf(name, sizeof(name), "conv%dx%d_%dto%d@%dx%d", c->kh, c->kw, c->Cin_p, c->Cout, H, W);
    else snprintf(name, sizeof(name), "conv%dx%d", c->kh, c->kw);
    static std::map<std::string, const char*> names;  // stable storage for trace names
    auto it = names.find(name);
    if (it == names.end()) it = names.emplace(name, strdup(name)).first;
    if (c->ww) {
        ConvArgs b = a;
        if (c->kh == 2) {       // the 2x2 layer as its 3x3 embedding (vfi_conv_create)
            b.ntaps = 9;
            b.tap_y0 = b.tap_x0 = -1;
        }
        if (conv_wino_eligible(b)) {
            b.w = c->ww;
            return conv_wino_launch(b, 0, (hipStream_t)stream, it->second);
        }
    }
    return conv_launch(a, 1, false, -1, (hipStream_t)stream, it->second);
}

int vfi_avgpool2(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int H, int W, int C, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && H >= 2 && W >= 2,
                "vfi_avgpool2: bad arguments (C, strides multiples of 4)");
    const long n = (long)N * (H / 2) * (W / 2) * (C / 4);
    TraceScope ts("avgpool2", (hipStream_t)stream);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, out_dev, out_cs, N, H, W, C / 4);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_upsample_nearest(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win, int Hout,
                         int Wout, int C, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0, "vfi_upsample_nearest: bad arguments");
    const long n = (long)N * Hout * Wout * (C / 4);
    TraceScope ts("upsample_nearest", (hipStream_t)stream);
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, out_dev, out_cs, N,
                       Hin, Win, Hout, Wout, C / 4, (float)Hin / (float)Hout, (float)Win / (float)Wout);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_resize_bilinear(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win, int Hout,
                        int Wout, int C, float mul, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && C > 0, "vfi_resize_bilinear: bad arguments");
    const long n = (long)N * Hout * Wout;
    TraceScope ts("resize_bilinear", (hipStream_t)stream);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, out_dev, out_cs, N,
                       Hin, Win, Hout, Wout, C, (float)Hin / (float)Hout, (float)Win / (float)Wout, mul);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_warp_film(const float* in_dev, int in_cs, const float* flow_dev, int flow_cs, float flow_mul, float* out_dev,
                  int out_cs, int N, int H, int W, int C, void* stream) {
    VFI_REQUIRE(in_dev && flow_dev && out_dev && C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && H > 1 && W > 1,
                "vfi_warp_film: bad arguments (C and strides multiples of 4)");
    // torch.linspace(-ls, ls, n) in fp32: start/end rounded from double, step = (end-start)/(n-1)
    const float sx1 = (float)(1.0 - 1.0 / (double)W), sx0 = -sx1, stepx = (sx1 - sx0) / (float)(W - 1);
    const float sy1 = (float)(1.0 - 1.0 / (double)H), sy0 = -sy1, stepy = (sy1 - sy0) / (float)(H - 1);
    const long n = (long)N * H * W * (C / 4);
    TraceScope ts("warp_film", (hipStream_t)stream);
    hipLaunchKernelGGL(warp_film_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, flow_dev, flow_cs,
                       flow_mul, out_dev, out_cs, N, H, W, C / 4, sx0, sx1, stepx, sy0, sy1, stepy);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_axpby(const float* a_dev, int a_cs, const float* b_dev, int b_cs, float* out_dev, int out_cs, int64_t pixels, int C,
              float alpha, float beta, void* stream) {
    VFI_REQUIRE(a_dev && out_dev && pixels > 0 && C > 0, "vfi_axpby: bad arguments");
    TraceScope ts("axpby", (hipStream_t)stream);
    const bool vec = C % 4 == 0 && a_cs % 4 == 0 && out_cs % 4 == 0 && (!b_dev || b_cs % 4 == 0) &&
                     ((((uintptr_t)a_dev | (uintptr_t)out_dev | (uintptr_t)b_dev) & 15) == 0) && pixels * (C / 4) < (int64_t)0xffffff00u;
    if (vec) {
        const unsigned total = (unsigned)(pixels * (C / 4));
        hipLaunchKernelGGL(axpby4_kernel, dim3((total + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, a_dev, a_cs, b_dev, b_cs, out_dev, out_cs, total,
                           (unsigned)(C / 4), alpha, beta);
        VFI_CHECK_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(axpby_kernel, dim3(nblk(pixels * C)), dim3(256), 0, (hipStream_t)stream, a_dev, a_cs, b_dev, b_cs, out_dev,
                       out_cs, (long)pixels, C, alpha, beta);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
