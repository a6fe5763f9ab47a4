/* ORACLE — test infrastructure only.  Never linked into or called by the product path.
 *
 * Plain-C restatement of the two custom CUDA ops on M2M's hot path, following the CUDA kernel TEXT of the
 * reference.  PINNED BY EXECUTION of that text: oracle/validate_m2m_vs_reference.py lets the reference's own
 * cuda_kernel() specialise its kernel strings, compiles them with g++ behind a serial __global__/atomicAdd shim
 * (oracle/stubs/cupy) and requires bit-equality with the functions below on edge-case inputs (oracle/VALIDATION_M2M.log;
 * outputs kept as tests/golden/m2m_ops_ref.npz; prebuilt kernels under oracle/_ref for the GPU box).  On a GPU the
 * reference's atomics commit in an unspecified order; both sides here use the thread-index order.
 *
 *   softsplat_out  vfi_models/ops/cupy_ops/softsplat.py:140-192   (launch :205-224, zero-init :201-203)
 *   costvol_out    vfi_models/ops/cupy_ops/costvol.py:4-43         (launch :143-179)
 *
 * Tensors are NCHW fp32 exactly as in the reference.  One loop iteration == one CUDA thread, visited in
 * thread-index order; fp32 arithmetic, no fused multiply-add (compile with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* softsplat.py:147-191: thread per (n,c,y,x); out[n,c,y',x'] += in * bilinear weight at 4 integer neighbours of
 * (x+fx, y+fy), each bounds-checked; non-finite target -> skip (:157-158). */
void oracle_softsplat_sum(const float* in, const float* flow, float* out, int N, int C, int H, int W) {
    memset(out, 0, sizeof(float) * (size_t)N * C * H * W);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const float fltX = (float)x + flow[(((size_t)n * 2 + 0) * H + y) * W + x];
                    const float fltY = (float)y + flow[(((size_t)n * 2 + 1) * H + y) * W + x];
                    if (!isfinite(fltX) || !isfinite(fltY)) continue;
                    const float fltIn = in[(((size_t)n * C + c) * H + y) * W + x];
                    const int nwX = (int)floorf(fltX), nwY = (int)floorf(fltY);
                    const int neX = nwX + 1, neY = nwY, swX = nwX, swY = nwY + 1, seX = nwX + 1, seY = nwY + 1;
                    const float wNW = ((float)seX - fltX) * ((float)seY - fltY);
                    const float wNE = (fltX - (float)swX) * ((float)swY - fltY);
                    const float wSW = ((float)neX - fltX) * (fltY - (float)neY);
                    const float wSE = (fltX - (float)nwX) * (fltY - (float)nwY);
                    float* o = out + ((size_t)n * C + c) * H * W;
                    if (nwX >= 0 && nwX < W && nwY >= 0 && nwY < H) o[(size_t)nwY * W + nwX] += fltIn * wNW;
                    if (neX >= 0 && neX < W && neY >= 0 && neY < H) o[(size_t)neY * W + neX] += fltIn * wNE;
                    if (swX >= 0 && swX < W && swY >= 0 && swY < H) o[(size_t)swY * W + swX] += fltIn * wSW;
                    if (seX >= 0 && seX < W && seY >= 0 && seY < H) o[(size_t)seY * W + seX] += fltIn * wSE;
                }
}

/* The same splat with the sources visited in REVERSE order (last pixel first): every contribution is the same product, only the
 * order of the additions into a target changes — what the reference's own atomicAdd kernel leaves undefined on a GPU
 * (softsplat.py:176-190).  Used by oracle/m2m_hot_certificate.py to measure how far the ORACLE's frame moves under a change of
 * summation order alone. */
void oracle_softsplat_sum_rev(const float* in, const float* flow, float* out, int N, int C, int H, int W) {
    memset(out, 0, sizeof(float) * (size_t)N * C * H * W);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int y = H - 1; y >= 0; --y)
                for (int x = W - 1; x >= 0; --x) {
                    const float fltX = (float)x + flow[(((size_t)n * 2 + 0) * H + y) * W + x];
                    const float fltY = (float)y + flow[(((size_t)n * 2 + 1) * H + y) * W + x];
                    if (!isfinite(fltX) || !isfinite(fltY)) continue;
                    const float fltIn = in[(((size_t)n * C + c) * H + y) * W + x];
                    const int nwX = (int)floorf(fltX), nwY = (int)floorf(fltY);
                    const int neX = nwX + 1, neY = nwY, swX = nwX, swY = nwY + 1, seX = nwX + 1, seY = nwY + 1;
                    const float wNW = ((float)seX - fltX) * ((float)seY - fltY);
                    const float wNE = (fltX - (float)swX) * ((float)swY - fltY);
                    const float wSW = ((float)neX - fltX) * (fltY - (float)neY);
                    const float wSE = (fltX - (float)nwX) * (fltY - (float)nwY);
                    float* o = out + ((size_t)n * C + c) * H * W;
                    if (seX >= 0 && seX < W && seY >= 0 && seY < H) o[(size_t)seY * W + seX] += fltIn * wSE;
                    if (swX >= 0 && swX < W && swY >= 0 && swY < H) o[(size_t)swY * W + swX] += fltIn * wSW;
                    if (neX >= 0 && neX < W && neY >= 0 && neY < H) o[(size_t)neY * W + neX] += fltIn * wNE;
                    if (nwX >= 0 && nwX < W && nwY >= 0 && nwY < H) o[(size_t)nwY * W + nwX] += fltIn * wNW;
                }
}

/* costvol.py:10-42: thread per (n,y,x); 81 output channels (dy outer, dx inner, both -4..4);
 * out = sum_c |one - two(shifted)| / C, out-of-bounds shift -> sum_c |one| / C. */
void oracle_costvol(const float* one, const float* two, float* out, int N, int C, int H, int W) {
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int ch = 0;
                for (int oy = y - 4; oy <= y + 4; ++oy)
                    for (int ox = x - 4; ox <= x + 4; ++ox) {
                        float v = 0.0f;
                        if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
                            for (int c = 0; c < C; ++c)
                                v += fabsf(one[(((size_t)n * C + c) * H + y) * W + x] -
                                           two[(((size_t)n * C + c) * H + oy) * W + ox]);
                        } else {
                            for (int c = 0; c < C; ++c) v += fabsf(one[(((size_t)n * C + c) * H + y) * W + x]);
                        }
                        out[(((size_t)n * 81 + ch) * H + y) * W + x] = v / (float)C;
                        ++ch;
                    }
            }
}
