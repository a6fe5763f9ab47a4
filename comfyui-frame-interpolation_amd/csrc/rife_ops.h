// Launchers of the non-convolution RIFE kernels (rife_ops.hip).
#pragma once
#include "vfi_common.h"

namespace vfi {

constexpr int kMaxTasks = 32;  // tasks per launch (kernel-argument table: 384 bytes)
struct RifeTasks {
    int slot0[kMaxTasks];
    int slot1[kMaxTasks];
    float t[kMaxTasks];
};

int warp_border_launch(const float* in, const float* flow, float* out, int N, int H, int W, int C, hipStream_t s);
int prep_frame_launch(const float* src, float* P, int H, int W, int C, int Hp, int Wp, hipStream_t s);
// arch 4.7: prep + encode.0 + encode.1 in one launch (exactly one of f32 / u8 non-null)
int encode47_fused_launch(const float* f32, const unsigned char* u8, float* P, const float* w0, const float* b0, const float* w1, const float* b1,
                          int H, int W, int C, int Hp, int Wp, hipStream_t s);
int encode47_batch_launch(int n, const void* const* srcs, bool u8, float* const* packs, const float* w0, const float* b0, const float* w1,
                          const float* b1, int H, int W, int C, int Hp, int Wp, hipStream_t s);
int prep_frame_u8_launch(const unsigned char* src, float* P, int H, int W, int C, int Hp, int Wp, hipStream_t s);
int f32_to_u8_launch(const float* in, unsigned char* out, long n, hipStream_t s);
// Head: first conv 3 -> CM (stride 2, optional LeakyReLU) into E [Hp/2][Wp/2][CM]; last layer CM -> CF transposed conv
// into pack planes 1..CF/4.  (CM, CF, act) = (16, 4, false) for 4.7, (32, 8, true) for 4.17.
int encode_conv_launch(const float* P, float* E, const float* w0, const float* b0, int CM, bool act, int Hp, int Wp,
                       hipStream_t s);
int encode_deconv_launch(const float* E, float* P, const float* w1, const float* b1, int CM, int CF, int Hp, int Wp,
                         hipStream_t s);
// NF = feature planes of the frame pack (1: 4 feature channels, 2: 8); CX = round_up(7 + 8*NF (+5 with flow), 8)
// FEAT (nullable): [B][2][Hp][Wp][4] features carried from the previous block (arch 4.26), inserted before the flow
int stage_in_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* F,
                    const float* M, const float* FEAT, float* X, int Hp, int Wp, int s, int CX, int NF, bool has_flow,
                    hipStream_t st);
// tp = planes of T per task (2: flow+mask, 4: arch 4.26 with 8 carried feature channels)
int flow_up_launch(const float* T, float* F, float* M, int B, int Hp, int Wp, int s, int tp, bool has_prev, hipStream_t st);
int feat_up_launch(const float* T, float* FEAT, int B, int Hp, int Wp, int s, hipStream_t st);
int stage_trans_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T, float* F,
                       float* X, int Hp, int Wp, int s_prev, int s_next, int NF, bool has_prev, hipStream_t st);
// last transition (scales 2 -> 1, NF = 1) fused into the next block's conv0.0 (24 -> 32 channels, stride 2, LeakyReLU):
// reads Fin, writes the updated flow to Fout (a different buffer) and conv0.0's output A0 [B][Hp/2][Wp/2][32]; X is never stored
int trans1_conv0a_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T, const float* Fin,
                         float* Fout, const float* wpk, const float* bias, float* A0, int Hp, int Wp, float slope, hipStream_t st);
// arch 4.26 (T with 4 planes, 8 carried feature channels, NF = 1); block scales 2*s_next -> s_next, s_next in {8,4,2,1}
int stage_trans_x_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T, float* F,
                         float* X, int Hp, int Wp, int s_prev, int s_next, bool has_prev, hipStream_t st);
int final_blend_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T,
                       const float* F, float* out, float* Fdbg, int H, int W, int Hp, int Wp, int s, int tp, hipStream_t st);
int planar4_up_launch(const float* X1, float* X, int B, int Hp, int Wp, int u, int CX, int flow_plane, hipStream_t st);
int t_down_launch(const float* T, float* T1, int B, int Hp, int Wp, int u, int tp, hipStream_t st);
int t_to_nhwc_launch(const float* T, float* out, int N, int Hq, int Wq, int C4, hipStream_t st);

}  // namespace vfi
