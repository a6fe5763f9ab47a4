// Cooperative forms of the CBAM steps of IFUNet whose one-thread-per-element bodies (ifunet_bodies.h) dominated the non-conv
// time of the first MI355X run (profiles/r02_ifunet_first_gpu_run.txt: cbam_gate 20.0 ms, channel_pool_partial 4.8 ms,
// cbam_scale_compress 3.2 ms of 76.6 ms per 1080p frame).  The bodies stay the specification (tests/hostcheck).
#pragma once
#include "ifunet_bodies.h"

namespace vfi {

int chan_pool_partial_wg_launch(const vfi_ifunet::PoolPartArgs& a, void* stream);     // C <= 256: a workgroup per strip
bool chan_pool_partial_wg_fits(const vfi_ifunet::PoolPartArgs& a);
int chan_pool_final_wave_launch(const vfi_ifunet::PoolFinalArgs& a, void* stream);    // a wave per (image, channel)
int cbam_gate_wg_launch(const vfi_ifunet::GateArgs& a, void* stream);                 // R <= 64: a workgroup per image
bool cbam_gate_wg_fits(const vfi_ifunet::GateArgs& a);
int cbam_scale_compress_wave_launch(const vfi_ifunet::ScaleCompArgs& a, void* stream);   // lanes over channels
int cbam_spatial_wave_launch(const vfi_ifunet::SpatialArgs& a, void* stream);            // taps and channels over the lanes

}  // namespace vfi
