// Wave-cooperative / matrix-core forms of the GMFlow steps whose one-thread-per-element bodies (gmfss_bodies.h) were the top
// of the GMFSS `prepare` profile in round 2.  The bodies stay the specification (tests/hostcheck runs them on the host);
// these launchers are what gmfss_ops.hip dispatches to on the device when the shape fits, else it launches the body.
#pragma once
#include "gmfss_bodies.h"

namespace vfi {

// nn.LayerNorm over C <= 256 channels: one wave per token, coalesced row reads, shuffle reductions.
bool layernorm_wave_fits(const vfi_gmfss::LayerNormArgs& a);
int layernorm_wave_launch(const vfi_gmfss::LayerNormArgs& a, void* stream);

// InstanceNorm2d partial sums, C <= 256: one workgroup per (strip, image) instead of C threads per strip.
bool instnorm_partial_wg_fits(const vfi_gmfss::InStatsArgs& a);
int instnorm_partial_wg_launch(const vfi_gmfss::InStatsArgs& a, void* stream);
int instnorm_final_wave_launch(const vfi_gmfss::InFinalArgs& a, void* stream);      // one wave per (image, channel)

// local_correlation_softmax with C = 128, radius 4 on the fp32 matrix cores (gmfss_match section of gmfss_fast.hip).
bool local_match_mfma_fits(const vfi_gmfss::LocalMatchArgs& a);
int local_match_mfma_launch(const vfi_gmfss::LocalMatchArgs& a, void* stream);

// local window flow propagation with C = 128, radius 1: 16 lanes per pixel.
bool local_prop_coop_fits(const vfi_gmfss::LocalPropArgs& a);
int local_prop_coop_launch(const vfi_gmfss::LocalPropArgs& a, void* stream);

}  // namespace vfi
